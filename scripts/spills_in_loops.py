#!/usr/bin/env python3
"""Where the spill code of a kernel sits: scratch loads / stores per loop of its assembly.
usage: spills_in_loops.py <file.s> <kernel name prefix>     (file from hipcc --cuda-device-only -S)"""
import re, sys, collections
src, pref = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
loop, depth = "-", 0
cnt = collections.OrderedDict()
size = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"^\.?LBB\d+_\d+:\s*;\s*(.*)$", l)
    if m or l.startswith("; %bb."):
        c = m.group(1) if m else l
        h = re.search(r"(?:in Loop: Header=|Parent Loop |=>This Loop Header: )(BB\d+_\d+)? ?Depth=(\d+)", c)
        h2 = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", c)
        h3 = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", c)
        if h2: loop, depth = h2.group(1), int(h2.group(2))
        elif h3: loop, depth = (m.group(0).split(":")[0].lstrip(".L") if m else "?"), int(h3.group(1))
        elif "Loop" not in c: loop, depth = "-", 0
        continue
    t = l.strip()
    if not t or t.startswith(";"): continue
    key = (loop, depth)
    cnt.setdefault(key, [0, 0])
    size[key] += 1
    if t.startswith("scratch_store"): cnt[key][0] += 1
    if t.startswith("scratch_load"): cnt[key][1] += 1
for (lp, d), (s, r) in cnt.items():
    if s or r: print(f"loop {lp:12s} depth {d}: {size[(lp, d)]:5d} instructions, {s:3d} scratch stores, {r:3d} scratch loads")
