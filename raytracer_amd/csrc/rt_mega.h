// rt_mega.h — all generations of a chunk in ONE persistent launch.
//
// The pipeline of rt_kernels.h runs ComputeColor's recursion (raytracer.cpp:382-401, 1248-1436) generation by
// generation: trace(g) -> shade(g) -> trace(shadow slots of g + children g + 1) -> ...  Every trace launch ends
// with an end game in which the last, longest rays finish at their own pace while most lanes of the chip have
// nothing left (~0.4 ms per launch at 1080p), and a depth-6 frame has eight of them.
//
// Here a trace workgroup (one per CU, 16 waves) is a small pipeline of its own:
//   * waves 0..14 run the traversal loop of rt_traverse.h in its mixed mode: a lane's ray is a closest-hit ray of
//     any generation or a shadow ray.  A miss and a shadow verdict are final where they are found; a closest-hit
//     ray that hit something is handed — its record index — to wave 15 through a ring (one per tracing wave:
//     single producer, single consumer).
//   * wave 15 is the workgroup's SHADING wave: it takes up to 64 finished records at a time, finishes
//     FindClosestHit (normal, material: hit_normal), decides the child rays (raytracer.cpp:1272-1352) and shadow
//     tests (raytracer.cpp:1371-1377) exactly as k_shade does, writes them into records / shadow slots it
//     sub-allocates from slabs (one global atomic per slab, not one per batch), and appends them to the workgroup's
//     two rings: child rays and shadow slots.
//   * tracing waves refill their idle lanes from those rings first — some waves shadow slots first, the others child
//     rays and then the chunk's primary-ray cursor — so a wave mostly holds rays of one kind.
//   * WHEN it is used (rt_hip.hip): for chunks of up to RT_MEGA_MAX_PRIMARIES primary rays.  One shading wave's chain
//     of dependent loads (~12 us per batch of 64) keeps up with fifteen tracing waves only while launches are short;
//     on large frames the per-bounce pipeline's throughput wins (2.9 against 2.1 Grays/s on the 4K mesh frame) and
//     its end games no longer matter.  Letting every wave shade its own finished rays instead (no dedicated wave)
//     was tried: the shading code inside the traversal kernel raised its spills from 56 to 293 registers and halved
//     its speed.  profiles/r02/ holds both measurements.
//   * nothing crosses workgroups: rays a workgroup spawns are traced by that workgroup (its waves share one
//     vector L1, so workgroup-scope release / acquire around the LDS ring words is all the ordering needed; no
//     agent-scope hand-off, no cross-XCD traffic).  The cursor hands primaries out in 8-pixel strips spread over
//     the image, so every workgroup gets the same mix of cheap and expensive pixels.
//   * a workgroup is finished when the primaries are gone and its count of unfinished items (rays in flight, ring
//     entries, records waiting to be shaded) is zero.
// What does NOT change: which rays exist, their bits, what each is tested against and in which order
// (trace_persistent is the same code), and the colour arithmetic — k_combine_mega evaluates ApplyShading's
// expression bottom-up over the same records (they carry their depth), so frames are byte-identical to the
// per-bounce pipeline's, which stays available (RT_FLAG_PER_BOUNCE_LAUNCHES) and is compared with this one in
// tests/test_gpu_parity.py.
#pragma once
#include "rt_kernels.h"

#define RT_MEGA_WAVES (RT_TRACE_BLOCK / 64)
#ifndef RT_MEGA_SHADERS
#define RT_MEGA_SHADERS 1           // shading waves per workgroup (the last ones); shading wave k serves the tracing waves t with t % SHADERS == k
#endif
#define RT_MEGA_TRACERS (RT_MEGA_WAVES - RT_MEGA_SHADERS)   // tracing waves per workgroup
#ifndef RT_MEGA_SQ_CAP
#define RT_MEGA_SQ_CAP 2048u        // entries of a tracing wave's ring to the shading wave (power of two)
#endif
#ifndef RT_MEGA_BATCH_WAIT
#define RT_MEGA_BATCH_WAIT 64       // polls (~0.3 us each) the shading wave waits for a full batch while the tracers have other work
#endif
#ifndef RT_MEGA_CQ_CAP
#define RT_MEGA_CQ_CAP (1u << 17)   // entries of a workgroup's ring of child rays (power of two)
#endif
#ifndef RT_MEGA_AQ_CAP
#define RT_MEGA_AQ_CAP (1u << 18)   // entries of a workgroup's ring of shadow slots (power of two)
#endif
#define RT_MEGA_REC_SLAB 2048u      // records / shadow slots a shading wave takes from the pool at a time
#define RT_MEGA_SH_SLAB 4096u
#define RT_MEGA_COMMIT_SPINS (1u << 22)
#ifndef RT_MEGA_MAX_PRIMARIES
#define RT_MEGA_MAX_PRIMARIES (1ull << 20)   // renders of up to this many primary rays take the one-launch pipeline
#endif
#define RT_ITEM_ANY 0x80000000u     // work item: bit 31 set = shadow slot, clear = record (closest hit)
#define RF_DEPTH_SHIFT 24           // link.w bits 24..28: depth of the record's ray (k_combine_mega)
#define RF_DEPTH_NONE 31u           // a record slot of a slab that was never used

// LDS words of the one-launch pipeline (beside TraceLds).  A ring has three cursors: `alloc` (producers reserve
// space), `tail` (entries below it are written: producers publish in the order they reserved), `head` (takers).
struct MegaLds {
    uint32_t work;                          // unfinished items of this workgroup
    uint32_t abort;                         // a wave hit an internal bound: everybody leaves
    uint32_t cq_head, cq_tail, cq_alloc;    // ring of child rays
    uint32_t aq_head, aq_tail, aq_alloc;    // ring of shadow slots
    uint32_t sq_head[RT_MEGA_TRACERS];      // per tracing wave ring to the shading wave (head: shading wave; tail: that wave)
    uint32_t sq_tail[RT_MEGA_TRACERS];
};

struct MegaArgs {
    uint32_t* cq;          // [gridDim.x][RT_MEGA_CQ_CAP]
    uint32_t* aq;          // [gridDim.x][RT_MEGA_AQ_CAP]
    uint32_t* sq;          // [gridDim.x][RT_MEGA_TRACERS][RT_MEGA_SQ_CAP]
    uint32_t n_primary;    // records [0, n_primary) are the chunk's primary rays (k_raygen_*)
    uint32_t any_waves;    // waves of a workgroup that take shadow slots first (the others: closest-hit rays first)
    uint32_t strict;       // 1: while primaries remain, a wave takes only its own kind
};

__device__ __forceinline__ uint32_t lds_read(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_write(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t wave_bcast0(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// One wave's side of the pipeline: items in, results out, and the shading of what it finished.
struct MegaIO {
    ClosestIO c;
    ShadowIO s;
    Control* ctl;
    MegaLds* M;
    uint32_t* cq;          // this workgroup's rings
    uint32_t* aq;
    uint32_t* sq;          // tracing wave: its ring to the shading wave
    uint32_t wave;
    bool pref_any;         // wave-uniform: this wave takes shadow slots first
    bool strict;
    uint32_t n_primary;
    // wave-uniform state
    uint32_t rec_next, rec_end, sh_next, sh_end;      // shading wave: what is left of its slabs
    unsigned long long n_prim_traced;                 // closest-hit rays of depth 0 among `traced`
    // per lane, between store() and flush()
    uint32_t push_rec;     // record to note for shading (0xFFFFFFFF: none)
    uint32_t n_final;      // 1: an item of this lane became final in this pass (miss, shadow verdict)
    bool last_primary;     // the item this lane loaded last is a primary ray

    __device__ __forceinline__ bool aborted() const { return lds_read(&M->abort) != 0u; }
    __device__ __forceinline__ void abort() { lds_write(&M->abort, 1u); }
    __device__ __forceinline__ bool ring_empty() const { return lds_read(&M->cq_tail) == lds_read(&M->cq_head) && lds_read(&M->aq_tail) == lds_read(&M->aq_head); }
    __device__ __forceinline__ bool done() const { return lds_read(&M->work) == 0u; }
    __device__ __forceinline__ uint32_t shade_room() const { return RT_MEGA_SQ_CAP - (lds_read(&M->sq_tail[wave]) - lds_read(&M->sq_head[wave])); }
    __device__ __forceinline__ void count(uint32_t n, uint32_t lane) { if (lane == 0 && n) atomicAdd(&M->work, n); }
    __device__ __forceinline__ void uncount(uint32_t n, uint32_t lane) { if (lane == 0 && n) atomicSub(&M->work, n); }

    // Up to n entries of one ring for the idle lanes with rank r0, r0 + 1, ...: entries are read BEFORE the head moves
    // past them (behind the head a producer may write again).  Wave-uniform result.
    __device__ __forceinline__ uint32_t ring_take(bool any_ring, uint32_t n, uint32_t lane, uint32_t rank, uint32_t r0, uint32_t& it) {
        uint32_t* head = any_ring ? &M->aq_head : &M->cq_head;
        const uint32_t* tail = any_ring ? &M->aq_tail : &M->cq_tail;
        const uint32_t* ring = any_ring ? aq : cq;
        const uint32_t mask = any_ring ? RT_MEGA_AQ_CAP - 1u : RT_MEGA_CQ_CAP - 1u;
        for (int attempt = 0; attempt < 4; attempt++) {
            uint32_t h = 0, t = 0;
            if (lane == 0) { h = lds_read(head); t = lds_read(tail); }
            h = wave_bcast0(h);
            t = wave_bcast0(t);
            const uint32_t k = min(n, t - h);
            if (k == 0) return 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // pairs with the producer's release before its tail store
            const bool me = rank >= r0 && rank - r0 < k;
            uint32_t v = 0;
            if (me) v = ring[(h + (rank - r0)) & mask];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the entries are in registers before the head moves
            uint32_t ok = 0;
            if (lane == 0) ok = atomicCAS(head, h, h + k) == h ? 1u : 0u;
            ok = wave_bcast0(ok);
            if (ok) { if (me) it = v | (any_ring ? RT_ITEM_ANY : 0u); return k; }
        }
        return 0u;
    }
    // Producer side: space for n entries (wave-uniform; false = the ring is full), and the publication of a finished
    // reservation — in reservation order, so that everything below `tail` is written.
    __device__ __forceinline__ bool ring_reserve(bool any_ring, uint32_t n, uint32_t lane, uint32_t& pos) {
        uint32_t* alloc = any_ring ? &M->aq_alloc : &M->cq_alloc;
        const uint32_t* head = any_ring ? &M->aq_head : &M->cq_head;
        const uint32_t cap = any_ring ? RT_MEGA_AQ_CAP : RT_MEGA_CQ_CAP;
        uint32_t p = 0, ok = 1;
        if (lane == 0) {
            p = atomicAdd(alloc, n);
            if (p + n - lds_read(head) > cap) ok = 0;        // (the reservation stays: the render fails anyway)
        }
        pos = wave_bcast0(p);
        return wave_bcast0(ok) != 0u;
    }
    __device__ __forceinline__ void ring_commit(bool any_ring, uint32_t pos, uint32_t n, uint32_t lane) {
        uint32_t* tail = any_ring ? &M->aq_tail : &M->cq_tail;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         // entries (and the rays / slots they name) before the tail
        uint32_t spins = 0;
        while (lds_read(tail) != pos) {                                // an earlier reservation is still being written
            __builtin_amdgcn_s_sleep(1);
            if (++spins > RT_MEGA_COMMIT_SPINS || aborted()) { abort(); if (lane == 0) atomicOr(&ctl->overflow, 32u); return; }
        }
        if (lane == 0) lds_write(tail, pos + n);
    }
    __device__ __forceinline__ uint32_t first_blocks() const { return 0u; }   // (MixedIO interface; the MEGA refill does not use them)
    __device__ __forceinline__ uint32_t first_count() const { return 0u; }

    __device__ __forceinline__ int load(uint32_t it, V3& o, V3& d, float& time, float& minT, bool& has, HitOut& h, bool& any) {
        if (it & RT_ITEM_ANY) { any = true; last_primary = false; return s.load(it & ~RT_ITEM_ANY, o, d, time, minT, has, h); }
        any = false;
        last_primary = it < n_primary;
        const int got = c.load(it, o, d, time, minT, has, h);
        if (got == 0)   // padding record of an edge tile: final at once (what k_shade writes for it)
            c.P.link[it] = make_uint4(RT_NO_CHILD, RT_NO_CHILD, 0u, RF_MISS | RF_DEAD);
        return got;
    }
    // items that needed no traversal are final (wave-uniform call)
    __device__ __forceinline__ void loaded(bool mine, int got, uint32_t lane) {
        uncount((uint32_t)__popcll(__ballot(mine && got != 2)), lane);
        n_prim_traced += __popcll(__ballot(mine && got != 0 && last_primary));
    }

    __device__ __forceinline__ void store(uint32_t it, bool verdict, const HitOut& h, float minT) {
        if (it & RT_ITEM_ANY) {
            s.store(it & ~RT_ITEM_ANY, verdict, h, minT);
            n_final = 1;
        } else {
            c.store(it, verdict, h, minT);
            if (verdict) {
                push_rec = it;
            } else {   // a miss is final: ComputeColor returns the background or black (raytracer.cpp:393-400)
                const uint32_t depth = (__float_as_uint(c.P.ray1[it].w) >> 8) & 31u;
                c.P.link[it] = make_uint4(RT_NO_CHILD, RT_NO_CHILD, 0u, RF_MISS | (depth << RF_DEPTH_SHIFT));
                n_final = 1;
            }
        }
    }
    // after the store block of a pass (wave-uniform call): hits are noted for shading, final items leave the count
    __device__ __forceinline__ void flush(uint32_t lane, uint64_t lanes_below) {
        const uint64_t pm = __ballot(push_rec != 0xFFFFFFFFu);
        if (pm != 0) {
            uint32_t t = 0;
            if (lane == 0) t = lds_read(&M->sq_tail[wave]);
            t = wave_bcast0(t);
            if (push_rec != 0xFFFFFFFFu) sq[(t + (uint32_t)__popcll(pm & lanes_below)) & (RT_MEGA_SQ_CAP - 1u)] = push_rec;
            // the records' hit words and the ring entries are written before the shading wave can see the new tail
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) lds_write(&M->sq_tail[wave], t + (uint32_t)__popcll(pm));
            push_rec = 0xFFFFFFFFu;
        }
        const uint32_t nf = (uint32_t)__popcll(__ballot(n_final != 0));
        if (nf) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uncount(nf, lane);
            n_final = 0;
        }
    }

    // The shading wave: n records, one per lane (k_shade's body), and what they spawn appended to the workgroup's rings.
    __device__ __forceinline__ void shade_batch(const DScene& S, uint32_t lane, uint64_t lanes_below, uint32_t n, uint32_t r, bool valid) {
        const Pool& P = c.P;
        const uint32_t n_lights = S.n_point_lights;
        uint32_t n_child = 0, n_shadow = 0, flags = RF_MISS, depth = 0;
        ShadeCtx x;
        float time = 0.f;
        x.type = RT_MAT_NONE;
        x.tir = false;
        x.entering = true;
        if (valid) {
            const float4 r0 = P.ray0[r], r1 = P.ray1[r];
            const uint4 h0 = P.hit0[r];
            const float2 h1 = P.hit1[r];
            time = r0.w;
            depth = (__float_as_uint(r1.w) >> 8) & 31u;
            V3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z);
            const float tHit = __uint_as_float(h0.x);
            V3 hit_x = o + d * tHit;
            int material_id = 1;
            V3 nrm = hit_normal(S, h0.y, h0.z, h1.x, h1.y, hit_x, material_id);
            P.nrm[r] = make_float4(nrm.x, nrm.y, nrm.z, __int_as_float(material_id));
            const DMaterial& mat = S.materials[material_id - 1];
            shade_ctx(S, mat, o, d, tHit, nrm, x);
            flags = 0;
            const bool can_recurse = (int)depth < S.max_depth;    // child depth = depth + 1 <= max (raytracer.cpp:384)
            if (x.type == RT_MAT_MIRROR || x.type == RT_MAT_CONDUCTOR) n_child = can_recurse ? 1 : 0;
            else if (x.type == RT_MAT_DIELECTRIC) n_child = can_recurse ? (x.tir ? 1 : 2) : 0;
            if (!(x.type == RT_MAT_DIELECTRIC && !x.entering)) {   // raytracer.cpp:1364-1366
                flags |= RF_LIGHTS;
                n_shadow = n_lights;
            }
        }
        // ---- records for the children: sub-allocated from this wave's slab
        uint32_t cincl = n_child;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(cincl, off, 64);
            if ((int)lane >= off) cincl += v;
        }
        const uint32_t n_children = __shfl(cincl, 63, 64);
        bool pool_full = false;
        if (n_children > rec_end - rec_next) {
            // the rest of the old slab is never used: say so (k_combine_mega walks every record slot)
            for (uint32_t k = rec_next + lane; k < rec_end; k += 64) P.link[k] = make_uint4(RT_NO_CHILD, RT_NO_CHILD, 0u, RF_MISS | RF_DEAD | (RF_DEPTH_NONE << RF_DEPTH_SHIFT));
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ctl->n_records, RT_MEGA_REC_SLAB);
            base = wave_bcast0(base);
            if (base + RT_MEGA_REC_SLAB > P.cap_records) { pool_full = true; rec_next = rec_end = 0; if (lane == 0) atomicOr(&ctl->overflow, 1u); }
            else { rec_next = base; rec_end = base + RT_MEGA_REC_SLAB; }
        }
        const uint32_t child0 = rec_next + cincl - n_child;
        // ---- shadow slots, light-major inside the batch: slot of (k-th lit record of K, light l) = base + l * K + k
        const uint64_t lit = __ballot(n_shadow != 0);
        const uint32_t K = (uint32_t)__popcll(lit), n_slots = K * n_lights;
        if (n_slots > sh_end - sh_next) {
            uint32_t base = 0;
            const uint32_t want = max(max(RT_MEGA_SH_SLAB, 1024u * n_lights), n_slots);   // (so a slab has an unused end of at most a sixteenth)
            if (lane == 0) base = atomicAdd(&ctl->n_shadow, want);
            base = wave_bcast0(base);
            if (base + want > P.cap_shadow) { pool_full = true; sh_next = sh_end = 0; if (lane == 0) atomicOr(&ctl->overflow, 2u); }
            else { sh_next = base; sh_end = base + want; }
        }
        const uint32_t sbase0 = sh_next, sbase = sh_next + (uint32_t)__popcll(lit & lanes_below);
        // ---- room on the rings
        uint32_t cpos = 0, apos = 0;
        bool ring_full = false;
        if (!pool_full && n_children && !ring_reserve(false, n_children, lane, cpos)) ring_full = true;
        if (!pool_full && !ring_full && n_slots && !ring_reserve(true, n_slots, lane, apos)) ring_full = true;
        // (a failed reservation leaves `alloc` advanced and never commits, so the next commit on that ring would wait for a
        // tail that cannot move: stop the workgroup here — the render fails with RT_ERR_OVERFLOW, not with a spin bound)
        if (ring_full) { if (lane == 0) atomicOr(&ctl->overflow, 64u); abort(); }
        const bool produce = !pool_full && !ring_full;
        if (produce) { rec_next += n_children; sh_next += n_slots; }
        if (valid) {
            uint32_t c0 = RT_NO_CHILD, c1 = RT_NO_CHILD;
            const uint32_t child_live = 1u | ((depth + 1u) << 8);
            if (produce && n_child) {
                c0 = child0;
                P.ray0[c0] = make_float4(x.refl_origin.x, x.refl_origin.y, x.refl_origin.z, time);
                P.ray1[c0] = make_float4(x.wr.x, x.wr.y, x.wr.z, __uint_as_float(child_live));
                cq[(cpos + (cincl - n_child)) & (RT_MEGA_CQ_CAP - 1u)] = c0;
                if (n_child == 2) {
                    c1 = child0 + 1;
                    P.ray0[c1] = make_float4(x.refr_origin.x, x.refr_origin.y, x.refr_origin.z, time);
                    P.ray1[c1] = make_float4(x.wt.x, x.wt.y, x.wt.z, __uint_as_float(child_live));
                    cq[(cpos + (cincl - n_child) + 1u) & (RT_MEGA_CQ_CAP - 1u)] = c1;
                }
            }
            if (n_shadow) {
                if (produce) {
                    for (uint32_t l = 0; l < n_shadow; l++) {
                        const uint32_t slot = sbase + l * K;
                        P.shadow_rec[slot] = r;
                        aq[(apos + (slot - sbase0)) & (RT_MEGA_AQ_CAP - 1u)] = slot;
                    }
                    flags |= K << RF_STRIDE_SHIFT;
                } else {
                    flags &= ~RF_LIGHTS;                          // the chunk is redone (pool) or the render fails (ring)
                }
            }
            P.link[r] = make_uint4(c0, c1, sbase, flags | (depth << RF_DEPTH_SHIFT));
        }
        // ---- the n shaded rays are finished, their children and shadow tests are new unfinished items: counted BEFORE
        // they can be taken, so the count never runs low
        const uint32_t n_items = produce ? n_children + n_slots : 0u;
        if (lane == 0) {
            if (n_items >= n) atomicAdd(&M->work, n_items - n); else atomicSub(&M->work, n - n_items);
        }
        if (produce && n_children) ring_commit(false, cpos, n_children, lane);
        if (produce && n_slots) ring_commit(true, apos, n_slots, lane);
    }
    // the slots left in this wave's last record slab are never used
    __device__ __forceinline__ void finish(uint32_t lane) {
        for (uint32_t k = rec_next + lane; k < rec_end; k += 64)
            c.P.link[k] = make_uint4(RT_NO_CHILD, RT_NO_CHILD, 0u, RF_MISS | RF_DEAD | (RF_DEPTH_NONE << RF_DEPTH_SHIFT));
    }
};

// The shading wave (wave 15).  Returns when the workgroup is finished.
__device__ void mega_shading_wave(const DScene& S, MegaIO& io, const uint32_t* blk, uint32_t* sq_all, uint32_t parts, uint32_t lane, uint32_t shader) {
    MegaLds& M = *io.M;
    uint32_t waited = 0, spins = 0;
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    for (;;) {
        if (lds_read(&M.abort) != 0u) break;
        // ---- what the tracing waves have handed over: up to 64 records, ring by ring
        uint32_t avail = 0, head = 0;
        if (lane < RT_MEGA_TRACERS && lane % RT_MEGA_SHADERS == shader) { head = lds_read(&M.sq_head[lane]); avail = lds_read(&M.sq_tail[lane]) - head; }
        uint32_t incl = avail;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += v;
        }
        const uint32_t total = __shfl(incl, RT_MEGA_TRACERS - 1, 64);
        if (total == 0) {
            if (lds_read(&blk[4]) >= parts && lds_read(&M.work) == 0u) break;     // primaries gone, nothing unfinished
            __builtin_amdgcn_s_sleep(8);
            if (++spins > RT_MEGA_IDLE_CAP) { if (lane == 0) atomicOr(&io.ctl->overflow, 32u); lds_write(&M.abort, 1u); break; }
            continue;
        }
        spins = 0;
        // A full batch shades 64 records for the price of one, and its shadow slots come out as runs of 64 towards one
        // light from neighbouring pixels.  So wait for one — unless the tracing waves are short of work (primaries gone,
        // or both rings empty): then whatever has finished is shaded at once, its children are what they wait for.
        if (total < 64u && waited < RT_MEGA_BATCH_WAIT && lds_read(&blk[4]) < parts && !io.ring_empty()) {
            waited++;
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        waited = 0;
        const uint32_t n = min(total, 64u);
        // ring w covers the batch positions [excl_w, incl_w); take from each what falls below n
        const uint32_t excl = incl - avail;
        const uint32_t taken = lane < RT_MEGA_TRACERS ? (excl >= n ? 0u : min(avail, n - excl)) : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t r = 0;
        bool valid = false;
#pragma unroll
        for (int w = 0; w < RT_MEGA_TRACERS; w++) {
            const uint32_t e_w = __shfl(excl, w, 64), t_w = __shfl(taken, w, 64), h_w = __shfl(head, w, 64);
            if (lane >= e_w && lane < e_w + t_w) {
                r = sq_all[(size_t)w * RT_MEGA_SQ_CAP + ((h_w + (lane - e_w)) & (RT_MEGA_SQ_CAP - 1u))];
                valid = true;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // entries read before their ring slots are given back
        if (lane < RT_MEGA_TRACERS && taken) lds_write(&M.sq_head[lane], head + taken);
        io.shade_batch(S, lane, lanes_below, n, r, valid);
    }
    io.finish(lane);
}

template <bool NEAR>
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_MIN_WAVES) k_trace_mega(DScene S, Pool P, Control* ctl, MegaArgs A, uint2* spill, uint4* cand) {
    __shared__ TraceLds<RT_TRACE_BLOCK> s_lds;
    __shared__ MegaLds s_mega;
    WaveWork ww;
    if (threadIdx.x < sizeof(MegaLds) / 4) reinterpret_cast<uint32_t*>(&s_mega)[threadIdx.x] = 0;
    TraceStack st = make_stack(s_lds, spill, cand, ww, S.stack_cap);      // (its __syncthreads covers s_mega as well)
    const uint32_t n_blocks = (A.n_primary + 63u) / 64u;
    const uint32_t parts = n_blocks >= RT_PARTS_MIN_BLOCKS ? (uint32_t)RT_PARTS : 1u;   // as trace_persistent computes it
    uint32_t* cq = A.cq + (size_t)blockIdx.x * RT_MEGA_CQ_CAP;
    uint32_t* aq = A.aq + (size_t)blockIdx.x * RT_MEGA_AQ_CAP;
    uint32_t* sq_all = A.sq + (size_t)blockIdx.x * RT_MEGA_TRACERS * RT_MEGA_SQ_CAP;
    MegaIO io{ClosestIO{S, P}, ShadowIO{S, P}, ctl, &s_mega, cq, aq, sq_all + (size_t)(ww.wave % RT_MEGA_TRACERS) * RT_MEGA_SQ_CAP, ww.wave,
              ww.wave >= RT_MEGA_TRACERS - A.any_waves, A.strict != 0, A.n_primary, 0u, 0u, 0u, 0u, 0ull, 0xFFFFFFFFu, 0u, false};
    if (ww.wave >= RT_MEGA_TRACERS) {
        mega_shading_wave(S, io, s_lds.blk, sq_all, parts, ww.lane, ww.wave - RT_MEGA_TRACERS);
        return;
    }
#ifdef RT_TRACE_STATS
    TraceStatsAcc acc = {};
    st.acc = &acc;
    acc.clk_start = wall_clock64();
#endif
    unsigned long long traced = 0, traced_any = 0;   // wave-uniform
    trace_persistent<2, true, NEAR, MegaIO, true>(S, io, 0u, A.n_primary, ctl->cursor_closest, st, traced, &traced_any);
#ifdef RT_TRACE_STATS
    report_trace_stats(ctl, acc, traced, acc.clk_start, st, ww);
#endif
    if (st.overflow) atomicOr(&ctl->overflow, (st.overflow & 48u) | ((st.overflow & ~48u) ? 4u : 0u));
    if (ww.lane == 0) {
        if (traced_any) atomicAdd(&ctl->rays_shadow, traced_any);
        if (io.n_prim_traced) atomicAdd(&ctl->rays_primary, io.n_prim_traced);
        if (traced - traced_any - io.n_prim_traced) atomicAdd(&ctl->rays_secondary, traced - traced_any - io.n_prim_traced);
    }
}

// k_combine for the one-launch pipeline: the records of depth `gen` are wherever the waves put them, so every
// record slot [0, n_records) is looked at and the ones of other depths (and never-used slab slots) are skipped.
__global__ void __launch_bounds__(256) k_combine_mega(DScene S, Pool P, Control* ctl, int gen, uint32_t n_primary) {
    // A launch that overflowed a pool or a ring, or was aborted, leaves record slots that nobody wrote (the tail of a slab
    // reservation that crossed the cap, children that were allocated and never traced): their link / normal / colour
    // words are whatever the buffers held, and trusting them means reading materials[garbage].  The chunk is redone or
    // the render fails (render_impl), so nothing is combined.
    if (ctl->overflow != 0u) return;
    const uint32_t end = min(ctl->n_records, P.cap_records);
    const uint32_t begin = gen == 0 ? 0u : n_primary, stop = gen == 0 ? n_primary : end;     // depth 0 = the primaries, nothing else
    for (uint32_t r = begin + blockIdx.x * blockDim.x + threadIdx.x; r < stop; r += gridDim.x * blockDim.x) {
        const uint4 lk = P.link[r];
        if (((lk.w >> RF_DEPTH_SHIFT) & 31u) != (uint32_t)gen) continue;
        combine_record(S, P, r, lk, gen);
    }
}
