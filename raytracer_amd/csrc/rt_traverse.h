// rt_traverse.h — wave64 two-level BVH traversal (closest-hit and any-hit).
//
// Restates, with the reference's exact visiting order and culling rules,
//   IntersectTopBVH   raytracer.cpp:527-756     (closest hit, mode 0)
//   InShadow's loop   raytracer.cpp:1030-1243   (first accepted hit, mode 1; mode 2: per lane either of the two)
//   IntersectMeshBVH  raytracer.cpp:758-811
//   IntersectAABB     raytracer.cpp:815-894
//   IntersectsTriangle_Bary raytracer.cpp:896-947, IntersectSphere :949-980
//
// Differences in FORM, none in result:
//   * one node record holds the boxes of the four GRANDCHILDREN of a reference
//     interior node (two levels collapsed), in the reference's depth-first
//     order, and all four are tested when the record is visited.  The first hit
//     slot is what the reference reaches next with the same minT.  Every other
//     hit slot is reached by the reference only after the subtrees before it;
//     its box entry distance tEnter is kept on the stack and re-checked against
//     the then-current minT at pop time.  Since IntersectAABB's verdict is
//     "min(minT, exit) < enter  or  (enter == 0 and min(minT, exit) <= 0)" and
//     the box was a hit at push time, the pop-time verdict is exactly
//     "minT < tEnter or (tEnter == 0 and minT <= 0)".  The intermediate node
//     between a record and its slots needs no test of its own: its box contains
//     the slot's box, every slab operation is monotone under rounding, and the
//     reference tests parent and child with the same minT (no primitive test in
//     between), so "child box hit" implies "parent box hit" bit for bit; and a
//     parent whose children all miss has no side effect in the reference.
//   * mesh leaves are POSTPONED: a lane that reaches a leaf records (leaf, tEnter)
//     in a small per-lane queue and keeps traversing; when its queue is full or
//     its traversal is exhausted it waits, and once every lane of the wave waits
//     the queued leaves are tested, in order, each first re-checked against the
//     current minT with the pop-time rule above.  This is exact, not approximate:
//     a leaf L is tested by the reference iff L's own box passes IntersectAABB
//     against the minT left by the leaves before it (every ancestor's box
//     contains L's box and is tested no later with a minT no smaller, so by
//     monotonicity it passes whenever L's does), and traversing with a stale,
//     larger minT can only add candidates, which that re-check removes again.
//     It lifts SIMD utilisation of the node loop from 21 % to ~55 %.
//   * the box test is branch-free for rays whose three 1/d are finite (no slab product can be a NaN then; a
//     node record holds the four slots' planes by axis and a lane loads the planes its ray meets first / last,
//     so the slab distances come out ordered without the reference's compare-and-swap); rays with an infinite
//     1/d (d == 0, or 1/d overflowing) take the reference's branches literally — rt_box.h, which is also
//     compiled for the host and compared with the oracle (tests/box_logic.cpp).
//   * both BVH levels run in ONE loop (a lane switches its active ray to object
//     space and back), so lanes at different levels share the node-fetch /
//     box-test instructions instead of serialising nested loops.
//   * 1/dir is computed once per active ray (the reference recomputes the same
//     quotient at every box, raytracer.cpp:822,842,862).
//   * scheduling (trace_persistent below): ray slots refilled from a strip-interleaved
//     cursor, subtrees handed to idle lanes of the workgroup over an LDS task board with
//     an exact replay of their candidate hits, big leaves tested by the whole wave.
//     None of it changes which leaves are tested against which minT, in which order.
// Non-finite geometry is out of contract: a NaN `t` is accepted by the reference's
// comparisons (raytracer.cpp:941) and would then win every later comparison; the
// whole-wave leaf test and the candidate replay order hits by `t` and would not
// reproduce that.
#pragma once
#include "rt_device.h"

#ifndef RT_STACK_LDS
#define RT_STACK_LDS 10          // entries per lane kept in LDS
#endif
#ifndef RT_LEAF_QUEUE
#define RT_LEAF_QUEUE 8          // mesh leaves a lane may postpone before the wave switches to triangle tests
#endif
#ifndef RT_LEAF_QUEUE_NEAR
#define RT_LEAF_QUEUE_NEAR RT_LEAF_QUEUE   // closest-hit rays in the near-first kernels
#endif
#ifndef RT_LEAF_QUEUE_ANY
#define RT_LEAF_QUEUE_ANY 4      // same for any-hit (shadow) rays: the first accepted hit ends the ray, so postpone less
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4           // __launch_bounds__ waves per SIMD for the trace kernels
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 24         // idle lanes of a wave that trigger a refill from the generation's cursor (16 until the transitions got cheaper)
#endif
#ifndef RT_STRIP
#define RT_STRIP 8               // closest-hit: consecutive items of a 64-item block handed out together (divides 64)
#define RT_STRIP_LOG2 3
#endif
#ifndef RT_STRIP_WIDE_LOG2
#define RT_STRIP_WIDE_LOG2 5     // ... 32 of them in launches of at least RT_WIDE_STRIP_BLOCKS blocks
#endif
#ifndef RT_WIDE_STRIP_BLOCKS
#define RT_WIDE_STRIP_BLOCKS 100000
#endif
#ifndef RT_TRI_UNROLL
#define RT_TRI_UNROLL 2          // triangles of a queued leaf a lane loads together
#endif
#ifndef RT_PARTS
#define RT_PARTS 8               // cursor partitions of a generation (XCDs of the chip)
#endif
#ifndef RT_PARTS_MIN_BLOCKS
#define RT_PARTS_MIN_BLOCKS 4096 // 64-item blocks below which a generation keeps one cursor
#endif
#define RT_CURSOR_STRIDE 32u     // words between the partitions' cursors (a 128-byte line each)
#ifndef RT_STRIP_ANY
#define RT_STRIP_ANY 64          // any-hit: 64 consecutive shadow slots are rays of neighbouring pixels towards one light (k_shade): keep blocks whole (8 / 16 / 32: 20 / 11 / 4 % slower)
#endif
// node phase exit: at most EXIT_FEW lanes still walking and at least EXIT_WAIT lanes waiting, either with queued leaves or
// (RT_WAIT_COUNTS_TRANSITIONS) for the round's transition: a finished ray to be retired, an instance to be entered or left.
// An owner parked until its helpers report is not waiting for the wave and is not counted, or the node phase would end
// before the helpers in the same wave took a step. 24/8 without transitions -> 16/16 with: mesh1m +3.5 %, inst10 +4 %
// (profiles/r02/l_exit_rule_sweep.txt)
#ifndef RT_EXIT_FEW
#define RT_EXIT_FEW 16
#endif
#ifndef RT_EXIT_WAIT
#define RT_EXIT_WAIT 16
#endif
#ifndef RT_EXIT_FEW_ANY
#define RT_EXIT_FEW_ANY 20
#endif
#ifndef RT_EXIT_WAIT_ANY
#define RT_EXIT_WAIT_ANY 1
#endif
#ifndef RT_REFILL_MIN_ANY
#define RT_REFILL_MIN_ANY 48
#endif
#ifndef RT_DONATE_PASSES
#define RT_DONATE_PASSES 4       // stack entries a busy lane may post per round (helpers included, so a subtree fans out)
#endif
#ifndef RT_WAIT_COUNTS_TRANSITIONS
#define RT_WAIT_COUNTS_TRANSITIONS 1
#endif
#ifndef RT_TAIL_STEPS
#define RT_TAIL_STEPS 16         // node steps per round once the cursor is drained: the wave then looks at the task board that often
#endif
#ifndef RT_REFILL_PATIENCE
#define RT_REFILL_PATIENCE 2     // rounds a wave waits for RT_REFILL_MIN idle lanes before it refills the ones it has
#endif
#ifndef RT_COOP_MIN
#define RT_COOP_MIN 32           // a leaf with at least this many triangles can be tested by the whole wave for one ray
#endif
#ifndef RT_EARLY_FINISH
#define RT_EARLY_FINISH 1        // a ray whose stack runs empty in the transition pass is retired in that pass
#endif
#ifndef RT_NODE_RUN
#define RT_NODE_RUN 1            // top-level walk: the nodes up to the next prim are taken in a loop of their own (0: one item per turn, node or prim)
#endif
#ifndef RT_FLAT_TOP
#define RT_FLAT_TOP 1            // the top level is walked in the transition block, item by item (rt_device.h: RT_TOPITEM_*); 0: as 4-wide records in the node phase
#endif
#ifndef RT_OB_BATCH
#define RT_OB_BATCH 2            // the fields the transitions read of an object come in one batch of loads: 1 entering, 2 also leaving (0: field by field)
#endif
#ifndef RT_MIXED_PARTS
#define RT_MIXED_PARTS 1         // the mixed launches' cursor is partitioned like the others' (0: one cursor)
#endif
#ifndef RT_STATIC_FIRST
#define RT_STATIC_FIRST 1        // the first batch of every wave is its own by position, no atomic (0: everything through the cursors)
#endif
#ifndef RT_REFILL_PASSES
#define RT_REFILL_PASSES 3       // refills per round: rays that end in their first transition (they miss every object box) leave lanes that
                                 // are refilled at once, and the transition section stays a loop (with 1 the compiler made two copies of it):
                                 // depth-first pixel loop +6 %, inst10 8K +1.6 %, the 1080p frames within the spread (profiles/r03/h_*)
#endif
#ifndef RT_CAND_MAX
#define RT_CAND_MAX 16           // candidate hits a ray can collect from donated subtrees before it falls back to tracing alone
#endif
// Stack entries per ray (LDS part + global spill part): at least the reference's stack[64] (raytracer.cpp:533,766),
// more when the scene's BVH depths ask for it — one stack holds the top-level entries, a sentinel and the mesh
// level, and a node record leaves up to three siblings per two reference levels, so a BVH the reference walks
// with depth + 1 <= 64 entries can need up to ~200 here.  DScene::stack_cap, computed at upload (rt_hip.hip).
#define RT_STACK_MIN_CAP 64
#define RT_STACK_CAP_LIMIT 4096
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 1024      // threads per workgroup of the trace kernels: one workgroup per CU, its 16 waves share the task board
#endif
#define RT_ORDERED_BLOCK 256     // rt_ordered.h: 256 VGPRs per lane, 2 waves per SIMD
#define RT_BOARD_PER_WAVE 4      // task-board slots per wave of a workgroup (16 words each)
#ifndef RT_ROUND_CAP
#define RT_ROUND_CAP (1u << 22)   // WORKING rounds of one wave in one trace call (rounds a finished wave spends lingering do not
                                  // count); a 64 M-sample chunk needs ~2e5, so a stuck wave reports within seconds, not minutes
#endif
#define RT_MEGA_IDLE_CAP (1u << 27)    // rounds a wave of the one-launch pipeline may wait without a ray (~0.5 us each)
#define RT_WAIT_ROUNDS_MAX 1000000u   // an owner never waits longer for its helpers (error 16 instead of a hang)
#ifndef RT_LINGER_SPINS
#define RT_LINGER_SPINS 20000    // rounds a finished wave keeps looking at the board while others of its workgroup work
#endif

// Diagnostic build only (-DRT_TRACE_STATS): per-wave step counts to measure SIMD
// utilisation of the traversal loop.  Never compiled into the shipped library.
#ifdef RT_TRACE_STATS
struct TraceStatsAcc {
    unsigned long long wave_node_steps, lane_node_steps, wave_tri_steps, lane_tri_steps, wave_outer, lane_rays, wave_batches, wave_pops, lane_pops;
    unsigned int ray_leaves, ray_nodes;
    unsigned long long ts_a, ts_b, ts_c, ts_d, ts_post_iters;   // tail: parts of the scheduling section
    unsigned long long tl_rounds, tl_sched, tl_node, tl_tri, tl_big, tl_trans;   // after the cursor was drained
    unsigned long long clk_start, r_node_w, r_node_l, r_tri_w, r_tri_l;   // activity histogram (flushed every round)
    unsigned long long t_sched, t_node, t_tri, t_big, t_trans, clk_exhausted, n_donations, n_coop, n_replays, n_elig, n_gate, n_casfail, n_claim, n_culled;
    unsigned long long x_block, x_retire, x_refill, x_passes;   // parts of t_trans
    unsigned long long y_fin, y_prims, y_pop, y_witers, y_liters, y_wbody, y_lbody;   // inside the transition block (kept by the first active lane)
};
#define RT_CLK(v) const unsigned long long v = wall_clock64()
#define RT_STAT(x) x
#ifdef RT_STATS_SHADOW
#define RT_HIST_ANY true
#else
#define RT_HIST_ANY false
#endif
#ifdef RT_RAY_HIST                 // per-ray work histogram (its same-address atomics distort every timing)
#define RT_RAYHIST(x) x
#else
#define RT_RAYHIST(x)
#endif
#define RT_HIST_BINS 128
#define RT_HIST_TICKS 2500ull     // 25 us per bin (wall_clock64 ticks are 10 ns)
__device__ unsigned long long g_hist[RT_HIST_BINS][4];
__device__ unsigned long long g_rayhist[2][20];   // [0]: rays by log2(leaves visited), [1]: by log2(node steps); owner lanes only
#else
#define RT_STAT(x)
#define RT_RAYHIST(x)
#define RT_CLK(v)
#endif

// Diagnostic build only (-DRT_TIMELINE=mode): chip-wide activity of the trace kernel of that MODE over time, in 10-us
// bins: rounds, ray slots in use, node steps and their active lanes, triangle steps and theirs.  A few atomics per
// round: light enough to leave the timing alone (unlike RT_TRACE_STATS).
#ifdef RT_TIMELINE
#define RT_TL_BINS 256
#define RT_TL_WAVES 4096
__device__ uint32_t g_tl[RT_TL_WAVES][RT_TL_BINS][8];   // a row per wave: plain read-modify-writes by its lane 0, summed on the host
__device__ unsigned long long g_tl_t0 = ~0ull;
// rays (owners) that finish after the cursor was drained: own node steps, leaves queued, lifetime, subtrees given away, rounds spent waiting for helpers
#define RT_TL_LOG 262144
__device__ uint32_t g_tl_log[RT_TL_LOG][8];
__device__ uint32_t g_tl_log_n;
__device__ unsigned long long g_tl_tail[8];   // after the drain, summed over waves with rays: ticks in sched / node / tri / big / transitions, rounds
__device__ unsigned long long g_tl_tail2[8];  // the same rounds: sched = claim / announce / post / rest; transitions = block / retire; replays, candidates
#define RT_TLC(v) const unsigned long long v = (unsigned long long)wall_clock64()
#define RT_TL(...) __VA_ARGS__
#else
#define RT_TL(...)
#define RT_TLC(v)
#endif

// Wave-wide votes of the hot loops, straight from the compare's lane mask (HIP's __ballot / __any go through an integer
// per lane: a v_cndmask and a v_cmp more per vote).
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ uint32_t wave_count(bool p) { return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }

typedef float f4a __attribute__((ext_vector_type(4)));
typedef f4a f4u __attribute__((aligned(4)));   // 4 floats at dword alignment: one global_load_dwordx4

struct TraceStack {
    uint2*    lds;        // [RT_STACK_LDS][64] for this wave: (link, bits of tEnter)
    uint2*    spill;      // [(cap-RT_STACK_LDS)][64] for this wave (global)
    uint2*    q;          // [RT_LEAF_QUEUE][64] postponed mesh leaves, in visiting order: (link, bits of tEnter)
    // workgroup-wide (LDS unless noted), indexed by the owner's slot = wave * 64 + lane:
    uint32_t* cand_cnt;   // [threads] candidates collected for the ray of that slot
    uint32_t* out_cnt;    // [threads] subtrees of that ray that other lanes still trace
    uint8_t*  cancel;     // [threads] any-hit found for that ray by a helper
    uint4*    cand;       // [threads][RT_CAND_MAX] (global): (triangle slot, t bits, leaf tEnter bits, -)
    uint32_t* board;      // [n_board][16] task board: state, home, link, tEnter, bound, ray (9), object, kind
    uint32_t* blk;        // [8] busy waves, ready tasks, idle lanes, post cursor, cursor drained
    uint32_t  n_board;
    uint32_t  slot;       // this lane's slot in the workgroup
    uint32_t  lane;
    uint32_t  sp;
    uint32_t  cap;        // DScene::stack_cap
    uint32_t  overflow;
#ifdef RT_TRACE_STATS
    TraceStatsAcc* acc;
#endif

    __device__ __forceinline__ void push(uint32_t link, float t) {
        if (sp < RT_STACK_LDS) {
            lds[sp * 64 + lane] = make_uint2(link, __float_as_uint(t));
        } else if (sp < cap) {
            spill[(sp - RT_STACK_LDS) * 64 + lane] = make_uint2(link, __float_as_uint(t));
        } else {
            overflow = 1;   // cannot happen for a scene that passed the upload (cap covers its BVH depths); reported, not ignored
            return;
        }
        sp++;
    }
    // push when the caller has established (wave-wide) that the entry lands in the LDS part
    __device__ __forceinline__ void push_lds(uint32_t link, float t) {
        lds[sp * 64 + lane] = make_uint2(link, __float_as_uint(t));
        sp++;
    }
    // the same without a branch: the entry is written above the top in any case (free space: the caller has established
    // that three more entries fit the LDS part) and the top only moves if `p` holds
    __device__ __forceinline__ void push_lds_if(bool p, uint32_t link, float t) {
        lds[sp * 64 + lane] = make_uint2(link, __float_as_uint(t));
        sp += p ? 1u : 0u;
    }
    __device__ __forceinline__ uint2 read_at(uint32_t pos) const {
        return pos < RT_STACK_LDS ? lds[pos * 64 + lane] : spill[(pos - RT_STACK_LDS) * 64 + lane];
    }
    __device__ __forceinline__ void write_at(uint32_t pos, uint2 e) {
        if (pos < RT_STACK_LDS) lds[pos * 64 + lane] = e; else spill[(pos - RT_STACK_LDS) * 64 + lane] = e;
    }
    // Pop the next entry the reference would not cull (see file header).  Slots whose subtree
    // was given to another lane hold RT_LINK_NONE and are skipped.
    __device__ __forceinline__ uint32_t pop(float tmax, float& t_out) {
        while (sp > 0) {
            --sp;
            uint2 e;
            if (sp < RT_STACK_LDS) {
                e = lds[sp * 64 + lane];
            } else {
                e = spill[(sp - RT_STACK_LDS) * 64 + lane];
                asm volatile("" : "+v"(e.x));   // keep the rare global path a branch (else both merge into one flat load)
            }
            const float t = __uint_as_float(e.y);
            if (e.x != RT_LINK_NONE && !((tmax < t) || (t == 0.0f && tmax <= 0.0f))) { t_out = t; return e.x; }
        }
        return RT_LINK_DONE;
    }
    // the same when the caller has established (wave-wide) that no lane's stack reaches into the global part
    __device__ __forceinline__ uint32_t pop_lds(float tmax, float& t_out) {
        while (sp > 0) {
            --sp;
            const uint2 e = lds[sp * 64 + lane];
            const float t = __uint_as_float(e.y);
            if (e.x != RT_LINK_NONE && !((tmax < t) || (t == 0.0f && tmax <= 0.0f))) { t_out = t; return e.x; }
        }
        return RT_LINK_DONE;
    }
};

#include "rt_box.h"   // ActiveRay, box_test, node_slabs4, tri_test, sphere_test, plane_test

struct HitOut {
    float    t;          // minT (world; see raytracer.cpp:605-609 for what "world t" means)
    uint32_t kind_index; // kind<<28 | object index, 0xFFFFFFFF = miss
    uint32_t tri;        // mesh: global DTri slot of the hit triangle
    float    beta, gamma;
};

// (object_space_ray and world_t, the per-object ray set-up of raytracer.cpp:559-606, are in rt_box.h: host-testable)

__device__ __forceinline__ uint32_t leaf_count(const DScene& S, uint32_t link, bool top) {
    uint32_t c = RT_LEAF_COUNT(link);
    if (c != RT_LEAF_BIG) return c;
    // unsplittable node with more than 30 prims (raytracer.cpp:183,317): binary search the side table
    const uint32_t key = RT_LEAF_FIRST(link) | (top ? 0x80000000u : 0u);
    uint32_t lo = 0, hi = S.n_big_leaves;
    while (lo + 1 < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (S.big_leaves[mid].first <= key) lo = mid; else hi = mid;
    }
    return S.big_leaves[lo].count;
}

// Is `link` a mesh leaf whose primitive count is in the side table (>= RT_LEAF_BIG prims)?
// Leaf bit + count code 31 = the top six bits; the three special links share them.
__device__ __forceinline__ bool is_big_leaf_link(uint32_t link) { return link >= 0xFC000000u && link < RT_LINK_NONE; }

__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

// The persistent trace loop of one wave: FindClosestHit's BVH part (MODE 0), InShadow's (MODE 1), or both
// at once with the kind given per item (MODE 2: k_trace_mixed, k_pixel_ordered), for the items [begin, end).
//
// Scheduling.  The 64 lanes are 64 ray slots.
//  * Refill: a lane whose ray is finished stores its result and goes idle; when
//    RT_REFILL_MIN lanes are idle the wave takes that many consecutive items from the
//    generation's cursor (one atomic) and hands them to the idle lanes.  So a wave is
//    not bound to the slowest ray of a fixed 64-ray batch, while refilling in runs
//    keeps neighbouring pixels together.
//  * Subtree donation: once the cursor is drained (a wave that sees it tells its
//    workgroup through LDS), a busy lane posts the bottom-most entries of its
//    mesh-level stack (the subtrees it would reach LAST) on the workgroup's task
//    board in LDS, and idle lanes of any wave of the workgroup take them; a wave
//    without rays stays while another wave of its workgroup is busy.  The OWNER of
//    the ray keeps the reference's exact sequential state for the part it traverses
//    itself, which precedes every posted subtree in visiting order.  A HELPER
//    traverses its subtree with a bound b that starts at the poster's bound and
//    RECORDS every triangle it would accept, (triangle, t, tEnter of its leaf),
//    lowering b only when t >= tEnter.  b is then never below the minT the reference
//    has at the same point (by induction over the visiting order: the reference
//    either accepts the same triangle, or rejects it because minT <= t, or culls the
//    leaf because minT < tEnter <= t), so every triangle the reference accepts inside
//    the subtree is among the recorded ones (box and triangle tests are monotone in
//    the bound).  When the owner's own part is done and its count of outstanding
//    subtrees is zero, it replays the candidates in visiting order (= ascending
//    triangle slot: the builder partitions the primitive array in place,
//    raytracer.cpp:247-262, and the left child is visited first) with the reference's
//    rule: skip if minT culls the leaf box, accept if t < minT.  That is the
//    reference's result exactly; what changes is that one long ray is traced by many
//    lanes.  A ray with more than RT_CAND_MAX candidates
//    re-traverses the mesh alone.  Bounded: a wave gives up after RT_ROUND_CAP rounds,
//    an owner after RT_WAIT_ROUNDS_MAX rounds of waiting (reported as errors).
//  * The node phase of a round ends when no lane can walk on — or when only a few
//    still walk while others wait with queued leaves (the walkers keep their place).
//
// IO::load(item, o, d, time, minT, has, out) -> int  sets up the ray (plane loop
//   included, raytracer.cpp:423-435 / :1022-1028) and returns 2, or stores a final
//   result itself and returns 1 (a ray that needs no traversal) or 0 (no ray);
//   IO::store(item, verdict, out, minT) stores the result of a traversed ray
//   (verdict = has / in-shadow).  MODE 2: load has one more argument, bool& any (set to the item's kind), and
//   the IO tells how many 64-item blocks of closest-hit items precede the any-hit items (first_blocks,
//   first_count): the cursor hands out the former strip-interleaved, then the latter block by block.
// MEGA (rt_mega.h: one launch for all generations of a chunk): MODE 2 with items that come from two sources — the
// primary rays [begin, end) through the cursor as above, and a work ring of the workgroup that its shading wave
// fills with the child rays and shadow slots of the rays this workgroup has finished (io.ring_*).  `exhausted` then
// only means "no primary ray is left"; a wave leaves when, in addition, the workgroup's count of unfinished items
// is zero (io.done()), and subtrees are donated while primaries are gone and the ring is empty.
template <int MODE, bool LINGER, bool NEAR, class IO, bool MEGA = false>
__device__ void trace_persistent(const DScene& S, IO& io, uint32_t begin, uint32_t end, uint32_t* cursor, TraceStack& st,
                                 unsigned long long& traced, unsigned long long* traced_any = nullptr) {
    static_assert(!MEGA || (MODE == 2 && LINGER), "the one-launch pipeline runs the mixed loop of a trace workgroup");
    // MODE 0: closest hit, 1: any-hit, 2: mixed — every lane's item says which it is (rt_ordered.h, where the
    // lanes of a wave ask for both kinds at once); `lane_any` is then a per-lane value, otherwise a constant
    constexpr bool MIXED = MODE == 2;
    constexpr bool ANY_ONLY = MODE == 1;
    bool lane_any = ANY_ONLY;
#define RT_ISANY (MIXED ? lane_any : ANY_ONLY)
    constexpr uint32_t QCAP = ANY_ONLY ? RT_LEAF_QUEUE_ANY : (NEAR ? RT_LEAF_QUEUE_NEAR : RT_LEAF_QUEUE);
    // (closest-hit items of a large launch go out in wider strips: with many blocks per wave the end of the launch balances
    // itself, and 32 neighbours in a wave instead of 8 share more of their traversal — inst10 8K -2.8 %, mesh1m 4K -0.5 %; on
    // a 1080p launch, 8 blocks per wave, the same strips cost 3 %)
    const uint32_t strip_shift_ = ANY_ONLY ? 6u : ((end - begin + 63u) / 64u >= (uint32_t)RT_WIDE_STRIP_BLOCKS ? (uint32_t)RT_STRIP_WIDE_LOG2 : (uint32_t)RT_STRIP_LOG2);
    const uint32_t STRIP = 1u << strip_shift_;
    constexpr uint32_t REFILL_MIN = ANY_ONLY ? RT_REFILL_MIN_ANY : RT_REFILL_MIN;
    constexpr uint32_t EXIT_FEW = ANY_ONLY ? RT_EXIT_FEW_ANY : RT_EXIT_FEW, EXIT_WAIT = ANY_ONLY ? RT_EXIT_WAIT_ANY : RT_EXIT_WAIT;
    static_assert(RT_STRIP_ANY == 64 && (1 << RT_STRIP_LOG2) == RT_STRIP && RT_STRIP_WIDE_LOG2 <= 6, "strips tile a 64-item block");
    static_assert(RT_LEAF_QUEUE_ANY <= RT_LEAF_QUEUE, "the LDS queue is sized by RT_LEAF_QUEUE");
    const uint32_t lane = st.lane;
    const uint64_t lanes_below = (1ull << lane) - 1ull;

    // ---- per-lane slot: a ray this lane owns, or a donated subtree of lane `home`'s ray
    bool alive = false;          // traversing (owner or helper)
    bool helper = false;
    uint32_t home = st.slot;     // workgroup slot (wave * 64 + lane) of the ray's owner
    uint32_t item = 0;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 0);
    float time = 0.f, minT = 0.f;
    bool has = false;
    HitOut out;
    out.t = 0.f; out.kind_index = 0xFFFFFFFFu; out.tri = 0; out.beta = 0.f; out.gamma = 0.f;
    ActiveRay ar;
    ar.set(ro, rd);
    float tmax = 0.f;            // culling bound of the active level
    bool in_mesh = false;
    uint32_t top_slot = 0;       // the ray's place in the top level: the next item (RT_FLAT_TOP), or the next slot of the leaf it is in
#if !RT_FLAT_TOP
    uint32_t top_remaining = 0;
#endif
    uint32_t cur_obj = 0;        // state of the mesh traversal in flight
    bool mesh_found = false;
    uint32_t mesh_tri = 0;
    float mesh_beta = 0.f, mesh_gamma = 0.f;
    uint32_t cur_kind_index = 0;
    uint32_t qn = 0;             // postponed leaves in the queue
    uint32_t cur = RT_LINK_DONE;
    float cur_t = 0.f;           // box entry distance of `cur`
    uint32_t don_next = 0;       // stack slot of the bottom-most mesh-level entry not yet given away
    uint32_t don_base = 0;       // stack slot above the sentinel of the mesh traversal in flight
    bool donated = false;        // owner: subtrees of the mesh traversal in flight were given away
    bool no_donate = false;      // owner: re-traversing alone after a candidate overflow
    bool exhausted = false;      // wave-uniform: the cursor has passed the last item
    bool was_busy = false;       // wave-uniform: this wave is counted in st.blk[0]
    uint32_t idle_posted = 0;    // wave-uniform: idle lanes this wave has announced in st.blk[2]
    uint32_t spins = 0, round = 0;
    uint32_t wait_rounds = 0;    // owner: rounds spent waiting for helpers (bounded, see RT_WAIT_ROUNDS_MAX)
    // Order in which the cursor hands out the items.  Items come in blocks of 64 neighbours (one 8x8 pixel
    // tile for primary rays).  Expensive rays cluster (silhouettes): a wave that takes a whole expensive
    // block is busy three times longer than the average wave and the kernel waits for it.  So the cursor
    // walks the blocks strip by strip: strip 0 (RT_STRIP items) of every block, then strip 1 of every block,
    // ...  A 64-lane refill then takes one strip from each of 64/RT_STRIP neighbouring blocks (for primary
    // rays a line of pixels), and the rays of one block are spread over 64/RT_STRIP waves and moments.
    const uint32_t n_blocks = (end - begin + 63u) / 64u;
    // A large generation is cut into RT_PARTS contiguous partitions, each with a cursor of its own, and a
    // workgroup starts on partition blockIdx % RT_PARTS: the workgroups that share an XCD work on one part of
    // the image (of the record pool), so that XCD's L2 holds that part's share of the BVH instead of all of it.
    // (mixed launches too: a partition holds its share of the closest-hit blocks, handed out first, and its share of the any-hit
    // blocks — with one cursor for all 4096 waves the refills themselves queued on that word)
    const uint32_t parts = (LINGER && (RT_MIXED_PARTS || !MIXED || MEGA) && n_blocks >= RT_PARTS_MIN_BLOCKS) ? (uint32_t)RT_PARTS : 1u;
    const uint32_t part0 = LINGER ? blockIdx.x % parts : 0u;
    uint32_t parts_done = 0;     // wave-uniform: partitions this wave knows to be drained
    // The first 64 items of every wave are its own by position — wave k of a partition's workgroups takes cursor values
    // [64 k, 64 k + 64) of that partition — and the cursors hand out what lies behind those: when a launch starts, every
    // wave of the chip would otherwise add to the same few words at once, and same-address atomics queue at ~60 ns each
    // (512 waves per cursor: the last one starts 30 us late; 4096 on the mixed launches' one cursor: 245 us).
    bool first_fill = RT_STATIC_FIRST && LINGER && !MEGA;   // wave-uniform: that batch is still to be taken
    const uint32_t waves_per_wg = blockDim.x >> 6;
    auto static_items = [&](uint32_t part) -> uint32_t {     // cursor values of partition `part` that are taken by position
        return (RT_STATIC_FIRST && LINGER && !MEGA) ? ((gridDim.x + parts - 1u - part) / parts) * waves_per_wg * 64u : 0u;
    };
    uint32_t stale = 0;          // wave-uniform: rounds with idle lanes since the last refill
    uint32_t idle_rounds = 0;    // MEGA, wave-uniform: consecutive rounds without a ray in this wave (bounded)
    // MEGA: a wave that takes shadow slots first runs with the any-hit kernel's phase thresholds, and an any-hit lane
    // postpones as few leaves as there (its first accepted hit ends it) — per wave / per lane instead of per kernel
    uint32_t exit_few = EXIT_FEW, exit_wait = EXIT_WAIT;
    if constexpr (MEGA) { if (io.pref_any) { exit_few = RT_EXIT_FEW_ANY; exit_wait = RT_EXIT_WAIT_ANY; } }
    st.sp = 0;
    RT_TL(unsigned long long tl_t0 = 0; uint32_t tl_ns = 0, tl_nl = 0, tl_ts = 0, tl_tlanes = 0;)
    RT_TL(unsigned long long tlt[6] = {0, 0, 0, 0, 0, 0};)
    RT_TL(unsigned long long tlu[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0;)
    RT_TL(unsigned long long tlr_t0 = 0; uint32_t tlr_nodes = 0, tlr_leaves = 0, tlr_posts = 0, tlr_waits = 0, tlr_rounds = 0;)
    RT_TL(if (MODE == RT_TIMELINE) { const unsigned long long now = (unsigned long long)wall_clock64(); if (lane == 0) atomicMin(&g_tl_t0, now); tl_t0 = now; })

    for (;;) {
        RT_CLK(c0);
        RT_TLC(q0);
        RT_TL(if (alive && !helper) tlr_rounds++;)
        RT_TL(if (MODE == RT_TIMELINE) {
            const unsigned long long t0g = *(volatile unsigned long long*)&g_tl_t0;
            const uint32_t bin = (uint32_t)min((unsigned long long)(RT_TL_BINS - 1), ((unsigned long long)wall_clock64() - t0g) / 1000ull);
            const uint32_t na = (uint32_t)__popcll(__ballot(alive)), nh = (uint32_t)__popcll(__ballot(alive && helper));
            for (int off = 32; off != 0; off >>= 1) {   // the counts sit in whichever lane was first active in a step
                tl_ns += (uint32_t)__shfl_xor((int)tl_ns, off); tl_nl += (uint32_t)__shfl_xor((int)tl_nl, off);
                tl_ts += (uint32_t)__shfl_xor((int)tl_ts, off); tl_tlanes += (uint32_t)__shfl_xor((int)tl_tlanes, off);
            }
            const uint32_t wv = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
            if (lane == 0 && wv < RT_TL_WAVES) {
                uint32_t* o = g_tl[wv][bin];
                o[0] += 1u; o[1] += na; o[2] += tl_ns; o[3] += tl_nl; o[4] += tl_ts; o[5] += tl_tlanes; o[6] += nh; o[7] += exhausted ? 1u : 0u;
            }
            tl_ns = tl_nl = tl_ts = tl_tlanes = 0;
        })
        if (LINGER && !exhausted && !first_fill) {   // partitions other waves of the workgroup have seen drained
            const uint32_t d = st.blk[4];
            if (d > parts_done) {
                parts_done = d;
                if (parts_done >= parts) { exhausted = true; RT_STAT(st.acc->clk_exhausted = wall_clock64();) }
            }
        }
        // ---- subtree donation over the workgroup's task board (see the header)
#ifdef RT_SPLIT_ALWAYS
        const bool split_now = S.donate_ok != 0;     // test build: exercise the donation path all the time
#elif defined(RT_NO_DONATE)
        const bool split_now = false;                // experiment: no end-game machinery at all
#else
        // (not in the any-hit kernel: shadow rays end at their first hit, few are long enough to be worth splitting, and
        // the kernel is 10 % faster without the board: 0.86 -> 0.77 ms on the 1080p mesh frame.  Any-hit rays of the
        // mixed kernels still take part.)
        // (nor in the depth-first pixel loop, LINGER == false: each of its trace calls holds at most one ray per pixel
        // slot and is "drained" from its first round on; the board cost 40 % of every call: 269 -> 158 us)
        bool split_now = LINGER && !ANY_ONLY && exhausted && S.donate_ok != 0;
        if constexpr (MEGA) split_now = split_now && io.ring_empty();   // idle lanes have nothing else to take
#endif
        if constexpr (MEGA) {
            if (io.aborted()) break;
        }
        round++;
        if (round > RT_ROUND_CAP) {                 // never spin forever on the GPU: report instead (error 32)
            st.overflow |= 32u;
            if constexpr (MEGA) io.abort();
            RT_STAT(if (lane == 0) printf("[stuck] block %d wave slot %u: alive %llx helper %llx inmesh %llx donated %llx cur0 %x busy %u ready %u idle %u out[me] %u\n", (int)blockIdx.x, st.slot, (unsigned long long)__ballot(alive), (unsigned long long)__ballot(helper), (unsigned long long)__ballot(in_mesh), (unsigned long long)__ballot(donated), cur, st.blk[0], st.blk[1], st.blk[2], st.out_cnt[st.slot]);)
            break;
        }
        // (MEGA: subtrees are posted only while split_now holds, which comes and goes with the work ring; posted
        // subtrees are taken, and cancellations seen, in every round — an owner must never wait on a task nobody takes)
        if (split_now || MEGA) {
            RT_CLK(s0);
            RT_TL(u0 = (unsigned long long)wall_clock64();)
            // (a) an idle lane takes a posted subtree
            if (!alive && st.blk[1] != 0) {
                const uint32_t start = (lane * 7u + st.slot + round) % st.n_board;
                for (uint32_t k = 0; k < 4u && !alive; k++) {
                    uint32_t* rec = st.board + ((start + k) % st.n_board) * 16u;
                    // (rec[15]: closest-hit or any-hit subtree — in rt_ordered.h the waves of a workgroup run both kinds of loop)
                    if (rec[0] == 2u && (MIXED || rec[15] == (ANY_ONLY ? 1u : 0u)) && atomicCAS(&rec[0], 2u, 3u) == 2u) {
                        // pairs with the poster's release fence: the record's words are read after the claim, not before
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        if (MIXED) lane_any = rec[15] != 0u;
                        home = rec[1];
                        cur = rec[2];
                        cur_t = __uint_as_float(rec[3]);
                        tmax = __uint_as_float(rec[4]);
                        ar.ox = __uint_as_float(rec[5]); ar.oy = __uint_as_float(rec[6]); ar.oz = __uint_as_float(rec[7]);
                        ar.dx = __uint_as_float(rec[8]); ar.dy = __uint_as_float(rec[9]); ar.dz = __uint_as_float(rec[10]);
                        ar.ix = __uint_as_float(rec[11]); ar.iy = __uint_as_float(rec[12]); ar.iz = __uint_as_float(rec[13]);
                        cur_obj = rec[14];
                        rec[0] = 0u;                         // slot free again
                        atomicSub(&st.blk[1], 1u);
                        RT_STAT(st.acc->n_claim++;)
                        alive = true;
                        helper = true;
                        in_mesh = true;
                        mesh_found = false;
                        donated = false;
                        no_donate = false;
                        qn = 0;
                        st.sp = 0;
                        st.push(RT_LINK_SENTINEL, -1.0f);
                        don_next = st.sp;
                        don_base = st.sp;
                    }
                }
            }
            RT_CLK(s1);
            RT_TL(u1 = (unsigned long long)wall_clock64();)
            // (b) the wave tells the workgroup how many of its lanes are idle
            {
                const uint32_t n_idle_now = split_now ? (uint32_t)__popcll(__ballot(!alive)) : 0u;
                if (n_idle_now != idle_posted) {
                    if (lane == 0) atomicAdd(&st.blk[2], n_idle_now - idle_posted);   // unsigned wrap = signed difference
                    idle_posted = n_idle_now;
                }
            }
            RT_CLK(s2);
            RT_TL(u2 = (unsigned long long)wall_clock64();)
            // (c) a busy lane posts the bottom-most entry of its mesh-level stack while idle lanes outnumber posted tasks
            RT_STAT(if (split_now && alive && in_mesh && !no_donate && don_next < st.sp) { st.acc->n_elig++; if (!(st.blk[2] > st.blk[1])) st.acc->n_gate++; })
            for (int pass = 0; pass < RT_DONATE_PASSES && split_now && alive && in_mesh && !no_donate && don_next < st.sp && st.blk[2] > st.blk[1] && st.blk[1] < st.n_board; pass++) {
                RT_STAT(if (st.lane == __ffsll((long long)__ballot(1)) - 1) st.acc->ts_post_iters++;)
                uint32_t* rec = st.board + (atomicAdd(&st.blk[3], 1u) % st.n_board) * 16u;
                const bool got_slot = atomicCAS(&rec[0], 0u, 1u) == 0u;
                RT_STAT(if (!got_slot) st.acc->n_casfail++;)
                if (got_slot) {
                    const uint2 e = st.read_at(don_next);
                    st.write_at(don_next, make_uint2(RT_LINK_NONE, 0u));
                    don_next++;
                    const float dt = __uint_as_float(e.y);
                    if (e.x != RT_LINK_NONE && !((tmax < dt) || (dt == 0.0f && tmax <= 0.0f))) {
                        if (!helper && !donated) {           // first subtree of this mesh traversal that leaves the owner
                            donated = true;
                            st.cand_cnt[st.slot] = 0;
                            st.cancel[st.slot] = 0;
                        }
                        atomicAdd(&st.out_cnt[home], 1u);
                        rec[1] = home; rec[2] = e.x; rec[3] = e.y; rec[4] = __float_as_uint(tmax);
                        rec[5] = __float_as_uint(ar.ox); rec[6] = __float_as_uint(ar.oy); rec[7] = __float_as_uint(ar.oz);
                        rec[8] = __float_as_uint(ar.dx); rec[9] = __float_as_uint(ar.dy); rec[10] = __float_as_uint(ar.dz);
                        rec[11] = __float_as_uint(ar.ix); rec[12] = __float_as_uint(ar.iy); rec[13] = __float_as_uint(ar.iz);
                        rec[14] = cur_obj;
                        rec[15] = RT_ISANY ? 1u : 0u;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        rec[0] = 2u;                         // ready
                        atomicAdd(&st.blk[1], 1u);
                        RT_STAT(st.acc->n_donations++;)
                        RT_TL(tlr_posts++;)
                    } else {
                        rec[0] = 0u;                         // the entry is culled by now: nothing to give
                    }
                }
            }
            RT_CLK(s3);
            RT_TL(u3 = (unsigned long long)wall_clock64();)
            // The slots given away are the bottom of the mesh-level stack: close the gap (usually nothing is left
            // above it), so that the stack stays short and in LDS instead of growing into its global part.
            if (alive && in_mesh && don_next > don_base) {
                const uint32_t gap = don_next - don_base;
                for (uint32_t pos = don_next; pos < st.sp; pos++) st.write_at(pos - gap, st.read_at(pos));
                st.sp -= gap;
                don_next = don_base;
            }
            // (d) any-hit found by somebody working on the same ray: helpers drop their subtree, the owner stops
            //     its own part (pops down to the sentinel) and finishes once its helpers are gone
            if (RT_ISANY && alive && (helper || donated) && st.cancel[home] != 0) {
                if (helper) {
                    atomicSub(&st.out_cnt[home], 1u);
                    alive = false; helper = false; in_mesh = false; cur = RT_LINK_DONE; st.sp = 0; qn = 0;
                } else if (cur != RT_LINK_SENTINEL) {
                    qn = 0;
                    cur = st.pop(-1.0f, cur_t);              // everything above the sentinel is culled by a negative bound
                }
            }
            RT_TL(if (MODE == RT_TIMELINE && exhausted) { const unsigned long long u4_ = (unsigned long long)wall_clock64(); tlu[0] += u1 - u0; tlu[1] += u2 - u1; tlu[2] += u3 - u2; tlu[3] += u4_ - u3; })
            RT_STAT({ const unsigned long long s4 = wall_clock64(); if (__any(alive)) { st.acc->ts_a += s1 - s0; st.acc->ts_b += s2 - s1; st.acc->ts_c += s3 - s2; st.acc->ts_d += s4 - s3; } })
        }

        // ---- nothing to do in this wave: leave, or stay to help the workgroup
        const bool any_alive = __any(alive);
        if constexpr (MEGA) {
            if (!any_alive) {
                if (was_busy) {
                    was_busy = false;
                    if (lane == 0) atomicSub(&st.blk[0], 1u);
                }
                if (exhausted && io.done()) {                // no primary left, nothing unfinished in the workgroup
                    if (idle_posted != 0 && lane == 0) atomicSub(&st.blk[2], idle_posted);
                    break;
                }
                if (io.ring_empty() && st.blk[1] == 0) __builtin_amdgcn_s_sleep(16);
                round--;                                     // waiting is not work
                if (++idle_rounds > RT_MEGA_IDLE_CAP) { st.overflow |= 32u; io.abort(); break; }
            } else {
                idle_rounds = 0;
            }
        }
        if (!MEGA && !any_alive && exhausted) {
            if (was_busy) {
                was_busy = false;
                if (lane == 0) atomicSub(&st.blk[0], 1u);
            }
            // A finished wave of a trace kernel stays while other waves of its workgroup work: they may post
            // subtrees.  Leaving early is always safe (a poster takes its own tasks back when its lanes go idle).
            if (!LINGER || !split_now || st.blk[0] == 0 || (++spins > RT_LINGER_SPINS && st.blk[1] == 0)) {
                if (idle_posted != 0 && lane == 0) atomicSub(&st.blk[2], idle_posted);
                break;
            }
            __builtin_amdgcn_s_sleep(32);
            round--;                                 // lingering is not work: RT_ROUND_CAP bounds working rounds
            continue;
        }
        if (any_alive && !was_busy) {
            was_busy = true;
            if (lane == 0) atomicAdd(&st.blk[0], 1u);
        }

        // ---- node phase: collapsed interior records of either level; mesh leaves are queued
        RT_CLK(c1);
        RT_TLC(q1);
        RT_STAT(if (st.lane == __ffsll((long long)__ballot(1)) - 1) st.acc->wave_outer++;)
        uint32_t phase_steps = 0;    // wave-uniform
        for (;;) {
            // A lane can make progress here while it sits on an interior record, or on a mesh
            // leaf it still has queue room for (big leaves wait for the triangle phase; idle
            // lanes sit on RT_LINK_DONE).  The phase ends for the WAVE when no lane can
            // (wave-uniform exit: the two phases must not be interleaved by the compiler).
            const bool leaf_here = (cur & RT_LINK_LEAF) != 0;
            const bool want = !leaf_here || (in_mesh && cur < 0xFC000000u && qn < ((MEGA && lane_any) ? (uint32_t)RT_LEAF_QUEUE_ANY : QCAP));
            {   // Leave the node phase also when only a few lanes still walk while others wait with queued
                // leaves: the walkers keep their place and go on in the next node phase.  (A wave completes a
                // step about every 1.3 us whatever the number of active lanes, so steps with few lanes cost
                // as much as full ones.)
                const uint32_t n_want = wave_count(want), n_wait = wave_count(!want && (qn != 0 || (RT_WAIT_COUNTS_TRANSITIONS && alive && !donated)));
                if (n_want == 0 || (n_want <= exit_few && n_wait >= exit_wait)) break;
                // end game (subtrees are being donated): short node phases, so that posted subtrees are taken and
                // new ones posted every few steps instead of once per ~30 — a long ray fans out over the idle lanes
                if (split_now && ++phase_steps > (uint32_t)RT_TAIL_STEPS) break;
            }
            // (wave-uniform, once per step: every stack entry touched in it — the pop after a queued leaf, up to three
            // pushes, the pop of a lane that hit nothing — lies in the LDS part; the global part is the rare case)
            const bool lds_only = !wave_any(want && st.sp + 3 > RT_STACK_LDS);
            if (want) {
                if (leaf_here) {
                    RT_TL(tlr_leaves++;)
                    RT_STAT(st.acc->ray_leaves++;)
                    st.q[qn * 64 + st.lane] = make_uint2(cur, __float_as_uint(cur_t));
                    qn++;
                    cur = lds_only ? st.pop_lds(tmax, cur_t) : st.pop(tmax, cur_t);
                }
                if (!(cur & RT_LINK_LEAF)) {
                    RT_TL({ const unsigned long long b_ = __ballot(1); if ((int)lane == __ffsll((long long)b_) - 1) { tl_ns++; tl_nl += (uint32_t)__popcll(b_); } })
                    RT_TL(tlr_nodes++;)
                    RT_STAT(st.acc->ray_nodes++;)
                    RT_STAT({ unsigned long long m = __ballot(1); if (st.lane == __ffsll((long long)m) - 1) { st.acc->wave_node_steps++; st.acc->lane_node_steps += __popcll(m); st.acc->r_node_w++; st.acc->r_node_l += __popcll(m); } })
                    // The record holds, per axis, the four slots' min planes and their max planes (rt_device.h).  A
                    // lane reads per axis the planes its ray meets first and the ones it meets last — min then max
                    // if 1/d >= 0 (+inf for d == 0 included), max then min otherwise — so the slab distances come
                    // out ordered without the reference's compare-and-swap (node_slabs4).
                    // (32-bit byte offsets from the uniform base — the upload refuses node arrays of 4 GB — so that the seven
                    // loads take the base from scalar registers and the lane's offset from one register each, instead of
                    // seven 64-bit address sums)
                    const char* nodes_b = reinterpret_cast<const char*>(S.nodes);
                    const uint32_t no = cur << 7;
                    const uint32_t sx = (__float_as_uint(ar.ix) >> 31) << 4, sy = (__float_as_uint(ar.iy) >> 31) << 4,
                                   sz = (__float_as_uint(ar.iz) >> 31) << 4;
                    // (records are 128-byte aligned: the other 16-byte half of an axis' 32 bytes is the offset ^ 16)
                    const uint32_t ox_ = no | sx, oy_ = no | sy, oz_ = no | sz;
                    const float4 nx = *reinterpret_cast<const float4*>(nodes_b + ox_), fx = *reinterpret_cast<const float4*>(nodes_b + (ox_ ^ 16u));
                    const float4 ny = *reinterpret_cast<const float4*>(nodes_b + oy_ + 32u), fy = *reinterpret_cast<const float4*>(nodes_b + (oy_ ^ 16u) + 32u);
                    const float4 nz = *reinterpret_cast<const float4*>(nodes_b + oz_ + 64u), fz = *reinterpret_cast<const float4*>(nodes_b + (oz_ ^ 16u) + 64u);
                    const uint4 lk = *reinterpret_cast<const uint4*>(nodes_b + no + 96u);
                    float te4[4];
                    bool hit4[4];
                    if (__builtin_expect(!wave_any(ray_is_odd(ar)), 1)) node_slabs4(ar, nx, fx, ny, fy, nz, fz, tmax, te4, hit4);
                    else node_slabs4_any(ar, nx, fx, ny, fy, nz, fz, sx != 0u, sy != 0u, sz != 0u, tmax, te4, hit4);
                    const float t0 = te4[0], t1 = te4[1], t2 = te4[2], t3 = te4[3];
                    // slot 0 is always valid; an empty slot carries RT_LINK_DONE
                    const bool h0 = hit4[0];
                    const bool h1 = hit4[1] & (lk.y != RT_LINK_DONE);
                    const bool h2 = hit4[2] & (lk.z != RT_LINK_DONE);
                    const bool h3 = hit4[3] & (lk.w != RT_LINK_DONE);
#ifdef RT_NEAR_TOP      // experiment: the top level nearest-first as well (outside the tolerance on overlapping objects, see below)
                    if constexpr (NEAR && !ANY_ONLY) {
#else
                    if constexpr (NEAR && !ANY_ONLY) if (in_mesh) {
#endif
                        // Non-parity order (RT_FLAG_NEAR_FIRST), inside a mesh BVH only: the hit slots are visited by
                        // increasing entry distance instead of in the reference's slot order, so the bound shrinks
                        // early and more of the later boxes are culled.  All triangles of a mesh are tested against
                        // one bound in one space, so the closest hit is the same set minimum and only candidates with
                        // equal t can come out differently.  The top level keeps the reference's order: there the
                        // bound is carried between object spaces through a Q_rsqrt-normalised direction
                        // (raytracer.cpp:572-580), and which of two nearby objects wins depends on the order.
                        const float INF = __builtin_inff();
                        float k0 = h0 ? t0 : INF, k1 = h1 ? t1 : INF, k2 = h2 ? t2 : INF, k3 = h3 ? t3 : INF;
                        uint32_t l0 = lk.x, l1 = lk.y, l2 = lk.z, l3 = lk.w;
#define RT_CSWAP(ka, la, kb, lb) { const bool s_ = ka > kb; const float kl_ = fminf(ka, kb), kh_ = fmaxf(ka, kb); const uint32_t ll_ = s_ ? lb : la, lh_ = s_ ? la : lb; ka = kl_; kb = kh_; la = ll_; lb = lh_; }
                        RT_CSWAP(k0, l0, k1, l1) RT_CSWAP(k2, l2, k3, l3) RT_CSWAP(k0, l0, k2, l2) RT_CSWAP(k1, l1, k3, l3) RT_CSWAP(k1, l1, k2, l2)
#undef RT_CSWAP
                        if (lds_only) {
                            if (k3 < INF) st.push_lds(l3, k3);
                            if (k2 < INF) st.push_lds(l2, k2);
                            if (k1 < INF) st.push_lds(l1, k1);
                        } else {
                            if (k3 < INF) st.push(l3, k3);
                            if (k2 < INF) st.push(l2, k2);
                            if (k1 < INF) st.push(l1, k1);
                        }
                        if (k0 < INF) { cur = l0; cur_t = k0; }
                        else cur = lds_only ? st.pop_lds(tmax, cur_t) : st.pop(tmax, cur_t);
                        continue;
                    }
                    // visit order = slot order: the first hit slot is next, the later ones are
                    // pushed last-first so that they pop in order
                    const bool p1 = h1 & h0, p2 = h2 & (h0 | h1), p3 = h3 & (h0 | h1 | h2);
                    if (lds_only) {
                        st.push_lds_if(p3, lk.w, t3);
                        st.push_lds_if(p2, lk.z, t2);
                        st.push_lds_if(p1, lk.y, t1);
                    } else {
                        if (p3) st.push(lk.w, t3);
                        if (p2) st.push(lk.z, t2);
                        if (p1) st.push(lk.y, t1);
                    }
                    // (selects, then one branch for the lanes that hit nothing)
                    cur = h0 ? lk.x : (h1 ? lk.y : (h2 ? lk.z : lk.w));
                    cur_t = h0 ? t0 : (h1 ? t1 : (h2 ? t2 : t3));
                    if (!(h0 | h1 | h2 | h3)) cur = lds_only ? st.pop_lds(tmax, cur_t) : st.pop(tmax, cur_t);
                }
            }
        }

        RT_CLK(c2);
        RT_TLC(q2);
        // A triangle of the mesh in flight passed the test with bound tmax (leaf box entry te):
        // the owner accepts it as the reference does; a helper records it (see the header).
        // Returns true when the lane is finished with its ray / subtree (any-hit).
        auto accept = [&](uint32_t ti, float t, float bb, float gg, float te) -> bool {
            if (helper) {
                const uint32_t k = atomicAdd(&st.cand_cnt[home], 1u);
                if (k < RT_CAND_MAX) {
                    // read by the owner, possibly in another wave: written through to L2 (agent scope), read past L1
                    uint32_t* cp = reinterpret_cast<uint32_t*>(&st.cand[home * RT_CAND_MAX + k]);
                    __hip_atomic_store(cp, ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cp + 1, __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cp + 2, __float_as_uint(te), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // (written as selects on purpose: with `tmax = t` under two different conditions the optimiser turns the two
            // stores into one store through a selected ADDRESS, and tmax / mesh_* then live in scratch memory for the
            // whole kernel — a scratch round trip per triangle step)
            const bool mine = !helper;
            tmax = (mine || t >= te) ? t : tmax;
            mesh_found = mesh_found || mine;
            mesh_tri = mine ? ti : mesh_tri;
            mesh_beta = mine ? bb : mesh_beta;
            mesh_gamma = mine ? gg : mesh_gamma;
            if (RT_ISANY) {
                // Untransformed mesh: worldT == t < testMinT == minT, so the
                // reference's verdict is already known.
                if (!(S.objects[cur_obj].flags & RT_OBJ_HAS_TRANSFORM)) {
                    if (helper || donated) st.cancel[home] = 1;
                    return true;
                }
            }
            return false;
        };

        // RT_FLAG_NEAR_FIRST only: leaves are no longer met in the reference's order, so a triangle that ties with
        // the best so far (equal t) is let through the test (bound one ulp up) and wins if it comes earlier in the
        // reference's order — triangle records are stored in that order.  Helpers record every tie; the owner
        // decides at the replay.  With this the mode returns the triangle the reference returns whenever the
        // reference's box culls agree with its triangle tests.
        auto tbound = [&]() -> float {
            if constexpr (NEAR) {
                if (!RT_ISANY && (helper || mesh_found) && tmax < RT_INF && tmax > 0.0f) return __uint_as_float(__float_as_uint(tmax) + 1u);
            }
            return tmax;
        };
        // (a rebuilt tree stores its triangles in its own leaf order: the reference's is then a side array)
        auto earlier = [&](uint32_t a, uint32_t b) -> bool { return S.tri_rank ? S.tri_rank[a] < S.tri_rank[b] : a < b; };
        auto wins = [&](float t, uint32_t ti) -> bool {
            if constexpr (NEAR) return RT_ISANY || helper || !mesh_found || t < tmax || earlier(ti, mesh_tri);
            return true;
        };

        // ---- triangle phase: the postponed leaves, in visiting order (raytracer.cpp:781-798)
        bool fin = false;        // this lane's ray / subtree is finished by an any-hit
        for (uint32_t qi = 0; qi < qn && !fin; qi++) {
            const uint2 qe = st.q[qi * 64 + st.lane];
            const uint32_t link = qe.x;
            const float te = __uint_as_float(qe.y);
            if ((tmax < te) || (te == 0.0f && tmax <= 0.0f)) continue;   // the reference would cull this leaf's box now
            uint32_t ti = RT_LEAF_FIRST(link);
            uint32_t n = RT_LEAF_COUNT(link);                            // < RT_LEAF_BIG: big leaves are never queued
            // RT_TRI_UNROLL triangles are loaded before the first is tested; they are tested one after the other
            // against the bound the previous one left, as the reference's loop does.  Past the leaf's end a lane
            // re-reads the leaf's first triangle.
            while (n != 0 && !fin) {
                RT_TL({ const unsigned long long b_ = __ballot(1); if ((int)lane == __ffsll((long long)b_) - 1) { tl_ts++; tl_tlanes += (uint32_t)__popcll(b_); } })
                RT_STAT({ unsigned long long m = __ballot(1); if (st.lane == __ffsll((long long)m) - 1) { st.acc->wave_tri_steps++; st.acc->lane_tri_steps += __popcll(m); st.acc->r_tri_w++; st.acc->r_tri_l += __popcll(m); } })
                const uint32_t m = n < (uint32_t)RT_TRI_UNROLL ? n : (uint32_t)RT_TRI_UNROLL;
                f4u ta[RT_TRI_UNROLL], tb[RT_TRI_UNROLL];
                float tc[RT_TRI_UNROLL];
#pragma unroll
                for (uint32_t u = 0; u < RT_TRI_UNROLL; u++) {
                    // 36-byte record as dwordx4 + dwordx4 + dword, all issued before the tests (32-bit byte offset: a leaf's
                    // first triangle is below RT_LEAF_MAX_FIRST, 36 x that is below 4 GB)
                    const float* tp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(S.tris) + (ti + (u < m ? u : 0u)) * 36u);
                    ta[u] = *reinterpret_cast<const f4u*>(tp);
                    tb[u] = *reinterpret_cast<const f4u*>(tp + 4);
                    tc[u] = tp[8];
                }
#pragma unroll
                for (uint32_t u = 0; u < RT_TRI_UNROLL; u++) {
                    float t, bb, gg;
                    if (u < m && !fin && tri_test(ar, ta[u].x, ta[u].y, ta[u].z, ta[u].w, tb[u].x, tb[u].y, tb[u].z, tb[u].w, tc[u], tbound(), t, bb, gg)) {
                        if (wins(t, ti + u) && accept(ti + u, t, bb, gg, te)) fin = true;
                    }
                }
                n -= m;
                ti += m;
            }
        }
        qn = 0;
        RT_CLK(c3);
        RT_TLC(q3);

        // ---- a big leaf (>= RT_LEAF_BIG triangles, raytracer.cpp:183: an unsplittable node) waiting at `cur`.
        // It follows the queued leaves in visiting order.  Below RT_COOP_MIN triangles the lane tests it
        // alone; larger ones (from RT_COOP_MIN on, unless many lanes wait with one) the whole wave tests for that lane, 64 triangles per step: the
        // sequential scan `if (t < minT) minT = t` ends at the first triangle that attains the smallest
        // t below the incoming bound, which is what the (t, index) minimum over the lanes selects.
        bool big = alive && !fin && in_mesh && is_big_leaf_link(cur);
        if (big && ((tmax < cur_t) || (cur_t == 0.0f && tmax <= 0.0f))) { big = false; cur = st.pop(tmax, cur_t); }
        const uint64_t big_lanes = __ballot(big);
        if (big_lanes != 0) {
            // the whole wave pays ~3 steps per leaf it tests together; a lane alone pays one step per triangle
            const uint32_t coop_min = max((uint32_t)RT_COOP_MIN, 3u * (uint32_t)__popcll(big_lanes));
            uint32_t big_first = 0, big_n = 0;
            if (big) {
                big_first = RT_LEAF_FIRST(cur);
                big_n = leaf_count(S, cur, false);
                if (big_n < coop_min) {
                    uint32_t ti = big_first;
                    for (uint32_t n = big_n; n != 0; n--, ti++) {
                        const float* tp = reinterpret_cast<const float*>(&S.tris[ti]);
                        const f4u ta = *reinterpret_cast<const f4u*>(tp), tb = *reinterpret_cast<const f4u*>(tp + 4);
                        const float tc = tp[8];
                        float t, bb, gg;
                        if (tri_test(ar, ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w, tc, tbound(), t, bb, gg)) {
                            if (wins(t, ti) && accept(ti, t, bb, gg, cur_t)) { fin = true; break; }
                        }
                    }
                }
            }
            uint64_t coop = __ballot(big && big_n >= coop_min);
            while (coop != 0) {                      // wave-uniform; every lane (idle ones too) takes part
                const int src = __ffsll((long long)coop) - 1;
                coop &= coop - 1;
                RT_STAT(st.acc->n_coop++;)
                ActiveRay br;
                br.ox = lane_bcast(ar.ox, src); br.oy = lane_bcast(ar.oy, src); br.oz = lane_bcast(ar.oz, src);
                br.dx = lane_bcast(ar.dx, src); br.dy = lane_bcast(ar.dy, src); br.dz = lane_bcast(ar.dz, src);
                br.ix = br.iy = br.iz = 0.f;         // not used by tri_test
                const float bound = lane_bcast(tbound(), src);
                const uint32_t first = lane_bcast(big_first, src), count = lane_bcast(big_n, src);
                float best_t = bound, best_b = 0.f, best_g = 0.f;
                uint32_t best_i = 0xFFFFFFFFu;
                for (uint32_t k = lane; k < count; k += 64) {
                    const float* tp = reinterpret_cast<const float*>(&S.tris[first + k]);
                    const f4u ta = *reinterpret_cast<const f4u*>(tp), tb = *reinterpret_cast<const f4u*>(tp + 4);
                    const float tc = tp[8];
                    float t, bb, gg;
                    if (tri_test(br, ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w, tc, bound, t, bb, gg) && t < best_t) {
                        best_t = t; best_i = first + k; best_b = bb; best_g = gg;
                    }
                }
                float red_t = best_t;
                uint32_t red_i = best_i;
                for (int off = 32; off != 0; off >>= 1) {
                    const float ot = __shfl_xor(red_t, off);
                    const uint32_t oi = (uint32_t)__shfl_xor((int)red_i, off);
                    if (ot < red_t || (ot == red_t && oi < red_i)) { red_t = ot; red_i = oi; }
                }
                if (red_i != 0xFFFFFFFFu) {          // wave-uniform after the butterfly
                    const int win = __ffsll((long long)__ballot(best_i == red_i)) - 1;
                    const float wb = lane_bcast(best_b, win), wg = lane_bcast(best_g, win);
                    if ((int)lane == src) {
                        if (wins(red_t, red_i) && accept(red_i, red_t, wb, wg, cur_t)) fin = true;
                    }
                }
            }
            if (big && !fin) cur = st.pop(tmax, cur_t);
        }

        // An owner that found its any-hit while helpers still trace parts of the same ray has told them
        // (st.cancel); it drops the rest of its own part and waits at the sentinel until they are gone, so
        // that no helper of this ray is left when the slot takes its next ray.
        if (RT_ISANY && fin && !helper && donated) {
            fin = false;
            if (cur != RT_LINK_SENTINEL) cur = st.pop(-1.0f, cur_t);   // (the sentinel may already have been popped)
        }

        // ---- per-lane transitions between the levels; then idle lanes are refilled from the cursor, and the new
        // rays make their first transition (top-level entry, raytracer.cpp:548-745) at once, in a second pass,
        // instead of sitting through a whole round before it
        RT_CLK(c4);
        RT_TLC(q4);
        for (int tpass = 0;; tpass++) {
        RT_CLK(xa);
        RT_TL(u4 = (unsigned long long)wall_clock64();)
        bool finished = fin;
        if (alive && !fin) do {
            if (!(cur & RT_LINK_LEAF)) break;          // still on an interior record (the node phase was left early)
            RT_STAT(const unsigned long long y0 = wall_clock64(); { st.acc->y_lbody++; if ((int)lane == __ffsll((unsigned long long)__ballot(1)) - 1) st.acc->y_wbody++; })
            if (in_mesh) {
                if (cur != RT_LINK_SENTINEL) {
                    // a mesh leaf is waiting (queue was full): it was accepted with an older minT, re-check it
                    if ((cur & RT_LINK_LEAF) && ((tmax < cur_t) || (cur_t == 0.0f && tmax <= 0.0f))) cur = st.pop(tmax, cur_t);
                    break;
                }
                if (helper) { finished = true; break; }        // the donated subtree is done
                bool redo = false;
                if (donated) {
                    RT_TL(if (st.out_cnt[st.slot] != 0) tlr_waits++;)
                    if (st.out_cnt[st.slot] != 0) {            // wait for the helpers (cur stays on the sentinel)
                        if (++wait_rounds > RT_WAIT_ROUNDS_MAX) { st.overflow |= 16u; } else break;
                    }
                    wait_rounds = 0;
                    if (RT_ISANY && st.cancel[st.slot] != 0) {   // somebody found the any-hit (untransformed mesh)
                        donated = false;
                        fin = true;
                        finished = true;
                        break;
                    }
                    // ---- replay the helpers' candidates in visiting order with the reference's rule
                    const uint32_t nc = st.cand_cnt[st.slot];
                    RT_TL(if (MODE == RT_TIMELINE) { atomicAdd(&g_tl_tail2[6], 1ull); atomicAdd(&g_tl_tail2[7], (unsigned long long)nc); })
                    if (nc > RT_CAND_MAX) {
                        redo = true;                           // too many to keep: trace this mesh again, alone
                    } else {
                        uint32_t last = 0;
                        bool first_round = true, from_cand = false;
                        for (uint32_t r = 0; r < nc; r++) {
                            uint4 best = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
                            for (uint32_t k = 0; k < nc; k++) {
                                // written by other lanes of the workgroup: read past the (non-coherent) vector L1
                                const uint32_t* cp = reinterpret_cast<const uint32_t*>(&st.cand[st.slot * RT_CAND_MAX + k]);
                                uint4 c;
                                c.x = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                c.y = __hip_atomic_load(cp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                c.z = __hip_atomic_load(cp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                c.w = 0u;
                                if ((first_round || c.x > last) && c.x < best.x) best = c;
                            }
                            first_round = false;
                            last = best.x;
                            const float ct = __uint_as_float(best.y), cte = __uint_as_float(best.z);
                            if ((tmax < cte) || (cte == 0.0f && tmax <= 0.0f)) continue;     // leaf culled by now
                            if (ct < tmax || (NEAR && !RT_ISANY && mesh_found && ct == tmax && (S.tri_rank ? S.tri_rank[best.x] < S.tri_rank[mesh_tri] : best.x < mesh_tri))) { tmax = ct; mesh_found = true; mesh_tri = best.x; from_cand = true; }
                        }
                        if (from_cand) {
                            // beta / gamma of the winner: the same test on the same operands gives the same bits
                            const float* tp = reinterpret_cast<const float*>(&S.tris[mesh_tri]);
                            float t;
                            (void)tri_test(ar, tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6], tp[7], tp[8], RT_INF, t, mesh_beta, mesh_gamma);
                        }
                    }
                    donated = false;
                }
                in_mesh = false;
                if (redo) {
                    no_donate = true;
                    top_slot--;
#if !RT_FLAT_TOP
                    top_remaining++;
#endif
                } else {
                    // ---- mesh traversal finished (IntersectMeshBVH returned), raytracer.cpp:596-620
                    no_donate = false;
                    if (mesh_found) {
#if RT_OB_BATCH > 1
                        DObject ob;                            // transform, motion and flags in one batch of loads
                        {
                            const float4* src = reinterpret_cast<const float4*>(&S.objects[cur_obj]);
                            float4* dst = reinterpret_cast<float4*>(&ob);
                            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; dst[8] = src[8];
                        }
#else
                        const DObject& ob = S.objects[cur_obj];
#endif
                        float wt = world_t(ob, ro, time, mk(ar.ox, ar.oy, ar.oz), mk(ar.dx, ar.dy, ar.dz), tmax);
                        if (wt < minT) {
                            if (RT_ISANY) { fin = true; finished = true; break; }
                            minT = wt;
                            has = true;
                            out.kind_index = cur_kind_index;
                            out.tri = mesh_tri;
                            out.beta = mesh_beta;
                            out.gamma = mesh_gamma;
                        }
                    }
                }
                ar.set(ro, rd);
                tmax = minT;
            } else {
                if (cur == RT_LINK_DONE) { finished = true; break; }   // traversal exhausted
#if !RT_FLAT_TOP
                top_slot = RT_LEAF_FIRST(cur);
                top_remaining = leaf_count(S, cur, true);
#endif
            }

            // ---- prims of the current top-level leaf, in order (raytracer.cpp:548-745)
            RT_STAT(const unsigned long long y1 = wall_clock64(); if ((int)lane == __ffsll((unsigned long long)__ballot(1)) - 1) st.acc->y_fin += y1 - y0;)
            bool entered = false;
#if RT_FLAT_TOP
            // The top level itself is walked here, in the reference's order (IntersectTopBVH, raytracer.cpp:536-745: the nodes
            // in pre-order, a node's box tested with the minT of the moment it is popped, its subtree skipped on a miss): one
            // item per turn of this loop, a node or a prim of the leaf before it.  `ar` is the world ray here.  (As 4-wide
            // records in the node phase, every top-level leaf cost the ray a round of its wave.)
            while (top_slot < S.n_top_items) {
                RT_STAT({ st.acc->y_liters++; if ((int)lane == __ffsll((unsigned long long)__ballot(1)) - 1) st.acc->y_witers++; })
#if RT_NODE_RUN
                // (the nodes up to the next prim in a loop of their own: the lanes of a wave then meet at the prims, whose
                // branch is the long one, instead of running it in every turn in which one of them has a prim)
                uint4 it0 = S.top_items[2u * top_slot], it1 = S.top_items[2u * top_slot + 1u];
                while (it1.w == RT_TOPITEM_NODE) {
                    float te_;
                    const bool hit_ = box_test(ar, __uint_as_float(it0.x), __uint_as_float(it0.y), __uint_as_float(it0.z), __uint_as_float(it0.w),
                                               __uint_as_float(it1.x), __uint_as_float(it1.y), minT, te_);
                    top_slot = hit_ ? top_slot + 1u : it1.z;
                    if (top_slot >= S.n_top_items) break;
                    it0 = S.top_items[2u * top_slot];
                    it1 = S.top_items[2u * top_slot + 1u];
                }
                if (top_slot >= S.n_top_items) break;
                top_slot++;
#else
                const uint4 it0 = S.top_items[2u * top_slot], it1 = S.top_items[2u * top_slot + 1u];
                if (it1.w == RT_TOPITEM_NODE) {
                    float te_;
                    const bool hit_ = box_test(ar, __uint_as_float(it0.x), __uint_as_float(it0.y), __uint_as_float(it0.z), __uint_as_float(it0.w),
                                               __uint_as_float(it1.x), __uint_as_float(it1.y), minT, te_);
                    top_slot = hit_ ? top_slot + 1u : it1.z;
                    continue;
                }
                top_slot++;
#endif
                DTopPrim tp;
                tp.kind = it0.x; tp.index = it0.y; tp.obj = it0.z; tp.remaining = 0;
#else
            while (top_remaining != 0) {
                RT_STAT({ st.acc->y_liters++; if ((int)lane == __ffsll((unsigned long long)__ballot(1)) - 1) st.acc->y_witers++; })
                const DTopPrim tp = S.top_prims[top_slot];
                top_slot++;
                top_remaining--;
#endif
#if RT_OB_BATCH
                // the object's transform, motion, flags, root link and root box in one batch of loads (read field by
                // field, each load waits behind the branch that needs it: four round trips instead of one)
                const DObject& obg = S.objects[tp.obj];
                DObject ob;
                {
                    const float4* src = reinterpret_cast<const float4*>(&obg);
                    float4* dst = reinterpret_cast<float4*>(&ob);
#pragma unroll
                    for (int k = 4; k <= 10; k++) dst[k] = src[k];
                }
#else
                const DObject& ob = S.objects[tp.obj];
                const DObject& obg = ob;
#endif
                V3 to, td;
                float testMinT;
                object_space_ray(ob, ro, rd, time, minT, to, td, testMinT);
                if (tp.kind == RT_KIND_MESH) {
                    if (ob.root_link == RT_LINK_DONE) continue;
                    ActiveRay mr;
                    mr.set(to, td);
                    float te;
                    if (!box_test(mr, ob.rminx, ob.rminy, ob.rminz, ob.rmaxx, ob.rmaxy, ob.rmaxz, testMinT, te)) continue;
                    ar = mr;
                    tmax = testMinT;
                    cur_obj = tp.obj;
                    cur_kind_index = (RT_KIND_MESH << 28) | tp.index;
                    mesh_found = false;
                    in_mesh = true;
                    st.push(RT_LINK_SENTINEL, -1.0f);
                    don_next = st.sp;
                        don_base = st.sp;
                    donated = false;
                    cur = ob.root_link;
                    cur_t = te;
                    entered = true;
                    break;
                } else if (tp.kind == RT_KIND_SPHERE) {
                    ActiveRay sr;
                    sr.set(to, td);
                    float t;
                    if (sphere_test(sr, obg.p[0], obg.p[1], obg.p[2], obg.p[3], testMinT, t)) {
                        float wt = world_t(obg, ro, time, to, td, t);
                        if (wt < minT) {
                            if (RT_ISANY) { fin = true; break; }
                            minT = wt;
                            has = true;
                            out.kind_index = (RT_KIND_SPHERE << 28) | tp.index;
                            out.tri = 0;
                            tmax = minT;
                        }
                    }
                } else {   // RT_KIND_TRIANGLE; or RT_KIND_MESHTRI, a triangle of a one-leaf mesh written into the walk (rt_device.h)
                    ActiveRay tr;
                    tr.set(to, td);
                    float t, bb, gg;
#if RT_FLAT_TOP && RT_INLINE_LEAF_MESH
                    const bool of_mesh = tp.kind == RT_KIND_MESHTRI;
                    const uint32_t mesh_tri_slot = it0.w;
                    const float* tq = of_mesh ? reinterpret_cast<const float*>(&S.tris[mesh_tri_slot]) : &obg.p[0];
#else
                    const bool of_mesh = false;
                    const uint32_t mesh_tri_slot = 0;
                    const float* tq = &obg.p[0];
#endif
                    if (tri_test(tr, tq[0], tq[1], tq[2], tq[3], tq[4], tq[5], tq[6], tq[7], tq[8], testMinT, t, bb, gg)) {
                        float wt = world_t(obg, ro, time, to, td, t);
                        if (wt < minT) {
                            if (RT_ISANY) { fin = true; break; }
                            minT = wt;
                            has = true;
                            out.kind_index = ((of_mesh ? RT_KIND_MESH : RT_KIND_TRIANGLE) << 28) | tp.index;
                            out.tri = of_mesh ? mesh_tri_slot : 0u;
                            out.beta = of_mesh ? bb : out.beta;
                            out.gamma = of_mesh ? gg : out.gamma;
                            tmax = minT;
                        }
                    }
                }
            }
            RT_STAT(const unsigned long long y2 = wall_clock64(); if ((int)lane == __ffsll((unsigned long long)__ballot(1)) - 1) st.acc->y_prims += y2 - y1;)
            if (fin) { finished = true; break; }
            if (!entered) {
                cur = st.pop(tmax, cur_t);
                // An empty stack is the end of the ray: seen here, not in the next round's pass (the ray kept its lane
                // for a round of its wave with nothing to do in it)
                if (RT_EARLY_FINISH && cur == RT_LINK_DONE) finished = true;
            }
        } while (0);
        RT_CLK(xb);
        RT_TL(u5 = (unsigned long long)wall_clock64();)

        if (alive && finished) {
            if (helper) {
                // the recorded candidates have reached L2 (the owner, in this workgroup, reads them past L1) before the count drops
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                atomicSub(&st.out_cnt[home], 1u);
            } else {
                RT_RAYHIST({ if (ANY_ONLY == RT_HIST_ANY) { atomicAdd(&g_rayhist[0][32 - __clz(st.acc->ray_leaves)], 1ull); atomicAdd(&g_rayhist[1][32 - __clz(st.acc->ray_nodes)], 1ull); } })
                io.store(item, RT_ISANY ? fin : has, out, minT);
                RT_TL(if (MODE == RT_TIMELINE && exhausted) {
                    const uint32_t k_ = atomicAdd(&g_tl_log_n, 1u);
                    if (k_ < RT_TL_LOG) {
                        const unsigned long long now_ = (unsigned long long)wall_clock64();
                        uint32_t* o_ = g_tl_log[k_];
                        o_[0] = tlr_nodes; o_[1] = tlr_leaves; o_[2] = (uint32_t)(now_ - tlr_t0); o_[3] = tlr_posts; o_[4] = tlr_waits; o_[5] = tlr_rounds; o_[6] = (uint32_t)(now_ - *(volatile unsigned long long*)&g_tl_t0); o_[7] = item;
                    }
                })
            }
            RT_STAT(st.acc->ray_leaves = 0; st.acc->ray_nodes = 0;)
            alive = false;
            helper = false;
            in_mesh = false;
            donated = false;
            no_donate = false;
            cur = RT_LINK_DONE;
            st.sp = 0;
        }
        if constexpr (MEGA) io.flush(lane, lanes_below);   // finished rays: hits to the shading wave, the others off the workgroup's count
        RT_CLK(xc);
        RT_TL(if (MODE == RT_TIMELINE && exhausted) { u6 = (unsigned long long)wall_clock64(); tlu[4] += u5 - u4; tlu[5] += u6 - u5; })
        RT_STAT(st.acc->x_block += xb - xa; st.acc->x_retire += xc - xb; st.acc->x_passes++;)
        if (tpass >= RT_REFILL_PASSES || (exhausted && !MEGA)) break;
        // ---- refill idle lanes
        bool refilled = false;
        if constexpr (MEGA) {
            const uint64_t idle = __ballot(!alive);
            uint32_t n_idle = (uint32_t)__popcll(idle);
            stale = n_idle != 0 ? stale + 1 : 0;
            if (n_idle >= (io.pref_any ? (uint32_t)RT_REFILL_MIN_ANY : (uint32_t)RT_REFILL_MIN) || stale > RT_REFILL_PATIENCE) {
                stale = 0;
                // every closest-hit ray in flight must find room in this wave's ring to the shading wave when it finishes
                const uint32_t in_flight = (uint32_t)__popcll(__ballot(alive && !helper && !lane_any));
                const uint32_t room = io.shade_room();
                n_idle = room > in_flight ? min(n_idle, room - in_flight) : 0u;
                const uint32_t rank = (uint32_t)__popcll(idle & lanes_below);
                // Sources, in this wave's order of preference.  A wave that prefers shadow slots takes them, then child rays,
                // then primaries; the others child rays, primaries, then shadow slots — so a wave mostly holds rays of
                // one kind (neighbouring slots towards one light; or closest-hit rays of neighbouring pixels) and only
                // mixes when its own kind has run dry.
                uint32_t it = 0;
                const uint32_t my_rank = !alive ? rank : 0xFFFFFFFFu;
                uint32_t n_ring = 0;
                if (n_idle && io.pref_any) n_ring += io.ring_take(true, n_idle, lane, my_rank, 0u, it);
                // (strict roles: a shadow wave takes closest-hit work only once the primaries are gone)
                const bool other_kind_ok = !io.strict || !io.pref_any || exhausted;
                if (n_idle > n_ring && other_kind_ok) n_ring += io.ring_take(false, n_idle - n_ring, lane, my_rank, n_ring, it);
                uint32_t n_prim = (exhausted || n_idle <= n_ring || !other_kind_ok) ? 0u : n_idle - n_ring;
                bool mine = !alive && rank < n_ring;
                if (n_prim != 0) {
                    const uint32_t part = (part0 + parts_done) % parts;
                    const uint32_t pb0 = (uint32_t)((uint64_t)n_blocks * part / parts);
                    const uint32_t pnb = (uint32_t)((uint64_t)n_blocks * (part + 1u) / parts) - pb0, p_padded = pnb * 64u;
                    io.count(n_prim, lane);                  // counted before they are taken: the count never runs low
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(cursor + part * RT_CURSOR_STRIDE, n_prim);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base + n_prim >= p_padded) {
                        parts_done++;
                        if (lane == 0) atomicMax(&st.blk[4], parts_done);
                        if (parts_done >= parts) { exhausted = true; RT_STAT(st.acc->clk_exhausted = wall_clock64();) }
                        else stale = RT_REFILL_PATIENCE + 1u;
                    }
                    bool valid = false;
                    if (!alive && rank >= n_ring && rank < n_idle) {
                        const uint32_t c = base + (rank - n_ring);
                        if (c < p_padded) {
                            const uint32_t strip = c >> strip_shift_, within = c & (STRIP - 1u);
                            const uint32_t i = begin + (pb0 + strip % pnb) * 64u + (strip / pnb) * STRIP + within;
                            if (i < end) { valid = true; it = i; }
                        }
                    }
                    io.uncount(n_prim - (uint32_t)__popcll(__ballot(valid)), lane);
                    mine = mine || valid;
                }
                if (!io.pref_any && n_idle > n_ring + n_prim && (!io.strict || exhausted)) {   // nothing else left for these lanes: shadow slots
                    const uint32_t k = io.ring_take(true, n_idle - n_ring - n_prim, lane, my_rank, n_ring + n_prim, it);
                    mine = mine || (!alive && rank >= n_ring + n_prim && rank < n_ring + n_prim + k);
                }
                int got = 0;
                if (mine) {
                    item = it;
                    got = io.load(it, ro, rd, time, minT, has, out, lane_any);
                }
                io.loaded(mine, got, lane);                  // rays that needed no traversal are finished already
                traced += __popcll(__ballot(got != 0));
                if (traced_any) *traced_any += __popcll(__ballot(got != 0 && lane_any));
                refilled = __any(got == 2);
                if (got == 2) {
                    RT_TL(tlr_t0 = (unsigned long long)wall_clock64(); tlr_nodes = tlr_leaves = tlr_posts = tlr_waits = tlr_rounds = 0;)
                    fin = false;
                    alive = true;
                    helper = false;
                    home = st.slot;
                    st.sp = 0;
                    in_mesh = false;
                    donated = false;
                    no_donate = false;
                    qn = 0;
                    ar.set(ro, rd);
                    tmax = minT;
                    cur = RT_LINK_DONE;
                    if (S.top_root_link != RT_LINK_DONE) {
                        float te;
                        if (box_test(ar, S.top_rmin[0], S.top_rmin[1], S.top_rmin[2], S.top_rmax[0], S.top_rmax[1], S.top_rmax[2], tmax, te)) {
#if RT_FLAT_TOP
                            cur = RT_LINK_TOPSCAN;
                            top_slot = 1;                        // (item 0 is the root node: its box is the one just tested)
#else
                            cur = S.top_root_link;
#endif
                            cur_t = te;
                        }
                    }
                }
            }
        } else
        if (!exhausted) {
            const uint64_t idle = __ballot(!alive);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            // A wave that keeps a few idle lanes for RT_REFILL_PATIENCE rounds refills them anyway: they
            // would otherwise wait for the slowest rays, and the attempt is also how the wave learns that
            // the cursor is drained (from then on idle lanes help the busy ones, see subtree donation).
            if (tpass == 0) stale = n_idle != 0 ? stale + 1 : 0;
            if (n_idle >= REFILL_MIN || (tpass == 0 && stale > RT_REFILL_PATIENCE)) {
                stale = 0;
                // the partition this workgroup draws from: its own (blockIdx % parts: workgroups that share an
                // XCD, hence an L2, share a partition), then the others in turn
                const uint32_t part = (part0 + parts_done) % parts;
                uint32_t pb0, pnb, pa0 = 0, pc_n = 0;         // this partition's blocks; MIXED: pb0 / pc_n closest-hit blocks, pa0 first any-hit block
                if constexpr (MIXED) {
                    const uint32_t nb1 = io.first_blocks(), nba = n_blocks - nb1;
                    pb0 = (uint32_t)((uint64_t)nb1 * part / parts);
                    pc_n = (uint32_t)((uint64_t)nb1 * (part + 1u) / parts) - pb0;
                    pa0 = (uint32_t)((uint64_t)nba * part / parts);
                    pnb = pc_n + ((uint32_t)((uint64_t)nba * (part + 1u) / parts) - pa0);
                } else {
                    pb0 = (uint32_t)((uint64_t)n_blocks * part / parts);
                    pnb = (uint32_t)((uint64_t)n_blocks * (part + 1u) / parts) - pb0;
                }
                const uint32_t p_padded = pnb * 64u;
                uint32_t base = 0;
                if (first_fill) {                            // (all 64 lanes are idle: nothing has been loaded yet)
                    base = ((blockIdx.x / parts) * waves_per_wg + (st.slot >> 6)) * 64u;
                    first_fill = false;
                } else {
                    if (lane == 0) base = atomicAdd(cursor + part * RT_CURSOR_STRIDE, n_idle);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + static_items(part);
                }
                if (base + n_idle >= p_padded) {
                    parts_done++;
                    if (LINGER && lane == 0) atomicMax(&st.blk[4], parts_done);
                    if (parts_done >= parts) { exhausted = true; RT_STAT(st.acc->clk_exhausted = wall_clock64();) }
                    else stale = RT_REFILL_PATIENCE + 1u;     // lanes left empty try the next partition at once
                }
                int got = 0;
                if (!alive) {
                    // cursor value -> item: strip-interleaved over the partition's 64-item blocks (see above)
                    const uint32_t c = base + (uint32_t)__popcll(idle & lanes_below);
                    if (c < p_padded) {
                        uint32_t i;
                        bool valid;
                        if constexpr (MIXED) {
                            // first the closest-hit items, strip-interleaved over their blocks; then the any-hit
                            // items block by block (item ids: the first kind's padded count, then the second kind) —
                            // the cheap any-hit rays are what is in flight when the cursor runs dry
                            if (c < pc_n * 64u) {
                                const uint32_t strip = c >> strip_shift_, within = c & (STRIP - 1u);
                                i = (pb0 + strip % pc_n) * 64u + (strip / pc_n) * STRIP + within;
                                valid = i < io.first_count();
                            } else {
                                i = (io.first_blocks() + pa0) * 64u + (c - pc_n * 64u);
                                valid = i < end;
                            }
                            i += begin;
                        } else {
                            const uint32_t strip = c >> strip_shift_, within = c & (STRIP - 1u);
                            i = begin + (pb0 + strip % pnb) * 64u + (strip / pnb) * STRIP + within;
                            valid = i < end;
                        }
                        if (valid) {
                            item = i;
                            if constexpr (MIXED) got = io.load(i, ro, rd, time, minT, has, out, lane_any);
                            else got = io.load(i, ro, rd, time, minT, has, out);
                        }
                    }
                }
                traced += __popcll(__ballot(got != 0));
                if (MIXED && traced_any) *traced_any += __popcll(__ballot(got != 0 && lane_any));
                refilled = __any(got == 2);
                if (got == 2) {
                    RT_TL(tlr_t0 = (unsigned long long)wall_clock64(); tlr_nodes = tlr_leaves = tlr_posts = tlr_waits = tlr_rounds = 0;)
                    fin = false;
                    alive = true;
                    helper = false;
                    home = st.slot;
                    st.sp = 0;
                    in_mesh = false;
                    donated = false;
                    no_donate = false;
                    qn = 0;
                    ar.set(ro, rd);
                    tmax = minT;
                    cur = RT_LINK_DONE;
                    if (S.top_root_link != RT_LINK_DONE) {
                        float te;
                        if (box_test(ar, S.top_rmin[0], S.top_rmin[1], S.top_rmin[2], S.top_rmax[0], S.top_rmax[1], S.top_rmax[2], tmax, te)) {
#if RT_FLAT_TOP
                            cur = RT_LINK_TOPSCAN;
                            top_slot = 1;                        // (item 0 is the root node: its box is the one just tested)
#else
                            cur = S.top_root_link;
#endif
                            cur_t = te;
                        }
                    }
                }
            }
        }
        RT_STAT(st.acc->x_refill += wall_clock64() - xc;)
        if (!refilled) break;
        }
        RT_TL(if (MODE == RT_TIMELINE && exhausted && any_alive) { const unsigned long long q5 = (unsigned long long)wall_clock64(); tlt[0] += q1 - q0; tlt[1] += q2 - q1; tlt[2] += q3 - q2; tlt[3] += q4 - q3; tlt[4] += q5 - q4; tlt[5]++; })
        RT_STAT({ unsigned long long b = (c0 - st.acc->clk_start) / RT_HIST_TICKS; if (b >= RT_HIST_BINS) b = RT_HIST_BINS - 1;
                  if (ANY_ONLY != RT_HIST_ANY) { st.acc->r_node_w = 0; st.acc->r_tri_w = 0; }
                  if (st.acc->r_node_w) { atomicAdd(&g_hist[b][0], st.acc->r_node_w); atomicAdd(&g_hist[b][1], st.acc->r_node_l); }
                  if (st.acc->r_tri_w) { atomicAdd(&g_hist[b][2], st.acc->r_tri_w); atomicAdd(&g_hist[b][3], st.acc->r_tri_l); }
                  st.acc->r_node_w = st.acc->r_node_l = st.acc->r_tri_w = st.acc->r_tri_l = 0; })
        RT_STAT({ if (exhausted) { const unsigned long long c5 = wall_clock64(); st.acc->tl_rounds++; st.acc->tl_sched += c1 - c0; st.acc->tl_node += c2 - c1; st.acc->tl_tri += c3 - c2; st.acc->tl_big += c4 - c3; st.acc->tl_trans += c5 - c4; } })
        RT_STAT({ const unsigned long long c5 = wall_clock64(); st.acc->t_sched += c1 - c0; st.acc->t_node += c2 - c1; st.acc->t_tri += c3 - c2; st.acc->t_big += c4 - c3; st.acc->t_trans += c5 - c4; })
    }
    RT_TL(if (MODE == RT_TIMELINE && lane == 0) { for (int k_ = 0; k_ < 6; k_++) atomicAdd(&g_tl_tail[k_], tlt[k_]); for (int k_ = 0; k_ < 6; k_++) atomicAdd(&g_tl_tail2[k_], tlu[k_]); })
}
#undef RT_ISANY
