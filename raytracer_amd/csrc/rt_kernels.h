// rt_kernels.h — the wavefront pipeline around the traversal:
//   raygen  -> [ trace_closest -> shade (spawns children + shadow slots) -> trace_shadow ] per depth
//           -> combine (deepest generation first) -> resolve (sum samples, 8-bit)
//
// The reference evaluates ComputeColor/ApplyShading recursively
// (raytracer.cpp:382-401, 1248-1436).  Here every ray of the recursion tree is
// a RECORD in a pool; generation g holds the rays of depth g.  Because the
// point-light path draws no random numbers (PerturbReflection returns early for
// roughness <= 0, raytracer.cpp:1441), the tree's rays can be traced in any
// order; the colour of a record is then computed bottom-up with exactly the
// reference's expression order, so sums round identically.
#pragma once
#include "rt_traverse.h"

#define RT_MAX_GEN 32
#define RT_NO_CHILD 0xFFFFFFFFu

// record flags (DLink.w)
#define RF_MISS        1u
#define RF_LIGHTS      2u    // local lighting is evaluated (not an exiting dielectric)
#define RF_DEAD        4u    // padding lane of an edge tile: no ray
#define RF_STRIDE_SHIFT 8     // link.w bits 8..23: distance between the record's shadow slots of consecutive lights (k_shade)

struct Pool {
    float4*   ray0;        // origin.xyz, time
    float4*   ray1;        // direction.xyz, live (bits: 1 = live)
    uint4*    hit0;        // t bits, kind_index, tri, -
    float2*   hit1;        // beta, gamma
    float4*   nrm;         // world normal (HitRecord::normal), material id bits
    uint4*    link;        // child0 (reflect), child1 (refract), shadow_base, flags
    float4*   color;       // ComputeColor result
    uint32_t* shadow_rec;  // per shadow slot: owning record
    uint8_t*  shadow_lit;  // per shadow slot: 1 = !InShadow
    uint32_t  cap_records, cap_shadow;
};

struct Control {
    uint32_t gstart[RT_MAX_GEN + 2];   // generation g = records [gstart[g], gstart[g+1])
    uint32_t sstart[RT_MAX_GEN + 2];   // shadow slots of generation g
    uint32_t n_records, n_shadow;      // allocation cursors
    uint32_t cursor_closest[RT_PARTS * RT_CURSOR_STRIDE], cursor_shadow[RT_PARTS * RT_CURSOR_STRIDE];   // one per partition, a line apart
    uint32_t overflow;                 // 1: record pool, 2: shadow pool, 4: traversal stack
    uint32_t pad;
    unsigned long long rays_primary, rays_secondary, rays_shadow;
    unsigned long long dbg[16];        // RT_TRACE_STATS builds only
};

// ----------------------------------------------------------------- pixel order
// A chunk is a run of 8x8 pixel tiles of the render rectangle; idx enumerates
// tile-major so that the 64 lanes of a wave start as one 8x8 pixel block.
struct ChunkGeom {
    int x0, y0, w, h;        // render rectangle inside the image
    int tiles_per_row;
    uint32_t tile_begin;     // first tile of this chunk
    uint32_t n_idx;          // 64 * tiles in this chunk
    int row_parts, row_part; // only tile rows ty with ty % row_parts == row_part are rendered (multi-GPU interleave)
};
__device__ __forceinline__ bool idx_to_pixel(const ChunkGeom& g, uint32_t idx, int& px, int& py) {
    uint32_t tile = g.tile_begin + (idx >> 6);
    uint32_t in = idx & 63u;
    int tx = (int)(tile % (uint32_t)g.tiles_per_row);
    int ty = (int)(tile / (uint32_t)g.tiles_per_row) * g.row_parts + g.row_part;
    int lx = tx * 8 + (int)(in & 7u), ly = ty * 8 + (int)(in >> 3);
    px = g.x0 + lx;
    py = g.y0 + ly;
    return lx < g.w && ly < g.h;
}

#include "rt_mt.h"   // the reference's per-pixel generator, restated (host-testable)
#include "rt_shade.h" // ray generation and shading arithmetic (host-testable)

__device__ __forceinline__ void write_ray(const Pool& P, uint32_t r, V3 o, V3 d, float time, bool live) {
    P.ray0[r] = make_float4(o.x, o.y, o.z, time);
    P.ray1[r] = make_float4(d.x, d.y, d.z, __uint_as_float(live ? 1u : 0u));
}

// Start of a chunk: generation 0 = the primary records.  Done by one thread of the ray-generation launch (nothing reads
// the control block before the first trace launch, which follows it on the stream).
__device__ __forceinline__ void init_chunk(Control* ctl, uint32_t n_primary) {
    for (int g = 0; g < RT_MAX_GEN + 2; g++) { ctl->gstart[g] = n_primary; ctl->sstart[g] = 0; }
    ctl->gstart[0] = 0;
    ctl->n_records = n_primary;
    ctl->n_shadow = 0;
    for (int p = 0; p < RT_PARTS; p++) { ctl->cursor_closest[p * RT_CURSOR_STRIDE] = 0; ctl->cursor_shadow[p * RT_CURSOR_STRIDE] = 0; }
    ctl->overflow = 0;
}

// num_samples == 1: one thread per pixel (raytracer.cpp:44-74 with N = 1: the
// shuffle of one element draws nothing, then five draws).
__global__ void __launch_bounds__(256) k_raygen_single(DCamera cam, ChunkGeom g, Pool P, Control* ctl, uint32_t n_primary) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && ctl) init_chunk(ctl, n_primary);
    if (idx >= g.n_idx) return;
    int px, py;
    if (!idx_to_pixel(g, idx, px, py)) {
        write_ray(P, idx, mk(0, 0, 0), mk(0, 0, 0), 0.f, false);
        return;
    }
    uint32_t u[5];
    mt_first5((uint32_t)(py * cam.image_w + px), u);
    // samples_per_side == 1: (0 + xi) / 1
    float sps = (float)cam.sps;
    float jx = (0 + u32_to_unit(u[0])) / sps;
    float jy = (0 + u32_to_unit(u[1])) / sps;
    float au = (0 + u32_to_unit(u[2])) / sps;
    float av = (0 + u32_to_unit(u[3])) / sps;
    float time = u32_to_unit(u[4]);
    V3 o, d;
    compute_ray(cam, px, py, jx, jy, au, av, o, d);
    write_ray(P, idx, o, d, time, true);
}

// num_samples > 1: one thread per pixel walks its sample sequence; the rays of
// pixel idx are records idx*N .. idx*N+N-1.
__global__ void __launch_bounds__(256) k_raygen_multi(DCamera cam, ChunkGeom g, Pool P, uint32_t* mt_scratch,
                                                      uint32_t* perm_scratch, size_t stride, Control* ctl, uint32_t n_primary) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && ctl) init_chunk(ctl, n_primary);
    if (idx >= g.n_idx) return;
    const int N = cam.num_samples, sps = cam.sps;
    int px, py;
    if (!idx_to_pixel(g, idx, px, py)) {
        for (int s = 0; s < N; s++) write_ray(P, idx * N + s, mk(0, 0, 0), mk(0, 0, 0), 0.f, false);
        return;
    }
    MtGlobal gen;
    gen.mt = mt_scratch + idx;
    gen.stride = stride;
    gen.seed((uint32_t)(py * cam.image_w + px));
    uint32_t* perm = perm_scratch + idx;
    mt_shuffle_identity(gen, perm, stride, N);
    const float fs = (float)sps;
    for (int s = 0; s < N; s++) {
        int sx = s % sps, sy = s / sps;
        float jx = (sx + gen.uniform()) / fs;
        float jy = (sy + gen.uniform()) / fs;
        int ai = (int)perm[(size_t)s * stride];
        int ax = ai % sps, ay = ai / sps;
        float au = (ax + gen.uniform()) / fs;
        float av = (ay + gen.uniform()) / fs;
        float time = gen.uniform();
        V3 o, d;
        compute_ray(cam, px, py, jx, jy, au, av, o, d);
        write_ray(P, idx * N + s, o, d, time, true);
    }
}

// ------------------------------------------------------------- trace kernels
struct WaveWork {
    uint32_t lane, wave;
    __device__ __forceinline__ WaveWork() : lane(threadIdx.x & 63u), wave(threadIdx.x >> 6) {}
    // Persistent-thread work fetch: one atomic per wave per 64 items.
    __device__ __forceinline__ uint32_t grab(uint32_t* cursor) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64u);
        return __builtin_amdgcn_readfirstlane(base);
    }
};

// LDS of a workgroup that runs the trace loop: per wave RT_STACK_LDS stack entries + RT_LEAF_QUEUE postponed
// leaves (8 bytes each per lane); per thread the owner-slot words of subtree donation; the task board.
#define RT_LDS_ENTRIES (RT_STACK_LDS + RT_LEAF_QUEUE)
template <int THREADS>
struct TraceLds {
    uint2    stack[(THREADS / 64) * RT_LDS_ENTRIES * 64];
    uint32_t cand_cnt[THREADS];
    uint32_t out_cnt[THREADS];
    uint32_t board[(THREADS / 64) * RT_BOARD_PER_WAVE * 16];
    uint32_t blk[8];
    uint32_t cursor[THREADS / 64];       // rt_ordered.h: per-wave item cursor
    uint8_t  cancel[THREADS];
};
template <int THREADS>
__device__ __forceinline__ TraceStack make_stack(TraceLds<THREADS>& L, uint2* spill, uint4* cand, const WaveWork& ww, uint32_t stack_cap) {
    // zero the workgroup-wide words once (all threads of the block call this together at kernel start)
    for (uint32_t i = threadIdx.x; i < (THREADS / 64) * RT_BOARD_PER_WAVE * 16; i += THREADS) L.board[i] = 0;
    L.cand_cnt[threadIdx.x] = 0;
    L.out_cnt[threadIdx.x] = 0;
    L.cancel[threadIdx.x] = 0;
    if (threadIdx.x < 8) L.blk[threadIdx.x] = 0;
    if (threadIdx.x < THREADS / 64) L.cursor[threadIdx.x] = 0;
    __syncthreads();
    TraceStack st;
    st.lds = L.stack + ww.wave * (RT_LDS_ENTRIES * 64);
    st.q = st.lds + RT_STACK_LDS * 64;
    st.cand_cnt = L.cand_cnt;
    st.out_cnt = L.out_cnt;
    st.cancel = L.cancel;
    st.board = L.board;
    st.blk = L.blk;
    st.n_board = (THREADS / 64) * RT_BOARD_PER_WAVE;
    st.slot = threadIdx.x;
    st.cand = cand + (size_t)blockIdx.x * THREADS * RT_CAND_MAX;
    size_t wave_slot = (size_t)blockIdx.x * (THREADS / 64) + ww.wave;
    st.cap = stack_cap;
    st.spill = spill + wave_slot * (size_t)((stack_cap - RT_STACK_LDS) * 64);
    st.lane = ww.lane;
    st.sp = 0;
    st.overflow = 0;
#ifdef RT_TRACE_STATS
    st.acc = nullptr;
#endif
    return st;
}

#ifdef RT_TRACE_STATS
__device__ unsigned long long g_wave_dbg[8192][16];   // per wave of k_trace_closest (diagnostic build)
__device__ unsigned long long g_ydbg[8];                // inside the transition block, summed over waves

// diagnostic build: totals into ctl->dbg, per-wave breakdown into g_wave_dbg (printed by render_impl)
__device__ void report_trace_stats(Control* ctl, const TraceStatsAcc& acc, unsigned long long traced, unsigned long long clk_start,
                                   const TraceStack& st, const WaveWork& ww) {
    atomicAdd(&ctl->dbg[0], acc.wave_node_steps); atomicAdd(&ctl->dbg[1], acc.lane_node_steps);
    atomicAdd(&ctl->dbg[2], acc.wave_tri_steps); atomicAdd(&ctl->dbg[3], acc.lane_tri_steps);
    atomicAdd(&ctl->dbg[4], acc.wave_outer);
    const uint32_t wg = blockIdx.x * (blockDim.x / 64) + ww.wave;
    if (ww.lane == 0) {
        atomicAdd(&ctl->dbg[5], traced);
        const unsigned long long clk_end = wall_clock64();
        if (wg < 8192) {
            unsigned long long* o = g_wave_dbg[wg];
            o[0] = clk_end - clk_start; o[1] = acc.clk_exhausted ? acc.clk_exhausted - clk_start : 0; o[2] = acc.t_sched; o[3] = acc.t_node;
            o[4] = acc.t_tri; o[5] = acc.t_big; o[6] = acc.t_trans; o[7] = 0; o[8] = 0; o[9] = 0;
            atomicAdd(&ctl->dbg[6], acc.x_block); atomicAdd(&ctl->dbg[7], acc.x_retire); atomicAdd(&ctl->dbg[8], acc.x_refill); atomicAdd(&ctl->dbg[9], acc.x_passes); atomicAdd(&ctl->dbg[10], acc.t_trans); atomicAdd(&ctl->dbg[11], acc.wave_outer ? 0ull : 0ull);
            o[10] = acc.n_donations; o[11] = acc.n_coop; o[12] = traced; o[13] = 0; o[14] = 0; o[15] = clk_start;
        }
    }
    if (acc.y_lbody) {
        atomicAdd(&g_ydbg[0], acc.y_fin); atomicAdd(&g_ydbg[1], acc.y_prims); atomicAdd(&g_ydbg[2], acc.y_witers); atomicAdd(&g_ydbg[3], acc.y_liters);
        atomicAdd(&g_ydbg[4], acc.y_wbody); atomicAdd(&g_ydbg[5], acc.y_lbody);
    }
    // step counts live in whichever lane was first active: sum them per wave (after lane 0's plain stores above)
    __threadfence();
    if (wg < 8192) {
        unsigned long long* o = g_wave_dbg[wg];
        if (acc.wave_outer) atomicAdd(&o[7], acc.wave_outer);
        if (acc.wave_node_steps) { atomicAdd(&o[8], acc.wave_node_steps); atomicAdd(&o[13], acc.lane_node_steps); }
        if (acc.wave_tri_steps) { atomicAdd(&o[9], acc.wave_tri_steps); atomicAdd(&o[14], acc.lane_tri_steps); }
    }
}
#endif

// FindClosestHit's search part (raytracer.cpp:405-444) for generation `gen`.
struct ClosestIO {
    const DScene& S;
    const Pool& P;
    __device__ __forceinline__ int load(uint32_t r, V3& o, V3& d, float& time, float& minT, bool& has, HitOut& h) const {
        const float4 r0 = P.ray0[r], r1 = P.ray1[r];
        h.kind_index = 0xFFFFFFFFu;
        h.tri = 0;
        h.beta = 0.f;
        h.gamma = 0.f;
        if (__float_as_uint(r1.w) == 0u) {   // padding record of an edge tile
            P.hit0[r] = make_uint4(__float_as_uint(RT_FLT_MAX), 0xFFFFFFFFu, 0u, 0u);
            P.hit1[r] = make_float2(0.f, 0.f);
            return 0;
        }
        o = mk(r0.x, r0.y, r0.z);
        d = mk(r1.x, r1.y, r1.z);
        time = r0.w;
        minT = RT_FLT_MAX;
        has = false;
        for (uint32_t i = 0; i < S.n_planes; i++) {   // raytracer.cpp:423-435
            float t;
            if (plane_test(o, d, S.planes[i], minT, t)) {
                minT = t;
                has = true;
                h.kind_index = (RT_KIND_PLANE << 28) | i;
            }
        }
        return 2;
    }
    __device__ __forceinline__ void store(uint32_t r, bool has, const HitOut& h, float minT) const {
        P.hit0[r] = make_uint4(__float_as_uint(minT), has ? h.kind_index : 0xFFFFFFFFu, h.tri, 0u);
        P.hit1[r] = make_float2(h.beta, h.gamma);
    }
};

template <bool NEAR>
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_MIN_WAVES) k_trace_closest(DScene S, Pool P, Control* ctl, int gen, uint2* spill, uint4* cand) {
    __shared__ TraceLds<RT_TRACE_BLOCK> s_lds;
    WaveWork ww;
    TraceStack st = make_stack(s_lds, spill, cand, ww, S.stack_cap);
    const uint32_t begin = ctl->gstart[gen], end = ctl->gstart[gen + 1];
#ifdef RT_TRACE_STATS
    TraceStatsAcc acc = {};
    st.acc = &acc;
    const unsigned long long clk_start = wall_clock64();
    acc.clk_start = clk_start;
#endif
    unsigned long long traced = 0;   // wave-uniform
    ClosestIO io{S, P};
    trace_persistent<0, true, NEAR>(S, io, begin, end, ctl->cursor_closest, st, traced);
    if (st.overflow) atomicOr(&ctl->overflow, (st.overflow & 48u) | ((st.overflow & ~48u) ? 4u : 0u));
#ifdef RT_TRACE_STATS
#ifndef RT_STATS_SHADOW
    report_trace_stats(ctl, acc, traced, clk_start, st, ww);
#endif
#endif
    if (traced && ww.lane == 0) atomicAdd(gen == 0 ? &ctl->rays_primary : &ctl->rays_secondary, traced);
}

// InShadow, raytracer.cpp:1008-1244, one shadow slot per (record, point light).
struct ShadowIO {
    const DScene& S;
    const Pool& P;
    __device__ __forceinline__ int load(uint32_t i, V3& so, V3& sd, float& time, float& minT, bool& has, HitOut& h) const {
        const uint32_t r = P.shadow_rec[i];
        if (r == 0xFFFFFFFFu) return 0;                        // slot of a block that overflowed the pool (k_shade)
        const uint4 lkr = P.link[r];
        const uint32_t light = (i - lkr.z) / ((lkr.w >> RF_STRIDE_SHIFT) & 0xFFFFu);
        const float4 r0 = P.ray0[r], r1 = P.ray1[r], nn = P.nrm[r];
        const float tHit = __uint_as_float(P.hit0[r].x);
        V3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z);
        V3 x = o + d * tHit;                                   // raytracer.cpp:447
        V3 n = mk(nn.x, nn.y, nn.z);
        V3 n_shading = (dot(n, d) < 0.0f) ? n : n * -1.0f;     // raytracer.cpp:1258-1265
        const DPointLight L = S.point_lights[light];
        V3 toLight = mk(L.px, L.py, L.pz) - x;
        float dist = vlen(toLight);
        so = x + n_shading * S.shadow_eps;
        sd = vdiv(toLight, dist);
        time = r0.w;
        minT = dist;
        has = false;
        for (uint32_t k = 0; k < S.n_planes; k++) {            // raytracer.cpp:1022-1028
            float t;
            if (plane_test(so, sd, S.planes[k], minT, t)) { P.shadow_lit[i] = 0; return 1; }
        }
        return 2;
    }
    __device__ __forceinline__ void store(uint32_t i, bool shadow, const HitOut&, float) const { P.shadow_lit[i] = shadow ? 0 : 1; }
};

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_MIN_WAVES) k_trace_shadow(DScene S, Pool P, Control* ctl, int gen, uint2* spill, uint4* cand) {
    __shared__ TraceLds<RT_TRACE_BLOCK> s_lds;
    WaveWork ww;
    TraceStack st = make_stack(s_lds, spill, cand, ww, S.stack_cap);
    const uint32_t begin = ctl->sstart[gen], end = ctl->sstart[gen + 1];
#ifdef RT_TRACE_STATS
    TraceStatsAcc acc = {};
    st.acc = &acc;
    const unsigned long long clk_start = wall_clock64();
    acc.clk_start = clk_start;
#endif
    unsigned long long traced = 0;
    ShadowIO io{S, P};
    trace_persistent<1, true, false>(S, io, begin, end, ctl->cursor_shadow, st, traced);
#if defined(RT_TRACE_STATS) && defined(RT_STATS_SHADOW)
    report_trace_stats(ctl, acc, traced, clk_start, st, ww);
#endif
    if (st.overflow) atomicOr(&ctl->overflow, (st.overflow & 48u) | ((st.overflow & ~48u) ? 4u : 0u));
    if (traced && ww.lane == 0) atomicAdd(&ctl->rays_shadow, traced);
}

// One launch for two independent sets of rays that both come out of k_shade(g): the shadow slots of generation g
// and the child rays of generation g + 1 (trace_persistent mode 2) — one end game instead of two per bounce.
struct MixedIO {
    ClosestIO c;
    ShadowIO s;
    uint32_t cbegin, n_c, sbegin;      // records [cbegin, cbegin + n_c), shadow slots from sbegin
    __device__ __forceinline__ uint32_t first_blocks() const { return (n_c + 63u) / 64u; }
    __device__ __forceinline__ uint32_t first_count() const { return n_c; }
    __device__ __forceinline__ int load(uint32_t i, V3& o, V3& d, float& time, float& minT, bool& has, HitOut& h, bool& any) const {
        const uint32_t pad = first_blocks() * 64u;
        if (i < pad) { any = false; return c.load(cbegin + i, o, d, time, minT, has, h); }
        any = true;
        return s.load(sbegin + (i - pad), o, d, time, minT, has, h);
    }
    __device__ __forceinline__ void store(uint32_t i, bool verdict, const HitOut& h, float minT) const {
        const uint32_t pad = first_blocks() * 64u;
        if (i < pad) c.store(cbegin + i, verdict, h, minT);
        else s.store(sbegin + (i - pad), verdict, h, minT);
    }
};

template <bool NEAR>
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_MIN_WAVES) k_trace_mixed(DScene S, Pool P, Control* ctl, int gen_shadow, int gen_closest, uint2* spill, uint4* cand) {
    __shared__ TraceLds<RT_TRACE_BLOCK> s_lds;
    WaveWork ww;
    TraceStack st = make_stack(s_lds, spill, cand, ww, S.stack_cap);
#ifdef RT_TRACE_STATS
    TraceStatsAcc acc = {};
    st.acc = &acc;
    acc.clk_start = wall_clock64();
#endif
    const uint32_t cbegin = ctl->gstart[gen_closest], n_c = ctl->gstart[gen_closest + 1] - cbegin;
    const uint32_t sbegin = ctl->sstart[gen_shadow], n_s = ctl->sstart[gen_shadow + 1] - sbegin;
    MixedIO io{ClosestIO{S, P}, ShadowIO{S, P}, cbegin, n_c, sbegin};
    unsigned long long traced = 0, traced_any = 0;   // wave-uniform
    trace_persistent<2, true, NEAR>(S, io, 0u, io.first_blocks() * 64u + n_s, ctl->cursor_closest, st, traced, &traced_any);
    if (st.overflow) atomicOr(&ctl->overflow, (st.overflow & 48u) | ((st.overflow & ~48u) ? 4u : 0u));
    if (ww.lane == 0) {
        if (traced_any) atomicAdd(&ctl->rays_shadow, traced_any);
        if (traced - traced_any) atomicAdd(gen_closest == 0 ? &ctl->rays_primary : &ctl->rays_secondary, traced - traced_any);
    }
}

// ------------------------------------------------------------------- shading
// Normal reconstruction of FindClosestHit, raytracer.cpp:447-517.
__device__ __forceinline__ V3 hit_normal(const DScene& S, uint32_t kind_index, uint32_t tri, float beta, float gamma,
                                         V3 hit_x, int& material_id) {
    const uint32_t kind = kind_index >> 28, index = kind_index & 0x0FFFFFFFu;
    V3 normal = mk(0, 0, 0);
    if (kind == RT_KIND_MESH) {
        const DObject& ob = S.objects[S.mesh_obj_base + index];
        material_id = ob.material_id;
        const DTriShade ts = S.tri_shade[tri];
        V3 objN;
        if (ob.flags & RT_OBJ_SMOOTH) {
            float4 a4 = S.vnormals[ts.i0 - 1], b4 = S.vnormals[ts.i1 - 1], c4 = S.vnormals[ts.i2 - 1];
            float alpha = 1.f - beta - gamma;
            objN = normalize((mk(a4.x, a4.y, a4.z) * alpha + mk(b4.x, b4.y, b4.z) * beta) + mk(c4.x, c4.y, c4.z) * gamma);
        } else {
            objN = mk(ts.nx, ts.ny, ts.nz);
        }
        const bool xf = (ob.flags & RT_OBJ_HAS_TRANSFORM) != 0;
        if (xf && (ob.flags & DOBJ_NEG_DET)) objN = objN * -1.0f;
        normal = xf ? normalize(xf_vector_transposed(ob.Minv, objN)) : objN;
    } else if (kind == RT_KIND_TRIANGLE) {
        const DObject& ob = S.objects[S.tri_obj_base + index];
        material_id = ob.material_id;
        V3 objN = mk(ob.p[9], ob.p[10], ob.p[11]);
        const bool xf = (ob.flags & RT_OBJ_HAS_TRANSFORM) != 0;
        normal = xf ? normalize(xf_vector_transposed(ob.Minv, objN)) : objN;
        if (xf && (ob.flags & DOBJ_NEG_DET)) normal = normal * -1.0f;
    } else if (kind == RT_KIND_SPHERE) {
        const DObject& ob = S.objects[S.sphere_obj_base + index];
        material_id = ob.material_id;
        V3 c = mk(ob.p[0], ob.p[1], ob.p[2]);
        const bool xf = (ob.flags & RT_OBJ_HAS_TRANSFORM) != 0;
        if (xf) {
            V3 objHit = xf_point(ob.Minv, hit_x);
            V3 objN = normalize(objHit - c);
            normal = normalize(xf_vector_transposed(ob.Minv, objN));
        } else {
            normal = normalize(hit_x - c);
        }
        if (xf && (ob.flags & DOBJ_NEG_DET)) normal = normal * -1.0f;
    } else {   // plane
        const DPlane& p = S.planes[index];
        material_id = p.material_id;
        normal = mk(p.nx, p.ny, p.nz);
    }
    return normal;
}

// Block-aggregated allocation of consecutive slots: every thread asks for `n`; one atomic per 256-thread block
// (same-address atomics from every wave cost ~5 ns each in sequence — with one per wave they, not the memory
// traffic, set k_shade's time).  Slots stay in thread order inside the block.  All threads of the block call it.
__device__ __forceinline__ uint32_t block_alloc(uint32_t* counter, uint32_t n, uint32_t* s_tmp /* [8] LDS */,
                                                uint32_t* block_base = nullptr, uint32_t* block_total = nullptr) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63) s_tmp[wave] = incl;            // wave totals
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t0 = s_tmp[0], t1 = s_tmp[1], t2 = s_tmp[2], t3 = s_tmp[3];
        const uint32_t total = t0 + t1 + t2 + t3;
        const uint32_t base = total ? atomicAdd(counter, total) : 0u;
        s_tmp[4] = base; s_tmp[5] = base + t0; s_tmp[6] = base + t0 + t1; s_tmp[7] = base + t0 + t1 + t2;
    }
    __syncthreads();
    const uint32_t r = s_tmp[4 + wave] + incl - n;
    if (block_base) { *block_base = s_tmp[4]; *block_total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3]; }
    __syncthreads();                               // s_tmp is reused by the next call
    return r;
}

// After the closest-hit search of generation `gen`: finish FindClosestHit
// (normal, material), decide which child rays ApplyShading will ask for
// (raytracer.cpp:1272-1352) and which shadow tests (raytracer.cpp:1371-1377),
// and append them to the pools.  This is the ray-compaction step: only live
// continuations get a slot (ballot/popcount prefix, one atomic per wave).
__global__ void __launch_bounds__(256) k_shade(DScene S, Pool P, Control* ctl, int gen) {
    __shared__ uint32_t s_tmp[8];
    const uint32_t begin = ctl->gstart[gen], end = ctl->gstart[gen + 1];
    const uint32_t span = end - begin;
    const uint32_t rounds = (span + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t off = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        const bool valid = off < span;
        const uint32_t r = begin + (valid ? off : 0);
        uint32_t n_child = 0, n_shadow = 0, flags = RF_MISS;
        ShadeCtx c;
        float time = 0.f;
        c.type = RT_MAT_NONE;
        c.tir = false;
        if (valid) {
            const float4 r0 = P.ray0[r], r1 = P.ray1[r];
            const uint4 h0 = P.hit0[r];
            time = r0.w;
            if (__float_as_uint(r1.w) == 0u) {
                flags = RF_MISS | RF_DEAD;
            } else if (h0.y != 0xFFFFFFFFu) {
                const float2 h1 = P.hit1[r];
                V3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z);
                const float tHit = __uint_as_float(h0.x);
                V3 hit_x = o + d * tHit;
                int material_id = 1;
                V3 n = hit_normal(S, h0.y, h0.z, h1.x, h1.y, hit_x, material_id);
                P.nrm[r] = make_float4(n.x, n.y, n.z, __int_as_float(material_id));
                const DMaterial& mat = S.materials[material_id - 1];
                shade_ctx(S, mat, o, d, tHit, n, c);
                flags = 0;
                const bool can_recurse = gen < S.max_depth;   // child depth = gen+1 <= max (raytracer.cpp:384)
                if (c.type == RT_MAT_MIRROR || c.type == RT_MAT_CONDUCTOR) n_child = can_recurse ? 1 : 0;
                else if (c.type == RT_MAT_DIELECTRIC) n_child = can_recurse ? (c.tir ? 1 : 2) : 0;
                if (!(c.type == RT_MAT_DIELECTRIC && !c.entering)) {   // raytracer.cpp:1364-1366
                    flags |= RF_LIGHTS;
                    n_shadow = S.n_point_lights;
                }
            }
        }
        uint32_t child = block_alloc(&ctl->n_records, n_child, s_tmp);
        // Shadow slots of a block: light-major — slot of (record with rank k among the block's K lit records,
        // light l) = base + l * K + k, so that 64 consecutive slots are rays towards ONE light from neighbouring
        // pixels (record-major interleaves the lights lane by lane: 7 % slower any-hit launches at 3840x2160).
        // link.z = slot of light 0, link.w carries K above the flag bits.
        uint32_t sblock = 0, stotal = 0;
        uint32_t sbase = block_alloc(&ctl->n_shadow, n_shadow, s_tmp, &sblock, &stotal);
        const uint32_t n_lights = S.n_point_lights;
        const uint32_t sK = n_lights ? stotal / n_lights : 0u;                  // every requester asks for n_lights
        if (n_shadow) sbase = sblock + (sbase - sblock) / n_lights;             // base + k
        if (!valid) continue;
        uint32_t c0 = RT_NO_CHILD, c1 = RT_NO_CHILD;
        if (n_child) {
            if (child + n_child <= P.cap_records) {
                c0 = child;
                write_ray(P, c0, c.refl_origin, c.wr, time, true);
                if (n_child == 2) {
                    c1 = child + 1;
                    write_ray(P, c1, c.refr_origin, c.wt, time, true);
                }
            } else {
                atomicOr(&ctl->overflow, 1u);
            }
        }
        if (n_shadow) {
            if (sblock + stotal <= P.cap_shadow) {
                for (uint32_t l = 0; l < n_shadow; l++) P.shadow_rec[sbase + l * sK] = r;
                flags |= sK << RF_STRIDE_SHIFT;
            } else {
                // the chunk will be redone; until then the slots below the cap must not name a record
                for (uint32_t l = 0; l < n_shadow; l++)
                    if (sbase + l * sK < P.cap_shadow) P.shadow_rec[sbase + l * sK] = 0xFFFFFFFFu;
                atomicOr(&ctl->overflow, 2u);
                flags &= ~RF_LIGHTS;
            }
        }
        P.link[r] = make_uint4(c0, c1, sbase, flags);
    }
}

// Close generation gen: fix the boundaries of the next generation and of this
// generation's shadow slots, and rewind the work cursors.
__global__ void k_advance(Control* ctl, Pool P, int gen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t nr = ctl->n_records < P.cap_records ? ctl->n_records : P.cap_records;
        uint32_t ns = ctl->n_shadow < P.cap_shadow ? ctl->n_shadow : P.cap_shadow;
        ctl->n_records = nr;
        ctl->n_shadow = ns;
        for (int g = gen + 2; g < RT_MAX_GEN + 2; g++) ctl->gstart[g] = nr;
        for (int g = gen + 1; g < RT_MAX_GEN + 2; g++) ctl->sstart[g] = ns;
        for (int p = 0; p < RT_PARTS; p++) { ctl->cursor_closest[p * RT_CURSOR_STRIDE] = 0; ctl->cursor_shadow[p * RT_CURSOR_STRIDE] = 0; }
    }
}

// ApplyShading's colour expression (raytracer.cpp:1251-1377) for one record of depth `gen`, children already combined.
__device__ __forceinline__ void combine_record(const DScene& S, const Pool& P, uint32_t r, const uint4 lk, int gen) {
    V3 color = mk(0, 0, 0);
    if (lk.w & RF_MISS) {
        if (gen == 0 && !(lk.w & RF_DEAD)) color = ldv(S.background);   // raytracer.cpp:393-400
        P.color[r] = make_float4(color.x, color.y, color.z, 0.f);
        return;
    }
    const float4 r0 = P.ray0[r], r1 = P.ray1[r], nn = P.nrm[r];
    const float tHit = __uint_as_float(P.hit0[r].x);
    V3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z), n = mk(nn.x, nn.y, nn.z);
    const DMaterial& mat = S.materials[__float_as_int(nn.w) - 1];
    ShadeCtx c;
    shade_ctx(S, mat, o, d, tHit, n, c);
    V3 child0 = mk(0, 0, 0), child1 = mk(0, 0, 0);   // ComputeColor beyond max depth: (0,0,0) (raytracer.cpp:384-386)
    if (lk.x != RT_NO_CHILD) { float4 v = P.color[lk.x]; child0 = mk(v.x, v.y, v.z); }
    if (lk.y != RT_NO_CHILD) { float4 v = P.color[lk.y]; child1 = mk(v.x, v.y, v.z); }
    V3 km = ldv(mat.km);
    if (c.type == RT_MAT_MIRROR) {
        color = color + emul(km, child0);
    } else if (c.type == RT_MAT_CONDUCTOR) {
        V3 reflectColor = emul(km, child0);
        color = color + reflectColor * c.Fr;
    } else if (c.type == RT_MAT_DIELECTRIC) {
        if (c.tir) {
            color = color + emul(km, child0);
        } else {
            V3 reflectColor = emul(km, child0);
            V3 refractColor = child1;
            if (!c.entering) {
                float dd = vlen(c.x - o);
                refractColor.x *= rt_expf(-mat.absorb_coef[0] * dd);
                refractColor.y *= rt_expf(-mat.absorb_coef[1] * dd);
                refractColor.z *= rt_expf(-mat.absorb_coef[2] * dd);
            }
            color = (color + reflectColor * c.Fr) + refractColor * (1.0f - c.Fr);
        }
    }
    if (lk.w & RF_LIGHTS) {
        color = color + emul(ldv(S.ambient), ldv(mat.ka));
        const uint32_t sK = (lk.w >> RF_STRIDE_SHIFT) & 0xFFFFu;
        for (uint32_t l = 0; l < S.n_point_lights; l++) {
            if (P.shadow_lit[lk.z + l * sK]) color = color + diffuse_specular(mat, S.point_lights[l], c.x, c.n_shading, c.w0);
        }
    }
    P.color[r] = make_float4(color.x, color.y, color.z, 0.f);
}

// ... for the records of generation gen (contiguous in the per-bounce pipeline).
__global__ void __launch_bounds__(256) k_combine(DScene S, Pool P, Control* ctl, int gen) {
    const uint32_t begin = ctl->gstart[gen], end = ctl->gstart[gen + 1];
    for (uint32_t r = begin + blockIdx.x * blockDim.x + threadIdx.x; r < end; r += gridDim.x * blockDim.x)
        combine_record(S, P, r, P.link[r], gen);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) {   // MathF.h:11-16
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

// raytracer.cpp:77-85: sum the samples in order, scale, clamp, truncate.
// COMBINE0: the records of generation 0 — the pixel's own samples, idx * N + s — are combined here, by the thread that
// sums them (their children are done: k_combine ran for the generations above), instead of by a launch of their own.
template <bool COMBINE0>
__global__ void __launch_bounds__(256) k_resolve(DScene S, DCamera cam, ChunkGeom g, Pool P, const DTriShade* tri_shade, float inv_n,
                                                 uint8_t* rgb, size_t row_stride, uint32_t* hit_ids) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.n_idx) return;
    int px, py;
    if (!idx_to_pixel(g, idx, px, py)) return;
    const int N = cam.num_samples;
    V3 sum = mk(0.0f, 0.0f, 0.0f);
    for (int s = 0; s < N; s++) {
        const uint32_t r = idx * (uint32_t)N + (uint32_t)s;
        if constexpr (COMBINE0) combine_record(S, P, r, P.link[r], 0);
        float4 c = P.color[r];
        sum = sum + mk(c.x, c.y, c.z);
    }
    V3 c = sum * inv_n;
    const int lx = px - g.x0, ly = py - g.y0;
    uint8_t* o = rgb + (size_t)ly * row_stride + (size_t)lx * 3;
    o[0] = (uint8_t)clampf(c.x, 0.0f, 255.0f);
    o[1] = (uint8_t)clampf(c.y, 0.0f, 255.0f);
    o[2] = (uint8_t)clampf(c.z, 0.0f, 255.0f);
    if (hit_ids) {
        const uint4 h0 = P.hit0[(size_t)idx * N];
        uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu;
        if (h0.y != 0xFFFFFFFFu) {
            w0 = h0.y;
            w1 = ((h0.y >> 28) == RT_KIND_MESH) ? tri_shade[h0.z].face : 0u;
        }
        uint32_t* ho = hit_ids + ((size_t)ly * g.w + lx) * 2;
        ho[0] = w0;
        ho[1] = w1;
    }
}
