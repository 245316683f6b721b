// sim_wave.cpp — development tool: lockstep simulation of one wave (64 rays) traversing the
// collapsed 4-wide BVH of an untransformed mesh under different leaf-postponing policies.
// Counts wave-level node steps / triangle steps to estimate SIMD utilisation before writing a kernel.
//   build: g++ -O2 -std=c++17 -Iinclude -Iraytracer_amd/host scripts/sim/sim_wave.cpp -o /tmp/sim_wave
//   usage: sim_wave scene.rtscene rays.bin K [mesh_bvh_index]
// rays.bin: float32 [n][8] = o.xyz, d.xyz, tmax, pad ; n multiple of 64, consecutive 64 = one wave.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#include "rt_scene_store.hpp"

struct N4 { float f[24]; uint32_t link[4]; uint32_t mask; };
static const uint32_t LEAF = 0x80000000u, DONE = 0xFFFFFFFFu, NONE = 0xFFFFFFFDu;
struct Tri { float a[3], e1[3], e2[3]; };

static std::vector<N4> nodes;
static std::vector<Tri> tris;
static std::vector<uint32_t> leafcount;   // by first index

static uint32_t leaf_link(const rt_bvh_node& n) { leafcount[n.left_or_first] = n.prim_count; return LEAF | (uint32_t)n.left_or_first; }
static uint32_t build(const rt_bvh_node* bn, uint32_t ref) {
    uint32_t slot = nodes.size(); nodes.emplace_back();
    N4 d; memset(&d, 0, sizeof(d)); for (int k = 0; k < 4; k++) d.link[k] = DONE;
    int s = 0; uint32_t L = bn[ref].left_or_first;
    auto put = [&](int sl, uint32_t r) {
        const rt_bvh_node& c = bn[r]; float* f = d.f + 6 * sl;
        f[0]=c.bmin[0]; f[1]=c.bmin[1]; f[2]=c.bmin[2]; f[3]=c.bmax[0]; f[4]=c.bmax[1]; f[5]=c.bmax[2];
        d.mask |= 1u << sl;
        d.link[sl] = c.prim_count > 0 ? leaf_link(c) : build(bn, r);
    };
    for (uint32_t ch : {L, L + 1}) { const rt_bvh_node& c = bn[ch]; if (c.prim_count > 0) put(s, ch); else { put(s, c.left_or_first); put(s + 1, c.left_or_first + 1); } s += 2; }
    nodes[slot] = d; return slot;
}
struct Ray { float o[3], d[3], inv[3]; };
static bool box(const Ray& r, const float* f, float tmax, float& te) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { float t0 = (f[a] - r.o[a]) * r.inv[a], t1 = (f[3 + a] - r.o[a]) * r.inv[a]; bool sw = t1 < t0; lo[a] = sw ? t1 : t0; hi[a] = sw ? t0 : t1; }
    float tMin = fmaxf(fmaxf(fmaxf(0.f, lo[0]), lo[1]), lo[2]), tMax = fminf(fminf(fminf(tmax, hi[0]), hi[1]), hi[2]);
    te = tMin; return tMin <= tMax && tMax > 0.f;
}
static float det3(float a11,float a12,float a13,float a21,float a22,float a23,float a31,float a32,float a33){return a11*(a22*a33-a23*a32)-a12*(a21*a33-a23*a31)+a13*(a21*a32-a22*a31);}
static bool tri(const Ray& r, const Tri& t, float minT, float& tout) {
    float bx=t.a[0]-r.o[0],by=t.a[1]-r.o[1],bz=t.a[2]-r.o[2];
    float detA=det3(t.e1[0],t.e2[0],r.d[0],t.e1[1],t.e2[1],r.d[1],t.e1[2],t.e2[2],r.d[2]);
    if (fabsf(detA)<1e-8f) return false;
    float beta=det3(bx,t.e2[0],r.d[0],by,t.e2[1],r.d[1],bz,t.e2[2],r.d[2])/detA; if (beta<-1e-8f) return false;
    float gamma=det3(t.e1[0],bx,r.d[0],t.e1[1],by,r.d[1],t.e1[2],bz,r.d[2])/detA; if (gamma<-1e-8f||beta+gamma>1.0f+1e-8f) return false;
    float tt=det3(t.e1[0],t.e2[0],bx,t.e1[1],t.e2[1],by,t.e1[2],t.e2[2],bz)/detA; if (tt<-1e-8f||tt>=minT) return false;
    tout=tt; return true;
}
struct Lane {
    Ray r; float tmax; uint32_t cur; float cur_t; std::vector<std::pair<uint32_t,float>> stack; std::vector<std::pair<uint32_t,float>> q; bool done;
    uint32_t best;
    uint32_t pop() { while (!stack.empty()) { auto e = stack.back(); stack.pop_back(); if (!((tmax < e.second) || (e.second == 0.f && tmax <= 0.f))) { cur_t = e.second; return e.first; } } return DONE; }
};
int main(int argc, char** argv) {
    RtSceneStore sc; if (!sc.load(argv[1])) return 1;
    int K = atoi(argv[3]); int mb = argc > 4 ? atoi(argv[4]) : (int)sc.mesh_bvhs.size() - 1;
    const rt_mesh_bvh& B = sc.mesh_bvhs[mb]; const rt_bvh_node* bn = sc.mesh_nodes.data() + B.node_offset;
    leafcount.assign(B.prim_count, 0); tris.resize(B.prim_count);
    for (uint32_t p = 0; p < B.prim_count; p++) { const rt_face& f = sc.faces[B.face_offset + sc.mesh_prim_face[B.prim_offset + p]];
        const float* a = sc.vertex_data[f.i0-1].pos; const float* b = sc.vertex_data[f.i1-1].pos; const float* c = sc.vertex_data[f.i2-1].pos;
        for (int k=0;k<3;k++){tris[p].a[k]=a[k];tris[p].e1[k]=a[k]-b[k];tris[p].e2[k]=a[k]-c[k];} }
    uint32_t root = bn[0].prim_count > 0 ? leaf_link(bn[0]) : build(bn, 0);
    FILE* f = fopen(argv[2], "rb"); std::vector<float> rays; { float buf[8]; while (fread(buf, 4, 8, f) == 8) rays.insert(rays.end(), buf, buf + 8); } fclose(f);
    size_t n = rays.size() / 8; unsigned long long wave_node = 0, lane_node = 0, wave_tri = 0, lane_tri = 0, outer = 0, nrays = 0, checksum = 0;
    for (size_t w = 0; w + 64 <= n; w += 64) {
        Lane L[64];
        for (int l = 0; l < 64; l++) { const float* r = &rays[(w + l) * 8]; Lane& a = L[l];
            for (int k=0;k<3;k++){a.r.o[k]=r[k];a.r.d[k]=r[3+k];a.r.inv[k]=r[3+k]==0.f?INFINITY:1.0f/r[3+k];}
            a.tmax = r[6]; a.done = false; a.best = 0xFFFFFFFFu; float te; a.cur = DONE; a.cur_t = 0;
            if (box(a.r, bn[0].bmin, a.tmax, te)) { a.cur = root; a.cur_t = te; } if (a.cur == DONE) a.done = true; nrays++; }
        for (;;) {
            bool any = false; for (auto& a : L) any |= !a.done; if (!any) break;
            outer++;
            // node phase: lanes run until queue has K+1 leaves (K postponed + current) or traversal exhausted
            for (;;) {
                int act = 0;
                for (auto& a : L) { if (a.done) continue; if (a.cur == DONE) continue; if ((int)a.q.size() > K) continue;
                    if (a.cur & LEAF) { a.q.push_back({a.cur, a.cur_t}); a.cur = a.pop(); continue_flag: ; }
                }
                // after moving leaves to queues, lanes with interior cur and room step
                for (auto& a : L) { if (a.done || a.cur == DONE || (a.cur & LEAF) || (int)a.q.size() > K) continue; act++;
                    const N4& nd = nodes[a.cur]; float t[4]; bool h[4];
                    for (int k = 0; k < 4; k++) h[k] = (nd.mask >> k & 1) && box(a.r, nd.f + 6 * k, a.tmax, t[k]);
                    uint32_t nx = NONE; float tn = 0;
                    for (int k = 3; k >= 0; k--) if (h[k]) { if (nx != NONE) a.stack.push_back({nx, tn}); nx = nd.link[k]; tn = t[k]; }
                    if (nx != NONE) { a.cur = nx; a.cur_t = tn; } else a.cur = a.pop();
                }
                if (!act) { bool more = false; for (auto& a : L) if (!a.done && a.cur != DONE && (a.cur & LEAF) && (int)a.q.size() <= K) more = true; if (!more) break; else continue; }
                wave_node++; lane_node += act;
            }
            // leaf phase: every lane drains its queue in order, one triangle per wave step
            std::vector<std::vector<uint32_t>> work(64);
            size_t maxw = 0;
            for (int l = 0; l < 64; l++) { Lane& a = L[l]; 
                // sequential semantics per lane: recheck, then triangles in order (tmax updates matter) -> simulate directly, count steps
                size_t steps = 0;
                for (auto& e : a.q) { if ((a.tmax < e.second) || (e.second == 0.f && a.tmax <= 0.f)) continue; uint32_t first = e.first & 0x7FFFFFFFu;
                    for (uint32_t i = 0; i < leafcount[first]; i++) { steps++; float tt; if (tri(a.r, tris[first + i], a.tmax, tt)) { a.tmax = tt; a.best = first + i; } } }
                a.q.clear(); lane_tri += steps; if (steps > maxw) maxw = steps;
                // recheck current position against the new tmax
                if (!a.done && a.cur != DONE && ((a.tmax < a.cur_t) || (a.cur_t == 0.f && a.tmax <= 0.f))) a.cur = a.pop();
                if (a.cur == DONE) a.done = true;
            }
            wave_tri += maxw;
        }
        for (auto& a : L) checksum = checksum * 1315423911ull + a.best;
    }
    printf("K=%d rays=%llu per-ray node %.2f tri %.2f | per batch: wave node steps %.1f (util %.3f) tri steps %.1f (util %.3f) outer %.1f | est VALU/batch %.0f checksum %llx\n",
           K, nrays, (double)lane_node / nrays, (double)lane_tri / nrays, 64.0 * wave_node / nrays, lane_node / (64.0 * wave_node), 64.0 * wave_tri / nrays,
           lane_tri / (64.0 * std::max(1ull, wave_tri)), 64.0 * outer / nrays, (64.0 * wave_node / nrays) * 135 + (64.0 * wave_tri / nrays) * 115, checksum);
    return 0;
}
