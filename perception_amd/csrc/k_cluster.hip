// k_cluster.hip - S5 Euclidean cluster extraction.
//
// Replaces pcl::search::KdTree + pcl::EuclideanClusterExtraction::extract (reference:
// object_detection/src/object_pose_detection.cpp:345-362: tolerance 0.02, min 200, max 25000).
// PCL's BFS over FLANN radius searches computes the connected components of the graph
// "d2(i,j) < (float)(tol*tol)", which do not depend on traversal order.  Here:
//   build   : fixed-radius spatial hash (cell edge = tol*(1+2^-10)), one bucket list per cell
//   hook    : every point tests the 27 neighbouring buckets and unions with lower-index
//             neighbours (lock-free union-find, root = smallest member index)
//   flatten : parent[i] = root, component sizes
//   rank    : components inside [min,max] ranked by (size desc, root asc)  (rule C5)
//   label   : labels + ordered per-cluster compaction (wave ballot/popcount prefix) into the
//             ICP source segments.
#include "kernels.hpp"

namespace cd {


__device__ __forceinline__ int ld_agent(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t cell_hash(int cx, int cy, int cz) {
    return ((uint32_t)cx * 73856093u ^ (uint32_t)cy * 19349663u ^ (uint32_t)cz * 83492791u) & (CELL_BUCKETS - 1);
}
__device__ __forceinline__ void cell_of(const float4& p, const float* o, float inv_cell, int& cx, int& cy, int& cz) {
    cx = (int)floorf(__fmul_rn(__fsub_rn(p.x, o[0]), inv_cell));
    cy = (int)floorf(__fmul_rn(__fsub_rn(p.y, o[1]), inv_cell));
    cz = (int)floorf(__fmul_rn(__fsub_rn(p.z, o[2]), inv_cell));
}

__global__ void __launch_bounds__(BLOCK) k_cluster_build(const float4* __restrict__ obj, int N,
                                                         const FrameState* __restrict__ fs, float inv_cell,
                                                         int* __restrict__ head, int* __restrict__ next,
                                                         int* __restrict__ parent, int* __restrict__ csize,
                                                         int* __restrict__ rank_of_root) {
    CD_FRONT_PRIO();
    const int f = blockIdx.y;
    const int n = fs[f].n_o;
    if (n <= 0 || fs[f].cl_done) return;      // finished by k_cluster_lds
    const size_t fbase = (size_t)f * N;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        int cx, cy, cz;
        cell_of(obj[fbase + i], fs[f].origin, inv_cell, cx, cy, cz);
        parent[fbase + i] = i;
        csize[fbase + i] = 0;
        rank_of_root[fbase + i] = -1;
        next[fbase + i] = atomicExch(&head[(size_t)f * CELL_BUCKETS + cell_hash(cx, cy, cz)], i);
    }
}

__device__ __forceinline__ int uf_find(int* par, int x) {
    for (;;) {
        const int p = ld_agent(par + x);
        if (p == x) return x;
        const int gp = ld_agent(par + p);
        if (gp == p) return p;
        st_agent(par + x, gp);   // path halving; parent pointers only ever move to an ancestor
        x = gp;
    }
}
__device__ __forceinline__ void uf_union(int* par, int a, int b) {
    a = uf_find(par, a);
    b = uf_find(par, b);
    while (a != b) {
        if (a < b) { const int t = a; a = b; b = t; }   // link the larger root under the smaller
        const int old = atomicCAS(par + a, a, b);
        if (old == a) return;
        a = uf_find(par, old);
        b = uf_find(par, b);
    }
}

// Cheap find for the hot loop: plain (L1-cacheable) loads, path halving by write-through stores.  A stale value is always
// a past parent, i.e. still an ancestor-or-self, so a stale "root" only makes the CAS below fail,
// after which the slow path re-reads with agent-scope atomics.
__device__ __forceinline__ int uf_find_cached(int* par, int x) {
    int p = par[x];
    while (p != x) {
        const int gp = par[p];
        if (gp != p) st_agent(par + x, gp);   // path halving: x is not a root, gp is one of its ancestors
        x = p;
        p = gp;
    }
    return x;
}

__global__ void __launch_bounds__(BLOCK) k_cluster_hook(const float4* __restrict__ obj, int N,
                                                        const FrameState* __restrict__ fs, float inv_cell, float r2,
                                                        const int* __restrict__ head, const int* __restrict__ next,
                                                        int* parent) {
    CD_FRONT_PRIO();
    const int f = blockIdx.y;
    const int n = fs[f].n_o;
    if (n <= 0 || fs[f].cl_done) return;      // finished by k_cluster_lds
    const size_t fbase = (size_t)f * N;
    const float4* P = obj + fbase;
    const int* nx = next + fbase;
    int* par = parent + fbase;
    const int* hd = head + (size_t)f * CELL_BUCKETS;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 p = P[i];
        int cx, cy, cz;
        cell_of(p, fs[f].origin, inv_cell, cx, cy, cz);
        int ri = uf_find_cached(par, i);   // root of i, kept in a register across its neighbours
        for (int a = -1; a <= 1; ++a)
            for (int b = -1; b <= 1; ++b)
                for (int c = -1; c <= 1; ++c) {
                    int j = hd[cell_hash(cx + a, cy + b, cz + c)];
                    while (j >= 0) {
                        if (j < i) {
                            const float4 q = P[j];
                            if (dist2(p.x, p.y, p.z, q.x, q.y, q.z) < r2) {
                                int rj = uf_find_cached(par, j);
                                while (ri != rj) {
                                    const int hi = ri > rj ? ri : rj, lo = ri > rj ? rj : ri;
                                    const int old = atomicCAS(par + hi, hi, lo);   // link the larger root under the smaller
                                    if (old == hi) { ri = lo; rj = lo; break; }
                                    ri = uf_find(par, old);                        // hi was no root any more: re-read coherently
                                    rj = uf_find(par, lo);
                                }
                                ri = ri < rj ? ri : rj;
                            }
                        }
                        j = nx[j];
                    }
                }
        if (ri != i) st_agent(par + i, ri);   // shortcut for later finds; i is not a root, so no CAS targets it
    }
}

__global__ void __launch_bounds__(BLOCK) k_cluster_flatten(int N, const FrameState* __restrict__ fs,
                                                           int* __restrict__ parent, int* __restrict__ csize) {
    CD_FRONT_PRIO();
    const int f = blockIdx.y;
    const int n = fs[f].n_o;
    if (n <= 0 || fs[f].cl_done) return;      // finished by k_cluster_lds
    int* par = parent + (size_t)f * N;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        int r = i;
        for (;;) {   // read-only walk to the root
            const int p = ld_agent(par + r);
            if (p == r) break;
            r = p;
        }
        if (r != i) st_agent(par + i, r);
        atomicAdd(&csize[(size_t)f * N + r], 1);
    }
}

// ---- LDS variant: one 1024-thread workgroup per frame, for frames with n_o <= CL_LDS_CAP ----------
// Clusters CELLS instead of points.  The cell edge is e = tol/sqrt(3) * (1 - 2^-10), so any two points of one cell are
// closer than tol (their squared distance is below 3 e^2 = tol^2 (1 - 2^-10)^2; the float roundings of the cell
// assignment and of dist2 are below 2^-12 relative while cell coordinates stay under 1024, which the kernel checks):
// a cell is connected by construction and needs no test at all.  Two cells are joined when ONE pair of their points
// passes the reference's predicate d2 < (float)(tol*tol) (strict, rule C3); points more than two cells apart on any axis
// cannot pass it ((3-1) e > tol), so a cell looks at the 62 "forward" cells of its 5x5x5 neighbourhood.  The components of
// the cell graph are exactly those of PCL's point graph; what is saved is the all-pairs work inside and between cells
// that are already connected (the point-pair version of this kernel spent 0.25 ms per batch on LDS reads).
//   1. every point inserts its packed cell key in an open-addressing table (LDS CAS), takes a ticket in its cell
//   2. exclusive scan of the cell counts, points scattered into LDS in cell order
//   3. hook, four lanes per occupied cell: first the 13 adjacent forward cells, then (after a barrier, when most
//      cells already share a root and are skipped on a root compare) the 49 cells two steps away
//   4. per component: smallest original index and size -> parent / csize as build + hook + flatten produce them
// A frame the table cannot hold (too many cells, or a cloud wider than 1018 cells) is left to the global-memory path:
// fs.cl_done says which frames are finished.
#ifdef CD_CLDBG
// per phase: sum over the workgroups and max over the workgroups of the time thread 0 spent (100 MHz ticks)
__device__ unsigned long long g_cl_dbg[8][2];
extern "C" int cd_debug_cluster(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cl_dbg), sizeof(g_cl_dbg)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[8][2]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cl_dbg), z, sizeof(z)); }
    return 0;
}
#define CL_PHASE(k) { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_cl_dbg[k][0], t_ - t0_); atomicMax(&g_cl_dbg[k][1], t_ - t0_); t0_ = t_; } }
#else
#define CL_PHASE(k)
#endif
constexpr int CL_LDS_CAP = 8192;
constexpr int CL_SLOTS = 4096;          // cell table (power of two)
constexpr int CL_MAX_CELLS = 3072;      // load factor <= 3/4
constexpr int CL_COORD_MIN = -1;        // a centroid may round one ulp below the cloud's minimum
constexpr int CL_COORD_MAX = 1018;      // 10-bit fields hold coordinate + 3, neighbours reach +-2
constexpr int CL_THREADS = 1024;
constexpr int CL_PER_THREAD = CL_LDS_CAP / CL_THREADS;
constexpr int CL_WAVES = CL_THREADS / WAVE;
constexpr int CL_LANES_PER_CELL = 4;

// packed key offsets of the 62 forward cells (code (dx+2) + 5 (dy+2) + 25 (dz+2) in 63..124): the 13 adjacent ones
// (every |d| <= 1) first, then the 49 that are two cells away on some axis
constexpr int CL_RING1 = 13, CL_FORWARD = 62;
struct ClOffsets {
    int d[CL_FORWARD];
    constexpr ClOffsets() : d{} {
        int m = 0;
        for (int ring = 0; ring < 2; ++ring)
            for (int code = 63; code < 125; ++code) {
                const int dx = code % 5 - 2, dy = (code / 5) % 5 - 2, dz = code / 25 - 2;
                const bool adjacent = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz <= 1;
                if (adjacent == (ring == 0)) d[m++] = dx + dy * 1024 + dz * 1048576;
            }
    }
};
__constant__ ClOffsets cl_offsets = ClOffsets();

__device__ __forceinline__ uint32_t cl_slot_hash(int key) {
    uint32_t h = (uint32_t)key * 2654435761u;
    return (h >> 17) & (CL_SLOTS - 1);
}
// find with path halving; parents live in LDS (coherent within the workgroup).  A halving store
// only ever re-points a NON-root at one of its ancestors, so it cannot disturb the root CAS.
__device__ __forceinline__ int lds_find(int* par, int x) {
    int p = par[x];
    while (p != x) {
        const int gp = par[p];
        if (gp != p) par[x] = gp;
        x = p;
        p = gp;
    }
    return x;
}
// slot of a cell key, or -1 (the table always keeps empty slots: CL_MAX_CELLS < CL_SLOTS)
__device__ __forceinline__ int cl_lookup(const int* s_key, int key) {
    int h = (int)cl_slot_hash(key);
    for (;;) {
        const int k = s_key[h];
        if (k == key) return h;
        if (k == -1) return -1;
        h = (h + 1) & (CL_SLOTS - 1);
    }
}
// cell `ha` against the cell at packed offset `delta`: union the two when one pair of points is closer than tol
__device__ __forceinline__ void cl_join(const int* s_key, const int* s_val, int* s_par, const float* s_x, const float* s_y,
                                        const float* s_z, int ha, int key, int delta, float r2, int& ra) {
    const int hb = cl_lookup(s_key, key + delta);
    if (hb < 0) return;
    int rb = lds_find(s_par, hb);
    if (rb == ra) return;
    const int a0 = s_val[ha], a1 = s_val[ha + 1], b0 = s_val[hb], b1 = s_val[hb + 1];
    bool hit = false;
    for (int a = a0; a < a1 && !hit; ++a) {
        const float px = s_x[a], py = s_y[a], pz = s_z[a];
        for (int b = b0; b < b1; b += 2) {   // two candidates per trip (the last one repeats at an odd end)
            const int bb = b + 1 < b1 ? b + 1 : b;
            const float d0 = dist2(px, py, pz, s_x[b], s_y[b], s_z[b]);
            const float d1 = dist2(px, py, pz, s_x[bb], s_y[bb], s_z[bb]);
            if (d0 < r2 || d1 < r2) { hit = true; break; }
        }
    }
    if (!hit) return;
    ra = lds_find(s_par, ra);
    rb = lds_find(s_par, rb);
    while (ra != rb) {
        const int big = ra > rb ? ra : rb, sml = ra > rb ? rb : ra;
        const int old = atomicCAS(&s_par[big], big, sml);   // link the larger root under the smaller
        if (old == big) { ra = sml; break; }
        ra = lds_find(s_par, ra);                          // big had been linked meanwhile: climb, retry
        rb = lds_find(s_par, rb);
    }
}

__global__ void __launch_bounds__(CL_THREADS) k_cluster_lds(const float4* __restrict__ obj, int N, FrameState* __restrict__ fs,
                                                            float inv_cell, float r2, int* __restrict__ parent,
                                                            int* __restrict__ csize, int* __restrict__ rank_of_root) {
    CD_FRONT_PRIO();
    __shared__ float s_x[CL_LDS_CAP], s_y[CL_LDS_CAP], s_z[CL_LDS_CAP];   // 96 KiB, cell order (s_x / s_y: per-component min / size at the end)
    __shared__ int s_key[CL_SLOTS];                                       // packed cell of the slot, -1 = empty
    __shared__ int s_val[CL_SLOTS + 1];                                   // points in the slot's cell, then their start
    __shared__ int s_par[CL_SLOTS];                                       // union-find over slots
    __shared__ unsigned short s_cells[CL_MAX_CELLS];                      // occupied slots, dense
    __shared__ int s_w[CL_WAVES];
    __shared__ int s_ncell, s_bail;
    const int f = blockIdx.x;
    const int n = fs[f].n_o;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (n <= 0 || n > CL_LDS_CAP) {           // larger frames take the global-memory path
        if (tid == 0) fs[f].cl_done = n <= 0 ? 1 : 0;
        return;
    }
    const size_t fbase = (size_t)f * N;
    const float4* __restrict__ P = obj + fbase;
    const float org[3] = {fs[f].origin[0], fs[f].origin[1], fs[f].origin[2]};
#ifdef CD_CLDBG
    unsigned long long t0_ = wall_clock64();
#endif
    for (int i = tid; i < CL_SLOTS; i += CL_THREADS) { s_key[i] = -1; s_val[i] = 0; s_par[i] = i; }
    if (tid == 0) { s_ncell = 0; s_bail = 0; }
    __syncthreads();
    float4 pt[CL_PER_THREAD];
    int sl[CL_PER_THREAD], tk[CL_PER_THREAD];
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) pt[k] = P[min(tid + k * CL_THREADS, n - 1)];   // (all loads first: see load_rows_clamped)
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) {
        const int i = tid + k * CL_THREADS;
        sl[k] = 0; tk[k] = 0;
        if (i < n) {
            int cx, cy, cz;
            cell_of(pt[k], org, inv_cell, cx, cy, cz);
            bool ok = cx >= CL_COORD_MIN && cy >= CL_COORD_MIN && cz >= CL_COORD_MIN && cx <= CL_COORD_MAX && cy <= CL_COORD_MAX && cz <= CL_COORD_MAX;
            if (ok) {
                const int key = (cx + 3) | ((cy + 3) << 10) | ((cz + 3) << 20);
                int h = (int)cl_slot_hash(key);
                ok = false;
                for (int probe = 0; probe < CL_SLOTS; ++probe) {
                    const int old = atomicCAS(&s_key[h], -1, key);
                    if (old == -1) {   // this point opened the cell
                        const int ci = atomicAdd(&s_ncell, 1);
                        if (ci < CL_MAX_CELLS) s_cells[ci] = (unsigned short)h;
                        ok = true;
                        break;
                    }
                    if (old == key) { ok = true; break; }
                    h = (h + 1) & (CL_SLOTS - 1);
                }
                sl[k] = h;
                if (ok) tk[k] = atomicAdd(&s_val[h], 1);
            }
            if (!ok) s_bail = 1;
        }
    }
    __syncthreads();
    const int ncell = s_ncell;
    if (s_bail || ncell > CL_MAX_CELLS) {     // uniform: leave the frame to the global-memory kernels
        if (tid == 0) fs[f].cl_done = 0;
        return;
    }
    {   // exclusive scan of the slot counts: four consecutive slots per thread
        const int q = 4 * tid;
        const int c0 = s_val[q], c1 = s_val[q + 1], c2 = s_val[q + 2], c3 = s_val[q + 3];
        const int sum = c0 + c1 + c2 + c3;
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int base = inc - sum;
        for (int v = 0; v < w; ++v) base += s_w[v];
        s_val[q] = base;
        s_val[q + 1] = base + c0;
        s_val[q + 2] = base + c0 + c1;
        s_val[q + 3] = base + c0 + c1 + c2;
        if (tid == 0) s_val[CL_SLOTS] = n;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) {
        const int i = tid + k * CL_THREADS;
        if (i < n) {
            const int pos = s_val[sl[k]] + tk[k];
            s_x[pos] = pt[k].x; s_y[pos] = pt[k].y; s_z[pos] = pt[k].z;
        }
    }
    __syncthreads();
    CL_PHASE(0)
    // hook.  item = (cell, lane of the cell); the lanes of a cell share its forward offsets round robin.
    const int items = ncell * CL_LANES_PER_CELL;
    for (int ring = 0; ring < 2; ++ring) {
        const int j0 = ring == 0 ? 0 : CL_RING1, j1 = ring == 0 ? CL_RING1 : CL_FORWARD;
        for (int it = tid; it < items; it += CL_THREADS) {
            const int ha = (int)s_cells[it / CL_LANES_PER_CELL], g = it % CL_LANES_PER_CELL;
            const int key = s_key[ha];
            int ra = lds_find(s_par, ha);
#pragma unroll 1
            for (int j = j0 + g; j < j1; j += CL_LANES_PER_CELL)
                cl_join(s_key, s_val, s_par, s_x, s_y, s_z, ha, key, cl_offsets.d[j], r2, ra);
        }
        __syncthreads();
    }
    CL_PHASE(1)
    // components -> (smallest original member index, size), kept at the root slot in the dead coordinate arrays
    int* s_min = reinterpret_cast<int*>(s_x);
    int* s_cnt = reinterpret_cast<int*>(s_y);
    for (int q = tid; q < CL_SLOTS; q += CL_THREADS) { s_min[q] = 0x7fffffff; s_cnt[q] = 0; }
    __syncthreads();
    int root[CL_PER_THREAD];
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) {
        const int i = tid + k * CL_THREADS;
        root[k] = 0;
        if (i < n) {
            root[k] = lds_find(s_par, sl[k]);
            atomicMin(&s_min[root[k]], i);
            atomicAdd(&s_cnt[root[k]], 1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) {
        const int i = tid + k * CL_THREADS;
        if (i < n) {
            const int r = s_min[root[k]];
            parent[fbase + i] = r;
            csize[fbase + i] = r == i ? s_cnt[root[k]] : 0;
            rank_of_root[fbase + i] = -1;
        }
    }
    if (tid == 0) fs[f].cl_done = 1;
    CL_PHASE(2)
}

// ---- the same for frames of MORE than CL_LDS_CAP object points (end of round 5): cells in LDS, points in global memory --------
// BASELINE config 5 (five cuboids in a 1 M-point frame: ~12 k object points) and the object launch values (leaf 0.001: ~18 k) put
// every frame beyond what k_cluster_lds holds, and the point-graph kernels below it (k_cluster_build / hook / flatten: linked
// lists and a union-find over POINTS in HBM) took 1.7 ms per 64 such frames.  The number of CELLS does not grow with the
// density, so the cell table, the cell counts and the union-find over cells still fit LDS; only the points move out: they are
// put in cell order into `sorted` (a per-frame scratch array: the ICP source buffer, which is written after this stage), each
// point's (slot, ticket) waits in parent / csize, which receive their final values at the end.  Same cells, same predicate,
// same components as k_cluster_lds.  Frames the table cannot hold keep cl_done = 0 for the point-graph kernels.
__device__ __forceinline__ void cl_join_g(const int* s_key, const int* s_val, int* s_par, const float4* __restrict__ S, int ha, int key, int delta,
                                          float r2, int& ra) {
    const int hb = cl_lookup(s_key, key + delta);
    if (hb < 0) return;
    int rb = lds_find(s_par, hb);
    if (rb == ra) return;
    const int a0 = s_val[ha], a1 = s_val[ha + 1], b0 = s_val[hb], b1 = s_val[hb + 1];
    bool hit = false;
    for (int a = a0; a < a1 && !hit; ++a) {
        const float4 p = S[a];
        for (int b = b0; b < b1; b += 4) {   // four candidates per trip, their loads in flight together (the last one repeats at the end)
            const float4 q0 = S[b], q1 = S[min(b + 1, b1 - 1)], q2 = S[min(b + 2, b1 - 1)], q3 = S[min(b + 3, b1 - 1)];
            const float d0 = dist2(p.x, p.y, p.z, q0.x, q0.y, q0.z), d1 = dist2(p.x, p.y, p.z, q1.x, q1.y, q1.z);
            const float d2 = dist2(p.x, p.y, p.z, q2.x, q2.y, q2.z), d3 = dist2(p.x, p.y, p.z, q3.x, q3.y, q3.z);
            if (d0 < r2 || d1 < r2 || d2 < r2 || d3 < r2) { hit = true; break; }
        }
    }
    if (!hit) return;
    ra = lds_find(s_par, ra);
    rb = lds_find(s_par, rb);
    while (ra != rb) {
        const int big = ra > rb ? ra : rb, sml = ra > rb ? rb : ra;
        const int old = atomicCAS(&s_par[big], big, sml);
        if (old == big) { ra = sml; break; }
        ra = lds_find(s_par, ra);
        rb = lds_find(s_par, rb);
    }
}

__global__ void __launch_bounds__(CL_THREADS) k_cluster_cells(const float4* __restrict__ obj, int N, FrameState* __restrict__ fs,
                                                              float inv_cell, float r2, int* __restrict__ parent,
                                                              int* __restrict__ csize, int* __restrict__ rank_of_root,
                                                              float4* __restrict__ sorted) {
    CD_FRONT_PRIO();
    __shared__ int s_key[CL_SLOTS];        // packed cell of the slot, -1 = empty; after the hook: per-component smallest member
    __shared__ int s_val[CL_SLOTS + 1];    // points in the slot's cell, then their start; after the hook: per-component size
    __shared__ int s_par[CL_SLOTS];        // union-find over slots
    __shared__ unsigned short s_cells[CL_MAX_CELLS];
    __shared__ int s_w[CL_WAVES];
    __shared__ int s_ncell, s_bail;
    const int f = blockIdx.x;
    const int n = fs[f].n_o;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (n <= 0 || fs[f].cl_done) return;       // (uniform) empty, or finished by k_cluster_lds
    const size_t fbase = (size_t)f * N;
    const float4* __restrict__ P = obj + fbase;
    float4* __restrict__ S = sorted + fbase;
    int* slot_of = parent + fbase;             // scratch until the end
    int* ticket_of = csize + fbase;
    const float org[3] = {fs[f].origin[0], fs[f].origin[1], fs[f].origin[2]};
    for (int i = tid; i < CL_SLOTS; i += CL_THREADS) { s_key[i] = -1; s_val[i] = 0; s_par[i] = i; }
    if (tid == 0) { s_ncell = 0; s_bail = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += CL_THREADS) {
        int cx, cy, cz;
        cell_of(P[i], org, inv_cell, cx, cy, cz);
        bool ok = cx >= CL_COORD_MIN && cy >= CL_COORD_MIN && cz >= CL_COORD_MIN && cx <= CL_COORD_MAX && cy <= CL_COORD_MAX && cz <= CL_COORD_MAX;
        int h = 0, tk = 0;
        if (ok) {
            const int key = (cx + 3) | ((cy + 3) << 10) | ((cz + 3) << 20);
            h = (int)cl_slot_hash(key);
            ok = false;
            for (int probe = 0; probe < CL_SLOTS; ++probe) {
                const int old = atomicCAS(&s_key[h], -1, key);
                if (old == -1) {
                    const int ci = atomicAdd(&s_ncell, 1);
                    if (ci < CL_MAX_CELLS) s_cells[ci] = (unsigned short)h;
                    ok = true;
                    break;
                }
                if (old == key) { ok = true; break; }
                h = (h + 1) & (CL_SLOTS - 1);
            }
            if (ok) tk = atomicAdd(&s_val[h], 1);
        }
        if (!ok) s_bail = 1;
        slot_of[i] = h;
        ticket_of[i] = tk;
    }
    __syncthreads();
    const int ncell = s_ncell;
    if (s_bail || ncell > CL_MAX_CELLS) return;   // (uniform) cl_done stays 0: the point-graph kernels take the frame
    {   // exclusive scan of the slot counts: four consecutive slots per thread
        const int q = 4 * tid;
        const int c0 = s_val[q], c1 = s_val[q + 1], c2 = s_val[q + 2], c3 = s_val[q + 3];
        const int sum = c0 + c1 + c2 + c3;
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int base = inc - sum;
        for (int v = 0; v < w; ++v) base += s_w[v];
        s_val[q] = base;
        s_val[q + 1] = base + c0;
        s_val[q + 2] = base + c0 + c1;
        s_val[q + 3] = base + c0 + c1 + c2;
        if (tid == 0) s_val[CL_SLOTS] = n;
    }
    __syncthreads();
    for (int i = tid; i < n; i += CL_THREADS) S[s_val[slot_of[i]] + ticket_of[i]] = P[i];   // (every thread reads back its own words)
    __syncthreads();   // (workgroup-scope release / acquire: the points are read by other waves of this workgroup below)
    const int items = ncell * CL_LANES_PER_CELL;
    for (int ring = 0; ring < 2; ++ring) {
        const int j0 = ring == 0 ? 0 : CL_RING1, j1 = ring == 0 ? CL_RING1 : CL_FORWARD;
        for (int it = tid; it < items; it += CL_THREADS) {
            const int ha = (int)s_cells[it / CL_LANES_PER_CELL], g = it % CL_LANES_PER_CELL;
            const int key = s_key[ha];
            int ra = lds_find(s_par, ha);
#pragma unroll 1
            for (int j = j0 + g; j < j1; j += CL_LANES_PER_CELL) cl_join_g(s_key, s_val, s_par, S, ha, key, cl_offsets.d[j], r2, ra);
        }
        __syncthreads();
    }
    int* s_min = s_key;   // (the keys and the cell starts are dead from here on)
    int* s_cnt = s_val;
    for (int q = tid; q < CL_SLOTS; q += CL_THREADS) { s_min[q] = 0x7fffffff; s_cnt[q] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += CL_THREADS) {
        const int r = lds_find(s_par, slot_of[i]);
        slot_of[i] = r;
        atomicMin(&s_min[r], i);
        atomicAdd(&s_cnt[r], 1);
    }
    __syncthreads();
    for (int i = tid; i < n; i += CL_THREADS) {
        const int root = slot_of[i];
        const int r = s_min[root];
        parent[fbase + i] = r;
        csize[fbase + i] = r == i ? s_cnt[root] : 0;
        rank_of_root[fbase + i] = -1;
    }
    if (tid == 0) fs[f].cl_done = 1;
}

// one block per frame
__global__ void __launch_bounds__(BLOCK) k_cluster_rank(int N, FrameState* __restrict__ fs, int enable, int min_sz,
                                                        int max_sz, const int* __restrict__ parent,
                                                        const int* __restrict__ csize, int* __restrict__ cand,
                                                        int* __restrict__ rank_of_root, int* __restrict__ sizes_sorted,
                                                        FrameState* __restrict__ mirror) {
    // mirror != nullptr: the frame's FrameState goes to the host's pinned copy from here (the last kernel of the stage that
    // writes it), instead of by a copy launch
    CD_FRONT_PRIO();
    __shared__ int s_w[WAVES_PER_BLOCK];
    const int f = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_o;
    const size_t fbase = (size_t)f * N;
    if (threadIdx.x < KICP) { fs[f].ksize[threadIdx.x] = 0; fs[f].koff[threadIdx.x] = 0; }
    if (!enable) {   // cuboid_detection flavour: the whole extracted cloud is the one ICP source
        if (threadIdx.x == 0) {
            fs[f].n_k = n > 0 ? 1 : 0;
            fs[f].ksize[0] = n;
            if (n > 0) sizes_sorted[fbase] = n;
        }
        if (mirror) {
            __syncthreads();
            if (threadIdx.x == 0) mirror[f] = fs[f];
        }
        return;
    }
    const int* par = parent + fbase;
    const int* cs = csize + fbase;
    int* cd = cand + fbase;
    const uint64_t lt = lanemask_lt();
    int K = 0;
    for (int c0 = 0; c0 < n; c0 += BLOCK) {
        const int i = c0 + threadIdx.x;
        bool flag = false;
        if (i < n && par[i] == i) {
            const int s = cs[i];
            flag = s >= min_sz && s <= max_sz;
        }
        const uint64_t bal = __ballot(flag);
        if (lane == 0) s_w[w] = __popcll(bal);
        __syncthreads();
        int wb = 0, tot = 0;
        for (int q = 0; q < WAVES_PER_BLOCK; ++q) { if (q < w) wb += s_w[q]; tot += s_w[q]; }
        if (flag) cd[K + wb + __popcll(bal & lt)] = i;
        K += tot;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += BLOCK) {
        const int root = cd[k], s = cs[root];
        int r = 0;
        for (int q = 0; q < K; ++q) {
            const int root2 = cd[q], s2 = cs[root2];
            r += (s2 > s || (s2 == s && root2 < root)) ? 1 : 0;
        }
        rank_of_root[fbase + root] = r;
        sizes_sorted[fbase + r] = s;
        if (r < KICP) fs[f].ksize[r] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        fs[f].n_k = K;
        int off = 0;
        for (int k = 0; k < KICP; ++k) { fs[f].koff[k] = off; off += fs[f].ksize[k]; }
        if (mirror) mirror[f] = fs[f];
    }
}

// Clusters are extracted for ICP KICP at a time: round `kbase` handles the clusters ranked kbase .. kbase+KICP-1 (a frame
// rarely has more than KICP; the host runs further rounds only for frames that do - opd.cpp:376 gives EVERY cluster its ICP).
__global__ void __launch_bounds__(BLOCK) k_label_count(int N, int T, const FrameState* __restrict__ fs, int enable,
                                                       const int* __restrict__ parent, const int* __restrict__ rank_of_root,
                                                       int* __restrict__ label, int* __restrict__ tile_cnt, int kbase) {
    CD_FRONT_PRIO();
    __shared__ int s_c[WAVES_PER_BLOCK][KICP];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_o;
    if (tile * TILE >= n) return;
    if (kbase > 0 && kbase >= fs[f].n_k) return;
    const size_t fbase = (size_t)f * N;
    const int base = tile * TILE + w * WAVE_SPAN + lane;
    int cnt[KICP];
#pragma unroll
    for (int k = 0; k < KICP; ++k) cnt[k] = 0;
    // (all parents, then all ranks, before any is used: two memory round trips instead of sixteen - see load_rows_clamped)
    int rt[ITEMS];
    if (enable) {
        load_rows_clamped<ITEMS>(parent + fbase, base, n, rt);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rt[j] = rank_of_root[fbase + rt[j]];
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rt[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        int lab = -1;
        if (e < n) {
            lab = rt[j];
            if (kbase == 0) label[fbase + e] = lab;
        }
#pragma unroll
        for (int k = 0; k < KICP; ++k) cnt[k] += __popcll(__ballot(lab == kbase + k));
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < KICP; ++k) s_c[w][k] = cnt[k];
    }
    __syncthreads();
    if (threadIdx.x < KICP) {
        const int k = threadIdx.x;
        tile_cnt[((size_t)f * KICP + k) * T + tile] = s_c[0][k] + s_c[1][k] + s_c[2][k] + s_c[3][k];
    }
}

// koff_tab: offsets of the round's clusters in the frame's ICP source segment, [F][KICP] (NULL in round 0: fs[f].koff)
__global__ void __launch_bounds__(BLOCK) k_label_scatter(const float4* __restrict__ obj, int N, int T,
                                                         const FrameState* __restrict__ fs, const int* __restrict__ label,
                                                         const int* __restrict__ tile_off, float4* __restrict__ src0,
                                                         float4* __restrict__ src, int kbase, const int* __restrict__ koff_tab) {
    CD_FRONT_PRIO();
    __shared__ int s_c[WAVES_PER_BLOCK][KICP];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_o;
    if (tile * TILE >= n) return;
    if (kbase > 0 && kbase >= fs[f].n_k) return;
    const size_t fbase = (size_t)f * N;
    const int base = tile * TILE + w * WAVE_SPAN + lane;
    int lab[ITEMS];
    int cnt[KICP];
#pragma unroll
    for (int k = 0; k < KICP; ++k) cnt[k] = 0;
    float4 pr[ITEMS];
    load_rows_clamped<ITEMS>(label + fbase, base, n, lab);   // (labels and points of all rows before any is used)
    load_rows_clamped<ITEMS>(obj + fbase, base, n, pr);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        lab[j] = e < n ? lab[j] - kbase : -1;   // rank within the round (other rounds' clusters fall outside 0..KICP-1)
#pragma unroll
        for (int k = 0; k < KICP; ++k) cnt[k] += __popcll(__ballot(lab[j] == k));
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < KICP; ++k) s_c[w][k] = cnt[k];
    }
    __syncthreads();
    int pos[KICP];
#pragma unroll
    for (int k = 0; k < KICP; ++k) {
        int p = (koff_tab ? koff_tab[f * KICP + k] : fs[f].koff[k]) + tile_off[((size_t)f * KICP + k) * T + tile];
        for (int q = 0; q < w; ++q) p += s_c[q][k];
        pos[k] = p;
    }
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
#pragma unroll
        for (int k = 0; k < KICP; ++k) {
            const uint64_t bal = __ballot(lab[j] == k);
            if (lab[j] == k) {
                const float4 p = pr[j];
                const int d = pos[k] + __popcll(bal & lt);
                src0[fbase + d] = p;
                src[fbase + d] = p;
            }
            pos[k] += __popcll(bal);
        }
    }
}

static inline int grid_for(int n_max) {
    int g = (n_max + BLOCK - 1) / BLOCK;
    return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

void launch_cluster_lds(hipStream_t s, const float4* obj, int N, int F, FrameState* fs, float inv_cell, float r2,
                        int* parent, int* csize, int* rank_of_root) {
    hipLaunchKernelGGL(k_cluster_lds, dim3(F), dim3(CL_THREADS), 0, s, obj, N, fs, inv_cell, r2, parent, csize, rank_of_root);
}
void launch_cluster_cells(hipStream_t s, const float4* obj, int N, int F, FrameState* fs, float inv_cell, float r2,
                          int* parent, int* csize, int* rank_of_root, float4* sorted) {
    hipLaunchKernelGGL(k_cluster_cells, dim3(F), dim3(CL_THREADS), 0, s, obj, N, fs, inv_cell, r2, parent, csize, rank_of_root, sorted);
}
void launch_cluster_build(hipStream_t s, const float4* obj, int N, int F, int Tact, const FrameState* fs, float inv_cell,
                          int* head, int* next, int* parent, int* csize, int* rank_of_root) {
    hipLaunchKernelGGL(k_cluster_build, dim3(grid_for(Tact * TILE), F), dim3(BLOCK), 0, s, obj, N, fs, inv_cell, head, next,
                       parent, csize, rank_of_root);
}
void launch_cluster_hook(hipStream_t s, const float4* obj, int N, int F, int Tact, const FrameState* fs, float inv_cell,
                         float r2, const int* head, const int* next, int* parent) {
    hipLaunchKernelGGL(k_cluster_hook, dim3(grid_for(Tact * TILE), F), dim3(BLOCK), 0, s, obj, N, fs, inv_cell, r2, head,
                       next, parent);
}
void launch_cluster_flatten(hipStream_t s, int N, int F, int Tact, const FrameState* fs, int* parent, int* csize) {
    hipLaunchKernelGGL(k_cluster_flatten, dim3(grid_for(Tact * TILE), F), dim3(BLOCK), 0, s, N, fs, parent, csize);
}
void launch_cluster_rank(hipStream_t s, int N, int F, FrameState* fs, int enable, int min_sz, int max_sz,
                         const int* parent, const int* csize, int* cand, int* rank_of_root, int* sizes_sorted, FrameState* mirror) {
    hipLaunchKernelGGL(k_cluster_rank, dim3(F), dim3(BLOCK), 0, s, N, fs, enable, min_sz, max_sz, parent, csize, cand,
                       rank_of_root, sizes_sorted, mirror);
}
void launch_label_count(hipStream_t s, int N, int F, int T, int Tact, const FrameState* fs, int enable, const int* parent,
                        const int* rank_of_root, int* label, int* tile_cnt, int kbase) {
    hipLaunchKernelGGL(k_label_count, dim3(Tact, F), dim3(BLOCK), 0, s, N, T, fs, enable, parent, rank_of_root, label,
                       tile_cnt, kbase);
}
void launch_label_scatter(hipStream_t s, const float4* obj, int N, int F, int T, int Tact, const FrameState* fs,
                          const int* label, const int* tile_off, float4* src0, float4* src, int kbase, const int* koff_tab) {
    hipLaunchKernelGGL(k_label_scatter, dim3(Tact, F), dim3(BLOCK), 0, s, obj, N, T, fs, label, tile_off, src0, src, kbase,
                       koff_tab);
}

}  // namespace cd
