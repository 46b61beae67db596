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

constexpr int CL_LDS_CAP_GLOBAL_SKIP = 8192;   // == CL_LDS_CAP: frames this small are clustered in LDS

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
    const int f = blockIdx.y;
    const int n = fs[f].n_o;
    if (n <= CL_LDS_CAP_GLOBAL_SKIP) return;   // handled by k_cluster_lds
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
    const int f = blockIdx.y;
    const int n = fs[f].n_o;
    if (n <= CL_LDS_CAP_GLOBAL_SKIP) return;   // handled by k_cluster_lds
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
    const int f = blockIdx.y;
    const int n = fs[f].n_o;
    if (n <= CL_LDS_CAP_GLOBAL_SKIP) return;   // handled by k_cluster_lds
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
// Points, the cell hash (heads + next links) and the union-find parents all live in LDS, so the
// hooking phase runs at LDS latency and is coherent by construction (one CU).  Produces exactly what
// build + hook + flatten produce (parent = root = smallest member index, component sizes).
constexpr int CL_LDS_CAP = 8192;
constexpr int CL_LDS_BUCKETS = 8192;
constexpr int CL_THREADS = 1024;
constexpr int CL_PER_THREAD = CL_LDS_CAP / CL_THREADS;

__device__ __forceinline__ uint32_t cell_hash_lds(int cx, int cy, int cz) {
    return ((uint32_t)cx * 73856093u ^ (uint32_t)cy * 19349663u ^ (uint32_t)cz * 83492791u) & (CL_LDS_BUCKETS - 1);
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

__global__ void __launch_bounds__(CL_THREADS) k_cluster_lds(const float4* __restrict__ obj, int N,
                                                            const FrameState* __restrict__ fs, float inv_cell, float r2,
                                                            int* __restrict__ parent, int* __restrict__ csize,
                                                            int* __restrict__ rank_of_root) {
    __shared__ int s_par[CL_LDS_CAP];         // 32 KiB
    __shared__ int s_next[CL_LDS_CAP];        // 32 KiB (reused as the size counters at the end)
    __shared__ int s_head[CL_LDS_BUCKETS];    // 32 KiB
    const int f = blockIdx.x;
    const int n = fs[f].n_o;
    if (n <= 0 || n > CL_LDS_CAP) return;     // larger frames take the global-memory path
    const size_t fbase = (size_t)f * N;
    const float4* __restrict__ P = obj + fbase;   // read-only in this kernel: plain cached loads
    const float org[3] = {fs[f].origin[0], fs[f].origin[1], fs[f].origin[2]};
    for (int i = threadIdx.x; i < CL_LDS_BUCKETS; i += CL_THREADS) s_head[i] = -1;
    for (int i = threadIdx.x; i < n; i += CL_THREADS) s_par[i] = i;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += CL_THREADS) {
        int cx, cy, cz;
        cell_of(P[i], org, inv_cell, cx, cy, cz);
        s_next[i] = atomicExch(&s_head[cell_hash_lds(cx, cy, cz)], i);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += CL_THREADS) {
        const float4 p = P[i];
        int cx, cy, cz;
        cell_of(p, org, inv_cell, cx, cy, cz);
        int ri = lds_find(s_par, i);
        // Each unordered pair of neighbouring cells is examined once: the own cell (pairs j < i) and the
        // 13 "forward" cells (dz > 0, or dz == 0 && dy > 0, or dz == dy == 0 && dx > 0), all their points.
        // (Two different cells may share a hash bucket; a pair met twice is just a redundant union.)
        for (int nb = 0; nb < 14; ++nb) {
            const int code = nb == 0 ? 13 : 13 + nb;               // 13 = (0,0,0); 14..26 = forward half
            const int a = code % 3 - 1, b = (code / 3) % 3 - 1, c = code / 9 - 1;
            int j = s_head[cell_hash_lds(cx + a, cy + b, cz + c)];
            while (j >= 0) {
                int jx, jy, jz;
                const float4 q = P[j];
                bool take;
                if (nb == 0) {
                    take = j < i;
                } else {   // the bucket may also hold points of other cells (hash collisions): keep exact cell matches only
                    cell_of(q, org, inv_cell, jx, jy, jz);
                    take = jx == cx + a && jy == cy + b && jz == cz + c;
                }
                // a neighbour that already hangs directly under i's root needs no find and no union (most pairs of a big
                // cluster once the first unions and path halvings have happened): one LDS read instead of a chain walk
                if (take && dist2(p.x, p.y, p.z, q.x, q.y, q.z) < r2 && s_par[j] != ri) {
                    int rj = lds_find(s_par, j);
                    while (ri != rj) {
                        const int hi = ri > rj ? ri : rj, lo = ri > rj ? rj : ri;
                        const int old = atomicCAS(&s_par[hi], hi, lo);   // link the larger root under the smaller
                        if (old == hi) { ri = lo; rj = lo; break; }
                        ri = lds_find(s_par, ri);                        // hi had been linked meanwhile: climb and retry
                        rj = lds_find(s_par, rj);
                    }
                    ri = ri < rj ? ri : rj;
                }
                j = s_next[j];
            }
        }
    }
    __syncthreads();
    int root[CL_PER_THREAD];
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CL_THREADS;
        root[k] = i < n ? lds_find(s_par, i) : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += CL_THREADS) s_next[i] = 0;   // links are dead: reuse as size counters
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CL_THREADS;
        if (i < n) atomicAdd(&s_next[root[k]], 1);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CL_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CL_THREADS;
        if (i < n) {
            parent[fbase + i] = root[k];
            csize[fbase + i] = s_next[i];
            rank_of_root[fbase + i] = -1;
        }
    }
}

// one block per frame
__global__ void __launch_bounds__(BLOCK) k_cluster_rank(int N, FrameState* __restrict__ fs, int enable, int min_sz,
                                                        int max_sz, const int* __restrict__ parent,
                                                        const int* __restrict__ csize, int* __restrict__ cand,
                                                        int* __restrict__ rank_of_root, int* __restrict__ sizes_sorted) {
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
    }
}

// Clusters are extracted for ICP KICP at a time: round `kbase` handles the clusters ranked kbase .. kbase+KICP-1 (a frame
// rarely has more than KICP; the host runs further rounds only for frames that do - opd.cpp:376 gives EVERY cluster its ICP).
__global__ void __launch_bounds__(BLOCK) k_label_count(int N, int T, const FrameState* __restrict__ fs, int enable,
                                                       const int* __restrict__ parent, const int* __restrict__ rank_of_root,
                                                       int* __restrict__ label, int* __restrict__ tile_cnt, int kbase) {
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
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        int lab = -1;
        if (e < n) {
            lab = enable ? rank_of_root[fbase + parent[fbase + e]] : 0;
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
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        lab[j] = e < n ? label[fbase + e] - kbase : -1;   // rank within the round (other rounds' clusters fall outside 0..KICP-1)
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
        const int e = base + j * WAVE;
#pragma unroll
        for (int k = 0; k < KICP; ++k) {
            const uint64_t bal = __ballot(lab[j] == k);
            if (lab[j] == k) {
                const float4 p = obj[fbase + e];
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

void launch_cluster_lds(hipStream_t s, const float4* obj, int N, int F, const FrameState* fs, float inv_cell, float r2,
                        int* parent, int* csize, int* rank_of_root) {
    static_assert(CL_LDS_CAP == CL_LDS_CAP_GLOBAL_SKIP, "LDS / global split must agree");
    hipLaunchKernelGGL(k_cluster_lds, dim3(F), dim3(CL_THREADS), 0, s, obj, N, fs, inv_cell, r2, parent, csize, rank_of_root);
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
                         const int* parent, const int* csize, int* cand, int* rank_of_root, int* sizes_sorted) {
    hipLaunchKernelGGL(k_cluster_rank, dim3(F), dim3(BLOCK), 0, s, N, fs, enable, min_sz, max_sz, parent, csize, cand,
                       rank_of_root, sizes_sorted);
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
