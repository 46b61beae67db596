// k_sort.hip - segmented stable LSD radix sort (8-bit digits), one segment per frame.
//
// VoxelGrid sorts the cropped points by voxel index (pcl::VoxelGrid::applyFilter uses
// std::sort on (idx, point) pairs; rule C2 makes it stable).  Single-histogram ("onesweep") organisation:
//   ghist   : ONE pass over the keys builds the per-frame digit histograms of ALL passes (the histogram of a digit does
//             not depend on the order of the elements)
//   scatter : per pass and ordered tile of 8192 pairs - stable rank of each pair inside its tile with wave ballots
//             ("match-any" over the 8 digit bits, 64-wide); the tile's bin counts go through a chained scan over the
//             frame's tiles (one thread per bin, common.hpp chained_scan) and give, with the exclusive scan of the
//             frame's digit histogram, where the tile's part of every bin starts; the tile is put in bin order in LDS and
//             written out as contiguous runs.
// Element order inside a tile is (wave, row, lane), so per-wave running bin counts kept in
// LDS plus a cross-wave prefix give the stable position.  A sort tile is 8192 pairs (1024 threads x 8 rows): with 256
// bins a tile sends ~32 consecutive pairs to each bin.  (Rounds 1-2 ran a per-tile histogram kernel and a scan kernel before
// every scatter: three reads of the keys more per sort.)
#include "kernels.hpp"

namespace cd {

constexpr int SORT_WAVES = SORT_BLOCK / WAVE;

// The first pass after a single-pass crop reads absolute coordinate fields (KeyPack) and turns them into PCL's voxel index,
// with the arithmetic of VoxelGrid::applyFilter: ijk = (int)(floor(p * inv_leaf) - min_b) in float, idx = i + j dx + k dx dy.
struct KeyGrid {
    float mb0, mb1, mb2;
    int d0, d01;
};
__device__ __forceinline__ KeyGrid key_grid(const FrameState& s) {
    KeyGrid g;
    g.mb0 = (float)s.min_b[0]; g.mb1 = (float)s.min_b[1]; g.mb2 = (float)s.min_b[2];
    g.d0 = s.div_b[0]; g.d01 = s.div_b[0] * s.div_b[1];
    return g;
}
__device__ __forceinline__ uint32_t voxel_key(uint32_t a, const KeyPack& kp, const KeyGrid& g) {
    const int fx = (int)(a & ((1u << kp.bi) - 1u)) + kp.ilo;
    const int fy = (int)((a >> kp.bi) & ((1u << kp.bj) - 1u)) + kp.jlo;
    const int fz = (int)(a >> (kp.bi + kp.bj)) + kp.klo;
    const int i0 = (int)__fsub_rn((float)fx, g.mb0), i1 = (int)__fsub_rn((float)fy, g.mb1), i2 = (int)__fsub_rn((float)fz, g.mb2);
    return (uint32_t)(i0 + i1 * g.d0 + i2 * g.d01);
}

#ifdef CD_SORTDBG
__device__ unsigned long long g_sort_dbg[8];
extern "C" int cd_debug_sort(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_dbg), sizeof(g_sort_dbg)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sort_dbg), z, sizeof(z)); }
    return 0;
}
#define SORT_PHASE(k) { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_sort_dbg[k], t_ - t0_); t0_ = t_; } }
#else
#define SORT_PHASE(k)
#endif
constexpr int SORT_MAX_PASSES = 4;
constexpr int GHIST_TILES = 4;   // sort tiles per workgroup of the histogram kernel (fewer global flushes)

// ghist[f][pass][digit] += occurrences, for every pass at once.  Neighbouring elements mostly carry the same digit
// (neighbouring pixels): only the first lane of a run of equal digits adds, the run length at once - the LDS serialises
// lanes that add to one address, which is the whole cost of a histogram.
__global__ void __launch_bounds__(SORT_BLOCK) k_radix_ghist(const uint32_t* __restrict__ kin, int N, int npass,
                                                            const FrameState* __restrict__ fs, uint32_t* __restrict__ ghist,
                                                            KeyPack kp) {
    CD_FRONT_PRIO();
    __shared__ uint32_t s_h[SORT_MAX_PASSES][RADIX];
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int n = fs[f].n_c;
    const int e0 = blockIdx.x * GHIST_TILES * SORT_TILE;
    if (e0 >= n) return;
    for (int q = threadIdx.x; q < SORT_MAX_PASSES * RADIX; q += SORT_BLOCK) (&s_h[0][0])[q] = 0;
    __syncthreads();
    const uint32_t* k = kin + (size_t)f * N;
    const KeyGrid g = key_grid(fs[f]);
    for (int t = 0; t < GHIST_TILES; ++t) {
        const int base = e0 + t * SORT_TILE + (threadIdx.x >> 6) * WAVE_SPAN + lane;
        if (e0 + t * SORT_TILE >= n) break;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = base + j * WAVE;
            const bool valid = e < n;
            uint32_t key = 0;
            if (valid) key = kp.enabled ? voxel_key(k[e], kp, g) : k[e];
            for (int p = 0; p < npass; ++p) {
                const uint32_t d = valid ? ((key >> (p * RADIX_BITS)) & (RADIX - 1)) : 0xffffffffu;   // past the end: own run
                const uint32_t prev = (uint32_t)__shfl_up((int)d, 1, 64);
                const uint64_t heads = __ballot(lane == 0 || d != prev);
                const uint64_t above = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
                const int next = above ? __ffsll((long long)above) - 1 : 64;
                if (((heads >> lane) & 1ull) && valid) atomicAdd(&s_h[p][d], (uint32_t)(next - lane));
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < npass * RADIX; q += SORT_BLOCK) {
        const uint32_t c = (&s_h[0][0])[q];
        if (c) atomicAdd(&ghist[(size_t)f * SORT_MAX_PASSES * RADIX + q], c);
    }
}

__global__ void __launch_bounds__(SORT_BLOCK) k_radix_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                         uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, int N,
                                                         int T, int Tact, int pass, int shift, FrameState* __restrict__ fs,
                                                         const uint32_t* __restrict__ ghist, int* __restrict__ state,
                                                         KeyPack kp, int use_runs, int* __restrict__ ticket) {
    CD_FRONT_PRIO();
    __shared__ int s_ticket;
    __shared__ unsigned short s_wh[SORT_WAVES][RADIX];   // per-wave bin counts, then the wave's offset inside the bin
    __shared__ uint32_t s_goff[RADIX];                   // where the tile's part of each bin starts in the frame
    __shared__ uint32_t s_bstart[RADIX];                 // where each bin starts inside the tile
    __shared__ uint32_t s_k[SORT_TILE], s_v[SORT_TILE];  // the tile, ordered by bin (64 KiB)
    __shared__ uint32_t s_ws[RADIX / WAVE], s_gs[RADIX / WAVE];
    // workgroup b works on frame b % F and takes its tile by ticket (see k_crop_fused): a tile's predecessors in the chained
    // scan are long done
    const int F = gridDim.x / Tact;   // (pass: row of the frame's histogram table; shift: bit position of the digit)
    const int f = blockIdx.x % F, tile = take_ticket(ticket + f * TICKET_PITCH, Tact, &s_ticket), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = use_runs ? fs[f].n_runs : fs[f].n_c;   // (runs: the elements are k_voxel_runs' (voxel index, start | length) pairs)
    if (tile * SORT_TILE >= n) return;
    const size_t fbase = (size_t)f * N;
#ifdef CD_SORTDBG
    unsigned long long t0_ = wall_clock64();
    if (threadIdx.x == 0) atomicAdd(&g_sort_dbg[7], 1ull);
#endif
    for (int q = threadIdx.x; q < SORT_WAVES * RADIX; q += SORT_BLOCK) (&s_wh[0][0])[q] = 0;
    __syncthreads();
    const int base = tile * SORT_TILE + w * WAVE_SPAN + lane;
    const uint64_t lt = lanemask_lt();
    const KeyGrid g = key_grid(fs[f]);
    uint32_t key[ITEMS], rank[ITEMS], val[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {   // all loads of the tile go out together (the values are only needed for the staging)
        const int e = base + j * WAVE;
        key[j] = e < n ? kin[fbase + e] : 0xffffffffu;
        val[j] = e < n ? (vin ? vin[fbase + e] : (uint32_t)e) : 0u;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        const bool valid = e < n;
        if (kp.enabled && valid) key[j] = voxel_key(key[j], kp, g);
        const uint32_t d = (key[j] >> shift) & (RADIX - 1);
        // match-any over the 8 digit bits: lanes that differ from this one in some bit are collected in two 32-bit halves
        // (per bit: sign-extended bit, one compare for the ballot, xor + or per half - written out this way because the
        // 64-bit select form "peers &= bit ? m : ~m" compiled to twice the vector instructions, and this loop is what the
        // scatter's ranking costs)
        uint32_t dl = 0u, dh = 0u;
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const int y = (int)(d << (31 - b)) >> 31;          // all ones where this lane's bit b is set
            const uint64_t m = __ballot(y < 0);
            dl |= (uint32_t)m ^ (uint32_t)y;
            dh |= (uint32_t)(m >> 32) ^ (uint32_t)y;
        }
        const uint64_t peers = ~(((uint64_t)dh << 32) | dl) & __ballot(valid);   // same digit, and a real element
        // every lane reads its bin's running count, then the LAST lane of each group of equal digits adds the group's size
        // (the LDS executes a wave's operations in order: all reads see the count before any of this row's updates)
        uint32_t bin_base = 0;
        if (valid) bin_base = s_wh[w][d];
        if (valid && (peers >> lane) == 1ull) s_wh[w][d] = (unsigned short)(bin_base + (uint32_t)__popcll(peers));
        rank[j] = bin_base + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    SORT_PHASE(0)
    if (threadIdx.x < RADIX) {   // per-digit exclusive prefix over the waves, then over the digits
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < SORT_WAVES; ++q) {
            const uint32_t c = s_wh[q][d];
            s_wh[q][d] = (unsigned short)run;
            run += c;
        }
        // where the frame's bin d starts: exclusive scan of the frame's digit histogram ...
        const uint32_t tot = ghist[((size_t)f * SORT_MAX_PASSES + pass) * RADIX + d];
        uint32_t inc = run, ginc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64), gu = __shfl_up(ginc, o, 64);
            if (lane >= o) { inc += u; ginc += gu; }
        }
        if (lane == 63) { s_ws[w] = inc; s_gs[w] = ginc; }
        s_bstart[d] = inc - run;   // both completed below with the totals of the lower waves
        // ... plus what the tiles before this one put into bin d (chained scan over the frame's tiles, one thread per bin)
        const int before = chained_scan(state + (size_t)f * T * RADIX + d, RADIX, tile, (int)run, &fs[f].scan_stalled);
        s_goff[d] = ginc - tot + (uint32_t)before;
    }
    __syncthreads();
    if (threadIdx.x < RADIX) {
        uint32_t add = 0, gadd = 0;
        for (int q = 0; q < w; ++q) { add += s_ws[q]; gadd += s_gs[q]; }
        s_bstart[threadIdx.x] += add;
        s_goff[threadIdx.x] += gadd;
    }
    __syncthreads();
    SORT_PHASE(1)
    // the tile in bin order, staged in LDS: the global stores below are then contiguous runs (one run per bin) instead of
    // 64 scattered 4-byte stores per instruction
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        if (e < n) {
            const uint32_t d = (key[j] >> shift) & (RADIX - 1);
            const uint32_t lp = s_bstart[d] + s_wh[w][d] + rank[j];
            if (CD_IN_RANGE(lp < (uint32_t)SORT_TILE, 2u)) {
                s_k[lp] = key[j];
                s_v[lp] = val[j];
            }
        }
    }
    __syncthreads();
    SORT_PHASE(2)
    const int cnt = min(SORT_TILE, n - tile * SORT_TILE);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int lp = threadIdx.x + j * SORT_BLOCK;
        if (lp < cnt) {
            const uint32_t k = s_k[lp];
            const uint32_t d = (k >> shift) & (RADIX - 1);
            const uint32_t dst = s_goff[d] + ((uint32_t)lp - s_bstart[d]);
            if (CD_IN_RANGE(dst < (uint32_t)n, 3u)) {
                kout[fbase + dst] = k;
                vout[fbase + dst] = s_v[lp];
            }
        }
    }
    SORT_PHASE(3)
}

// All passes of one sort.  Tact = sort tiles that hold data (max over the frames).  ghist [F][SORT_MAX_PASSES][RADIX] and
// state [npass][F][Tact][RADIX] are zeroed here.  kp.enabled: key[0] holds the
// absolute coordinate fields of k_crop_fused; the first pass writes voxel indices.  Returns the index of the buffers
// that hold the sorted keys / the permutation, or -1 when the scan state could not be zeroed.
int launch_radix_sort(hipStream_t s, uint32_t* const key[2], uint32_t* const val[2], int N, int F, int Tact, int npass,
                      FrameState* fs, uint32_t* ghist, int* state, KeyPack kp, int* ticket) {
    if (npass <= 0) return 0;
    // (a failure is also left in hipGetLastError, which the caller's LAUNCH() reads; -1 makes it explicit)
    if (hipMemsetAsync(ghist, 0, sizeof(uint32_t) * (size_t)F * SORT_MAX_PASSES * RADIX, s) != hipSuccess) return -1;
    if (hipMemsetAsync(state, 0, sizeof(int) * (size_t)npass * F * Tact * RADIX, s) != hipSuccess) return -1;
    const int G = (Tact + GHIST_TILES - 1) / GHIST_TILES;
    hipLaunchKernelGGL(k_radix_ghist, dim3(G, F), dim3(SORT_BLOCK), 0, s, key[0], N, npass, fs, ghist, kp);
    int cur = 0;
    for (int pass = 0; pass < npass; ++pass) {
        hipLaunchKernelGGL(k_radix_scatter, dim3(Tact * F), dim3(SORT_BLOCK), 0, s, key[cur], pass ? val[cur] : nullptr, key[cur ^ 1],
                           val[cur ^ 1], N, Tact, Tact, pass, pass * RADIX_BITS, fs, ghist, state + (size_t)pass * F * Tact * RADIX, kp, 0, ticket);
        kp.enabled = 0;   // later passes read voxel indices
        cur ^= 1;
    }
    return cur;
}

// ---- S1 by runs ------------------------------------------------------------------------
// The cropped points are in image order, and neighbouring pixels of a row mostly fall into the same voxel: on the bench
// frames 121 k cropped points are 51 k runs of equal voxel index (17.7 k voxels).  A stable sort of the RUNS by voxel index
// leaves every voxel's points in ascending input order just as a stable sort of the points does (rule C2), with 2.4 x fewer
// elements to move; the centroid kernel then reads each run's points contiguously (k_voxel_centroid_runs).
// k_voxel_runs: one pass over the keys of an ordered tile - voxel index (from the absolute coordinate fields of a single-pass
// crop), run heads (a run also ends at every multiple of 64, so its length is known inside the row), (voxel index, start |
// length << 20) written at the frame's running run count (chained scan over the tiles), n_runs, and the digit histograms of
// all sort passes over the run keys (what k_radix_ghist does for a sort of the points).
constexpr int RUN_SHIFT = 20;   // start < 2^20 (CD_MAX_POINTS), length <= 64
__global__ void __launch_bounds__(SORT_BLOCK) k_voxel_runs(const uint32_t* __restrict__ kin, int N, int T, int Tact, int npass,
                                                           FrameState* __restrict__ fs, uint32_t* __restrict__ ghist,
                                                           int* __restrict__ state, uint32_t* __restrict__ kout,
                                                           uint32_t* __restrict__ vout, KeyPack kp, int* __restrict__ ticket) {
    CD_FRONT_PRIO();
    __shared__ uint32_t s_h[SORT_MAX_PASSES][RADIX];
    __shared__ int s_cnt[SORT_WAVES];
    __shared__ int s_out0, s_ticket;
    const int F = gridDim.x / Tact;
    const int f = blockIdx.x % F, tile = take_ticket(ticket + f * TICKET_PITCH, Tact, &s_ticket), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_c;
    if (tile * SORT_TILE >= n) return;
    for (int q = threadIdx.x; q < SORT_MAX_PASSES * RADIX; q += SORT_BLOCK) (&s_h[0][0])[q] = 0;
    const size_t fbase = (size_t)f * N;
    const uint32_t* k = kin + fbase;
    const KeyGrid g = key_grid(fs[f]);
    const int base = tile * SORT_TILE + w * WAVE_SPAN + lane;
    uint32_t key[ITEMS];
    uint64_t heads[ITEMS];
    int nvalid[ITEMS], wtot = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        const bool valid = e < n;
        key[j] = 0xffffffffu;
        if (valid) key[j] = kp.enabled ? voxel_key(k[e], kp, g) : k[e];
        const uint32_t prev = (uint32_t)__shfl_up((int)key[j], 1, 64);
        heads[j] = __ballot(valid && (lane == 0 || key[j] != prev));
        nvalid[j] = __popcll(__ballot(valid));
        wtot += __popcll(heads[j]);
    }
    if (lane == 0) s_cnt[w] = wtot;
    __syncthreads();
    int pos = 0, nheads = 0;
    for (int q = 0; q < SORT_WAVES; ++q) { if (q < w) pos += s_cnt[q]; nheads += s_cnt[q]; }
    if (threadIdx.x == 0) {
        const int excl = chained_scan(state + (size_t)f * T, 1, tile, nheads, &fs[f].scan_stalled);
        s_out0 = excl;
        if ((tile + 1) * SORT_TILE >= n) fs[f].n_runs = excl + nheads;
    }
    __syncthreads();
    const int out0 = s_out0;
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool is_head = (heads[j] >> lane) & 1ull;
        if (is_head) {
            const uint64_t above = lane == 63 ? 0ull : heads[j] & ~((2ull << lane) - 1ull);
            const int next = above ? __ffsll((long long)above) - 1 : nvalid[j];
            const size_t o = fbase + (size_t)(out0 + pos + __popcll(heads[j] & lt));
            kout[o] = key[j];
            vout[o] = (uint32_t)(base + j * WAVE) | ((uint32_t)(next - lane) << RUN_SHIFT);
        }
        // digit histograms of the run keys.  Above the lowest digit the heads of a row (64 neighbouring pixels) nearly always
        // share the digit: one add of their number instead of that many adds to one LDS word, which the LDS would serialise
        // (30 % of the kernel).
        if (heads[j]) {
            const int first = __ffsll((long long)heads[j]) - 1;
            for (int p = 0; p < npass; ++p) {
                const uint32_t d = (key[j] >> (p * RADIX_BITS)) & (RADIX - 1);
                if (p > 0) {
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
                    if (__ballot(is_head && d != d0) == 0ull) {
                        if (lane == first) atomicAdd(&s_h[p][d0], (uint32_t)__popcll(heads[j]));
                        continue;
                    }
                }
                if (is_head) atomicAdd(&s_h[p][d], 1u);
            }
        }
        pos += __popcll(heads[j]);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < npass * RADIX; q += SORT_BLOCK) {
        const uint32_t c = (&s_h[0][0])[q];
        if (c) atomicAdd(&ghist[(size_t)f * SORT_MAX_PASSES * RADIX + q], c);
    }
}

// The sort of a batch by runs: key[0] holds the crop's keys; the runs go to key[1] / val[1] and the passes alternate from
// there.  `tile_state` [F][T] (zeroed here) carries the chained scan of the run counts.  Returns the index of the buffers that
// hold the sorted run keys / payloads, or -1 when a scan state could not be zeroed.
int launch_radix_sort_runs(hipStream_t s, uint32_t* const key[2], uint32_t* const val[2], int N, int F, int T, int Tact, int npass,
                           FrameState* fs, uint32_t* ghist, int* state, int* tile_state, KeyPack kp, int* ticket) {
    if (npass <= 0) return 0;
    if (hipMemsetAsync(ghist, 0, sizeof(uint32_t) * (size_t)F * SORT_MAX_PASSES * RADIX, s) != hipSuccess) return -1;
    if (hipMemsetAsync(state, 0, sizeof(int) * (size_t)npass * F * Tact * RADIX, s) != hipSuccess) return -1;
    if (hipMemsetAsync(tile_state, 0, sizeof(int) * (size_t)F * T, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_voxel_runs, dim3(Tact * F), dim3(SORT_BLOCK), 0, s, key[0], N, T, Tact, npass, fs, ghist, tile_state, key[1],
                       val[1], kp, ticket);
    kp.enabled = 0;
    int cur = 1;
    for (int pass = 0; pass < npass; ++pass) {
        hipLaunchKernelGGL(k_radix_scatter, dim3(Tact * F), dim3(SORT_BLOCK), 0, s, key[cur], val[cur], key[cur ^ 1], val[cur ^ 1], N,
                           Tact, Tact, pass, pass * RADIX_BITS, fs, ghist, state + (size_t)pass * F * Tact * RADIX, kp, 1, ticket);
        cur ^= 1;
    }
    return cur;
}

// The scatters of a sort whose run records and histograms k_crop_runs has already written (key[0] / val[0], ghist): one pass
// per digit of the packed key that varies in some frame, lowest first.  `state` [ndigits][F][Tact][RADIX] is zeroed here.
int launch_radix_scatter_runs(hipStream_t s, uint32_t* const key[2], uint32_t* const val[2], int N, int F, int Tact, const int* digits, int ndigits,
                              FrameState* fs, const uint32_t* ghist, int* state, int* ticket) {
    if (ndigits <= 0) return 0;
    if (hipMemsetAsync(state, 0, sizeof(int) * (size_t)ndigits * F * Tact * RADIX, s) != hipSuccess) return -1;
    KeyPack none;
    none.enabled = 0; none.bi = none.bj = 0; none.ilo = none.jlo = none.klo = 0;
    int cur = 0;
    for (int q = 0; q < ndigits; ++q) {
        hipLaunchKernelGGL(k_radix_scatter, dim3(Tact * F), dim3(SORT_BLOCK), 0, s, key[cur], val[cur], key[cur ^ 1], val[cur ^ 1], N,
                           Tact, Tact, digits[q], digits[q] * RADIX_BITS, fs, ghist, state + (size_t)q * F * Tact * RADIX, none, 1, ticket);
        cur ^= 1;
    }
    return cur;
}

}  // namespace cd
