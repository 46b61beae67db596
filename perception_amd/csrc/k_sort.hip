// k_sort.hip - segmented stable LSD radix sort (8-bit digits), one segment per frame.
//
// VoxelGrid sorts the cropped points by voxel index (pcl::VoxelGrid::applyFilter uses
// std::sort on (idx, point) pairs; rule C2 makes it stable).  One pass = three kernels over
// ordered tiles of 2048 pairs:
//   hist    : per-tile digit histogram (LDS atomics)            -> hist[f][tile][digit]  (digit fastest: every
//             access of the three kernels to it is coalesced)
//   scan    : per frame, exclusive scan in (digit, tile) order  -> global base of every bin
//   scatter : stable rank of each pair inside its tile with wave ballots ("match-any" over
//             the 8 digit bits, 64-wide), then one scattered store per pair.
// Element order inside a tile is (wave, row, lane), so per-wave running bin counts kept in
// LDS plus a cross-wave prefix give the stable position.  A sort tile is 8192 pairs (1024 threads x 8 rows): with 256
// bins a tile sends ~32 consecutive pairs to each bin, so the scattered 4-byte stores fill whole 128-byte lines (with
// 2048-pair tiles the scatter wrote twice the bytes it stored).
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

__global__ void __launch_bounds__(SORT_BLOCK) k_radix_hist(const uint32_t* __restrict__ kin, int N, int T, int shift,
                                                           const FrameState* __restrict__ fs, uint32_t* __restrict__ hist,
                                                           KeyPack kp) {
    __shared__ uint32_t s_h[RADIX];
    const int f = blockIdx.y, tile = blockIdx.x;
    const int n = fs[f].n_c;
    if (tile * SORT_TILE >= n) return;
    if (threadIdx.x < RADIX) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* k = kin + (size_t)f * N;
    const int base = tile * SORT_TILE + (threadIdx.x >> 6) * WAVE_SPAN + (threadIdx.x & 63);
    const KeyGrid g = key_grid(fs[f]);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        // Neighbouring elements mostly carry the same digit (neighbouring pixels, or an input ordered by the lower digits):
        // only the first lane of a run of equal digits adds, the run length at once - the LDS serialises lanes that add to
        // one address, which was the whole cost of this kernel.
        uint32_t d = 0xffffffffu;   // past the end: a run of its own that adds nothing
        if (e < n) {
            const uint32_t key = kp.enabled ? voxel_key(k[e], kp, g) : k[e];
            d = (key >> shift) & (RADIX - 1);
        }
        const uint32_t prev = (uint32_t)__shfl_up((int)d, 1, 64);
        const uint64_t heads = __ballot(lane == 0 || d != prev);
        const uint64_t above = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
        const int next = above ? __ffsll((long long)above) - 1 : 64;
        if (((heads >> lane) & 1ull) && e < n) atomicAdd(&s_h[d], (uint32_t)(next - lane));
    }
    __syncthreads();
    if (threadIdx.x < RADIX) hist[((size_t)f * T + tile) * RADIX + threadIdx.x] = s_h[threadIdx.x];
}

// one block per frame; thread d owns digit d's row of Tact tiles
__global__ void __launch_bounds__(BLOCK) k_radix_scan(int T, const FrameState* __restrict__ fs, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_w[WAVES_PER_BLOCK];
    const int f = blockIdx.x, d = threadIdx.x, w = d >> 6, lane = d & 63;
    const int n = fs[f].n_c;
    if (n <= 0) return;
    const int tact = (n + SORT_TILE - 1) / SORT_TILE;
    uint32_t* col = hist + (size_t)f * T * RADIX + d;
    uint32_t sum = 0;
    for (int t = 0; t < tact; ++t) {
        const uint32_t v = col[(size_t)t * RADIX];
        col[(size_t)t * RADIX] = sum;
        sum += v;
    }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = inc - sum;
    for (int q = 0; q < w; ++q) base += s_w[q];
    for (int t = 0; t < tact; ++t) col[(size_t)t * RADIX] += base;
}

__global__ void __launch_bounds__(SORT_BLOCK) k_radix_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                         uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, int N,
                                                         int T, int shift, const FrameState* __restrict__ fs,
                                                         const uint32_t* __restrict__ hist, KeyPack kp) {
    __shared__ unsigned short s_wh[SORT_WAVES][RADIX];   // per-wave bin counts, then the wave's offset inside the bin
    __shared__ uint32_t s_goff[RADIX];                   // where the tile's part of each bin starts in the frame
    __shared__ uint32_t s_bstart[RADIX];                 // where each bin starts inside the tile
    __shared__ uint32_t s_k[SORT_TILE], s_v[SORT_TILE];  // the tile, ordered by bin (64 KiB)
    __shared__ uint32_t s_ws[RADIX / WAVE];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_c;
    if (tile * SORT_TILE >= n) return;
    const size_t fbase = (size_t)f * N;
    for (int q = threadIdx.x; q < SORT_WAVES * RADIX; q += SORT_BLOCK) (&s_wh[0][0])[q] = 0;
    if (threadIdx.x < RADIX) s_goff[threadIdx.x] = hist[((size_t)f * T + tile) * RADIX + threadIdx.x];
    __syncthreads();
    const int base = tile * SORT_TILE + w * WAVE_SPAN + lane;
    const uint64_t lt = lanemask_lt();
    const KeyGrid g = key_grid(fs[f]);
    uint32_t key[ITEMS], rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        const bool valid = e < n;
        key[j] = valid ? kin[fbase + e] : 0xffffffffu;
        if (kp.enabled && valid) key[j] = voxel_key(key[j], kp, g);
        const uint32_t d = (key[j] >> shift) & (RADIX - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
        uint32_t bin_base = 0;
        if (valid && lane == leader) {
            bin_base = s_wh[w][d];
            s_wh[w][d] = (unsigned short)(bin_base + (uint32_t)__popcll(peers));
        }
        bin_base = __shfl(bin_base, leader, 64);
        rank[j] = bin_base + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    if (threadIdx.x < RADIX) {   // per-digit exclusive prefix over the waves, then over the digits
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < SORT_WAVES; ++q) {
            const uint32_t c = s_wh[q][d];
            s_wh[q][d] = (unsigned short)run;
            run += c;
        }
        uint32_t inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        if (lane == 63) s_ws[w] = inc;
        s_bstart[d] = inc - run;   // completed below with the totals of the lower waves
    }
    __syncthreads();
    if (threadIdx.x < RADIX) {
        uint32_t add = 0;
        for (int q = 0; q < w; ++q) add += s_ws[q];
        s_bstart[threadIdx.x] += add;
    }
    __syncthreads();
    // the tile in bin order, staged in LDS: the global stores below are then contiguous runs (one run per bin) instead of
    // 64 scattered 4-byte stores per instruction
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        if (e < n) {
            const uint32_t d = (key[j] >> shift) & (RADIX - 1);
            const uint32_t lp = s_bstart[d] + s_wh[w][d] + rank[j];
            s_k[lp] = key[j];
            s_v[lp] = vin ? vin[fbase + e] : (uint32_t)e;
        }
    }
    __syncthreads();
    const int cnt = min(SORT_TILE, n - tile * SORT_TILE);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int lp = threadIdx.x + j * SORT_BLOCK;
        if (lp < cnt) {
            const uint32_t k = s_k[lp];
            const uint32_t d = (k >> shift) & (RADIX - 1);
            const uint32_t dst = s_goff[d] + ((uint32_t)lp - s_bstart[d]);
            kout[fbase + dst] = k;
            vout[fbase + dst] = s_v[lp];
        }
    }
}

// T = sort tiles per frame the histogram is laid out for, Tact = sort tiles that hold data (max over the frames)
// kp.enabled: kin holds the absolute coordinate fields of k_crop_fused (first pass only); kout gets voxel indices
void launch_radix_pass(hipStream_t s, const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int N,
                       int F, int T, int Tact, int shift, const FrameState* fs, uint32_t* hist, KeyPack kp) {
    hipLaunchKernelGGL(k_radix_hist, dim3(Tact, F), dim3(SORT_BLOCK), 0, s, kin, N, T, shift, fs, hist, kp);
    hipLaunchKernelGGL(k_radix_scan, dim3(F), dim3(BLOCK), 0, s, T, fs, hist);
    hipLaunchKernelGGL(k_radix_scatter, dim3(Tact, F), dim3(SORT_BLOCK), 0, s, kin, vin, kout, vout, N, T, shift, fs, hist, kp);
}

}  // namespace cd
