// k_sort.hip - segmented stable LSD radix sort (8-bit digits), one segment per frame.
//
// VoxelGrid sorts the cropped points by voxel index (pcl::VoxelGrid::applyFilter uses
// std::sort on (idx, point) pairs; rule C2 makes it stable).  One pass = three kernels over
// ordered tiles of 2048 pairs:
//   hist    : per-tile digit histogram (LDS atomics)            -> hist[f][digit][tile]
//   scan    : per frame, exclusive scan in (digit, tile) order  -> global base of every bin
//   scatter : stable rank of each pair inside its tile with wave ballots ("match-any" over
//             the 8 digit bits, 64-wide), then one scattered store per pair.
// Element order inside a tile is (wave, row, lane), so per-wave running bin counts kept in
// LDS plus a cross-wave prefix give the stable position.
#include "kernels.hpp"

namespace cd {

__global__ void __launch_bounds__(BLOCK) k_radix_hist(const uint32_t* __restrict__ kin, int N, int T, int shift,
                                                      const FrameState* __restrict__ fs, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[RADIX];
    const int f = blockIdx.y, tile = blockIdx.x;
    const int n = fs[f].n_c;
    if (tile * TILE >= n) return;
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* k = kin + (size_t)f * N;
    const int base = tile * TILE + (threadIdx.x >> 6) * WAVE_SPAN + (threadIdx.x & 63);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        if (e < n) atomicAdd(&s_h[(k[e] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    hist[((size_t)f * RADIX + threadIdx.x) * T + tile] = s_h[threadIdx.x];
}

// one block per frame; thread d owns digit d's row of Tact tiles
__global__ void __launch_bounds__(BLOCK) k_radix_scan(int T, const FrameState* __restrict__ fs, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_w[WAVES_PER_BLOCK];
    const int f = blockIdx.x, d = threadIdx.x, w = d >> 6, lane = d & 63;
    const int n = fs[f].n_c;
    if (n <= 0) return;
    const int tact = (n + TILE - 1) / TILE;
    uint32_t* row = hist + ((size_t)f * RADIX + d) * T;
    uint32_t sum = 0;
    for (int t = 0; t < tact; ++t) {
        const uint32_t v = row[t];
        row[t] = sum;
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
    for (int t = 0; t < tact; ++t) row[t] += base;
}

__global__ void __launch_bounds__(BLOCK) k_radix_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                         uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, int N,
                                                         int T, int shift, const FrameState* __restrict__ fs,
                                                         const uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_wh[WAVES_PER_BLOCK][RADIX];
    __shared__ uint32_t s_goff[RADIX];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_c;
    if (tile * TILE >= n) return;
    const size_t fbase = (size_t)f * N;
#pragma unroll
    for (int q = 0; q < WAVES_PER_BLOCK; ++q) s_wh[q][threadIdx.x] = 0;
    s_goff[threadIdx.x] = hist[((size_t)f * RADIX + threadIdx.x) * T + tile];
    __syncthreads();
    const int base = tile * TILE + w * WAVE_SPAN + lane;
    const uint64_t lt = lanemask_lt();
    uint32_t key[ITEMS], rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        const bool valid = e < n;
        key[j] = valid ? kin[fbase + e] : 0xffffffffu;
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
            s_wh[w][d] = bin_base + (uint32_t)__popcll(peers);
        }
        bin_base = __shfl(bin_base, leader, 64);
        rank[j] = bin_base + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    {   // per-digit exclusive prefix over the 4 waves
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < WAVES_PER_BLOCK; ++q) {
            const uint32_t c = s_wh[q][d];
            s_wh[q][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        if (e < n) {
            const uint32_t d = (key[j] >> shift) & (RADIX - 1);
            const uint32_t dst = s_goff[d] + s_wh[w][d] + rank[j];
            kout[fbase + dst] = key[j];
            vout[fbase + dst] = vin ? vin[fbase + e] : (uint32_t)e;
        }
    }
}

void launch_radix_pass(hipStream_t s, const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int N,
                       int F, int T, int Tact, int shift, const FrameState* fs, uint32_t* hist) {
    hipLaunchKernelGGL(k_radix_hist, dim3(Tact, F), dim3(BLOCK), 0, s, kin, N, T, shift, fs, hist);
    hipLaunchKernelGGL(k_radix_scan, dim3(F), dim3(BLOCK), 0, s, T, fs, hist);
    hipLaunchKernelGGL(k_radix_scatter, dim3(Tact, F), dim3(BLOCK), 0, s, kin, vin, kout, vout, N, T, shift, fs, hist);
}

}  // namespace cd
