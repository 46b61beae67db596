// k_voxel.hip - S0 crop (two PassThrough filters) + S1 VoxelGrid.
//
// Replaces pcl::PassThrough<PCLPointCloud2>::filter x2 and pcl::VoxelGrid<PCLPointCloud2>::filter
// (reference: cuboid_detection/src/ground_plane_segmentation.cpp:53-73,
//  object_detection/src/object_pose_detection.cpp:273-298).
//
// Data layout: every per-point array is frame-major with pitch N (= points per frame).
// An "ordered tile" is 2048 consecutive elements; wave w of the 256-thread block owns the
// contiguous 512-element span [w*512,(w+1)*512) as 8 rows of 64 lanes, so every row is one
// fully coalesced 1 KiB (16 B/lane) access and element order == (wave, row, lane) order.
// Order-preserving stream compaction = per-row wave ballot + popcount prefix, one LDS
// exchange of the 4 wave totals, plus a per-frame exclusive scan of the tile totals.
#include "kernels.hpp"

namespace cd {

// ALL rows of a wave, loaded before any of them is used.  Written as "load a row, test it, load the next" (rounds 1-4: the load
// sat inside `if (e < N)`) every row's load was followed by s_waitcnt vmcnt(0) - a wave paid one memory round trip per row,
// eight in a row, and the crop moved its bytes at 4.2 TB/s only by keeping many waves resident.  Here the index is clamped
// instead of tested (the caller still ignores rows past N) and the record format is decided once, outside the loop.
template <int R>
__device__ __forceinline__ void load_rows(const char* __restrict__ in, size_t stride, size_t fbase, int first, int N, int rgb_off,
                                          float (&x)[R], float (&y)[R], float (&z)[R], uint32_t (&rgb)[R]) {
    if (stride == 16 && rgb_off == 12) {
        float4 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = *reinterpret_cast<const float4*>(in + (fbase + (size_t)min(first + j * WAVE, N - 1)) * 16);
#pragma unroll
        for (int j = 0; j < R; ++j) { x[j] = v[j].x; y[j] = v[j].y; z[j] = v[j].z; rgb[j] = __float_as_uint(v[j].w); }
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const char* p = in + (fbase + (size_t)min(first + j * WAVE, N - 1)) * stride;
            x[j] = *reinterpret_cast<const float*>(p);
            y[j] = *reinterpret_cast<const float*>(p + 4);
            z[j] = *reinterpret_cast<const float*>(p + 8);
            rgb[j] = rgb_off >= 0 ? *reinterpret_cast<const uint32_t*>(p + rgb_off) : 0u;
        }
    }
}

// PCL keeps a point iff xyz finite and !(v > max) && !(v < min) with v promoted to double;
// the double limits were folded on the host into the exactly equivalent float limits.
__device__ __forceinline__ bool crop_keep(float x, float y, float z, const CropLimits& L) {
    const bool fin = (fabsf(x) <= 3.402823466e38f) && (fabsf(y) <= 3.402823466e38f) && (fabsf(z) <= 3.402823466e38f);
    return fin && z >= L.zlo && z <= L.zhi && x >= L.xlo && x <= L.xhi;
}

// ---- pass A: survivors per tile + min/max of the survivors ---------------------------
__global__ void __launch_bounds__(BLOCK) k_crop_count(const char* __restrict__ in, size_t stride, int N, int rgb_off,
                                                      CropLimits lim, int T, FrameState* __restrict__ fs,
                                                      int* __restrict__ tile_cnt) {
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ float s_mn[WAVES_PER_BLOCK][3], s_mx[WAVES_PER_BLOCK][3];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t fbase = (size_t)f * N;
    const int base = tile * TILE + w * WAVE_SPAN;
    int cnt = 0;
    float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    float mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    float qx[ITEMS], qy[ITEMS], qz[ITEMS];
    uint32_t qc[ITEMS];
    load_rows<ITEMS>(in, stride, fbase, base + lane, N, rgb_off, qx, qy, qz, qc);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE + lane;
        bool keep = false;
        if (e < N) {
            const float x = qx[j], y = qy[j], z = qz[j];
            keep = crop_keep(x, y, z, lim);
            if (keep) {
                mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
                mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
            }
        }
        cnt += __popcll(__ballot(keep));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
    }
    if (lane == 0) {
        s_cnt[w] = cnt;
        for (int a = 0; a < 3; ++a) { s_mn[w][a] = mn[a]; s_mx[w][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < WAVES_PER_BLOCK; ++k) tot += s_cnt[k];
        tile_cnt[(size_t)f * T + tile] = tot;
        if (tot > 0) {
            for (int a = 0; a < 3; ++a) {
                float lo = s_mn[0][a], hi = s_mx[0][a];
                for (int k = 1; k < WAVES_PER_BLOCK; ++k) { lo = fminf(lo, s_mn[k][a]); hi = fmaxf(hi, s_mx[k][a]); }
                atomicMin(&fs[f].mn[a], f2ord(lo));
                atomicMax(&fs[f].mx[a], f2ord(hi));
            }
        }
    }
}

// ---- per-"row" exclusive scan of tile totals (rows = frames, or frames x bins) --------
// counts[row*T + t] -> exclusive prefix in place; total -> totals[row*total_pitch].
__global__ void __launch_bounds__(BLOCK) k_scan_tiles(ScanJob j0, ScanJob j1, int T, int total_pitch) {
    // blockIdx.y picks the job (two independent scans of a stage in one launch); a job's total per row goes to `totals` and -
    // mirror != nullptr - to the same field of the host's pinned FrameState copy
    CD_FRONT_PRIO();
    __shared__ int s_w[WAVES_PER_BLOCK];
    __shared__ int s_base;
    const ScanJob job = blockIdx.y ? j1 : j0;
    const int row = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int* c = job.counts + (size_t)row * T;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += BLOCK) {
        const int t = t0 + threadIdx.x;
        const int v = t < T ? c[t] : 0;
        int inc = v;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int wb = 0;
        for (int k = 0; k < w; ++k) wb += s_w[k];
        const int base = s_base;
        if (t < T) c[t] = base + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == BLOCK - 1) s_base = base + wb + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (job.totals) job.totals[(size_t)row * total_pitch] = s_base;
        if (job.mirror) job.mirror[(size_t)row * total_pitch] = s_base;
    }
}

// ---- voxel grid geometry from the min/max (VoxelGrid::applyFilter prologue) ------------
// after k_crop_runs: digit d of the packed cell keys varies in a frame when its histogram has two non-empty bins.  One wave per
// frame, four coalesced 256-byte reads per digit.
// One wave per frame (one launch where rounds 4-5 had k_digit_vary + k_voxel_setup + the copy of the FrameState array to its
// host mirror): the digits that vary, the grid geometry (lane 0), and - mirror != nullptr - the frame's whole FrameState
// written to the device-visible pinned mirror the host reads after its synchronisation.
__device__ __forceinline__ void voxel_setup_frame(FrameState& s, float leaf) {
    s.key_bits = 0;
    s.n_cropped = s.n_c;
    s.origin[0] = s.origin[1] = s.origin[2] = 0.f;
    for (int a = 0; a < 3; ++a) { s.min_b[a] = 0; s.div_b[a] = 0; }
    if (s.n_c <= 0) return;
    const float inv = __fdiv_rn(1.0f, leaf);
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(s.mn[a]); mx[a] = ord2f(s.mx[a]); s.origin[a] = mn[a]; }
    // "Leaf size is too small for the input dataset. Integer indices would overflow."
    const long long dx = (long long)__fmul_rn(__fsub_rn(mx[0], mn[0]), inv) + 1;
    const long long dy = (long long)__fmul_rn(__fsub_rn(mx[1], mn[1]), inv) + 1;
    const long long dz = (long long)__fmul_rn(__fsub_rn(mx[2], mn[2]), inv) + 1;
    long long cells = 1;
    bool bad = (dx * dy * dz) > 2147483647ll;
    for (int a = 0; a < 3; ++a) {
        const int lo = (int)floorf(__fmul_rn(mn[a], inv));
        const int hi = (int)floorf(__fmul_rn(mx[a], inv));
        s.min_b[a] = lo;
        s.div_b[a] = hi - lo + 1;
        cells *= (long long)(hi - lo + 1);
        if (cells > 2147483647ll) bad = true;
    }
    if (bad) { s.status = CD_ERR_LEAF_TOO_SMALL; s.n_c = 0; s.n_runs = 0; return; }   // (n_runs: what k_crop_runs' sort and centroids go by)
    int bits = 1;
    while (bits < 31 && (1ll << bits) < cells) ++bits;
    s.key_bits = bits;
}
__global__ void __launch_bounds__(WAVE) k_voxel_setup(FrameState* __restrict__ fs, float leaf, const uint32_t* __restrict__ ghist,
                                                      FrameState* __restrict__ mirror) {
    CD_FRONT_PRIO();
    const int f = blockIdx.x, lane = threadIdx.x;
    int vary = 0;
    if (ghist) {   // after k_crop_runs: digit d of the packed cell keys varies in a frame when its histogram has two non-empty bins
        for (int d = 0; d < 4; ++d) {
            int bins = 0;
            for (int q = 0; q < RADIX / WAVE; ++q) bins += __popcll(__ballot(ghist[((size_t)f * 4 + d) * RADIX + q * WAVE + lane] != 0u));
            if (bins > 1) vary |= 1 << d;
        }
    }
    if (lane != 0) return;
    FrameState s = fs[f];
    if (ghist) s.digit_vary = vary;
    voxel_setup_frame(s, leaf);
    fs[f] = s;
    if (mirror) mirror[f] = s;
}

// ---- pass B: ordered compaction + voxel key -------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_crop_compact(const char* __restrict__ in, size_t stride, int N, int pitch,
                                                        int rgb_off, CropLimits lim, int T, float leaf,
                                                        const FrameState* __restrict__ fs,
                                                        const int* __restrict__ tile_off, float4* __restrict__ cpt,
                                                        uint32_t* __restrict__ keys) {
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (fs[f].n_c <= 0) return;
    const size_t fbase = (size_t)f * N;        // input records
    const size_t obase = (size_t)f * pitch;    // internal arrays
    const int base = tile * TILE + w * WAVE_SPAN;
    float px[ITEMS], py[ITEMS], pz[ITEMS];
    uint32_t pc[ITEMS];
    uint64_t bal[ITEMS];
    int wtot = 0;
    load_rows<ITEMS>(in, stride, fbase, base + lane, N, rgb_off, px, py, pz, pc);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE + lane;
        const bool keep = e < N && crop_keep(px[j], py[j], pz[j], lim);
        bal[j] = __ballot(keep);
        wtot += __popcll(bal[j]);
    }
    if (lane == 0) s_cnt[w] = wtot;
    __syncthreads();
    int pos = tile_off[(size_t)f * T + tile];
    for (int k = 0; k < w; ++k) pos += s_cnt[k];
    const float inv = __fdiv_rn(1.0f, leaf);
    const int mb0 = fs[f].min_b[0], mb1 = fs[f].min_b[1], mb2 = fs[f].min_b[2];
    const int d0 = fs[f].div_b[0], d01 = fs[f].div_b[0] * fs[f].div_b[1];
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((bal[j] >> lane) & 1ull) {
            const int r = pos + __popcll(bal[j] & lt);
            const int i0 = (int)__fsub_rn(floorf(__fmul_rn(px[j], inv)), (float)mb0);
            const int i1 = (int)__fsub_rn(floorf(__fmul_rn(py[j], inv)), (float)mb1);
            const int i2 = (int)__fsub_rn(floorf(__fmul_rn(pz[j], inv)), (float)mb2);
            cpt[obase + r] = make_float4(px[j], py[j], pz[j], __uint_as_float(pc[j]));
            keys[obase + r] = (uint32_t)(i0 + i1 * d0 + i2 * d01);
        }
        pos += __popcll(bal[j]);
    }
}

// ---- single pass: crop + ordered compaction + min/max + absolute voxel coordinates -------
// One read of the input instead of two.  The exclusive prefix of the tile totals inside a frame comes from chained_scan().
// Workgroup b works on frame b % F and takes the frame's next tile by ticket (take_ticket, common.hpp): the tiles in flight
// at any time are a few per frame, so a tile's predecessors are usually long done and its look-back ends at the first
// word; a scan that gives up (it cannot: the guard of chained_scan) sends the batch to the two-pass path.
__global__ void __launch_bounds__(BLOCK) k_crop_fused(const char* __restrict__ in, size_t stride, int N, int pitch,
                                                      int rgb_off, CropLimits lim, int T, int Tin, float leaf, KeyPack kp,
                                                      FrameState* __restrict__ fs, int* __restrict__ state,
                                                      float4* __restrict__ cpt,
                                                      uint32_t* __restrict__ keys, int* __restrict__ ticket) {
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ float s_mn[WAVES_PER_BLOCK][3], s_mx[WAVES_PER_BLOCK][3];
    __shared__ int s_excl, s_ticket;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int F = gridDim.x / Tin;
    const int f = blockIdx.x % F, tile = take_ticket(ticket + f * TICKET_PITCH, Tin, &s_ticket);
    const size_t fbase = (size_t)f * N;        // input records
    const size_t obase = (size_t)f * pitch;    // internal arrays
    const int base = tile * TILE + w * WAVE_SPAN;
    float px[ITEMS], py[ITEMS], pz[ITEMS];
    uint32_t pc[ITEMS];
    uint64_t bal[ITEMS];
    int wtot = 0;
    float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    float mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    load_rows<ITEMS>(in, stride, fbase, base + lane, N, rgb_off, px, py, pz, pc);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE + lane;
        bool keep = false;
        if (e < N) {
            keep = crop_keep(px[j], py[j], pz[j], lim);
            if (keep) {
                mn[0] = fminf(mn[0], px[j]); mn[1] = fminf(mn[1], py[j]); mn[2] = fminf(mn[2], pz[j]);
                mx[0] = fmaxf(mx[0], px[j]); mx[1] = fmaxf(mx[1], py[j]); mx[2] = fmaxf(mx[2], pz[j]);
            }
        }
        bal[j] = __ballot(keep);
        wtot += __popcll(bal[j]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
    }
    if (lane == 0) {
        s_cnt[w] = wtot;
        for (int a = 0; a < 3; ++a) { s_mn[w][a] = mn[a]; s_mx[w][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < WAVES_PER_BLOCK; ++k) tot += s_cnt[k];
        const int excl = chained_scan(state + (size_t)f * T, 1, tile, tot, &fs[f].crop_overflow);
        s_excl = excl;
        if (tile == Tin - 1) fs[f].n_c = excl + tot;
        if (tot > 0) {
            for (int a = 0; a < 3; ++a) {
                float lo = s_mn[0][a], hi = s_mx[0][a];
                for (int k = 1; k < WAVES_PER_BLOCK; ++k) { lo = fminf(lo, s_mn[k][a]); hi = fmaxf(hi, s_mx[k][a]); }
                atomicMin(&fs[f].mn[a], f2ord(lo));
                atomicMax(&fs[f].mx[a], f2ord(hi));
            }
        }
    }
    __syncthreads();
    int pos = s_excl;
    for (int k = 0; k < w; ++k) pos += s_cnt[k];
    const float inv = __fdiv_rn(1.0f, leaf);
    const float jlo = (float)kp.jlo, jhi = (float)(kp.jlo + ((1 << kp.bj) - 1));
    const uint64_t lt = lanemask_lt();
    bool over = false;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((bal[j] >> lane) & 1ull) {
            const int r = pos + __popcll(bal[j] & lt);
            const float fy = floorf(__fmul_rn(py[j], inv));
            const bool fits = fy >= jlo && fy <= jhi;
            over = over || !fits;
            const uint32_t ui = (uint32_t)((int)floorf(__fmul_rn(px[j], inv)) - kp.ilo);
            const uint32_t uj = fits ? (uint32_t)((int)fy - kp.jlo) : 0u;
            const uint32_t uk = (uint32_t)((int)floorf(__fmul_rn(pz[j], inv)) - kp.klo);
            if (r < pitch) {
                cpt[obase + r] = make_float4(px[j], py[j], pz[j], __uint_as_float(pc[j]));
                keys[obase + r] = ui | (uj << kp.bi) | (uk << (kp.bi + kp.bj));
            }
        }
        pos += __popcll(bal[j]);
    }
    if (__ballot(over) != 0ull && lane == 0) fs[f].crop_overflow = 1;
}


// ---- single pass, round 4: crop + ordered compaction + min/max + the RUNS and their digit histograms ------------------------
// k_crop_fused wrote a 4-byte cell key per kept point and k_voxel_runs read them all back to find the runs of equal voxel
// index, turn the keys into PCL's idx = i + j dx + k dx dy and build the sort's histograms.  None of that needs a second
// pass: two neighbouring kept points are in one voxel exactly when their packed cell keys are equal, and the packed key
// (x | y << bi | z << (bi + bj), every field a monotone function of its coordinate) orders the voxels exactly as idx does
// (k, then j, then i) - so the SORT can run on the packed keys themselves, whose digits are known right here.  This kernel
// therefore writes the kept points (16 bytes each), ONE record per run - (packed key, start | length << 20), a run being
// consecutive kept points of one row of 64 input points with equal keys - and adds the runs' digits to the frame's four
// histograms; which of the four digits vary at all, k_voxel_setup reads off those histograms afterwards (the y field is wide
// because y is unbounded, but a frame's y cells span a few hundred values: digit 2 is constant unless they straddle a 512-cell
// block - the host biases the field so that they do not).
// Dropped: 4 bytes written per kept point, k_voxel_runs (a read of those keys, 0.12 ms), the key conversion in the first
// scatter.  The chained scan carries both running counts (points, runs) in one 64-bit word.
// What bounds it (round 4, profiles/r04_crop_variants.txt): a tile lives ~14 us - ticket, loads, scan and stores are round
// trips in series - so the kernel moves bytes in proportion to the TILES IN FLIGHT per CU, i.e. to resident waves x rows per
// wave.  443 us with 86 registers (5 waves per SIMD) and one load per round trip; 400 with the keys parked in LDS between
// the phases (80 registers, 6 waves); 342 with all eight loads of a wave issued together (load_rows) = 5.5 TB/s.  Fewer rows
// per wave on more threads goes the other way (4 rows x 512 threads 427 us, 2 x 1024 966 us: half the tiles in flight each
// time), 16 rows x 128 threads gains nothing (349), and the ticket costs 3 us.
constexpr int RUN_SHIFT_C = 20;   // as k_sort.hip's RUN_SHIFT: start < 2^20, length <= 64
template <bool DIRECT>
__global__ void __launch_bounds__(BLOCK) k_crop_runs(const char* __restrict__ in, size_t stride, int N, int pitch,
                                                     int rgb_off, CropLimits lim, int T, int Tin, float leaf, KeyPack kp,
                                                     FrameState* __restrict__ fs, unsigned long long* __restrict__ state,
                                                     float4* __restrict__ cpt, uint32_t* __restrict__ rkeys, uint32_t* __restrict__ rvals,
                                                     uint32_t* __restrict__ ghist, int* __restrict__ ticket) {
    // DIRECT (16-byte input records x y z rgb): the kept points are NOT copied - a run's record carries its start in the INPUT
    // (runs are pieces of a row of the input, so their points are consecutive there too) and the centroid kernel reads the input
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK], s_rcnt[WAVES_PER_BLOCK];
    __shared__ float s_mn[WAVES_PER_BLOCK][3], s_mx[WAVES_PER_BLOCK][3];
    __shared__ uint32_t s_h[4][RADIX];
    __shared__ unsigned short s_rk[TILE];   // the tile's run heads in order, as element numbers: their keys are in s_key (for the histograms;
                                            // 16 bits each: 16.6 KB of LDS per workgroup, eight workgroups per CU instead of seven at 20.6 KB)
    __shared__ uint32_t s_key[TILE];  // every element's key between the two phases (eight registers less per lane: one more wave per SIMD)
    __shared__ int s_excl, s_rexcl, s_ticket;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int F = gridDim.x / Tin;
    const int f = blockIdx.x % F, tile = take_ticket(ticket + f * TICKET_PITCH, Tin, &s_ticket);
    for (int q = threadIdx.x; q < 4 * RADIX; q += BLOCK) (&s_h[0][0])[q] = 0;
    const size_t fbase = (size_t)f * N;        // input records
    const size_t obase = (size_t)f * pitch;    // internal arrays
    const int base = tile * TILE + w * WAVE_SPAN;
    const float inv = __fdiv_rn(1.0f, leaf);
    const float jlo = (float)kp.jlo, jhi = (float)(kp.jlo + ((1 << kp.bj) - 1));
    float px[ITEMS], py[ITEMS], pz[ITEMS];
    uint32_t pc[ITEMS];
    uint64_t bal[ITEMS], heads[ITEMS];
    int wtot = 0, rtot = 0;
    float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    float mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    bool over = false;
    load_rows<ITEMS>(in, stride, fbase, base + lane, N, rgb_off, px, py, pz, pc);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE + lane;
        bool keep = false;
        uint32_t key = 0xffffffffu;
        if (e < N) {
            keep = crop_keep(px[j], py[j], pz[j], lim);
            if (keep) {
                mn[0] = fminf(mn[0], px[j]); mn[1] = fminf(mn[1], py[j]); mn[2] = fminf(mn[2], pz[j]);
                mx[0] = fmaxf(mx[0], px[j]); mx[1] = fmaxf(mx[1], py[j]); mx[2] = fmaxf(mx[2], pz[j]);
                const float fy = floorf(__fmul_rn(py[j], inv));
                const bool fits = fy >= jlo && fy <= jhi;
                over = over || !fits;
                const uint32_t ui = (uint32_t)((int)floorf(__fmul_rn(px[j], inv)) - kp.ilo);
                const uint32_t uj = fits ? (uint32_t)((int)fy - kp.jlo) : 0u;
                const uint32_t uk = (uint32_t)((int)floorf(__fmul_rn(pz[j], inv)) - kp.klo);
                key = ui | (uj << kp.bi) | (uk << (kp.bi + kp.bj));
            }
        }
        bal[j] = __ballot(keep);
        // a run starts at a kept point whose left neighbour in the row is not kept or lies in another cell
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
        const bool prev_kept = lane > 0 && ((bal[j] >> (lane - 1)) & 1ull);
        heads[j] = __ballot(keep && !(prev_kept && prev == key));
        s_key[w * WAVE_SPAN + j * WAVE + lane] = key;   // (read back by this same lane only)
        wtot += __popcll(bal[j]);
        rtot += __popcll(heads[j]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
    }
    if (lane == 0) {
        s_cnt[w] = wtot; s_rcnt[w] = rtot;
        for (int a = 0; a < 3; ++a) { s_mn[w][a] = mn[a]; s_mx[w][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0, rt = 0;
        for (int k = 0; k < WAVES_PER_BLOCK; ++k) { tot += s_cnt[k]; rt += s_rcnt[k]; }
        int excl = 0, rexcl = 0;
        chained_scan2(state + (size_t)f * T, tile, tot, rt, &excl, &rexcl, &fs[f].crop_overflow);
        s_excl = excl; s_rexcl = rexcl;
        if (tile == Tin - 1) { fs[f].n_c = excl + tot; fs[f].n_runs = rexcl + rt; }
        if (tot > 0) {
            for (int a = 0; a < 3; ++a) {
                float lo = s_mn[0][a], hi = s_mx[0][a];
                for (int k = 1; k < WAVES_PER_BLOCK; ++k) { lo = fminf(lo, s_mn[k][a]); hi = fmaxf(hi, s_mx[k][a]); }
                atomicMin(&fs[f].mn[a], f2ord(lo));
                atomicMax(&fs[f].mx[a], f2ord(hi));
            }
        }
    }
    __syncthreads();
    const int rexcl0 = s_rexcl;
    int pos = s_excl, rpos = rexcl0;
    for (int k = 0; k < w; ++k) { pos += s_cnt[k]; rpos += s_rcnt[k]; }
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool kept = (bal[j] >> lane) & 1ull;
        const int r = pos + __popcll(bal[j] & lt);
        if (!DIRECT && kept && r < pitch) cpt[obase + r] = make_float4(px[j], py[j], pz[j], __uint_as_float(pc[j]));
        const bool is_head = (heads[j] >> lane) & 1ull;
        if (is_head) {
            // the run ends at the next head or at the first point of the row that is not kept, whichever comes first
            const uint64_t stop = (~bal[j] | heads[j]) & (lane == 63 ? 0ull : ~((2ull << lane) - 1ull));
            const int next = stop ? __ffsll((long long)stop) - 1 : 64;
            const int ro = rpos + __popcll(heads[j] & lt);
            const uint32_t key = s_key[w * WAVE_SPAN + j * WAVE + lane];
            s_rk[ro - rexcl0] = (unsigned short)(w * WAVE_SPAN + j * WAVE + lane);
            if (ro < pitch && CD_IN_RANGE(ro >= 0, 5u)) {
                rkeys[obase + ro] = key;
                rvals[obase + ro] = (uint32_t)(DIRECT ? base + j * WAVE + lane : r) | ((uint32_t)(next - lane) << RUN_SHIFT_C);
            }
        }
        pos += __popcll(bal[j]);
        rpos += __popcll(heads[j]);
    }
    if (__ballot(over) != 0ull && lane == 0) fs[f].crop_overflow = 1;
    __syncthreads();
    // digit histograms of the tile's run keys, 64 runs per wave trip over the tile's compacted run keys (a row of 64 pixels
    // holds ~10 runs: per row it is 32 trips per wave instead of ~6; the whole histogram costs 18 of the kernel's 342 us).
    // Lowest digit: one add per run (x changes from run to run).  Upper digits: neighbouring runs mostly share them, and adds
    // to one LDS word would serialise - the first lane of each stretch of equal digits adds its length.
    {
        int nr = 0;
        for (int k = 0; k < WAVES_PER_BLOCK; ++k) nr += s_rcnt[k];
        for (int q0 = w * WAVE; q0 < nr; q0 += BLOCK) {
            const int q = q0 + lane;
            const bool on = q < nr;
            const uint32_t kq = on ? s_key[s_rk[q]] : 0u;
            if (on) atomicAdd(&s_h[0][kq & (RADIX - 1)], 1u);
#pragma unroll
            for (int d = 1; d < 4; ++d) {
                const uint32_t dg = on ? ((kq >> (d * RADIX_BITS)) & (RADIX - 1)) : 0xffffffffu;   // past the end: own stretch
                const uint32_t prev = (uint32_t)__shfl_up((int)dg, 1, 64);
                const uint64_t hd = __ballot(lane == 0 || dg != prev);
                const uint64_t above = lane == 63 ? 0ull : hd & ~((2ull << lane) - 1ull);
                const int next = above ? __ffsll((long long)above) - 1 : 64;
                if (((hd >> lane) & 1ull) && on) atomicAdd(&s_h[d][dg], (uint32_t)(next - lane));
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 4 * RADIX; q += BLOCK) {
        const uint32_t c = (&s_h[0][0])[q];
        if (c) atomicAdd(&ghist[(size_t)f * 4 * RADIX + q], c);
    }
}

// ---- voxel run heads in the sorted key array -------------------------------------------
// heads[j] = ballot of "element first + j * 64 + lane starts a voxel" (its key differs from its left neighbour's).  All keys
// of the wave's rows are loaded first (clamped indices, no branch around a load: see load_rows); a lane's left neighbour is the
// lane below it, lane 0 takes the element before the row from memory.
template <int R>
__device__ __forceinline__ void head_ballots(const uint32_t* __restrict__ k, int first, int n, uint64_t (&heads)[R]) {
    const int lane = threadIdx.x & 63;
    uint32_t cur[R], left[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int e = first + j * WAVE + lane;
        cur[j] = k[max(min(e, n - 1), 0)];
        left[j] = k[max(min(e - 1, n - 1), 0)];   // (the same cache lines as cur: no second trip to memory)
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int e = first + j * WAVE + lane;
        heads[j] = __ballot(e < n && (e == 0 || cur[j] != left[j]));
    }
}

// One QUAD per voxel.  The 4 lanes fetch 4 consecutive members of the voxel's run at once (indices,
// then points: members are mostly neighbouring pixels, so the 4 reads usually share a cache line),
// and every lane of the quad replays the same strictly sequential float32 sum - ascending input
// order inside the voxel, rule C2, exactly PCL's loop - on DPP quad broadcasts; one division by the
// float count.  rgb is averaged as three float channels and re-packed.
template <int I>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true));
}
template <int I>
__device__ __forceinline__ int quad_bcast_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true);
}

__global__ void __launch_bounds__(BLOCK) k_voxel_centroid(const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ vals,
                                                          const float4* __restrict__ cpt, int N, int T, int Tact, int rgb_on,
                                                          FrameState* __restrict__ fs, int* __restrict__ state,
                                                          float4* __restrict__ vox, int* __restrict__ ticket) {
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ int s_head[TILE];     // sorted positions of the voxel heads of this tile, in order
    __shared__ int s_out0, s_ticket;
    // workgroup b works on frame b % F and takes its tile by ticket (see k_crop_fused): the position of a tile's first voxel in
    // the frame's output comes from a chained scan of the tiles' head counts, so the heads are found once (no count pass + scan)
    const int F = gridDim.x / Tact;
    const int f = blockIdx.x % F, tile = take_ticket(ticket + f * TICKET_PITCH, Tact, &s_ticket), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_c;
    if (tile * TILE >= n) return;
    const size_t fbase = (size_t)f * N;
    const uint32_t* k = keys + fbase;
    const uint32_t* v = vals + fbase;
    const int base = tile * TILE + w * WAVE_SPAN;
    uint64_t bal[ITEMS];
    int wtot = 0;
    head_ballots<ITEMS>(k, base, n, bal);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) wtot += __popcll(bal[j]);
    if (lane == 0) s_cnt[w] = wtot;
    __syncthreads();
    int pos = 0, nheads = 0;
    for (int q = 0; q < WAVES_PER_BLOCK; ++q) { if (q < w) pos += s_cnt[q]; nheads += s_cnt[q]; }
    if (threadIdx.x == 0) {
        const int excl = chained_scan(state + (size_t)f * T, 1, tile, nheads, &fs[f].scan_stalled);
        s_out0 = excl;
        if ((tile + 1) * TILE >= n) fs[f].n_v = excl + nheads;
    }
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((bal[j] >> lane) & 1ull) s_head[pos + __popcll(bal[j] & lt)] = base + j * WAVE + lane;
        pos += __popcll(bal[j]);
    }
    __syncthreads();
    const int out0 = s_out0;
    const int quad = threadIdx.x >> 2, ql = threadIdx.x & 3;
    for (int h = quad; h < nheads; h += BLOCK / 4) {
        const int e0 = s_head[h];
        const uint32_t key = k[e0];
        float sx = 0.f, sy = 0.f, sz = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        int cnt = 0;
        for (int e = e0;; e += 4) {
            const int me = e + ql;
            // key and index of the candidate member are fetched together (one round trip before the gather, not two)
            uint32_t kk = ~key, vv = 0;
            if (me < n) { kk = k[me]; vv = v[me]; }
            const bool mine = kk == key;                  // members of a run are contiguous in sorted order
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mine) p = cpt[fbase + vv];
            const int m0 = quad_bcast_i<0>(mine), m1 = quad_bcast_i<1>(mine), m2 = quad_bcast_i<2>(mine), m3 = quad_bcast_i<3>(mine);
#define CD_ACC(I, M)                                                                                   \
            if (M) {                                                                                   \
                sx = __fadd_rn(sx, quad_bcast<I>(p.x)); sy = __fadd_rn(sy, quad_bcast<I>(p.y));        \
                sz = __fadd_rn(sz, quad_bcast<I>(p.z));                                                \
                if (rgb_on) {                                                                          \
                    const uint32_t u = __float_as_uint(quad_bcast<I>(p.w));                            \
                    cr += (float)((u >> 16) & 0xff); cg += (float)((u >> 8) & 0xff); cb += (float)(u & 0xff); \
                }                                                                                      \
                ++cnt;                                                                                 \
            }
            CD_ACC(0, m0) CD_ACC(1, m1) CD_ACC(2, m2) CD_ACC(3, m3)
#undef CD_ACC
            if (!m3) break;    // the run ended inside this group of four (runs are contiguous)
        }
        if (ql == 0) {
            const float c = (float)cnt;
            uint32_t packed = 0;
            if (rgb_on) {
                const int R = (int)__fdiv_rn(cr, c), G = (int)__fdiv_rn(cg, c), B = (int)__fdiv_rn(cb, c);
                packed = ((uint32_t)R << 16) | ((uint32_t)G << 8) | (uint32_t)B;
            }
            vox[fbase + out0 + h] = make_float4(__fdiv_rn(sx, c), __fdiv_rn(sy, c), __fdiv_rn(sz, c), __uint_as_float(packed));
        }
    }
}

// The same over SORTED RUNS (k_voxel_runs, k_sort.hip): keys = voxel index per run, vals = start | length << 20 of the run in the
// cropped points, which are in input order.  A voxel's runs are consecutive in the sorted order and, the sort being stable, in
// ascending input order; inside a run the points are consecutive: the quad replays the same sequential sum, reading each run
// as one contiguous piece instead of one 16-byte gather per point.
__global__ void __launch_bounds__(BLOCK) k_voxel_centroid_runs(const uint32_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ vals,
                                                               const float4* __restrict__ cpt, int N, int T, int Tact, int rgb_on,
                                                               FrameState* __restrict__ fs, int* __restrict__ state,
                                                               float4* __restrict__ vox, int* __restrict__ ticket, int pts_pitch) {
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ int s_head[TILE];
    __shared__ int s_out0, s_ticket;
    const int F = gridDim.x / Tact;
    const int f = blockIdx.x % F, tile = take_ticket(ticket + f * TICKET_PITCH, Tact, &s_ticket), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_runs;
    if (tile * TILE >= n) return;
    const size_t fbase = (size_t)f * N;
    const uint32_t* k = keys + fbase;
    const uint32_t* v = vals + fbase;
    const float4* pts = cpt + (size_t)f * pts_pitch;   // (the cropped points, or - k_crop_runs' direct form - the input records themselves)
    const int base = tile * TILE + w * WAVE_SPAN;
    uint64_t bal[ITEMS];
    int wtot = 0;
    head_ballots<ITEMS>(k, base, n, bal);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) wtot += __popcll(bal[j]);
    if (lane == 0) s_cnt[w] = wtot;
    __syncthreads();
    int pos = 0, nheads = 0;
    for (int q = 0; q < WAVES_PER_BLOCK; ++q) { if (q < w) pos += s_cnt[q]; nheads += s_cnt[q]; }
    if (threadIdx.x == 0) {
        const int excl = chained_scan(state + (size_t)f * T, 1, tile, nheads, &fs[f].scan_stalled);
        s_out0 = excl;
        if ((tile + 1) * TILE >= n) fs[f].n_v = excl + nheads;
    }
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((bal[j] >> lane) & 1ull) s_head[pos + __popcll(bal[j] & lt)] = base + j * WAVE + lane;
        pos += __popcll(bal[j]);
    }
    __syncthreads();
    const int out0 = s_out0;
    const int quad = threadIdx.x >> 2, ql = threadIdx.x & 3;
    // Latency plan (round 4): a wave spent ~10 dependent memory round trips per group of 16 voxels - the voxel's key, its run
    // records, the first four points of four runs, then one more trip for EVERY run index that is longer than four points in
    // some quad (13 % of the runs are), and all of it again for the 20 % of voxels with more than four runs.  Now the first
    // four records of a voxel are fetched one voxel AHEAD and carry the key (record e0 is the voxel's first run), the next
    // four records are fetched while the current four are summed, and points 4..7 of every run go out together with points
    // 0..3: ~2.6 round trips per group (one per four runs), the rest is the summation itself.
    // Colour: exact integer channel sums (v_dot4_u32_u8: one instruction per channel) converted once - equal to PCL's float
    // sums as long as every partial sum is below 2^24, i.e. for voxels of at most 65 536 points; a larger voxel (a fifth of a
    // 640 x 480 frame in one 5 mm cell) is re-summed in float by the loop at the end.
    int h = quad, e0 = 0;
    uint32_t kk = 0, rec = 0;
    bool have = false;
    if (h < nheads) {
        e0 = s_head[h];
        have = e0 + ql < n;
        if (have) { kk = k[e0 + ql]; rec = v[e0 + ql]; }
    }
    while (h < nheads) {
        const int hn = h + BLOCK / 4;
        int ne0 = 0;
        uint32_t nkk = 0, nrec = 0;
        bool nhave = false;
        if (hn < nheads) {
            ne0 = s_head[hn];
            nhave = ne0 + ql < n;
            if (nhave) { nkk = k[ne0 + ql]; nrec = v[ne0 + ql]; }
        }
        const uint32_t key = (uint32_t)quad_bcast_i<0>((int)kk);
        float sx = 0.f, sy = 0.f, sz = 0.f;
        uint32_t ir = 0, ig = 0, ib = 0;
        int cnt = 0;
#define CD_ADD(P, J)                                                                                   \
        {                                                                                              \
            sx = __fadd_rn(sx, quad_bcast<J>(P.x)); sy = __fadd_rn(sy, quad_bcast<J>(P.y));            \
            sz = __fadd_rn(sz, quad_bcast<J>(P.z));                                                    \
            if (rgb_on) {                                                                              \
                const uint32_t u = __float_as_uint(quad_bcast<J>(P.w));                                \
                ir = __builtin_amdgcn_udot4(u, 0x00010000u, ir, false);                                \
                ig = __builtin_amdgcn_udot4(u, 0x00000100u, ig, false);                                \
                ib = __builtin_amdgcn_udot4(u, 0x00000001u, ib, false);                                \
            }                                                                                          \
        }
#define CD_RUN(I, P, Q)                                                                                \
        if (len##I > 0) {                                                                              \
            CD_ADD(P, 0)                                                                               \
            if (len##I > 1) CD_ADD(P, 1)                                                               \
            if (len##I > 2) CD_ADD(P, 2)                                                               \
            if (len##I > 3) CD_ADD(P, 3)                                                               \
            if (len##I > 4) {                                                                          \
                CD_ADD(Q, 0)                                                                           \
                if (len##I > 5) CD_ADD(Q, 1)                                                           \
                if (len##I > 6) CD_ADD(Q, 2)                                                           \
                if (len##I > 7) CD_ADD(Q, 3)                                                           \
                for (int i = 8; i < len##I; i += 4) {   /* longer runs: the rest, four points per trip */ \
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);                                        \
                    if (i + ql < len##I) t = pts[start##I + i + ql];                           \
                    CD_ADD(t, 0)                                                                       \
                    if (len##I - i > 1) CD_ADD(t, 1)                                                   \
                    if (len##I - i > 2) CD_ADD(t, 2)                                                   \
                    if (len##I - i > 3) CD_ADD(t, 3)                                                   \
                }                                                                                      \
            }                                                                                          \
        }
        for (int eb = e0;; eb += 4) {
            // lane ql of the quad holds record eb + ql (kk, rec, have)
            const int mylen = (have && kk == key) ? (int)(rec >> 20) : 0, mystart = (int)(rec & ((1u << 20) - 1u));
            const int len0 = quad_bcast_i<0>(mylen), len1 = quad_bcast_i<1>(mylen), len2 = quad_bcast_i<2>(mylen), len3 = quad_bcast_i<3>(mylen);
            const int start0 = quad_bcast_i<0>(mystart), start1 = quad_bcast_i<1>(mystart), start2 = quad_bcast_i<2>(mystart), start3 = quad_bcast_i<3>(mystart);
            // the next four records, in case the voxel goes on (runs of a voxel are contiguous in the sorted order, so its
            // members are a prefix of every group of four)
            uint32_t kk2 = 0, rec2 = 0;
            const bool have2 = len3 > 0 && eb + 4 + ql < n;
            if (have2) { kk2 = k[eb + 4 + ql]; rec2 = v[eb + 4 + ql]; }
            cnt += (len0 + len1) + (len2 + len3);   // (the voxel's point count is the sum of its runs' lengths: not counted point by point)
            float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0, q0 = p0, q1 = p0, q2 = p0, q3 = p0;
            if (ql < len0) p0 = pts[start0 + ql];
            if (ql < len1) p1 = pts[start1 + ql];
            if (ql < len2) p2 = pts[start2 + ql];
            if (ql < len3) p3 = pts[start3 + ql];
            if (4 + ql < len0) q0 = pts[start0 + 4 + ql];
            if (4 + ql < len1) q1 = pts[start1 + 4 + ql];
            if (4 + ql < len2) q2 = pts[start2 + 4 + ql];
            if (4 + ql < len3) q3 = pts[start3 + 4 + ql];
            // the additions replay the input order: run by run, point by point
            CD_RUN(0, p0, q0) CD_RUN(1, p1, q1) CD_RUN(2, p2, q2) CD_RUN(3, p3, q3)
            if (len3 == 0) break;    // the voxel's runs ended inside this group of four
            kk = kk2; rec = rec2; have = have2;
        }
#undef CD_RUN
#undef CD_ADD
        float cr = (float)ir, cg = (float)ig, cb = (float)ib;
        if (rgb_on && cnt > 65536) {   // partial sums beyond 2^24 round in PCL's float accumulation: replay it
            cr = cg = cb = 0.f;
            for (int e = e0; e < n && k[e] == key; ++e) {
                const uint32_t r = v[e];
                const int st = (int)(r & ((1u << 20) - 1u)), ln = (int)(r >> 20);
                for (int i = 0; i < ln; ++i) {
                    const uint32_t u = __float_as_uint(pts[st + i].w);
                    cr += (float)((u >> 16) & 0xff); cg += (float)((u >> 8) & 0xff); cb += (float)(u & 0xff);
                }
            }
        }
        if (ql == 0) {
            const float c = (float)cnt;
            uint32_t packed = 0;
            if (rgb_on) {
                const int R = (int)__fdiv_rn(cr, c), G = (int)__fdiv_rn(cg, c), B = (int)__fdiv_rn(cb, c);
                packed = ((uint32_t)R << 16) | ((uint32_t)G << 8) | (uint32_t)B;
            }
            vox[fbase + out0 + h] = make_float4(__fdiv_rn(sx, c), __fdiv_rn(sy, c), __fdiv_rn(sz, c), __uint_as_float(packed));
        }
        h = hn; e0 = ne0; kk = nkk; rec = nrec; have = nhave;
    }
}

// The same, ONE LANE PER VOXEL (end of round 5).  The quad form above spends four lanes on one sequential sum (every lane of a
// quad executes every addition of its voxel) and sixteen voxels share a wave's instruction stream: 465 vector instructions per
// sixteen voxels, a fifth of the pipeline's vector instructions - and with several batches in flight vector issue is what the
// batches compete for.  Here a lane walks its own voxel: a window of four run records (fetched together;
// the next window while this one is summed), and per trip eight points of the window's flat point sequence - run by run, point
// by point: the input order of rule C2 - with all eight loads in flight together; slots beyond the voxel read nothing and add
// +0, which leaves a sum that started at +0 unchanged bit for bit.  6.8 points in 2.9 runs per voxel on the bench frames: most
// voxels are one trip.  A wave's trips = the longest voxel's among its 64; every lane divides and stores for its own voxel.
__global__ void __launch_bounds__(BLOCK) k_voxel_centroid_lanes(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ vals,
                                                                const float4* __restrict__ cpt, int N, int T, int Tact, int rgb_on,
                                                                FrameState* __restrict__ fs, int* __restrict__ state,
                                                                float4* __restrict__ vox, int* __restrict__ ticket, int pts_pitch) {
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ int s_head[TILE];
    __shared__ int s_out0, s_ticket;
    const int F = gridDim.x / Tact;
    const int f = blockIdx.x % F, tile = take_ticket(ticket + f * TICKET_PITCH, Tact, &s_ticket), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_runs;
    if (tile * TILE >= n) return;
    const size_t fbase = (size_t)f * N;
    const uint32_t* k = keys + fbase;
    const uint32_t* v = vals + fbase;
    const float4* pts = cpt + (size_t)f * pts_pitch;   // (the cropped points, or - k_crop_runs' direct form - the input records themselves)
    const int base = tile * TILE + w * WAVE_SPAN;
    uint64_t bal[ITEMS];
    int wtot = 0;
    head_ballots<ITEMS>(k, base, n, bal);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) wtot += __popcll(bal[j]);
    if (lane == 0) s_cnt[w] = wtot;
    __syncthreads();
    int pos = 0, nheads = 0;
    for (int q = 0; q < WAVES_PER_BLOCK; ++q) { if (q < w) pos += s_cnt[q]; nheads += s_cnt[q]; }
    if (threadIdx.x == 0) {
        const int excl = chained_scan(state + (size_t)f * T, 1, tile, nheads, &fs[f].scan_stalled);
        s_out0 = excl;
        if ((tile + 1) * TILE >= n) fs[f].n_v = excl + nheads;
    }
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((bal[j] >> lane) & 1ull) s_head[pos + __popcll(bal[j] & lt)] = base + j * WAVE + lane;
        pos += __popcll(bal[j]);
    }
    __syncthreads();
    const int out0 = s_out0;
    constexpr uint32_t START = (1u << 20) - 1u;
    // a window of four run records (the voxel's runs are a prefix of it); a wave takes the tile's voxels 64 at a time, 256 apart
    // (fetching a lane's NEXT voxel's first window ahead was measured: nothing - nine registers for a round trip that is not
    // the bound)
    for (int h0 = w * WAVE; h0 < nheads; h0 += BLOCK) {   // (uniform per wave: ballots inside)
        const int h = h0 + lane;
        bool alive = h < nheads;
        int e = 0, i = 0, cnt = 0;
        uint32_t key = 0, wk0 = 0, wk1 = 0, wk2 = 0, wk3 = 0, wv0 = 0, wv1 = 0, wv2 = 0, wv3 = 0;
        if (alive) {
            e = s_head[h];
            key = k[e];
            wk0 = key; wv0 = v[e];
            wk1 = wk2 = wk3 = ~key;
            if (e + 1 < n) { wk1 = k[e + 1]; wv1 = v[e + 1]; }
            if (e + 2 < n) { wk2 = k[e + 2]; wv2 = v[e + 2]; }
            if (e + 3 < n) { wk3 = k[e + 3]; wv3 = v[e + 3]; }
        }
        uint32_t nk0 = 0, nk1 = 0, nk2 = 0, nk3 = 0, nv0 = 0, nv1 = 0, nv2 = 0, nv3 = 0;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        uint32_t ir = 0, ig = 0, ib = 0;
        while (ballot64(alive)) {
            const bool m0 = alive && wk0 == key, m1 = m0 && wk1 == key, m2 = m1 && wk2 == key, m3 = m2 && wk3 == key;
            const int l0 = m0 ? (int)(wv0 >> 20) : 0, l1 = m1 ? (int)(wv1 >> 20) : 0, l2 = m2 ? (int)(wv2 >> 20) : 0, l3 = m3 ? (int)(wv3 >> 20) : 0;
            const int c1 = l0, c2 = c1 + l1, c3 = c2 + l2, tot = c3 + l3;
            const int s0 = (int)(wv0 & START), s1 = (int)(wv1 & START) - c1, s2 = (int)(wv2 & START) - c2, s3 = (int)(wv3 & START) - c3;   // (start - first flat index of the run)
            if (m3 && i == 0) {   // first trip of a full window: the voxel may go on
                nk0 = nk1 = nk2 = nk3 = ~key;
                if (e + 4 < n) { nk0 = k[e + 4]; nv0 = v[e + 4]; }
                if (e + 5 < n) { nk1 = k[e + 5]; nv1 = v[e + 5]; }
                if (e + 6 < n) { nk2 = k[e + 6]; nv2 = v[e + 6]; }
                if (e + 7 < n) { nk3 = k[e + 7]; nv3 = v[e + 7]; }
            }
            // eight points of the window's flat sequence (run by run, point by point), all eight loads in flight together
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 P[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int idx = i + t;
                const int off = idx >= c3 ? s3 : (idx >= c2 ? s2 : (idx >= c1 ? s1 : s0));
                P[t] = z;
                if (idx < tot && CD_IN_RANGE(off + idx >= 0 && off + idx < pts_pitch, 4u)) P[t] = pts[off + idx];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                sx = __fadd_rn(sx, P[t].x); sy = __fadd_rn(sy, P[t].y); sz = __fadd_rn(sz, P[t].z);
                if (rgb_on) {
                    const uint32_t u = __float_as_uint(P[t].w);
                    ir = __builtin_amdgcn_udot4(u, 0x00010000u, ir, false);
                    ig = __builtin_amdgcn_udot4(u, 0x00000100u, ig, false);
                    ib = __builtin_amdgcn_udot4(u, 0x00000001u, ib, false);
                }
            }
            i += 8;
            if (alive && i >= tot) {   // the window is summed
                cnt += tot;
                alive = m3 && nk0 == key;
                e += 4; i = 0;
                wk0 = nk0; wk1 = nk1; wk2 = nk2; wk3 = nk3; wv0 = nv0; wv1 = nv1; wv2 = nv2; wv3 = nv3;
            }
        }
        if (h < nheads) {
            float cr = (float)ir, cg = (float)ig, cb = (float)ib;
            if (rgb_on && cnt > 65536) {   // partial sums beyond 2^24 round in PCL's float accumulation: replay it
                cr = cg = cb = 0.f;
                for (int ee = s_head[h]; ee < n && k[ee] == key; ++ee) {
                    const uint32_t r = v[ee];
                    const int s0 = (int)(r & START), ln = (int)(r >> 20);
                    for (int i = 0; i < ln; ++i) {
                        const uint32_t u = __float_as_uint(pts[s0 + i].w);
                        cr += (float)((u >> 16) & 0xff); cg += (float)((u >> 8) & 0xff); cb += (float)(u & 0xff);
                    }
                }
            }
            const float c = (float)cnt;
            uint32_t packed = 0;
            if (rgb_on) {
                const int R = (int)__fdiv_rn(cr, c), G = (int)__fdiv_rn(cg, c), B = (int)__fdiv_rn(cb, c);
                packed = ((uint32_t)R << 16) | ((uint32_t)G << 8) | (uint32_t)B;
            }
            vox[fbase + out0 + h] = make_float4(__fdiv_rn(sx, c), __fdiv_rn(sy, c), __fdiv_rn(sz, c), __uint_as_float(packed));
        }
    }
}

// ---- host launchers ---------------------------------------------------------------------
void launch_crop_count(hipStream_t s, const void* in, size_t stride, int N, int F, int rgb_off, CropLimits lim, int T,
                       FrameState* fs, int* tile_cnt) {
    const int Tin = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL(k_crop_count, dim3(Tin, F), dim3(BLOCK), 0, s, (const char*)in, stride, N, rgb_off, lim, T, fs, tile_cnt);
}
void launch_scan_tiles(hipStream_t s, int* counts, int rows, int T, int* totals, int total_pitch) {
    const ScanJob j{counts, totals, nullptr};
    hipLaunchKernelGGL(k_scan_tiles, dim3(rows, 1), dim3(BLOCK), 0, s, j, j, T, total_pitch);
}
void launch_scan_tiles2(hipStream_t s, const ScanJob& a, const ScanJob& b, int rows, int T, int total_pitch) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(rows, 2), dim3(BLOCK), 0, s, a, b, T, total_pitch);
}
void launch_voxel_setup(hipStream_t s, FrameState* fs, int F, float leaf, const uint32_t* ghist, FrameState* mirror) {
    hipLaunchKernelGGL(k_voxel_setup, dim3(F), dim3(WAVE), 0, s, fs, leaf, ghist, mirror);
}
void launch_crop_compact(hipStream_t s, const void* in, size_t stride, int N, int pitch, int F, int rgb_off, CropLimits lim,
                         int T, float leaf, const FrameState* fs, const int* tile_off, float4* cpt, uint32_t* keys) {
    const int Tin = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL(k_crop_compact, dim3(Tin, F), dim3(BLOCK), 0, s, (const char*)in, stride, N, pitch, rgb_off, lim, T, leaf,
                       fs, tile_off, cpt, keys);
}
void launch_crop_fused(hipStream_t s, const void* in, size_t stride, int N, int pitch, int F, int rgb_off, CropLimits lim,
                       int T, float leaf, KeyPack kp, FrameState* fs, int* state, float4* cpt, uint32_t* keys, int* ticket) {
    const int Tin = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL(k_crop_fused, dim3(Tin * F), dim3(BLOCK), 0, s, (const char*)in, stride, N, pitch, rgb_off, lim, T, Tin,
                       leaf, kp, fs, state, cpt, keys, ticket);
}
void launch_crop_runs(hipStream_t s, const void* in, size_t stride, int N, int pitch, int F, int rgb_off, CropLimits lim,
                      int T, float leaf, KeyPack kp, FrameState* fs, unsigned long long* state, float4* cpt, uint32_t* rkeys,
                      uint32_t* rvals, uint32_t* ghist, int* ticket, int direct) {
    const int Tin = (N + TILE - 1) / TILE;
    if (direct)
        hipLaunchKernelGGL(k_crop_runs<true>, dim3(Tin * F), dim3(BLOCK), 0, s, (const char*)in, stride, N, pitch, rgb_off, lim, T, Tin,
                           leaf, kp, fs, state, cpt, rkeys, rvals, ghist, ticket);
    else
        hipLaunchKernelGGL(k_crop_runs<false>, dim3(Tin * F), dim3(BLOCK), 0, s, (const char*)in, stride, N, pitch, rgb_off, lim, T, Tin,
                           leaf, kp, fs, state, cpt, rkeys, rvals, ghist, ticket);
}
void launch_voxel_centroid(hipStream_t s, const uint32_t* keys, const uint32_t* vals, const float4* cpt, int N, int F,
                           int T, int Tact, int rgb_on, FrameState* fs, int* state, float4* vox, int* ticket) {
    hipLaunchKernelGGL(k_voxel_centroid, dim3(Tact * F), dim3(BLOCK), 0, s, keys, vals, cpt, N, T, Tact, rgb_on, fs, state, vox, ticket);
}
void launch_voxel_centroid_runs(hipStream_t s, const uint32_t* keys, const uint32_t* vals, const float4* cpt, int N, int F,
                                int T, int Tact, int rgb_on, FrameState* fs, int* state, float4* vox, int* ticket, int lanes, int pts_pitch) {
    if (lanes) {
        hipLaunchKernelGGL(k_voxel_centroid_lanes, dim3(Tact * F), dim3(BLOCK), 0, s, keys, vals, cpt, N, T, Tact, rgb_on, fs, state, vox, ticket, pts_pitch);
        return;
    }
    hipLaunchKernelGGL(k_voxel_centroid_runs, dim3(Tact * F), dim3(BLOCK), 0, s, keys, vals, cpt, N, T, Tact, rgb_on, fs, state, vox, ticket, pts_pitch);
}

}  // namespace cd
