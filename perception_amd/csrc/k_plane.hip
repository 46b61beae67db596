// k_plane.hip - S2 RANSAC plane segmentation + S3 extract (+S3b second z crop).
//
// Replaces pcl::SACSegmentation<PointXYZ>::segment (SACMODEL_PLANE, SAC_RANSAC, optimize) and
// pcl::ExtractIndices / pcl::PassThrough that follow it (reference:
// cuboid_detection/src/ground_plane_segmentation.cpp:76-101,
// object_detection/src/object_pose_detection.cpp:300-336).
//
// PCL re-seeds its sampler with 12345 for every frame, so the hypothesis sequence is a pure
// function of the voxel cloud: k_ransac_sample replays PCL's drawIndexSample / isSampleGood /
// computeModelCoefficients serially (one lane per frame, shuffle state in an LDS hash map) and
// emits the first h_target plane models; k_ransac_count then scores ALL of them in ONE pass
// over the cloud: each lane keeps 8 points in registers (coalesced 16 B loads), the plane
// coefficients are wave-uniform (scalar loads), the inlier test is reduced per hypothesis with
// a 64-wide ballot + popcount and one atomic per wave.  The host replays PCL's sequential
// adaptive-k stop rule over the count vector (bit-exact: counts are integers).
#include <algorithm>

#include "kernels.hpp"

namespace cd {

// ---- sampler: PCL SampleConsensusModel::getSamples + plane model -------------------------
__global__ void __launch_bounds__(WAVE) k_ransac_sample(const float4* __restrict__ vox, int N,
                                                        FrameState* __restrict__ fs, const int* __restrict__ rnd,
                                                        int h_target, const int* __restrict__ active,
                                                        float4* __restrict__ models, int* __restrict__ valid) {
    CD_FRONT_PRIO();
    __shared__ int s_key[SAMPLER_MAP];
    __shared__ int s_val[SAMPLER_MAP];
    const int f = blockIdx.x;
    if (active && !active[f]) return;
    for (int i = threadIdx.x; i < SAMPLER_MAP; i += WAVE) s_key[i] = -1;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int n = fs[f].n_v;
    const float4* P = vox + (size_t)f * N;
    float4* M = models + (size_t)f * MAX_HYP;
    int* V = valid + (size_t)f * MAX_HYP;
    if (n < 3) { fs[f].n_hyp = 0; fs[f].sampler_exhausted = 1; return; }
    auto slot = [&](int pos) {
        uint32_t h = ((uint32_t)pos * 2654435761u) & (SAMPLER_MAP - 1);
        while (s_key[h] != -1 && s_key[h] != pos) h = (h + 1) & (SAMPLER_MAP - 1);
        return (int)h;
    };
    int rp = 0, h = 0, used = 0, exhausted = 0;
    auto get = [&](int pos) { const int h = slot(pos); return s_key[h] == pos ? s_val[h] : pos; };
    auto set = [&](int pos, int v) { const int h = slot(pos); if (s_key[h] == -1) ++used; s_key[h] = pos; s_val[h] = v; };
    while (h < h_target) {
        bool good = false;
        int s0 = 0, s1 = 0, s2 = 0;
        for (int it = 0; it < 1000; ++it) {            // max_sample_checks_
            if (rp + 3 > RND_TABLE || used + 6 > (SAMPLER_MAP * 3) / 4) { exhausted = 1; break; }
            for (int i = 0; i < 3; ++i) {               // drawIndexSample
                const uint32_t r = (uint32_t)rnd[rp++];
                const int j = i + (int)(r % (uint32_t)(n - i));
                const int vi = get(i), vj = get(j);
                set(i, vj); set(j, vi);
            }
            s0 = get(0); s1 = get(1); s2 = get(2);
            const float4 p0 = P[s0], p1 = P[s1], p2 = P[s2];
            // isSampleGood: (p1-p0)/(p2-p0) component ratios not all equal
            const float q0 = __fdiv_rn(p1.x - p0.x, p2.x - p0.x);
            const float q1 = __fdiv_rn(p1.y - p0.y, p2.y - p0.y);
            const float q2 = __fdiv_rn(p1.z - p0.z, p2.z - p0.z);
            if ((q0 != q1) || (q2 != q1)) { good = true; break; }
        }
        if (!good) { exhausted = 1; break; }
        // computeModelCoefficients
        const float4 p0 = P[s0], p1 = P[s1], p2 = P[s2];
        const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
        const float bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
        const float q0 = __fdiv_rn(ax, bx), q1 = __fdiv_rn(ay, by), q2 = __fdiv_rn(az, bz);
        int ok = 1;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 == q1 && q2 == q1) {
            ok = 0;   // collinear: PCL skips the hypothesis (++skipped_count)
        } else {
            float mx = __fsub_rn(__fmul_rn(ay, bz), __fmul_rn(az, by));
            float my = __fsub_rn(__fmul_rn(az, bx), __fmul_rn(ax, bz));
            float mz = __fsub_rn(__fmul_rn(ax, by), __fmul_rn(ay, bx));
            const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(mx, mx), __fmul_rn(my, my)), __fmul_rn(mz, mz)));
            mx = __fdiv_rn(mx, nrm); my = __fdiv_rn(my, nrm); mz = __fdiv_rn(mz, nrm);
            const float d = __fmul_rn(-1.0f, __fadd_rn(__fadd_rn(__fmul_rn(mx, p0.x), __fmul_rn(my, p0.y)), __fmul_rn(mz, p0.z)));
            m = make_float4(mx, my, mz, d);
        }
        M[h] = m;
        V[h] = ok;
        ++h;
    }
    fs[f].n_hyp = h;
    fs[f].sampler_exhausted = exhausted;
}

// ---- batched per-hypothesis inlier count --------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_ransac_count(const float4* __restrict__ vox, int N,
                                                        const FrameState* __restrict__ fs,
                                                        const float4* __restrict__ models, const int* __restrict__ valid,
                                                        const int* __restrict__ active, int h0, int h1, float thr,
                                                        int* __restrict__ counts) {
    CD_FRONT_PRIO();
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (active && !active[f]) return;
    const int n = fs[f].n_v;
    if (tile * TILE >= n) return;
    const int hmax = min(h1, fs[f].n_hyp);
    const float4* P = vox + (size_t)f * N;
    const int base = tile * TILE + w * WAVE_SPAN + lane;
    float px[ITEMS], py[ITEMS], pz[ITEMS];
    bool in[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        in[j] = e < n;
        const float4 p = P[e < n ? e : n - 1];   // (clamped, not tested: the eight loads go out together)
        px[j] = p.x; py[j] = p.y; pz[j] = p.z;
    }
    const float4* M = models + (size_t)f * MAX_HYP;
    const int* V = valid + (size_t)f * MAX_HYP;
    for (int h = h0; h < hmax; ++h) {
        if (!V[h]) continue;
        const float4 m = M[h];
        int c = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            c += __popcll(__ballot(in[j] && plane_dist(m.x, m.y, m.z, m.w, px[j], py[j], pz[j]) < thr));
        if (lane == 0 && c) atomicAdd(&counts[(size_t)f * MAX_HYP + h], c);
    }
}

__device__ __forceinline__ bool plane_inlier(const float4& m, int have, float thr, const float4& p) {
    return have && plane_dist(m.x, m.y, m.z, m.w, p.x, p.y, p.z) < thr;
}

// ---- 9 fixed-point moments of the inliers of the chosen model (rule C4) --------------------
// computeMeanAndCovarianceMatrix's accumulators: xx xy xz yy yz zz x y z (+ count)
__global__ void __launch_bounds__(BLOCK) k_plane_cov(const float4* __restrict__ vox, int N,
                                                     const FrameState* __restrict__ fs, const float4* __restrict__ model,
                                                     const int* __restrict__ have, float thr,
                                                     unsigned long long* __restrict__ sums) {
    CD_FRONT_PRIO();
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_v;
    if (tile * TILE >= n || !have[f]) return;
    const float4 m = model[f];
    const float4* P = vox + (size_t)f * N;
    const int base = tile * TILE + w * WAVE_SPAN + lane;
    unsigned long long S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float4 pr[ITEMS];
    load_rows_clamped<ITEMS>(P, base, n, pr);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        if (e < n) {
            const float4 p = pr[j];
            if (plane_inlier(m, 1, thr, p)) {
                S[0] += (unsigned long long)fixq(__fmul_rn(p.x, p.x), FIX_SHIFT);
                S[1] += (unsigned long long)fixq(__fmul_rn(p.x, p.y), FIX_SHIFT);
                S[2] += (unsigned long long)fixq(__fmul_rn(p.x, p.z), FIX_SHIFT);
                S[3] += (unsigned long long)fixq(__fmul_rn(p.y, p.y), FIX_SHIFT);
                S[4] += (unsigned long long)fixq(__fmul_rn(p.y, p.z), FIX_SHIFT);
                S[5] += (unsigned long long)fixq(__fmul_rn(p.z, p.z), FIX_SHIFT);
                S[6] += (unsigned long long)fixq(p.x, FIX_SHIFT);
                S[7] += (unsigned long long)fixq(p.y, FIX_SHIFT);
                S[8] += (unsigned long long)fixq(p.z, FIX_SHIFT);
                S[9] += 1ull;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const unsigned long long t = wave_sum_u64(S[k]);
        if (lane == 0 && t) atomicAdd(&sums[(size_t)f * 10 + k], t);
    }
}

// ---- S3: flags by the refined model, ordered compaction of (plane inliers) and (objects) ----
// bbox_filter.cpp:30-51: projection accumulated in double, stored to float, float division, strict compares
__device__ __forceinline__ bool within_bbox(const BBoxGate& g, float x, float y, float z) {
    float u = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(g.P[0], (double)x), __dmul_rn(g.P[1], (double)y)), __dmul_rn(g.P[2], (double)z)), g.P[3]);
    float v = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(g.P[4], (double)x), __dmul_rn(g.P[5], (double)y)), __dmul_rn(g.P[6], (double)z)), g.P[7]);
    const float w = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(g.P[8], (double)x), __dmul_rn(g.P[9], (double)y)), __dmul_rn(g.P[10], (double)z)), g.P[11]);
    u = __fdiv_rn(u, w);
    v = __fdiv_rn(v, w);
    return (g.rect[0] < u && u < g.rect[2]) && (g.rect[1] < v && v < g.rect[3]);
}

__device__ __forceinline__ void extract_flags(const float4& m, int have, float thr, int negative, int crop2, float z2lo,
                                              float z2hi, const BBoxGate& g, const float4& p, bool& inl, bool& obj) {
    if (g.enable == 2) {   // cd_bbox_filter: the index output is the set of points inside the rectangle
        inl = within_bbox(g, p.x, p.y, p.z);
        obj = false;
        return;
    }
    inl = plane_inlier(m, have, thr, p);
    obj = negative ? !inl : inl;
    if (obj && crop2) obj = (p.z >= z2lo) && (p.z <= z2hi);   // voxel centroids are finite
    if (obj && g.enable) obj = within_bbox(g, p.x, p.y, p.z);
}

__global__ void __launch_bounds__(BLOCK) k_plane_flag_count(const float4* __restrict__ vox, int N, int T,
                                                            const FrameState* __restrict__ fs,
                                                            const float4* __restrict__ model, const int* __restrict__ have,
                                                            float thr, int negative, int crop2, float z2lo, float z2hi, BBoxGate gate,
                                                            int* __restrict__ cnt_plane, int* __restrict__ cnt_obj) {
    CD_FRONT_PRIO();
    __shared__ int s_a[WAVES_PER_BLOCK], s_b[WAVES_PER_BLOCK];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_v;
    if (tile * TILE >= n) return;
    const float4 m = model[f];
    const int hv = have[f];
    const float4* P = vox + (size_t)f * N;
    const int base = tile * TILE + w * WAVE_SPAN + lane;
    int ca = 0, cb = 0;
    float4 pr[ITEMS];
    load_rows_clamped<ITEMS>(P, base, n, pr);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        bool inl = false, obj = false;
        if (e < n) extract_flags(m, hv, thr, negative, crop2, z2lo, z2hi, gate, pr[j], inl, obj);
        ca += __popcll(__ballot(inl));
        cb += __popcll(__ballot(obj));
    }
    if (lane == 0) { s_a[w] = ca; s_b[w] = cb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt_plane[(size_t)f * T + tile] = s_a[0] + s_a[1] + s_a[2] + s_a[3];
        cnt_obj[(size_t)f * T + tile] = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    }
}

__global__ void __launch_bounds__(BLOCK) k_extract_scatter(const float4* __restrict__ vox, int N, int T,
                                                           const FrameState* __restrict__ fs,
                                                           const float4* __restrict__ model, const int* __restrict__ have,
                                                           float thr, int negative, int crop2, float z2lo, float z2hi, BBoxGate gate,
                                                           const int* __restrict__ off_plane, const int* __restrict__ off_obj,
                                                           int* __restrict__ plane_idx, float4* __restrict__ obj_out) {
    CD_FRONT_PRIO();
    __shared__ int s_a[WAVES_PER_BLOCK], s_b[WAVES_PER_BLOCK];
    const int f = blockIdx.y, tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = fs[f].n_v;
    if (tile * TILE >= n) return;
    const size_t fbase = (size_t)f * N;
    const float4 m = model[f];
    const int hv = have[f];
    const float4* P = vox + fbase;
    const int base = tile * TILE + w * WAVE_SPAN + lane;
    float4 p[ITEMS];
    uint64_t ba[ITEMS], bb[ITEMS];
    int ca = 0, cb = 0;
    load_rows_clamped<ITEMS>(P, base, n, p);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        bool inl = false, obj = false;
        if (e < n) extract_flags(m, hv, thr, negative, crop2, z2lo, z2hi, gate, p[j], inl, obj);
        ba[j] = __ballot(inl);
        bb[j] = __ballot(obj);
        ca += __popcll(ba[j]);
        cb += __popcll(bb[j]);
    }
    if (lane == 0) { s_a[w] = ca; s_b[w] = cb; }
    __syncthreads();
    int pa = off_plane[(size_t)f * T + tile], pb = off_obj[(size_t)f * T + tile];
    for (int q = 0; q < w; ++q) { pa += s_a[q]; pb += s_b[q]; }
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE;
        if ((ba[j] >> lane) & 1ull) plane_idx[fbase + pa + __popcll(ba[j] & lt)] = e;
        if ((bb[j] >> lane) & 1ull) obj_out[fbase + pb + __popcll(bb[j] & lt)] = p[j];
        pa += __popcll(ba[j]);
        pb += __popcll(bb[j]);
    }
}

void launch_ransac_sample(hipStream_t s, const float4* vox, int N, int F, FrameState* fs, const int* rnd_table,
                          int h_target, const int* active, float4* models, int* valid) {
    hipLaunchKernelGGL(k_ransac_sample, dim3(F), dim3(WAVE), 0, s, vox, N, fs, rnd_table, h_target, active, models, valid);
}
void launch_ransac_count(hipStream_t s, const float4* vox, int N, int F, int Tact, const FrameState* fs,
                         const float4* models, const int* valid, const int* active, int h0, int h1, float thr, int* counts) {
    hipLaunchKernelGGL(k_ransac_count, dim3(Tact, F), dim3(BLOCK), 0, s, vox, N, fs, models, valid, active, h0, h1, thr, counts);
}
void launch_plane_cov(hipStream_t s, const float4* vox, int N, int F, int Tact, const FrameState* fs, const float4* model,
                      const int* have, float thr, unsigned long long* sums) {
    hipLaunchKernelGGL(k_plane_cov, dim3(Tact, F), dim3(BLOCK), 0, s, vox, N, fs, model, have, thr, sums);
}
void launch_plane_flag_count(hipStream_t s, const float4* vox, int N, int F, int T, int Tact, const FrameState* fs,
                             const float4* model, const int* have, float thr, int negative, int crop2, float z2lo,
                             float z2hi, const BBoxGate& gate, int* cnt_plane, int* cnt_obj) {
    hipLaunchKernelGGL(k_plane_flag_count, dim3(Tact, F), dim3(BLOCK), 0, s, vox, N, T, fs, model, have, thr, negative,
                       crop2, z2lo, z2hi, gate, cnt_plane, cnt_obj);
}
void launch_extract_scatter(hipStream_t s, const float4* vox, int N, int F, int T, int Tact, const FrameState* fs,
                            const float4* model, const int* have, float thr, int negative, int crop2, float z2lo,
                            float z2hi, const BBoxGate& gate, const int* off_plane, const int* off_obj, int* plane_idx, float4* obj) {
    hipLaunchKernelGGL(k_extract_scatter, dim3(Tact, F), dim3(BLOCK), 0, s, vox, N, T, fs, model, have, thr, negative,
                       crop2, z2lo, z2hi, gate, off_plane, off_obj, plane_idx, obj);
}

// ---- S3 as a call of its own: pcl::ExtractIndices<PCLPointCloud2> on whole records --------------------------------------
// (the fused path extracts inside k_extract_scatter; these serve cd_extract.)  Records are `words` 4-byte words.
__global__ void __launch_bounds__(BLOCK) k_mark_indices(const int* __restrict__ idx, int m, int n, int* __restrict__ flag) {
    CD_FRONT_PRIO();
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < m) { const int k = idx[i]; if (k >= 0 && k < n) flag[k] = 1; }
}
// pcl::PassThrough<PCLPointCloud2> as a call of its own (cd_passthrough): flag[i] = 1 for the records it REMOVES - x, y or z
// not finite, the field value not finite, or the value outside [lo, hi] compared as double (negative: inside (lo, hi)), exactly
// PCL's `distance_value > filter_limit_max_ || distance_value < filter_limit_min_` (negative: `<` and `>`)
__global__ void __launch_bounds__(BLOCK) k_passthrough_mark(const char* __restrict__ in, size_t stride, int n, int field_off, double lo, double hi,
                                                            int negative, int* __restrict__ flag) {
    CD_FRONT_PRIO();
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const char* p = in + (size_t)i * stride;
    const float x = *reinterpret_cast<const float*>(p), y = *reinterpret_cast<const float*>(p + 4), z = *reinterpret_cast<const float*>(p + 8);
    bool remove = !((fabsf(x) <= 3.402823466e38f) && (fabsf(y) <= 3.402823466e38f) && (fabsf(z) <= 3.402823466e38f));
    if (field_off >= 0) {
        const float vf = *reinterpret_cast<const float*>(p + field_off);
        const double v = (double)vf;
        remove = remove || !(fabsf(vf) <= 3.402823466e38f);
        remove = remove || (negative ? (v < hi && v > lo) : (v > hi || v < lo));
    }
    flag[i] = remove ? 1 : 0;
}
// positions i with flag[i] == 0, ascending (ordered compaction: ballots + chained scan over the tiles)
__global__ void __launch_bounds__(BLOCK) k_select_unmarked(const int* __restrict__ flag, int n, int* __restrict__ state,
                                                           FrameState* __restrict__ fs, int* __restrict__ out, int* __restrict__ ticket) {
    CD_FRONT_PRIO();
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ int s_excl, s_ticket;
    const int tile = take_ticket(ticket, (int)gridDim.x, &s_ticket), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int base = tile * TILE + w * WAVE_SPAN;
    uint64_t bal[ITEMS];
    int wtot = 0;
    int fl[ITEMS];
    load_rows_clamped<ITEMS>(flag, base + lane, n, fl);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = base + j * WAVE + lane;
        bal[j] = ballot64(e < n && fl[j] == 0);
        wtot += __popcll(bal[j]);
    }
    if (lane == 0) s_cnt[w] = wtot;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < WAVES_PER_BLOCK; ++k) tot += s_cnt[k];
        const int excl = chained_scan(state, 1, tile, tot, &fs[0].scan_stalled);
        s_excl = excl;
        if ((tile + 1) * TILE >= n) fs[0].n_plane = excl + tot;
    }
    __syncthreads();
    int pos = s_excl;
    for (int k = 0; k < w; ++k) pos += s_cnt[k];
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((bal[j] >> lane) & 1ull) out[pos + __popcll(bal[j] & lt)] = base + j * WAVE + lane;
        pos += __popcll(bal[j]);
    }
}
__global__ void __launch_bounds__(BLOCK) k_gather_records(const uint32_t* __restrict__ in, int words, const int* __restrict__ idx,
                                                          int m, uint32_t* __restrict__ out) {
    CD_FRONT_PRIO();
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (size_t)m * words) return;
    const int r = (int)(t / words), wd = (int)(t % words);
    out[t] = in[(size_t)idx[r] * words + wd];
}
// float4 points (x, y, z, packed rgb) -> records of `words` 4-byte words in a PointCloud2 layout: x,y,z in words 0..2, the
// colour in word rgb_word (< 0: none), word 3 = pad3 when it is not the colour word (pcl::PointXYZ keeps 1.0f there and
// pcl::toROSMsg ships the struct as it is), every other word zero.  One thread per output word: coalesced stores.
__global__ void __launch_bounds__(BLOCK) k_pack_records(const float4* __restrict__ pts, int m, int words, int rgb_word, uint32_t pad3,
                                                        uint32_t* __restrict__ out) {
    CD_FRONT_PRIO();
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (size_t)m * words) return;
    const int r = (int)(t / words), wd = (int)(t % words);
    const float4 p = pts[r];
    uint32_t v = 0u;
    if (wd == 0) v = __float_as_uint(p.x);
    else if (wd == 1) v = __float_as_uint(p.y);
    else if (wd == 2) v = __float_as_uint(p.z);
    else if (wd == rgb_word) v = __float_as_uint(p.w);
    else if (wd == 3) v = pad3;
    out[t] = v;
}
void launch_pack_records(hipStream_t s, const float4* pts, int m, int words, int rgb_word, uint32_t pad3, void* out) {
    const size_t tot = (size_t)m * words;
    if (tot > 0) hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((tot + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, pts, m, words, rgb_word, pad3, (uint32_t*)out);
}
// Small transfers between the context's PINNED host mirrors (device-visible) and device memory as ordinary kernels of the
// context's stream instead of hipMemcpyAsync.  Why: the runtime hands such copies to an SDMA engine, where a copy waits IN THE
// ENGINE'S RING for the kernel before it on its stream - with batches in flight that kernel may be queued for milliseconds
// behind other contexts' ICP launches, and every other context's copy on the same engine waits behind it (measured:
// hipMemcpyAsync calls of ALL contexts blocking 5-8 ms at once, tools/hip_api_long_calls.sh).  A kernel is ordered by its own
// stream only.  Rows of `width` bytes (a multiple of 4), `rows` of them, pitches in bytes; rows = 1 for a plain copy.
__global__ void __launch_bounds__(BLOCK) k_copy_rows(CopyList L) {
    CD_FRONT_PRIO();
    const CopySeg sg = L.seg[blockIdx.y];
    for (int r = blockIdx.z; r < sg.rows; r += gridDim.z)
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < sg.width_w; i += gridDim.x * BLOCK) sg.dst[(size_t)r * sg.dpitch_w + i] = sg.src[(size_t)r * sg.spitch_w + i];
}
// every zero-initialised scratch array of a fused batch call in ONE launch (blockIdx.y = region) instead of a dozen fill kernels
__global__ void __launch_bounds__(BLOCK) k_zero_regions(ZeroRegions r) {
    CD_FRONT_PRIO();
    if ((int)blockIdx.y == r.n) {   // the FrameState array's initial value: zero, mn[] = the ordered-uint encoding of +max
        constexpr size_t W = sizeof(FrameState) / 4, MN0 = offsetof(FrameState, mn) / 4;
        uint32_t* p = reinterpret_cast<uint32_t*>(r.fs);
        const size_t n = W * (size_t)r.nfs;
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
            const size_t w = i % W;
            p[i] = (w >= MN0 && w < MN0 + 3) ? 0xffffffffu : 0u;
        }
        return;
    }
    uint32_t* p = r.ptr[blockIdx.y];
    const size_t n = r.words[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) p[i] = 0u;
}
void launch_zero_regions(hipStream_t s, const ZeroRegions& r) {
    const int ny = r.n + (r.fs && r.nfs > 0 ? 1 : 0);
    if (ny > 0) hipLaunchKernelGGL(k_zero_regions, dim3(128, ny), dim3(BLOCK), 0, s, r);
}
// several transfers in ONE launch (blockIdx.y = transfer): consecutive copies of a stage cost one stream operation
void launch_copy_list(hipStream_t s, const CopyList& L) {
    if (L.n <= 0) return;
    int gx = 1, gz = 1;
    for (int k = 0; k < L.n; ++k) {
        gx = std::max(gx, std::min(64, (L.seg[k].width_w + BLOCK - 1) / BLOCK));
        gz = std::max(gz, std::min(L.seg[k].rows, 256));
    }
    hipLaunchKernelGGL(k_copy_rows, dim3(gx, L.n, gz), dim3(BLOCK), 0, s, L);
}
void launch_copy_rows(hipStream_t s, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int rows) {
    if (width == 0 || rows <= 0) return;
    CopyList L;
    L.n = 1;
    L.seg[0] = CopySeg{(uint32_t*)dst, (const uint32_t*)src, dpitch / 4, spitch / 4, (int)(width / 4), rows};
    launch_copy_list(s, L);
}
void launch_passthrough_mark(hipStream_t s, const void* in, size_t stride, int n, int field_off, double lo, double hi, int negative, int* flag) {
    if (n > 0) hipLaunchKernelGGL(k_passthrough_mark, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, (const char*)in, stride, n, field_off, lo, hi, negative, flag);
}
void launch_mark_indices(hipStream_t s, const int* idx, int m, int n, int* flag) {
    if (m > 0) hipLaunchKernelGGL(k_mark_indices, dim3((m + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, idx, m, n, flag);
}
void launch_select_unmarked(hipStream_t s, const int* flag, int n, int* state, FrameState* fs, int* out, int* ticket) {
    hipLaunchKernelGGL(k_select_unmarked, dim3((n + TILE - 1) / TILE), dim3(BLOCK), 0, s, flag, n, state, fs, out, ticket);
}
void launch_gather_records(hipStream_t s, const void* in, int words, const int* idx, int m, void* out) {
    const size_t tot = (size_t)m * words;
    if (tot > 0) hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((tot + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, (const uint32_t*)in, words, idx, m, (uint32_t*)out);
}

}  // namespace cd
