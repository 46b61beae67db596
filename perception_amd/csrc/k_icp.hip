// k_icp.hip - S6 point-to-point ICP of every cluster against a template.
//
// Replaces pcl::IterativeClosestPoint<PointXYZ,PointXYZ>::align + getFitnessScore (reference:
// cuboid_detection/src/iterative_closest_point.cpp:170-182,
// object_detection/src/object_pose_detection.cpp:220-235; maxIter 5000, transformation
// epsilon 1e-9, euclidean fitness epsilon = icp_fitness_score, no max correspondence distance,
// no rejectors).
//
// One launch == one PCL iteration for ALL clusters of ALL frames in the batch:
//   prologue (one lane per block, redundantly - it is ~2 us of scalar work and removes every
//             inter-workgroup hand-off): TransformationEstimationSVD (pcl::umeyama, Eigen
//             JacobiSVD restated in float32) from the 16 fixed-point moments the previous
//             launch accumulated, T_final <- T*T_final, DefaultConvergenceCriteria;
//   body    : X <- T*X (in place), exact brute-force nearest neighbour of every source point
//             over the template staged through LDS in 2048-point (32 KiB) chunks - every lane
//             reads the same LDS address (broadcast), ties keep the lowest template index -,
//             then the 16 moments {sum p, sum q, sum q p^T, sum d2} as 64-bit fixed point:
//             wave butterfly + one atomic per wave (order-free, hence bit-reproducible).
// State is double-buffered by launch parity and the moment buffers rotate mod 3, so no block
// ever reads a word another block writes in the same launch.
// No MFMA: the only matrix objects are 3x3; the pair loop is 8 dependent f32 VALU ops.
#include "kernels.hpp"

namespace cd {

constexpr int ICP_TILE = BLOCK;   // source points per work item (1 per lane)

struct Rot { float c, s; };
__device__ __forceinline__ Rot rot_mul(const Rot& a, const Rot& b) { return {a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }
__device__ __forceinline__ Rot rot_T(const Rot& a) { return {a.c, -a.s}; }
__device__ __forceinline__ void apply_left(float M[3][3], int p, int q, const Rot& j) {
    for (int i = 0; i < 3; ++i) {
        const float x = M[p][i], y = M[q][i];
        M[p][i] = j.c * x + j.s * y;
        M[q][i] = -j.s * x + j.c * y;
    }
}
__device__ __forceinline__ void apply_right(float M[3][3], int p, int q, const Rot& j) {
    for (int i = 0; i < 3; ++i) {
        const float x = M[i][p], y = M[i][q];
        M[i][p] = j.c * x - j.s * y;
        M[i][q] = j.s * x + j.c * y;
    }
}
__device__ __forceinline__ Rot make_jacobi(float x, float y, float z) {
    if (y == 0.f) return {1.f, 0.f};
    const float tau = (x - z) / (2.0f * fabsf(y));
    const float w = sqrtf(tau * tau + 1.0f);
    const float t = tau > 0.f ? 1.0f / (tau + w) : 1.0f / (tau - w);
    const float sign_t = t > 0.f ? 1.0f : -1.0f;
    const float n = 1.0f / sqrtf(t * t + 1.0f);
    Rot r;
    r.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    r.c = n;
    return r;
}
// Eigen 3.2 JacobiSVD<Matrix3f>(ComputeFullU|ComputeFullV): two-sided Jacobi, float32
__device__ void jacobi_svd3(const float A[3][3], float U[3][3], float S[3], float V[3][3]) {
    const float precision = 2.0f * 1.1920928955078125e-07f;
    const float consider_zero = 2.0f * 1.401298464324817e-45f;
    float scale = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = fmaxf(scale, fabsf(A[i][j]));
    if (scale == 0.f) scale = 1.f;
    float W[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            W[i][j] = A[i][j] / scale;
            U[i][j] = V[i][j] = (i == j) ? 1.f : 0.f;
        }
    bool finished = false;
    for (int sweep = 0; sweep < 64 && !finished; ++sweep) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const float thr = fmaxf(consider_zero, precision * fmaxf(fabsf(W[p][p]), fabsf(W[q][q])));
                if (fabsf(W[p][q]) > thr || fabsf(W[q][p]) > thr) {
                    finished = false;
                    const float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                    Rot rot1;
                    const float t = m00 + m11, d = m10 - m01;
                    if (t == 0.f) {
                        rot1.c = 0.f;
                        rot1.s = d > 0.f ? 1.f : -1.f;
                    } else {
                        const float u = d / t;
                        rot1.c = 1.0f / sqrtf(1.0f + u * u);
                        rot1.s = rot1.c * u;
                    }
                    const float n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11;
                    const float n11 = -rot1.s * m01 + rot1.c * m11;
                    const Rot j_right = make_jacobi(n00, n01, n11);
                    const Rot j_left = rot_mul(rot1, rot_T(j_right));
                    apply_left(W, p, q, j_left);
                    apply_right(U, p, q, rot_T(j_left));
                    apply_right(W, p, q, j_right);
                    apply_right(V, p, q, j_right);
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        const float a = fabsf(W[i][i]);
        S[i] = a;
        if (a != 0.f) {
            const float f = W[i][i] / a;
            for (int r = 0; r < 3; ++r) U[r][i] *= f;
        }
    }
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        float mxv = S[i];
        for (int k = i + 1; k < 3; ++k)
            if (S[k] > mxv) { mxv = S[k]; pos = k; }
        if (mxv == 0.f) break;
        if (pos != i) {
            float t = S[i]; S[i] = S[pos]; S[pos] = t;
            for (int r = 0; r < 3; ++r) {
                t = U[r][i]; U[r][i] = U[r][pos]; U[r][pos] = t;
                t = V[r][i]; V[r][i] = V[r][pos]; V[r][pos] = t;
            }
        }
    }
    for (int i = 0; i < 3; ++i) S[i] *= scale;
}
__device__ __forceinline__ float det3(const float m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}
__device__ __forceinline__ double unfix(unsigned long long s, int shift) { return ldexp((double)(long long)s, -shift); }

// pcl::umeyama(src, dst, false) from the fixed-point moments.  A: [0..2] sum p, [3..5] sum q,
// [6..14] sum q_a p_b, [15] sum d2.
__device__ void umeyama_from_moments(const unsigned long long* A, int n, float T[16]) {
    double mp[3], mq[3];
    for (int a = 0; a < 3; ++a) {
        mp[a] = unfix(A[a], FIX_SHIFT) / (double)n;
        mq[a] = unfix(A[3 + a], FIX_SHIFT) / (double)n;
    }
    float sigma[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) sigma[a][b] = (float)(unfix(A[6 + 3 * a + b], FIX_SHIFT) / (double)n - mq[a] * mp[b]);
    float U[3][3], S[3], V[3][3];
    jacobi_svd3(sigma, U, S, V);
    float sd[3] = {1.f, 1.f, 1.f};
    if (det3(sigma) < 0.f) sd[2] = -1.f;
    int rank = 0;
    for (int i = 0; i < 3; ++i)
        if (!(fabsf(S[i]) <= fabsf(S[0]) * 1e-5f)) ++rank;
    if (rank == 2) {
        sd[0] = 1.f; sd[1] = 1.f;
        sd[2] = (det3(U) * det3(V) > 0.f) ? 1.f : -1.f;
    }
    float R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i][j] = ((U[i][0] * sd[0]) * V[j][0] + (U[i][1] * sd[1]) * V[j][1]) + (U[i][2] * sd[2]) * V[j][2];
    const float mpf[3] = {(float)mp[0], (float)mp[1], (float)mp[2]};
    const float mqf[3] = {(float)mq[0], (float)mq[1], (float)mq[2]};
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[i][j];
        T[4 * i + 3] = mqf[i] - ((R[i][0] * mpf[0] + R[i][1] * mpf[1]) + R[i][2] * mpf[2]);
    }
    T[15] = 1.f;
}

__device__ __forceinline__ void xform(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    oy = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    oz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}

#ifdef CD_STATS
__device__ unsigned long long g_icp_stats[4];
extern "C" int cd_debug_icp_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_stats), sizeof(g_icp_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_stats), z, sizeof(z)); }
    return 0;
}
#endif

// Exact NN of (x,y,z) over the template, staged through LDS; every thread of the block calls it.
//
// Pruning is exact, not approximate.  The template is cut into runs of 64 consecutive points,
// each with an axis-aligned box [lo,hi] (built at cd_set_template).  For a query q the bound
//   e_a = max(lo_a - q_a, q_a - hi_a, 0),  lb = (e_x*e_x + e_y*e_y) + e_z*e_z
// evaluated with the canonical association satisfies lb <= d2(q,p) for EVERY p in the run in
// float arithmetic, because each step (rounded subtraction, square, rounded sum) is monotone in
// |component|.  A run is skipped only when lb > best for all 64 lanes (wave vote), so no
// candidate that could improve on or tie the current best is ever dropped.
// Seeding: (best, bi) enter as (nextafter(d(q, seed)), seed); an ascending strict-< scan from
// that state still ends at the lowest index achieving the global minimum (rule C5), and the run
// holding the true NN can never be pruned since its lb <= d_min < best.
__device__ __forceinline__ void nn_search(const float4* __restrict__ tpl, const float4* __restrict__ blo,
                                          const float4* __restrict__ bhi, int m, float4* s_tpl, float4* s_lo,
                                          float4* s_hi, float x, float y, float z, float& best, int& bi) {
    for (int c0 = 0; c0 < m; c0 += ICP_TPL_CHUNK) {
        const int cn = min(ICP_TPL_CHUNK, m - c0);
        const int nsub = (cn + ICP_SUB - 1) / ICP_SUB;
        __syncthreads();
        for (int k = threadIdx.x; k < cn; k += BLOCK) s_tpl[k] = tpl[c0 + k];
        if (threadIdx.x < nsub) {
            s_lo[threadIdx.x] = blo[c0 / ICP_SUB + threadIdx.x];
            s_hi[threadIdx.x] = bhi[c0 / ICP_SUB + threadIdx.x];
        }
        __syncthreads();
        for (int s = 0; s < nsub; ++s) {
            const float4 L = s_lo[s], H = s_hi[s];
            const float ex = fmaxf(fmaxf(L.x - x, x - H.x), 0.f);
            const float ey = fmaxf(fmaxf(L.y - y, y - H.y), 0.f);
            const float ez = fmaxf(fmaxf(L.z - z, z - H.z), 0.f);
            const float lb = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
#ifdef CD_STATS
            if ((threadIdx.x & 63) == 0) atomicAdd(&g_icp_stats[0], 1ull);
            { const unsigned long long need = __popcll(__ballot(lb <= best)); if ((threadIdx.x & 63) == 0 && need) { atomicAdd(&g_icp_stats[1], 1ull); atomicAdd(&g_icp_stats[2], need); } }
#endif
            if (__any(lb <= best)) {                       // wave-uniform; inactive lanes carry best = -1
                const int j0 = s * ICP_SUB, j1 = min(j0 + ICP_SUB, cn);
#pragma unroll 8
                for (int j = j0; j < j1; ++j) {
                    const float4 t = s_tpl[j];
                    const float d = dist2(x, y, z, t.x, t.y, t.z);
                    if (d < best) { best = d; bi = c0 + j; }   // strict: lowest index wins ties (C5)
                }
            }
        }
    }
}


// seed for a search that has no previous neighbour: best over the first point of every run
__device__ __forceinline__ void nn_seed_coarse(const float4* __restrict__ tpl, int m, float x, float y, float z,
                                               float& best, int& bi) {
    best = 3.402823466e38f;
    bi = 0;
    for (int j = 0; j < m; j += ICP_SUB) {
        const float4 t = tpl[j];
        const float d = dist2(x, y, z, t.x, t.y, t.z);
        if (d < best) { best = d; bi = j; }
    }
}
__device__ __forceinline__ float next_up_nonneg(float d) { return __uint_as_float(__float_as_uint(d) + 1u); }

__global__ void __launch_bounds__(BLOCK) k_icp_iter(int it, const IcpWork* __restrict__ work,
                                                    const IcpCluster* __restrict__ cl, IcpState* __restrict__ st,
                                                    unsigned long long* __restrict__ acc, const float4* __restrict__ tpl,
                                                    const float4* __restrict__ tlo, const float4* __restrict__ thi,
                                                    float4* __restrict__ src, int* __restrict__ nn, IcpParams prm) {
    __shared__ float4 s_tpl[ICP_TPL_CHUNK];
    __shared__ float4 s_lo[ICP_TPL_CHUNK / ICP_SUB], s_hi[ICP_TPL_CHUNK / ICP_SUB];
    __shared__ float s_T[16];
    __shared__ int s_done;
    const IcpWork wk = work[blockIdx.x];
    const IcpCluster c = cl[wk.cluster];
    const IcpState* sin = st + (size_t)wk.cluster * 2 + (it & 1);
    IcpState* sout = st + (size_t)wk.cluster * 2 + ((it + 1) & 1);
    const int lane = threadIdx.x & 63;
    if (sin->done) {
        if (wk.tile == 0 && threadIdx.x == 0) *sout = *sin;
        return;
    }
    if (threadIdx.x == 0) {
        IcpState so = *sin;
        int done = 0;
        if (it > 0) {
            const unsigned long long* A = acc + ((size_t)wk.cluster * 3 + (it - 1) % 3) * 16;
            float T[16];
            umeyama_from_moments(A, c.n, T);
            // final_transformation_ = transformation_ * final_transformation_
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    so.Tfinal[4 * i + j] = ((T[4 * i] * sin->Tfinal[j] + T[4 * i + 1] * sin->Tfinal[4 + j]) +
                                            T[4 * i + 2] * sin->Tfinal[8 + j]) + T[4 * i + 3] * sin->Tfinal[12 + j];
            so.iters = sin->iters + 1;
            // DefaultConvergenceCriteria::hasConverged
            if (so.iters >= prm.max_iter) {
                done = 1;
            } else {
                const double cos_angle = 0.5 * (double)(((T[0] + T[5]) + T[10]) - 1.0f);
                const double translation_sqr = (double)((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
                if (cos_angle >= prm.rot_thr && translation_sqr <= prm.trans_eps) {
                    done = 1;
                } else {
                    const double mse = unfix(A[15], FIX_SHIFT_D2) / (double)c.n;
                    if (fabs(mse - sin->prev_mse) < prm.abs_mse) done = 1;
                    else if (fabs(mse - sin->prev_mse) / sin->prev_mse < prm.rel_mse) done = 1;
                    so.prev_mse = mse;
                }
            }
            so.done = done;
            so.converged = done;
            for (int i = 0; i < 16; ++i) s_T[i] = T[i];
        }
        s_done = done;
        if (wk.tile == 0) *sout = so;
    }
    if (wk.tile == 0 && threadIdx.x >= 64 && threadIdx.x < 80)
        acc[((size_t)wk.cluster * 3 + (it + 1) % 3) * 16 + (threadIdx.x - 64)] = 0ull;
    __syncthreads();
    const int i = wk.tile * ICP_TILE + threadIdx.x;
    const bool active = i < c.n;
    float x = 0.f, y = 0.f, z = 0.f;
    if (active) {
        const float4 p = src[c.src_off + i];
        x = p.x; y = p.y; z = p.z;
        if (it > 0) {
            float T[16];
#pragma unroll
            for (int k = 0; k < 12; ++k) T[k] = s_T[k];
            float ox, oy, oz;
            xform(T, x, y, z, ox, oy, oz);
            x = ox; y = oy; z = oz;
            src[c.src_off + i] = make_float4(x, y, z, p.w);
        }
    }
    if (s_done) return;
    float best = -1.0f;   // inactive lanes never vote for a run and never update
    int bi = 0;
    const float4* tp = tpl + c.tpl_off;
    if (active) {
        if (it > 0) {
            bi = nn[c.src_off + i];                    // previous iteration's neighbour
            const float4 q0 = tp[bi];
            best = dist2(x, y, z, q0.x, q0.y, q0.z);
        } else {
            nn_seed_coarse(tp, c.tpl_m, x, y, z, best, bi);
        }
        best = best < 3.0e38f ? next_up_nonneg(best) : __uint_as_float(0x7f800000u);
    }
    nn_search(tp, tlo + c.tpl_off / ICP_SUB, thi + c.tpl_off / ICP_SUB, c.tpl_m, s_tpl, s_lo, s_hi, x, y, z, best, bi);
    if (active) nn[c.src_off + i] = bi;
    unsigned long long S[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) S[k] = 0ull;
    if (active) {
        const float4 q = tpl[c.tpl_off + bi];
        const float pv[3] = {x, y, z}, qv[3] = {q.x, q.y, q.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            S[a] = (unsigned long long)fixq(pv[a], FIX_SHIFT);
            S[3 + a] = (unsigned long long)fixq(qv[a], FIX_SHIFT);
#pragma unroll
            for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] = (unsigned long long)fixq(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
        }
        S[15] = (unsigned long long)fixq(best, FIX_SHIFT_D2);
    }
    unsigned long long* A = acc + ((size_t)wk.cluster * 3 + it % 3) * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned long long t = wave_sum_u64(S[k]);
        if (lane == 0) atomicAdd(&A[k], t);
    }
}

// getFitnessScore(): mean squared NN distance of T_final * (original source)
__global__ void __launch_bounds__(BLOCK) k_icp_fitness(const IcpWork* __restrict__ work, const IcpCluster* __restrict__ cl,
                                                       const IcpState* __restrict__ st, int parity,
                                                       unsigned long long* __restrict__ accf,
                                                       const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                                       const float4* __restrict__ thi, const float4* __restrict__ src0,
                                                       const int* __restrict__ nn) {
    __shared__ float4 s_tpl[ICP_TPL_CHUNK];
    __shared__ float4 s_lo[ICP_TPL_CHUNK / ICP_SUB], s_hi[ICP_TPL_CHUNK / ICP_SUB];
    const IcpWork wk = work[blockIdx.x];
    const IcpCluster c = cl[wk.cluster];
    const IcpState* s = st + (size_t)wk.cluster * 2 + parity;
    if (s->status != CD_OK) return;
    const int i = wk.tile * ICP_TILE + threadIdx.x;
    const bool active = i < c.n;
    float x = 0.f, y = 0.f, z = 0.f;
    if (active) {
        const float4 p = src0[c.src_off + i];
        float T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = s->Tfinal[k];
        xform(T, p.x, p.y, p.z, x, y, z);
    }
    float best = -1.0f;
    int bi = 0;
    const float4* tp = tpl + c.tpl_off;
    if (active) {
        bi = nn[c.src_off + i];                        // last iteration's neighbour as the seed
        const float4 q0 = tp[bi];
        best = dist2(x, y, z, q0.x, q0.y, q0.z);
        best = best < 3.0e38f ? next_up_nonneg(best) : __uint_as_float(0x7f800000u);
    }
    nn_search(tp, tlo + c.tpl_off / ICP_SUB, thi + c.tpl_off / ICP_SUB, c.tpl_m, s_tpl, s_lo, s_hi, x, y, z, best, bi);
    const unsigned long long v = active ? (unsigned long long)fixq(best, FIX_SHIFT_D2) : 0ull;
    const unsigned long long t = wave_sum_u64(v);
    if ((threadIdx.x & 63) == 0) atomicAdd(&accf[wk.cluster], t);
}

void launch_icp_iter(hipStream_t s, int it, int n_work, const IcpWork* work, const IcpCluster* cl, IcpState* st,
                     unsigned long long* acc, const float4* tpl, const float4* tlo, const float4* thi, float4* src,
                     int* nn, IcpParams prm) {
    if (n_work <= 0) return;
    hipLaunchKernelGGL(k_icp_iter, dim3(n_work), dim3(BLOCK), 0, s, it, work, cl, st, acc, tpl, tlo, thi, src, nn, prm);
}
void launch_icp_fitness(hipStream_t s, int n_work, const IcpWork* work, const IcpCluster* cl, const IcpState* st,
                        int parity, unsigned long long* accf, const float4* tpl, const float4* tlo, const float4* thi,
                        const float4* src0, const int* nn) {
    if (n_work <= 0) return;
    hipLaunchKernelGGL(k_icp_fitness, dim3(n_work), dim3(BLOCK), 0, s, work, cl, st, parity, accf, tpl, tlo, thi, src0, nn);
}

}  // namespace cd
