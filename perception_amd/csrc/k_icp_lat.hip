// k_icp_lat.hip - S6 point-to-point ICP against a LATTICE template: closed-form nearest neighbour.
//
// Replaces pcl::IterativeClosestPoint<PointXYZ,PointXYZ>::align + getFitnessScore (reference:
// cuboid_detection/src/iterative_closest_point.cpp:170-182, object_detection/src/object_pose_detection.cpp:220-235) for the
// templates cuboid_detection/templates/make_cuboid.py:38-55 writes: face k = meshgrid of two of the three axis tables
// X, Y, Z at a constant third coordinate, first axis fastest, faces one after the other.  cd_set_template verifies that
// structure bit by bit against the uploaded points (lattice_detect, cuboid_hip.hip) and hands over IcpLattice (common.hpp);
// every other template keeps the pruned searches of k_icp.hip.
//
// The search.  The canonical squared distance (rule C1/C5, common.hpp dist2) to point (i, j) of a face with constant z = c is
//     d2(i, j) = fl(fl(fx(i) + fy(j)) + fz),   fx(i) = fl(fl(qx - X[i])^2),  fy(j) = fl(fl(qy - Y[j])^2),  fz = fl(fl(qz - c)^2)
// (the constant term takes the place of its axis for the faces of constant x or y).  Every rounded operation is monotone:
// fl(a + b) does not decrease when a or b grows, fl(q - t) does not increase when t grows, squaring is monotone in |.|.  Hence
//   (1) min over the face = d2(i*, j*) with i* = argmin fx, j* = argmin fy - the two axes separate;
//   (2) fx is unimodal in i (tables ascend), so i* is the table entry nearest to qx: with tables uniform to 1/16 of a step
//       (verified on the host) it is one of ig - 1, ig, ig + 1 for ig = clamp(rint((qx - X[0]) / step)), all three evaluated
//       exactly - one 16-byte LDS read of the window (X[ig-1], X[ig], X[ig+1]);
//   (3) the points of the face that TIE with the minimum (equal float d2; rule C5 wants the lowest original index =
//       lowest slow-axis index, then lowest fast-axis index) form, along each axis, a contiguous run that contains i* / j*:
//       walk the slow axis down while d2(i*, j - 1) == min, then the fast axis.  A tie needs f(lower neighbour) - f(min) below
//       the rounding of the sum, so the walk is only entered by lanes whose gap is <= 2^-21 * min (a conservative filter:
//       it only ever sends too many lanes to the exact test);
//   (4) faces are consecutive in the file, so across faces the lexicographic (d2, index) minimum is the FIRST face that
//       reaches the minimum.
// Checked against brute force with the oracle's arithmetic on all five reference cuboid templates incl. the 21 400-point
// six-face one: tests/test_gpu_lattice.py (near, far, mid-cell, +-300 m along a face normal - a tie walk across the whole face).
// ~100 vector instructions per query with no divergence, against ~2400 per 64-query pass of the pruned searches, and no
// 116 KB template image: a workgroup needs <= 12 KB of LDS, so ICP workgroups stop monopolising CUs.
//
// Launch shape: ONE workgroup per cluster, all iterations and the fitness pass inside it (as k_icp_cluster): per iteration
// every wave takes passes of 64 points (X <- T X in place, search, 16 fixed-point moment sums of rule C4 kept in registers),
// the sums meet in LDS, thread 0 runs Umeyama + SVD + the convergence tests (icp_solve.hpp, same code as k_icp_solve), two
// barriers.  The hardware's workgroup dispatcher does the load balancing the persistent kernels did with queues and hand-overs.
// No MFMA: there is no dense contraction here (3x3 matrices only).
#include "kernels.hpp"
#include "icp_solve.hpp"

namespace cd {

// face descriptors: what the per-query face loop needs as wave-uniform values (scalar registers after unrolling with constant
// indices) - the constant coordinate and two all-ones / all-zeros words that say which axis it replaces - the rest as LDS
// words read per lane after the loop (s_face[f] = constant axis, fast axis, first index, constant coordinate)
struct LatFaces {
    int nface;
    unsigned m0[LAT_MAX_FACES], m1[LAT_MAX_FACES], m2[LAT_MAX_FACES];   // ~0 when the face's constant axis is x / y / z
    float c[LAT_MAX_FACES];
    int nx, ny, nz, tox, toy, toz;   // entries and first entry of the axis tables (separate scalars: an array indexed by a face's axis would be put into scratch memory)
    float ox, oy, oz, ivx, ivy, ivz;
};

// call from every thread of the workgroup; ends with a barrier
template <int THREADS>
__device__ __forceinline__ void lat_stage(const IcpLattice* __restrict__ L, LatFaces& F, float4* s_tab, int4* s_face) {
    F.nface = L->nface;
#pragma unroll
    for (int f = 0; f < LAT_MAX_FACES; ++f) { F.m0[f] = L->m0[f]; F.m1[f] = L->m1[f]; F.m2[f] = L->m2[f]; F.c[f] = L->c[f]; }
    F.nx = L->n[0]; F.ny = L->n[1]; F.nz = L->n[2]; F.tox = L->toff[0]; F.toy = L->toff[1]; F.toz = L->toff[2];
    F.ox = L->o[0]; F.oy = L->o[1]; F.oz = L->o[2]; F.ivx = L->inv[0]; F.ivy = L->inv[1]; F.ivz = L->inv[2];
    for (int i = threadIdx.x; i < L->ntab; i += THREADS) s_tab[i] = L->tab[i];
    if (threadIdx.x < LAT_MAX_FACES) {
        const int f = threadIdx.x, w = L->w[f], u = L->fast[f], v = 3 - w - u;
        s_face[f] = make_int4(w, u, L->base[f], __float_as_int(L->c[f]));
        s_face[LAT_MAX_FACES + f] = make_int4(L->toff[u], L->toff[v < 0 || v > 2 ? 0 : v], L->n[u], 0);   // tables of its fast and slow axis (unused faces: anything)
    }
    __syncthreads();
}

// m ? a : b for a wave-uniform all-ones / all-zeros word m: ONE v_bfi_b32 with the mask as its scalar operand (written as
// (m & a) | (~m & b) the compiler keeps m AND ~m in scalar registers and issues v_and + v_and_or)
__device__ __forceinline__ float lat_pick(unsigned m, float a, float b) {
    float r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(m), "v"(a), "v"(b));
    return r;
}

// one axis: the table entry nearest to q.  f = fl(fl(q - T[i])^2) at the minimum, t = T[i], gap = f(i - 1) - f(i) (+inf at
// i = 0; 0 = not known: the exact test decides).  Every axis has a table (one that no face varies along has the single entry 0).
__device__ __forceinline__ void lat_axis(const float4* s_tab, int toff, int n, float o, float inv, float q, float& f, int& i, float& t,
                                         float& gap) {
    int ig = __float2int_rn(__fmul_rn(__fsub_rn(q, o), inv));   // (saturating conversion; NaN -> 0)
    ig = min(max(ig, 0), n - 1);
    const float4 W = s_tab[toff + ig];
    const float d0 = __fsub_rn(q, W.x), d1 = __fsub_rn(q, W.y), d2 = __fsub_rn(q, W.z);
    const float f0 = __fmul_rn(d0, d0), f1 = __fmul_rn(d1, d1), f2 = __fmul_rn(d2, d2);
    const float g_mid = __fsub_rn(f0, f1), g_hi = __fsub_rn(f1, f2);
    // (f is unimodal along the table, so f1 > f0 and f1 > f2 cannot both hold: at most one of lo / hi)
    const bool lo = f0 < f1;
    const bool hi = f2 < f1;
    f = f1; i = ig; t = W.y; gap = g_mid;
    f = lo ? f0 : f; i = lo ? ig - 1 : i; t = lo ? W.x : t; gap = lo ? 0.f : gap;
    f = hi ? f2 : f; i = hi ? ig + 1 : i; t = hi ? W.z : t; gap = hi ? g_hi : gap;
}

// d2 of a face point from the three per-axis terms, canonical association (x + y) + z
__device__ __forceinline__ float lat_sum(float ax, float ay, float az) { return __fadd_rn(__fadd_rn(ax, ay), az); }

struct LatHit {
    float d;            // canonical squared distance to the nearest template point
    float nx, ny, nz;   // that point
    int face;           // its face
    int ix, iy, iz;     // table indices (the constant axis' entry is not used)
};

// TIES = false: d only (getFitnessScore needs no neighbour).  NF = faces evaluated (3 or LAT_MAX_FACES; the launch's templates
// have at most that many - entries beyond a template's own faces never win)
template <bool TIES, int NF>
__device__ __forceinline__ LatHit lat_nearest(const float4* s_tab, const int4* s_face, const LatFaces& F, float qx, float qy, float qz) {
    float fx, fy, fz, tx, ty, tz, gx, gy, gz;
    LatHit h;
    lat_axis(s_tab, F.tox, F.nx, F.ox, F.ivx, qx, fx, h.ix, tx, gx);
    lat_axis(s_tab, F.toy, F.ny, F.oy, F.ivy, qy, fy, h.iy, ty, gy);
    lat_axis(s_tab, F.toz, F.nz, F.oz, F.ivz, qz, fz, h.iz, tz, gz);
    h.d = __uint_as_float(0x7f800000u); h.face = 0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const unsigned m0 = F.m0[f], m1 = F.m1[f], m2 = F.m2[f];
        const float dc = __fsub_rn(lat_pick(m0, qx, lat_pick(m1, qy, qz)), F.c[f]);
        const float fc = __fmul_rn(dc, dc);
        const float d = lat_sum(lat_pick(m0, fc, fx), lat_pick(m1, fc, fy), lat_pick(m2, fc, fz));
        const bool better = d < h.d;   // strict: the first face that reaches the minimum keeps it (4); NaN (no such face): never
        h.d = better ? d : h.d;
        if (TIES) h.face = better ? f : h.face;
    }
    h.nx = tx; h.ny = ty; h.nz = tz;
    if (TIES) {
        // the winning face's constant coordinate replaces its axis' table value; (3): lanes whose lower neighbour along an
        // in-plane axis may round to the same sum take the exact walk
        const int4 fd = s_face[h.face];
        const float c = __int_as_float(fd.w);
        h.nx = fd.x == 0 ? c : tx; h.ny = fd.x == 1 ? c : ty; h.nz = fd.x == 2 ? c : tz;
        const float gsel = fd.x == 0 ? fminf(gy, gz) : (fd.x == 1 ? fminf(gx, gz) : fminf(gx, gy));
        const bool maybe = gsel <= __fmul_rn(h.d, 4.76837158203125e-07f);   // 2^-21
        if (ballot64(maybe)) {
            // rare (a lane enters when its gap is within the rounding of the sum: iteration 0, mid-cell queries): one face at a time
            for (int f = 0; f < F.nface; ++f) {
                const bool mine = maybe && h.face == f;
                if (!ballot64(mine)) continue;
                const int4 gd = s_face[f];
                const int w = __builtin_amdgcn_readfirstlane(gd.x), u = __builtin_amdgcn_readfirstlane(gd.y), v = 3 - w - u;   // constant, fast, slow axis
                const float cc = __int_as_float(__builtin_amdgcn_readfirstlane(gd.w));
                const int4 ge = s_face[LAT_MAX_FACES + f];
                const int toff_u = __builtin_amdgcn_readfirstlane(ge.x), toff_v = __builtin_amdgcn_readfirstlane(ge.y);
                const float qw = w == 0 ? qx : (w == 1 ? qy : qz), qu = u == 0 ? qx : (u == 1 ? qy : qz), qv = v == 0 ? qx : (v == 1 ? qy : qz);
                const float dc = __fsub_rn(qw, cc);
                const float fc = __fmul_rn(dc, dc);
                float fu = u == 0 ? fx : (u == 1 ? fy : fz), fv = v == 0 ? fx : (v == 1 ? fy : fz);
                float tu = u == 0 ? tx : (u == 1 ? ty : tz), tv = v == 0 ? tx : (v == 1 ? ty : tz);
                int iu = u == 0 ? h.ix : (u == 1 ? h.iy : h.iz), iv = v == 0 ? h.ix : (v == 1 ? h.iy : h.iz);
                // d2 with the terms of axes u, v, w put back on x, y, z
                auto d_of = [&](float a_u, float a_v) {
                    const float ax = w == 0 ? fc : (u == 0 ? a_u : a_v), ay = w == 1 ? fc : (u == 1 ? a_u : a_v), az = w == 2 ? fc : (u == 2 ? a_u : a_v);
                    return lat_sum(ax, ay, az);
                };
                // slow axis first (its index is the high part of the original index), then the fast axis at that row
                for (;;) {
                    const int j = max(iv - 1, 0);
                    const float t = s_tab[toff_v + j].y;
                    const float dd = __fsub_rn(qv, t);
                    const float a = __fmul_rn(dd, dd);
                    const bool go = mine && iv > 0 && d_of(fu, a) == h.d;
                    iv = go ? j : iv; fv = go ? a : fv; tv = go ? t : tv;
                    if (!ballot64(go)) break;
                }
                for (;;) {
                    const int i = max(iu - 1, 0);
                    const float t = s_tab[toff_u + i].y;
                    const float dd = __fsub_rn(qu, t);
                    const float a = __fmul_rn(dd, dd);
                    const bool go = mine && iu > 0 && d_of(a, fv) == h.d;
                    iu = go ? i : iu; fu = go ? a : fu; tu = go ? t : tu;
                    if (!ballot64(go)) break;
                }
                const bool ux = mine && u == 0, uy = mine && u == 1, uz = mine && u == 2, vx = mine && v == 0, vy = mine && v == 1, vz = mine && v == 2;
                h.ix = ux ? iu : (vx ? iv : h.ix); h.nx = ux ? tu : (vx ? tv : h.nx);
                h.iy = uy ? iu : (vy ? iv : h.iy); h.ny = uy ? tu : (vy ? tv : h.ny);
                h.iz = uz ? iu : (vz ? iv : h.iz); h.nz = uz ? tu : (vz ? tv : h.nz);
            }
        }
    }
    return h;
}

// original index of a hit (the diagnostic entry point; the ICP itself only needs the neighbour's coordinates)
__device__ __forceinline__ int lat_index(const int4* s_face, const LatFaces& F, const LatHit& h) {
    const int4 fd = s_face[h.face];
    const int w = fd.x, u = fd.y, v = 3 - w - u;
    const int iu = u == 0 ? h.ix : (u == 1 ? h.iy : h.iz), iv = v == 0 ? h.ix : (v == 1 ? h.iy : h.iz);
    return fd.z + iv * s_face[LAT_MAX_FACES + h.face].z + iu;
}

// workgroups per CU the register allocation is held to: four waves per SIMD (128 registers: the single-lane solve is what needs them)
#define LAT_WG_PER_CU(threads) ((threads) == 256 ? 4 : ((threads) == 512 ? 2 : 1))

// one PCL iteration's state update from the moment sums (what k_icp_solve / pipe_solve do): TransformationEstimationSVD,
// final_transformation_ = transformation_ * final_transformation_, DefaultConvergenceCriteria::hasConverged.  Returns done.
__device__ __forceinline__ int lat_solve(IcpState* so, const unsigned long long* A, int n, const IcpParams& prm) {
    float Tn[16];
    umeyama_from_moments(A, n, Tn);
    float Tf[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            Tf[4 * i + j] = ((Tn[4 * i] * so->Tfinal[j] + Tn[4 * i + 1] * so->Tfinal[4 + j]) + Tn[4 * i + 2] * so->Tfinal[8 + j]) +
                            Tn[4 * i + 3] * so->Tfinal[12 + j];
    for (int i = 0; i < 16; ++i) so->Tfinal[i] = Tf[i];
    so->iters += 1;
    int done = 0;
    if (so->iters >= prm.max_iter) {
        done = 1;
    } else {
        const double cos_angle = 0.5 * (double)(((Tn[0] + Tn[5]) + Tn[10]) - 1.0f);
        const double translation_sqr = (double)((Tn[3] * Tn[3] + Tn[7] * Tn[7]) + Tn[11] * Tn[11]);
        if (cos_angle >= prm.rot_thr && translation_sqr <= prm.trans_eps) {
            done = 1;
        } else {
            const double mse = unfix(A[15], FIX_SHIFT_D2) / (double)n;
            if (fabs(mse - so->prev_mse) < prm.abs_mse) done = 1;
            else if (fabs(mse - so->prev_mse) / so->prev_mse < prm.rel_mse) done = 1;
            so->prev_mse = mse;
        }
    }
    for (int i = 0; i < 16; ++i) so->T[i] = Tn[i];
    return done;
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS, LAT_WG_PER_CU(THREADS)) k_icp_lat(const int* __restrict__ order, const IcpCluster* __restrict__ cl, IcpState* __restrict__ st,
                                                     unsigned long long* __restrict__ accf, const IcpLattice* __restrict__ lats,
                                                     float4* __restrict__ src, const float4* __restrict__ src0, IcpParams prm) {
    __shared__ float4 s_tab[LAT_MAX_TAB];
    __shared__ int4 s_face[2 * LAT_MAX_FACES];
    __shared__ unsigned long long s_acc[16];
    __shared__ IcpState s_so;
    __shared__ int s_done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = THREADS / WAVE;
    const int k = order[blockIdx.x];
    const IcpCluster c = cl[k];
    if (st[2 * (size_t)k].done) return;   // host pre-marked (too few points / no template); uniform
    LatFaces F;
    if (threadIdx.x < 16) s_acc[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) { s_so = st[2 * (size_t)k]; s_done = 0; }
    lat_stage<THREADS>(lats + c.slot, F, s_tab, s_face);
    const bool nf3 = F.nface <= 3;   // (uniform)
    float4* pts = src + c.src_off;
    const float4* pts0 = src0 + c.src_off;
    const int n = c.n, npass = (n + 63) >> 6;
    for (int it = 0;; ++it) {
        float T[12];
        if (it > 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_so.T[i])));   // (uniform: scalar registers)
        }
        unsigned long long S[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0ull;
        for (int pss = wave; pss < npass; pss += NW) {
            const int myq = (pss << 6) + lane;
            const bool act = myq < n;
            const float4 p = pts[act ? myq : n - 1];
            float px = p.x, py = p.y, pz = p.z;
            if (it > 0) {   // X <- T*X, written back by the lane that owns the point
                xform(T, p.x, p.y, p.z, px, py, pz);
                if (act) pts[myq] = make_float4(px, py, pz, p.w);
            }
            const LatHit h = (nf3 ? lat_nearest<true, 3>(s_tab, s_face, F, px, py, pz) : lat_nearest<true, LAT_MAX_FACES>(s_tab, s_face, F, px, py, pz));
            // float -> fixed point (rule C4) in four instructions per term (fixq_fast, common.hpp), valid while every term stays
            // below 2^50 / 2^shift; a wave with a lane outside that range takes the general conversion - same integers either way
            const float pv[3] = {px, py, pz}, qv[3] = {h.nx, h.ny, h.nz};
            const float big_c = fmaxf(fmaxf(fmaxf(fabsf(pv[0]), fabsf(pv[1])), fabsf(pv[2])), fmaxf(fmaxf(fabsf(qv[0]), fabsf(qv[1])), fabsf(qv[2])));
            const bool fast = ballot64(act && !(big_c < 256.f && h.d < 16384.f)) == 0ull;
            if (act) {
                if (fast) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        S[a] += fixq_fast(pv[a], FIX_SHIFT);
                        S[3 + a] += fixq_fast(qv[a], FIX_SHIFT);
#pragma unroll
                        for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] += fixq_fast(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
                    }
                    S[15] += fixq_fast(h.d, FIX_SHIFT_D2);
                } else {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        S[a] += (unsigned long long)fixq(pv[a], FIX_SHIFT);
                        S[3 + a] += (unsigned long long)fixq(qv[a], FIX_SHIFT);
#pragma unroll
                        for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] += (unsigned long long)fixq(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
                    }
                    S[15] += (unsigned long long)fixq(h.d, FIX_SHIFT_D2);
                }
            }
        }
        if (wave < npass) wave_fold_to_lds(S, 16, s_acc);
        __syncthreads();
        if (threadIdx.x == 0) {
            s_done = lat_solve(&s_so, s_acc, n, prm);
            for (int i = 0; i < 16; ++i) s_acc[i] = 0ull;
        }
        __syncthreads();
        if (s_done) break;
    }
    // final X <- T*X (PCL transforms before it tests convergence), then getFitnessScore() of Tfinal * original source
    {
        float T[12], Tf[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            T[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_so.T[i])));
            Tf[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_so.Tfinal[i])));
        }
        unsigned long long S[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0ull;
        for (int pss = wave; pss < npass; pss += NW) {
            const int myq = (pss << 6) + lane;
            const bool act = myq < n;
            const float4 p = pts[act ? myq : n - 1];
            float ox, oy, oz;
            xform(T, p.x, p.y, p.z, ox, oy, oz);
            if (act) pts[myq] = make_float4(ox, oy, oz, p.w);
            const float4 p0 = pts0[act ? myq : n - 1];
            float qx, qy, qz;
            xform(Tf, p0.x, p0.y, p0.z, qx, qy, qz);
            const LatHit h = (nf3 ? lat_nearest<false, 3>(s_tab, s_face, F, qx, qy, qz) : lat_nearest<false, LAT_MAX_FACES>(s_tab, s_face, F, qx, qy, qz));
            const bool fast = ballot64(act && !(h.d < 16384.f)) == 0ull;
            if (act) S[0] += fast ? fixq_fast(h.d, FIX_SHIFT_D2) : (unsigned long long)fixq(h.d, FIX_SHIFT_D2);
        }
        if (wave < npass) wave_fold_to_lds(S, 1, s_acc);
        __syncthreads();
        if (threadIdx.x == 0) {
            s_so.done = 1;
            s_so.converged = 1;
            st[2 * (size_t)k] = s_so;
            st[2 * (size_t)k + 1] = s_so;
            accf[k] = s_acc[0];
        }
    }
}

// diagnostic / test entry: nearest template point of arbitrary queries (original index and canonical d2)
__global__ void __launch_bounds__(BLOCK) k_lat_nn(const IcpLattice* __restrict__ L, const float4* __restrict__ q, int n, int* __restrict__ out_idx,
                                                  float* __restrict__ out_d2) {
    __shared__ float4 s_tab[LAT_MAX_TAB];
    __shared__ int4 s_face[2 * LAT_MAX_FACES];
    LatFaces F;
    lat_stage<BLOCK>(L, F, s_tab, s_face);
    for (int base = blockIdx.x * BLOCK; base < n; base += gridDim.x * BLOCK) {   // (whole waves stay together: ballots inside)
        const int i = base + threadIdx.x;
        const float4 p = q[i < n ? i : n - 1];
        const LatHit h = lat_nearest<true, LAT_MAX_FACES>(s_tab, s_face, F, p.x, p.y, p.z);
        const LatHit g = lat_nearest<false, 3>(s_tab, s_face, F, p.x, p.y, p.z);   // (the three-face form: equal d only when nface <= 3)
        if (i < n) { out_idx[i] = lat_index(s_face, F, h); out_d2[i] = (F.nface > 3 || h.d == g.d) ? h.d : __uint_as_float(0x7fc00000u); }
    }
}

void launch_icp_lat(hipStream_t s, int nitems, int threads, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                    const IcpLattice* lats, float4* src, const float4* src0, IcpParams prm) {
    if (nitems <= 0) return;
    if (threads >= 1024) hipLaunchKernelGGL(k_icp_lat<1024>, dim3(nitems), dim3(1024), 0, s, order, cl, st, accf, lats, src, src0, prm);
    else if (threads >= 512) hipLaunchKernelGGL(k_icp_lat<512>, dim3(nitems), dim3(512), 0, s, order, cl, st, accf, lats, src, src0, prm);
    else hipLaunchKernelGGL(k_icp_lat<256>, dim3(nitems), dim3(256), 0, s, order, cl, st, accf, lats, src, src0, prm);
}
void launch_lat_nn(hipStream_t s, const IcpLattice* lat, const float4* q, int n, int* out_idx, float* out_d2) {
    if (n <= 0) return;
    const int g = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_lat_nn, dim3(g < 1024 ? g : 1024), dim3(BLOCK), 0, s, lat, q, n, out_idx, out_d2);
}

}  // namespace cd
