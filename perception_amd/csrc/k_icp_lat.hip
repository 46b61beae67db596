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
// ~215 vector instructions per pass of 64 queries (transform 18, three axes 51, faces 52, tie filter 13, moment terms 67, loop 14),
// no divergence outside the rare tie walk, against ~2400 per pass of the pruned searches - and no 116 KB template image: a slot
// needs 8.7 KB of LDS, so ICP workgroups stop monopolising CUs.
//
// Launch shape: a workgroup keeps one or several clusters going, every cluster's whole ICP inside it (k_icp_lat<CPW, WPC> below).
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
    float ox, oy, oz, ivx, ivy, ivz;   // (ox = -X[0] / step, ivx = 1 / step: IcpLattice::noi, inv)
};

// call from every thread of the workgroup; ends with a barrier
template <int THREADS>
__device__ __forceinline__ void lat_stage(const IcpLattice* __restrict__ L, LatFaces& F, float4* s_tab, int4* s_face) {
    F.nface = L->nface;
#pragma unroll
    for (int f = 0; f < LAT_MAX_FACES; ++f) { F.m0[f] = L->m0[f]; F.m1[f] = L->m1[f]; F.m2[f] = L->m2[f]; F.c[f] = L->c[f]; }
    F.nx = L->n[0]; F.ny = L->n[1]; F.nz = L->n[2]; F.tox = L->toff[0]; F.toy = L->toff[1]; F.toz = L->toff[2];
    F.ox = L->noi[0]; F.oy = L->noi[1]; F.oz = L->noi[2]; F.ivx = L->inv[0]; F.ivy = L->inv[1]; F.ivz = L->inv[2];
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
// i = 0; 0 = not known: the exact test decides), i = ig + di with di in {-1, 0, +1} (left as two flags: only the rare tie walk
// needs the index).  Every axis has a table (one that no face varies along has the single entry 0).
// The window's centre ig only has to be within one entry of the nearest one - all three of the window are evaluated exactly -
// so it is taken from ONE fused multiply-add and a conversion that rounds half up (v_cvt_rpi_i32_f32): q * inv - o * inv sits
// within 1e-4 entries of (q - o) * inv at these magnitudes, against the 7/16 of an entry the uniformity check leaves to spare.
__device__ __forceinline__ void lat_axis(const float4* s_tab, int toff, int n, float o_inv_neg, float inv, float q, float& f, int& ig_out, bool& lo_out,
                                         bool& hi_out, float& t, float& gap) {
    int ig;
    {
        const float pos = __builtin_fmaf(q, inv, o_inv_neg);   // (an approximate position: not one of the canonical operations)
        asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(ig) : "v"(pos));   // floor(pos + 0.5), saturating; NaN -> 0
    }
    ig = min(max(ig, 0), n - 1);   // (one v_med3_i32)
    const float4 W = s_tab[toff + ig];
    const float d0 = __fsub_rn(q, W.x), d1 = __fsub_rn(q, W.y), d2 = __fsub_rn(q, W.z);
    const float f0 = __fmul_rn(d0, d0), f1 = __fmul_rn(d1, d1), f2 = __fmul_rn(d2, d2);
    const float g_mid = __fsub_rn(f0, f1), g_hi = __fsub_rn(f1, f2);
    // (f is unimodal along the table, so f1 > f0 and f1 > f2 cannot both hold: at most one of lo / hi)
    const bool lo = f0 < f1;
    const bool hi = f2 < f1;
    f = f1; t = W.y; gap = g_mid;
    f = lo ? f0 : f; t = lo ? W.x : t; gap = lo ? 0.f : gap;
    f = hi ? f2 : f; t = hi ? W.z : t; gap = hi ? g_hi : gap;
    ig_out = ig; lo_out = lo; hi_out = hi;
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
    int cx, cy, cz;
    bool lox, hix, loy, hiy, loz, hiz;
    LatHit h;
    lat_axis(s_tab, F.tox, F.nx, F.ox, F.ivx, qx, fx, cx, lox, hix, tx, gx);
    lat_axis(s_tab, F.toy, F.ny, F.oy, F.ivy, qy, fy, cy, loy, hiy, ty, gy);
    lat_axis(s_tab, F.toz, F.nz, F.oz, F.ivz, qz, fz, cz, loz, hiz, tz, gz);
    h.ix = h.iy = h.iz = 0;
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
            h.ix = cx + (hix ? 1 : 0) - (lox ? 1 : 0); h.iy = cy + (hiy ? 1 : 0) - (loy ? 1 : 0); h.iz = cz + (hiz ? 1 : 0) - (loz ? 1 : 0);
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

// ---------------------------------------------------------------------------------------------------------------------------
// k_icp_lat<CPW, WPC>: a workgroup keeps CPW clusters ("slots") going, WPC waves each, and works in ROUNDS of two phases:
//   work  : the waves of a slot take the passes (64 points each) of the slot's current step - a PCL iteration (X <- T X in
//           place, nearest neighbours, the 16 fixed-point moment sums of rule C4 kept in registers and folded into the slot's
//           LDS accumulators) or, once converged, the final transform + getFitnessScore() pass;            -- barrier --
//   solve : lane s of wave 0 finishes slot s's step: Umeyama + SVD + convergence tests (lat_solve: the same code as
//           k_icp_solve), or the write-back of a finished cluster and the refill of the slot from the launch's cluster queue.
//           The solves of a workgroup's slots run SIDE BY SIDE in the lanes of one wave.                     -- barrier --
// Why this shape (measured, tools/lat_threads_serial.sh): a step of a 1 400-point cluster is ~19 us of passes for one wave
// and ~9.6 us of single-lane solve (3 300 dependent instructions: IEEE divisions and square roots of the Jacobi sweeps).
// One cluster per workgroup of four waves spends two thirds of its life with three waves parked behind one lane; CPW
// clusters per workgroup pay that solve once per round for all of them.  <1, 4> / <1, 16> (one cluster, many waves) remain the
// latency shapes: a launch that has the GPU to itself, a single frame.
// Each slot has its own copy of its template's tables (a launch may mix templates: BASELINE config 5), restaged when a refill
// brings a cluster of another template.
// ---------------------------------------------------------------------------------------------------------------------------
#ifdef CD_LAT_TIMERS
// phase times of k_icp_lat, summed over workgroups (thread 0's clock64 cycles): [0] work phase, [1] wait at the barrier after it,
// [2] solve phase, [3] wait at the barrier after it, [4] rounds, [5] staging
__device__ unsigned long long g_lat_stats[8];
extern "C" int cd_debug_lat_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lat_stats), sizeof(g_lat_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lat_stats), z, sizeof(z)); }
    return 0;
}
#define LAT_T(k) { const long long tn_ = clock64(); tph[k] += tn_ - tlast; tlast = tn_; }
#else
#define LAT_T(k)
#endif

// Rule C4's fixed-point term rint(v * 2^SHIFT) WITHOUT taking it out of the double: u = (double)v + 1.5 * 2^(52 - SHIFT) is ONE
// rounded addition (v is exact as a double; nearest even at 2^-SHIFT, the spacing of doubles in [2^(52-SHIFT), 2^(53-SHIFT))),
// and for |v * 2^SHIFT| < 2^50 the BITS of u, read as an integer, are C + rint(v * 2^SHIFT) with C = the bits of the constant.
// So a lane adds the raw bits to its 64-bit accumulator - cvt, one v_add_f64 with the constant as a scalar operand, one 64-bit
// add - and since every POINT adds exactly one term to each sum, the solver takes n x C off each sum again (mod 2^64, like the
// sums themselves).  (Until the end of round 5 this was fma(v, 2^SHIFT, 1.5 * 2^52): the same integers, but the compiler
// rebuilt the addend in a register pair before every fma - two moves per term.)
template <int SHIFT>
struct LatFix {
    static constexpr unsigned long long C = ((unsigned long long)(1023 + 52 - SHIFT) << 52) | (1ull << 51);   // bits of 1.5 * 2^(52 - SHIFT)
    static __device__ __forceinline__ unsigned long long bits(float v) {
        return (unsigned long long)__double_as_longlong(__dadd_rn((double)v, __longlong_as_double((long long)C)));
    }
};
static_assert(LatFix<32>::C == 0x4138000000000000ull && LatFix<36>::C == 0x40f8000000000000ull, "constants of lat_fix");
static_assert(FIX_SHIFT == 32 && FIX_SHIFT_D2 == 36, "k_icp_lat's moment sums are written for these scales");

struct LatSlot {
    int k, src_off, n, tslot;   // the cluster (index, first point, points, template slot)
    int phase;                  // LAT_ITER / LAT_FIT / LAT_EMPTY
    int gen;                    // bumped by every refill: the slot's waves restage tables and face descriptors
    int ready;                  // the work phase of this round added a step's sums
    int pad;
};
enum { LAT_ITER = 0, LAT_FIT = 1, LAT_EMPTY = 2 };

#ifndef CD_LAT_WAVES_PER_EU
#define CD_LAT_WAVES_PER_EU 4   // four waves per SIMD: 128 registers - the single-lane solve is what needs them
#endif
template <int CPW, int WPC>
__global__ void __launch_bounds__(CPW * WPC * WAVE, CD_LAT_WAVES_PER_EU)
k_icp_lat(int nitems, const int* __restrict__ order, const IcpCluster* __restrict__ cl, IcpState* __restrict__ st,
          unsigned long long* __restrict__ accf, const IcpLattice* __restrict__ lats, float4* __restrict__ src,
          const float4* __restrict__ src0, int* __restrict__ queue, unsigned long long* __restrict__ busy,
          unsigned long long* __restrict__ busy_out, IcpParams prm) {
    // (order / cl / st / accf may be the host's pinned arrays: every access to them is one cluster's record at a refill or at
    // the end of its ICP.  busy_out != nullptr: the last workgroup to finish writes the launch's wave-time there - queue[1]
    // counts the finished workgroups)
    __shared__ float4 s_tab[CPW][LAT_MAX_TAB];
    __shared__ int4 s_face[CPW][2 * LAT_MAX_FACES];
    __shared__ unsigned long long s_acc[CPW][16];
    __shared__ IcpState s_so[CPW];
    __shared__ LatSlot s_slot[CPW];
    __shared__ int s_flags[2];   // [0] every slot is empty, [1] some slot was refilled in this round
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = wave / WPC, sub = wave % WPC;
    const long long wg_t0 = wall_clock64();   // (100 MHz: the workgroup's lifetime goes to accf[nitems_all], see the end)
    // refill of slot s by one lane: the next live cluster of the queue (largest first), or nothing left
    auto refill = [&](int s) {
        LatSlot& sl = s_slot[s];
        for (int i = 0; i < 16; ++i) s_acc[s][i] = 0ull;
        sl.ready = 0;
        for (;;) {
            const int item = atomicAdd(queue, 1);
            if (item >= nitems) { sl.phase = LAT_EMPTY; return; }
            const int k = order[item];
            if (st[2 * (size_t)k].done) continue;   // host pre-marked (too few points / no template)
            const IcpCluster c = cl[k];
            sl.k = k; sl.src_off = c.src_off; sl.n = c.n; sl.tslot = c.slot;
            s_so[s] = st[2 * (size_t)k];
            sl.phase = LAT_ITER;
            sl.gen += 1;
            return;
        }
    };
    if (threadIdx.x < CPW) { s_slot[threadIdx.x].gen = 0; refill(threadIdx.x); }
    if (threadIdx.x == 0) { s_flags[0] = 0; s_flags[1] = 1; }
    __syncthreads();
    int my_gen = 0;
#ifdef CD_LAT_TIMERS
    long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
    for (;;) {
        if (s_flags[1]) {   // (uniform) some slot has a new cluster: its waves stage the tables of the cluster's template
            const LatSlot sl = s_slot[slot];
            const bool mine = sl.phase != LAT_EMPTY && sl.gen != my_gen;
            if (mine) {
                const IcpLattice* __restrict__ L = lats + sl.tslot;
                for (int i = sub * WAVE + lane; i < L->ntab; i += WPC * WAVE) s_tab[slot][i] = L->tab[i];
                if (sub == 0 && lane < LAT_MAX_FACES) {
                    const int f = lane, w = L->w[f], u = L->fast[f], v = 3 - w - u;
                    s_face[slot][f] = make_int4(w, u, L->base[f], __float_as_int(L->c[f]));
                    s_face[slot][LAT_MAX_FACES + f] = make_int4(L->toff[u], L->toff[v < 0 || v > 2 ? 0 : v], L->n[u], 0);
                }
                my_gen = sl.gen;
            }
            __syncthreads();
        }
        LAT_T(5)
        // ---- work phase ----
        {
            // (the slot's words as scalars; the face words of its template are re-read every round - a few scalar loads that hit
            // the constant cache - rather than carried across the solve phase in forty scalar registers)
            const int phase = __builtin_amdgcn_readfirstlane(s_slot[slot].phase), n = __builtin_amdgcn_readfirstlane(s_slot[slot].n);
            const int src_off = __builtin_amdgcn_readfirstlane(s_slot[slot].src_off), tslot = __builtin_amdgcn_readfirstlane(s_slot[slot].tslot);
            const int npass = (n + 63) >> 6;
            float4* pts = src + src_off;
            const float4* tab = s_tab[slot];
            const int4* face = s_face[slot];
            LatFaces F;
            {
                const IcpLattice* __restrict__ L = lats + (phase == LAT_EMPTY ? 0 : tslot);
                F.nface = L->nface;
#pragma unroll
                for (int f = 0; f < LAT_MAX_FACES; ++f) { F.m0[f] = L->m0[f]; F.m1[f] = L->m1[f]; F.m2[f] = L->m2[f]; F.c[f] = L->c[f]; }
                F.nx = L->n[0]; F.ny = L->n[1]; F.nz = L->n[2]; F.tox = L->toff[0]; F.toy = L->toff[1]; F.toz = L->toff[2];
                F.ox = L->noi[0]; F.oy = L->noi[1]; F.oz = L->noi[2]; F.ivx = L->inv[0]; F.ivy = L->inv[1]; F.ivz = L->inv[2];
            }
            const bool nf3 = F.nface <= 3;
            if (phase == LAT_ITER) {
                const bool moved = s_so[slot].iters > 0;   // X <- T*X from the second iteration on
                float T[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) T[i] = s_so[slot].T[i];   // (uniform values in vector registers: as scalars they were spilled and re-read lane by lane in every pass)
                unsigned long long S[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) S[i] = 0ull;
                // (the points of the next pass are requested before the current pass is worked on: one wave per SIMD per cluster has
                // nothing else to cover an L2 round trip with)
                float4 pnext = pts[min((sub << 6) + lane, n - 1)];
                for (int pss = sub; pss < npass; pss += WPC) {
                    const int myq = (pss << 6) + lane;
                    const bool act = myq < n;
                    const float4 p = pnext;
                    pnext = pts[min(((pss + WPC) << 6) + lane, n - 1)];
                    float px = p.x, py = p.y, pz = p.z;
                    if (moved) {   // written back by the lane that owns the point
                        xform(T, p.x, p.y, p.z, px, py, pz);
                        if (act) pts[myq] = make_float4(px, py, pz, p.w);
                    }
                    const LatHit h = nf3 ? lat_nearest<true, 3>(tab, face, F, px, py, pz) : lat_nearest<true, LAT_MAX_FACES>(tab, face, F, px, py, pz);
                    // the 16 moment terms of rule C4, accumulated as raw bits (LatFix above: the solver takes the constants off
                    // again).  Valid while every term stays below 2^50 / 2^shift; a point outside that range (coordinates beyond
                    // 256 m, a neighbour more than 128 m away) takes the general conversion - the same integers either way.  Lanes
                    // without a point add nothing.
                    if (act) {
                        const float pv[3] = {px, py, pz}, qv[3] = {h.nx, h.ny, h.nz};
                        const float big_c = fmaxf(fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz)), fmaxf(fmaxf(fabsf(h.nx), fabsf(h.ny)), fabsf(h.nz)));
                        if (big_c < 256.f && h.d < 16384.f) {
#pragma unroll
                            for (int a = 0; a < 3; ++a) {
                                S[a] += LatFix<FIX_SHIFT>::bits(pv[a]);
                                S[3 + a] += LatFix<FIX_SHIFT>::bits(qv[a]);
#pragma unroll
                                for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] += LatFix<FIX_SHIFT>::bits(__fmul_rn(qv[a], pv[b]));
                            }
                            S[15] += LatFix<FIX_SHIFT_D2>::bits(h.d);
                        } else {
#pragma unroll
                            for (int a = 0; a < 3; ++a) {
                                S[a] += (unsigned long long)fixq(pv[a], FIX_SHIFT) + LatFix<FIX_SHIFT>::C;
                                S[3 + a] += (unsigned long long)fixq(qv[a], FIX_SHIFT) + LatFix<FIX_SHIFT>::C;
#pragma unroll
                                for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] += (unsigned long long)fixq(__fmul_rn(qv[a], pv[b]), FIX_SHIFT) + LatFix<FIX_SHIFT>::C;
                            }
                            S[15] += (unsigned long long)fixq(h.d, FIX_SHIFT_D2) + LatFix<FIX_SHIFT_D2>::C;
                        }
                    }
                }
                if (sub < npass) wave_fold_to_lds(S, 16, s_acc[slot]);
                if (sub == 0 && lane == 0) s_slot[slot].ready = 1;
            } else if (phase == LAT_FIT) {
                // final X <- T*X (PCL transforms before it tests convergence), then getFitnessScore() of Tfinal * original source
                const float4* pts0 = src0 + src_off;
                float T[12], Tf[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    T[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_so[slot].T[i])));
                    Tf[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_so[slot].Tfinal[i])));
                }
                unsigned long long S[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) S[i] = 0ull;
                float4 pnext = pts[min((sub << 6) + lane, n - 1)], p0next = pts0[min((sub << 6) + lane, n - 1)];
                for (int pss = sub; pss < npass; pss += WPC) {
                    const int myq = (pss << 6) + lane;
                    const bool act = myq < n;
                    const float4 p = pnext, p0 = p0next;
                    pnext = pts[min(((pss + WPC) << 6) + lane, n - 1)];
                    p0next = pts0[min(((pss + WPC) << 6) + lane, n - 1)];
                    float ox, oy, oz;
                    xform(T, p.x, p.y, p.z, ox, oy, oz);
                    if (act) pts[myq] = make_float4(ox, oy, oz, p.w);
                    float qx, qy, qz;
                    xform(Tf, p0.x, p0.y, p0.z, qx, qy, qz);
                    const LatHit h = nf3 ? lat_nearest<false, 3>(tab, face, F, qx, qy, qz) : lat_nearest<false, LAT_MAX_FACES>(tab, face, F, qx, qy, qz);
                    if (act) S[0] += h.d < 16384.f ? LatFix<FIX_SHIFT_D2>::bits(h.d) : (unsigned long long)fixq(h.d, FIX_SHIFT_D2) + LatFix<FIX_SHIFT_D2>::C;
                }
                if (sub < npass) wave_fold_to_lds(S, 1, s_acc[slot]);
            }
        }
        LAT_T(0)
        __syncthreads();
        LAT_T(1)
        // ---- solve phase: lane s of wave 0 <-> slot s (everything below happens in wave 0, whose LDS operations execute in
        // program order: the flag is cleared before a refill raises it, and thread 0 sees the other lanes' phases) ----
        if (threadIdx.x == 0) s_flags[1] = 0;
        if (threadIdx.x < CPW) {
            const int s = threadIdx.x;
            LatSlot& sl = s_slot[s];
            // (every point added the constant of its scale to each sum it touched: see LatFix)
            const unsigned long long off = (unsigned long long)sl.n * LatFix<FIX_SHIFT>::C, off_d = (unsigned long long)sl.n * LatFix<FIX_SHIFT_D2>::C;
            if (sl.phase == LAT_ITER && sl.ready) {
                for (int i = 0; i < 15; ++i) s_acc[s][i] -= off;
                s_acc[s][15] -= off_d;
                if (lat_solve(&s_so[s], s_acc[s], sl.n, prm)) sl.phase = LAT_FIT;
                for (int i = 0; i < 16; ++i) s_acc[s][i] = 0ull;
                sl.ready = 0;
            } else if (sl.phase == LAT_FIT) {
                s_so[s].done = 1;
                s_so[s].converged = 1;
                st[2 * (size_t)sl.k] = s_so[s];
                st[2 * (size_t)sl.k + 1] = s_so[s];
                accf[sl.k] = s_acc[s][0] - off_d;
                refill(s);
                if (sl.phase != LAT_EMPTY) s_flags[1] = 1;
            }
        }
        if (threadIdx.x == 0) {
            int all_empty = 1;
            for (int s = 0; s < CPW; ++s) all_empty &= s_slot[s].phase == LAT_EMPTY ? 1 : 0;
            s_flags[0] = all_empty;
        }
        LAT_T(2)
        __syncthreads();
        LAT_T(3)
#ifdef CD_LAT_TIMERS
        tph[4] += 1;
#endif
        if (s_flags[0]) break;
    }
#ifdef CD_LAT_TIMERS
    if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) atomicAdd(&g_lat_stats[i], (unsigned long long)tph[i]);
#endif
    // the wave-time this launch cost: lifetime of the workgroup (100 MHz ticks) x its waves, summed over the workgroups
    // (cd_timing.icp_wave_ms: what a batch's ICP holds of the chip's wave slots, the throughput roof with batches in flight)
    if (threadIdx.x == 0 && busy) {
        atomicAdd(busy, (unsigned long long)(wall_clock64() - wg_t0) * (unsigned long long)(CPW * WPC));
        if (busy_out) {
            __threadfence();
            if (atomicAdd(queue + 1, 1) == (int)gridDim.x - 1) {
                __threadfence();
                *busy_out = atomicAdd(busy, 0ull);
            }
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

// shape = clusters per workgroup * 256 + waves per cluster (1 | 2 | 4 | 8 x 1 | 2 | 4 | 8 | 16, at most 16 waves per workgroup)
template <int CPW, int WPC>
static void launch_lat_shape(hipStream_t s, int nitems, int n_wg, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                             const IcpLattice* lats, float4* src, const float4* src0, int* queue, unsigned long long* busy, unsigned long long* busy_out,
                             IcpParams prm) {
    hipLaunchKernelGGL((k_icp_lat<CPW, WPC>), dim3(n_wg), dim3(CPW * WPC * WAVE), 0, s, nitems, order, cl, st, accf, lats, src, src0, queue, busy, busy_out, prm);
}
void launch_icp_lat(hipStream_t s, int nitems, int cpw, int wpc, int n_wg, const int* order, const IcpCluster* cl, IcpState* st,
                    unsigned long long* accf, const IcpLattice* lats, float4* src, const float4* src0, int* queue, unsigned long long* busy,
                    unsigned long long* busy_out, IcpParams prm) {
    if (nitems <= 0 || n_wg <= 0) return;
#define CD_LAT_CASE(C, W) if (cpw == C && wpc == W) return launch_lat_shape<C, W>(s, nitems, n_wg, order, cl, st, accf, lats, src, src0, queue, busy, busy_out, prm);
    CD_LAT_CASE(1, 1) CD_LAT_CASE(1, 2) CD_LAT_CASE(1, 4) CD_LAT_CASE(1, 8) CD_LAT_CASE(1, 16)
    CD_LAT_CASE(2, 1) CD_LAT_CASE(2, 2) CD_LAT_CASE(2, 4)
    CD_LAT_CASE(4, 1) CD_LAT_CASE(4, 2) CD_LAT_CASE(4, 4)
    CD_LAT_CASE(8, 1) CD_LAT_CASE(8, 2)
#undef CD_LAT_CASE
    launch_lat_shape<1, 4>(s, nitems, n_wg, order, cl, st, accf, lats, src, src0, queue, busy, busy_out, prm);
}
void launch_lat_nn(hipStream_t s, const IcpLattice* lat, const float4* q, int n, int* out_idx, float* out_d2) {
    if (n <= 0) return;
    const int g = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_lat_nn, dim3(g < 1024 ? g : 1024), dim3(BLOCK), 0, s, lat, q, n, out_idx, out_d2);
}

}  // namespace cd
