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
#include "icp_solve.hpp"

namespace cd {


// ---------------------------------------------------------------------------------------
// Exact nearest-neighbour search, "transposed": a wave owns ONE query at a time and its 64
// lanes scan 64 template points per step out of LDS (one conflict-free ds_read_b128 per lane).
//
// The template is cut into runs of 64 consecutive points, each with an axis-aligned box
// [lo,hi] (built at cd_set_template).  For a query q the bound
//   e_a = max(lo_a - q_a, q_a - hi_a, 0),  lb = (e_x*e_x + e_y*e_y) + e_z*e_z
// evaluated with the canonical association satisfies lb <= d2(q,p) for EVERY p in the run in
// float arithmetic, because each step (rounded subtraction, square, rounded sum) is monotone
// in |component|.  Lane l keeps the boxes of runs l and l+64 of the staged chunk in registers,
// so ONE ballot per 64 runs yields the wave-uniform set of runs that can still hold a point
// with d2 <= best; every other run is skipped without touching it.  This is pruning, not
// approximation: no candidate that could improve on or tie the current best is dropped.
// Seeding: best enters as nextafter(d2(q, seed)) with the previous iteration's neighbour as
// seed; an ascending strict-< scan from that state still ends at the lowest index achieving
// the global minimum (rule C5) - each lane keeps the first minimum of its own ascending
// subsequence and the wave reduction takes the lexicographic minimum of (d2 bits, index).
// ---------------------------------------------------------------------------------------
constexpr int ICPT_THREADS = 1024;                 // 16 waves, 4 per SIMD
constexpr int ICPT_WAVES = ICPT_THREADS / WAVE;
constexpr int ICPT_TPL_LDS = ICP_TPL_LDS;          // template points resident in LDS per pass (119 runs, 119 KiB)
constexpr int ICPT_IMG = ICPT_TPL_LDS + ICP_SUB;   // + one pad run

#if defined(CD_STATS) || defined(CD_TIMERS)
__device__ unsigned long long g_icp_stats[16];
extern "C" int cd_debug_icp_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_stats), sizeof(g_icp_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_stats), z, sizeof(z)); }
    return 0;
}
#endif

#ifdef CD_DONDBG
// hand-over timeline (tools/probe_handover.py): [1] latest workgroup end, [2] ticks spent waiting, [8 + b] waits begun,
// [24 + b] clusters published, [40 + b] clusters taken in quarter-millisecond bin b of the workgroup's own life (100 MHz ticks)
__device__ unsigned long long g_don_dbg[64];
extern "C" int cd_debug_don(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_don_dbg), sizeof(g_don_dbg)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_don_dbg), z, sizeof(z)); }
    return 0;
}
#define DON_DBG(slot0, t0) atomicAdd(&g_don_dbg[(slot0) + min(15, (int)((wall_clock64() - (t0)) / 25000))], 1ull)
#else
#define DON_DBG(slot0, t0)
#endif

#ifdef CD_STATS
// far queries (seed ball wider than the grid walk's reach) by iteration class (0: it < 3, 1: it < 16, 2: later) and by the
// radius of the seed ball in grid cells (bucket 15: 15 or more)
__device__ unsigned long long g_icp_rhist[48];
extern "C" int cd_debug_icp_rhist(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_rhist), sizeof(g_icp_rhist)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[48] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_rhist), z, sizeof(z)); }
    return 0;
}
#endif

#ifdef CD_ITSTATS
__device__ unsigned long long g_icp_it[32][4];   // per ICP iteration (31 = fitness pass): pass cycles, passes, far queries, patches visited
extern "C" int cd_debug_icp_it(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_it), sizeof(g_icp_it)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[32][4]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_it), z, sizeof(z)); }
    return 0;
}
__device__ int g_icp_cur_it;   // iteration bucket of the pass a wave is in (approximate: last writer wins; only read by the stats)
#endif

#ifdef CD_TIMERS
#define CD_PHASE(n) { const long long tn_ = clock64(); tph[n] += tn_ - tlast; tlast = tn_; }
#else
#define CD_PHASE(n)
#endif

__device__ __forceinline__ float next_up_nonneg(float d) { return __uint_as_float(__float_as_uint(d) + 1u); }
__device__ __forceinline__ float seed_bound(float d0) { return d0 < 3.0e38f ? next_up_nonneg(d0) : __uint_as_float(0x7f800000u); }

// (e_a = max(lo_a - q_a, q_a - hi_a, 0) is |q_a - clamp(q_a, lo_a, hi_a)|: one v_med3_f32 and one subtraction per axis instead of
// two subtractions and a v_max3 - the squares are the same floats, the sign is gone after squaring.  An inverted (empty) box
// gives 0 instead of "never": it is not pruned, which costs a test and no correctness.)
__device__ __forceinline__ float box_lb(const float4& L, const float4& H, float x, float y, float z) {
    const float ex = __fsub_rn(x, __builtin_amdgcn_fmed3f(x, L.x, H.x));
    const float ey = __fsub_rn(y, __builtin_amdgcn_fmed3f(y, L.y, H.y));
    const float ez = __fsub_rn(z, __builtin_amdgcn_fmed3f(z, L.z, H.z));
    return __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
}

// min over the 64 lanes of a non-negative float, by DPP row shifts + row broadcasts (no LDS).
// Non-negative floats order like their bit patterns, so the reduction is an unsigned integer min.
__device__ __forceinline__ float wave_min_f32_nonneg(float v) {
    unsigned x = __float_as_uint(v);
#define CD_DPP_MIN(ctrl, rowmask) x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)x, ctrl, rowmask, 0xf, false));
    CD_DPP_MIN(0x111, 0xf)   // row_shr:1
    CD_DPP_MIN(0x112, 0xf)   // row_shr:2
    CD_DPP_MIN(0x114, 0xf)   // row_shr:4
    CD_DPP_MIN(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row minimum
    CD_DPP_MIN(0x142, 0xa)   // row_bcast:15 into rows 1,3
    CD_DPP_MIN(0x143, 0xc)   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave minimum
#undef CD_DPP_MIN
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)x, 63));
}

// Boxes of runs `lane` and `lane+64` of the staged chunk, kept in registers.
struct RunBoxes { float4 L0, H0, L1, H1; };

// Stage template chunk [c0, c0+cn) into LDS: the chunk, its last run padded to 64 points, plus one
// whole pad run; pad points sit at +inf with original index INT_MAX, so their distance is +inf and
// they can never win (removes every bounds predicate from the search).  Ends with a barrier.
__device__ __forceinline__ void stage_chunk(const float4* __restrict__ tpl, const float4* __restrict__ blo,
                                            const float4* __restrict__ bhi, int c0, int cn, float4* s_tpl, RunBoxes& bx) {
    const int lane = threadIdx.x & 63;
    const int nruns = (cn + ICP_SUB - 1) / ICP_SUB;
    const float inf = __uint_as_float(0x7f800000u);
    __syncthreads();
    for (int k = threadIdx.x; k < (nruns + 1) * ICP_SUB; k += ICPT_THREADS)
        s_tpl[k] = k < cn ? tpl[c0 + k] : make_float4(inf, inf, inf, __int_as_float(0x7fffffff));
    bx.L0 = make_float4(inf, inf, inf, 0.f); bx.H0 = bx.L0; bx.L1 = bx.L0; bx.H1 = bx.L0;
    if (lane < nruns) { bx.L0 = blo[c0 / ICP_SUB + lane]; bx.H0 = bhi[c0 / ICP_SUB + lane]; }
    if (lane + 64 < nruns) { bx.L1 = blo[c0 / ICP_SUB + lane + 64]; bx.H1 = bhi[c0 / ICP_SUB + lane + 64]; }
    __syncthreads();
}

// Per-wave query state: lane k of a wave holds query (wave + 16*k) of the slice.
struct QueryRegs { float px, py, pz, pbest; int pbi, poi; };

// Templates that do not fit LDS are searched chunk by chunk.  With a chunk table (IcpGrid: whole k-d subtrees, i.e. compact
// regions, with their boxes) a workgroup stages a chunk only when its box comes within the running bound of SOME query of
// the workgroup - once the cloud sits on the template a slice of neighbouring queries needs one or two chunks, not all.
// Exact for the same reason as the run boxes: the chunk box's lower bound is below that of every run box inside it, so a
// skipped chunk holds only runs that the search would have pruned.  Returns the wave's queries (lane mask) that need the
// chunk; *any_in_wg says whether the workgroup needs it at all.  Contains two barriers: call from all threads.
__device__ __forceinline__ unsigned long long chunk_needed(const IcpGrid& g, int ci, const QueryRegs& q, int nk, int* s_flag,
                                                           bool* any_in_wg) {
    const int lane = threadIdx.x & 63;
    const float4 L = make_float4(g.chunk_lo[ci][0], g.chunk_lo[ci][1], g.chunk_lo[ci][2], 0.f);
    const float4 H = make_float4(g.chunk_hi[ci][0], g.chunk_hi[ci][1], g.chunk_hi[ci][2], 0.f);
    const unsigned long long m = ballot64(lane < nk && box_lb(L, H, q.px, q.py, q.pz) <= q.pbest);
    if (threadIdx.x == 0) *s_flag = 0;
    __syncthreads();
    if (lane == 0 && m) atomicOr(s_flag, 1);
    __syncthreads();
    *any_in_wg = *s_flag != 0;
    return m;
}

// Batched fetch of the wave's queries + seeds (one round of global loads per wave).
// use_prev: seeds are the previous launch's neighbours (nn[]); coarse: also try the first point of
// every run (worth its m/64 tests while the cloud still moves a lot between launches).
__device__ __forceinline__ void fetch_queries(const float4* __restrict__ tpl, int m, const float4* pts, int nq, bool use_prev,
                                              bool coarse, bool apply_T, const float* Tm, const int* nn, QueryRegs& q, int& nk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int myq = wave + ICPT_WAVES * lane;
    nk = nq > wave ? (nq - wave + ICPT_WAVES - 1) / ICPT_WAVES : 0;
    q.px = q.py = q.pz = q.pbest = 0.f;
    q.pbi = 0; q.poi = 0x7fffffff;
    if (lane < nk) {
        const float4 p = pts[myq];
        q.px = p.x; q.py = p.y; q.pz = p.z;
        if (apply_T) xform(Tm, p.x, p.y, p.z, q.px, q.py, q.pz);
        q.pbest = 3.402823466e38f;
        if (use_prev) {
            q.pbi = nn[myq];
            const float4 q0 = tpl[q.pbi];
            q.pbest = dist2(q.px, q.py, q.pz, q0.x, q0.y, q0.z);
        }
        if (coarse || !use_prev) {
            for (int j = 0; j < m; j += ICP_SUB) {
                const float4 t = tpl[j];
                const float d = dist2(q.px, q.py, q.pz, t.x, t.y, t.z);
                if (d < q.pbest) { q.pbest = d; q.pbi = j; }
            }
        }
        q.pbest = seed_bound(q.pbest);
        q.poi = __float_as_int(tpl[q.pbi].w);
    }
}


__device__ __forceinline__ unsigned long long lanes_below(int nk) { return nk >= 64 ? ~0ull : ((1ull << nk) - 1ull); }

// ---------------------------------------------------------------------------------------
// Exact nearest-neighbour search, lane-per-query over the template's uniform grid.
//
// Once ICP has pulled the cloud onto the template, a query's seed (its neighbour of the previous
// iteration) is millimetres away, and the ball of that radius touches a handful of grid cells.
// The template is stored sorted by cell (IcpGrid, common.hpp), so the cells cx0..cx1 of one
// (cy,cz) row are one contiguous range of stored points: a lane walks the rows of its ball's
// cell box and tests every point in them with the canonical dist2 and the lexicographic
// (d2, original index) update.  Exactness: with bound = nextafter(d2(q, seed)) and
//   rr = sqrt(bound)*(1+2e-6),  r_a = rr + 4e-7*|q_a| + 1e-7   (slack > rounding of q_a -+ r_a),
// every template point p outside the cell box has |q_a - p_a| > rr on some axis a (the cell
// coordinate is the same monotone float function on host and device), hence float
// d2(q,p) >= rr^2*(1-5u) > bound: it can neither beat nor tie the seed.  Every point inside the
// box is tested.  So the result equals the full scan's.  Queries whose ball is wider than
// IcpParams::grid_rc cells (early iterations) are left to the wave-per-query search below.
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ int grid_coord(float v, float o, float inv, int n) {
    const float t = floorf(__fmul_rn(__fsub_rn(v, o), inv));
    return t >= (float)(n - 1) ? n - 1 : (t > 0.f ? (int)t : 0);
}

// q.pbest holds the seed bound on entry; `act` lanes search, the others idle through the loop.
//
// Row pruning: the cell box is the bounding cube of the ball, but the ball itself only reaches a few of its
// rows.  For a row (cy,cz), e_y / e_z = distance from q to the row's slab on that axis, reduced by a slack
// that covers every float rounding of the cell boundaries (so they are lower bounds of |q_y-p_y|, |q_z-p_z|
// for every point p filed in the row).  A point that can still beat or tie the seed has real
// dx^2 <= bound*(1+5u) - (dy^2+dz^2) <= rem := rr2 - e2*(1-1e-6); rem < 0 skips the row, otherwise the x-range
// shrinks to the cells of q_x -+ sqrt(rem).
// The point loop has no divergent branch: a lane without a point to test reads the +inf pad point `pad`, whose key can never
// win.  (A select-only version of the row step was measured too: same time, so the row step keeps its early-outs.)
// SH = bits of the stored position in the low word of a key: 13 for an LDS-resident template (positions and original
// indices < 2^13, "no index" = 0x7fffffff), 16 for a template read from global memory (both < 2^16, "no index" = all ones).
template <int SH>
struct KeyFmt {
    static constexpr unsigned NONE = SH == 13 ? 0x7fffffffu : 0xffffffffu;
    static constexpr unsigned POS_MASK = (1u << SH) - 1u;
};
// PK: the template image's .w already holds the key's low word (original index << SH | stored position): k_icp_pipe re-labels
// its LDS image that way once, which takes one vector instruction per tested point out of both searches
template <int SH = 13, bool PK = false, bool SK = false>
__device__ __forceinline__ void grid_search(const float4* s_tpl, const unsigned short* s_cs, const IcpGrid& g, bool act, float rr,
                                            QueryRegs& q, int pad) {
    // running minimum as (d2 bits : original index): the lexicographic update of rule C5 is then ONE unsigned 64-bit compare
    // (no branch, no tie special case; the seed bound enters with "no index" = INT_MAX, so the seed point itself beats it).
    // SK (k_icp_pipe / k_icp_pipe_big since round 4): the caller hands over the SEED'S OWN KEY - q.pbest = d2(q, seed) exactly,
    // q.poi = the seed's key word ("no index" when there is no seed) - so the minimum starts at the seed and only a strictly
    // better candidate (closer, or as close with a lower original index) moves it: same result, and a query whose neighbour
    // did not change needs no write-back
    unsigned long long lkey = ((unsigned long long)__float_as_uint(q.pbest) << 32) | (unsigned long long)(SK ? (unsigned)q.poi : KeyFmt<SH>::NONE);
    const float slx = __fadd_rn(__fmul_rn(4.0e-7f, __fadd_rn(__fadd_rn(fabsf(q.px), fabsf(g.ox)), __fmul_rn((float)g.nx, g.cell))), 1.0e-7f);
    const float sly = __fadd_rn(__fmul_rn(4.0e-7f, __fadd_rn(__fadd_rn(fabsf(q.py), fabsf(g.oy)), __fmul_rn((float)g.ny, g.cell))), 1.0e-7f);
    const float slz = __fadd_rn(__fmul_rn(4.0e-7f, __fadd_rn(__fadd_rn(fabsf(q.pz), fabsf(g.oz)), __fmul_rn((float)g.nz, g.cell))), 1.0e-7f);
    const float ryy = __fadd_rn(rr, sly), rzz = __fadd_rn(rr, slz);
    const int y0 = grid_coord(__fsub_rn(q.py, ryy), g.oy, g.inv, g.ny), y1 = grid_coord(__fadd_rn(q.py, ryy), g.oy, g.inv, g.ny);
    const int z1 = grid_coord(__fadd_rn(q.pz, rzz), g.oz, g.inv, g.nz);
    int ry = y0, rz = grid_coord(__fsub_rn(q.pz, rzz), g.oz, g.inv, g.nz);
    const float rr2 = __fmul_rn(__fmul_rn(rr, rr), 1.0f + 1.0e-6f);
    const float huge = 3.0e38f;
    // Row advance WITHOUT divergent control flow (round 4): written with nested ifs (if (need) { if (rz > z1) .. else { .. if (rem
    // >= 0) .. } }, rounds 1-3) it compiled to ~35 scalar instructions per trip - exec-mask saves and restores, the live booleans
    // as scalar masks - on a scalar unit that sixteen waves share: 4.31 -> 4.19 ms for the kernel (profiles/r04_salu_diet_ab.txt).
    // Here "needs a row" is (i >= b && rz <= z1) - a lane whose walk is over, or that takes no part, has rz beyond z1 and
    // b = INT_MIN - every lane computes the row's range, and selects decide who takes it: one ballot per trip.
    int i = 0, b = act ? 0 : (-0x7fffffff - 1);
    if (!act) rz = z1 + 1;
    bool more = act;   // (only read by the loop exit below)
    for (;;) {
        for (;;) {
            const bool adv = i >= b && rz <= z1;
            if (!ballot64(adv)) break;
#ifdef CD_STATS
            { const unsigned long long nb_ = ballot64(adv); if ((threadIdx.x & 63) == 0) { atomicAdd(&g_icp_stats[4], 1ull); atomicAdd(&g_icp_stats[5], (unsigned long long)__popcll(nb_)); } }
#endif
            const int cz = min(rz, g.nz - 1);   // (address arithmetic of lanes that are done stays inside the table)
            const float ylo = ry == 0 ? -huge : __fadd_rn(g.oy, __fmul_rn((float)ry, g.cell));
            const float yhi = ry == g.ny - 1 ? huge : __fadd_rn(g.oy, __fmul_rn((float)(ry + 1), g.cell));
            const float zlo = cz == 0 ? -huge : __fadd_rn(g.oz, __fmul_rn((float)cz, g.cell));
            const float zhi = cz == g.nz - 1 ? huge : __fadd_rn(g.oz, __fmul_rn((float)(cz + 1), g.cell));
            const float ey = fmaxf(__fsub_rn(fmaxf(__fsub_rn(ylo, q.py), __fsub_rn(q.py, yhi)), sly), 0.f);
            const float ez = fmaxf(__fsub_rn(fmaxf(__fsub_rn(zlo, q.pz), __fsub_rn(q.pz, zhi)), slz), 0.f);
            const float e2 = __fmul_rn(__fadd_rn(__fmul_rn(ey, ey), __fmul_rn(ez, ez)), 1.0f - 1.0e-6f);
            const float rem = __fsub_rn(rr2, e2);
            const bool hit = adv && rem >= 0.f;
            const float rx = __fadd_rn(__fmul_rn(__builtin_amdgcn_sqrtf(fmaxf(rem, 0.f)), 1.0f + 1.0e-6f), slx);
            const int row = (cz * g.ny + ry) * g.nx;
            const int ni = s_cs[row + grid_coord(__fsub_rn(q.px, rx), g.ox, g.inv, g.nx)];
            const int nb = s_cs[row + grid_coord(__fadd_rn(q.px, rx), g.ox, g.inv, g.nx) + 1];
            i = hit ? ni : i;
            b = hit ? nb : b;
            const bool wrap = ry >= y1;
            rz = (adv && wrap) ? rz + 1 : rz;
            ry = adv ? (wrap ? y0 : ry + 1) : ry;
        }
        more = i < b;
        if (!ballot64(more)) break;
        // B: test the points of the current ranges, two per trip.  No lane is switched off: a lane whose range is used up (or
        // whose walk is over) simply tests the points that follow it - real template points, which can never displace the true
        // neighbour - clamped to the +inf pad point.  The low word of the key is (original index << 13 | stored position), so
        // the winner's position needs no select of its own.
        while (ballot64(i < b)) {
#ifdef CD_STATS
            { const unsigned long long pb_ = ballot64(i < b); if ((threadIdx.x & 63) == 0) { atomicAdd(&g_icp_stats[6], 1ull); atomicAdd(&g_icp_stats[7], (unsigned long long)__popcll(pb_)); } }
#endif
            if constexpr (SH == 16) {
                // template in global memory: a trip costs an L2 round trip, so four points are in flight per trip
                const int i0 = min(i, pad), i1 = min(i + 1, pad), i2 = min(i + 2, pad), i3 = min(i + 3, pad);
                const float4 t = s_tpl[(unsigned)i0];
                const float4 u = s_tpl[(unsigned)i1];
                const float4 v = s_tpl[(unsigned)i2];
                const float4 w = s_tpl[(unsigned)i3];
                const float d = dist2(q.px, q.py, q.pz, t.x, t.y, t.z);
                const float e = dist2(q.px, q.py, q.pz, u.x, u.y, u.z);
                const float f = dist2(q.px, q.py, q.pz, v.x, v.y, v.z);
                const float h = dist2(q.px, q.py, q.pz, w.x, w.y, w.z);
                const unsigned long long kd_ = ((unsigned long long)__float_as_uint(d) << 32) | (((unsigned)__float_as_int(t.w) << SH) | (unsigned)i0);
                const unsigned long long ke_ = ((unsigned long long)__float_as_uint(e) << 32) | (((unsigned)__float_as_int(u.w) << SH) | (unsigned)i1);
                const unsigned long long kf_ = ((unsigned long long)__float_as_uint(f) << 32) | (((unsigned)__float_as_int(v.w) << SH) | (unsigned)i2);
                const unsigned long long kh_ = ((unsigned long long)__float_as_uint(h) << 32) | (((unsigned)__float_as_int(w.w) << SH) | (unsigned)i3);
                lkey = kd_ < lkey ? kd_ : lkey;
                lkey = ke_ < lkey ? ke_ : lkey;
                lkey = kf_ < lkey ? kf_ : lkey;
                lkey = kh_ < lkey ? kh_ : lkey;
                i += 4;
            } else {
            // two points per trip, (i, i + 1), clamped together: one address, the second point 16 bytes on
            const int i0 = min(i, pad - 1), i1 = i0 + 1;
            const float4 t = s_tpl[(unsigned)i0];
            const float4 u = s_tpl[(unsigned)i1];
            // (PK images are stored (x, y, key word, z))
            const float d = dist2(q.px, q.py, q.pz, t.x, t.y, PK ? t.w : t.z);
            const float e = dist2(q.px, q.py, q.pz, u.x, u.y, PK ? u.w : u.z);
            unsigned long long kd_ = ((unsigned long long)__float_as_uint(d) << 32) | (PK ? (unsigned)__float_as_int(t.z) : (((unsigned)__float_as_int(t.w) << SH) | (unsigned)i0));
            unsigned long long ke_ = ((unsigned long long)__float_as_uint(e) << 32) | (PK ? (unsigned)__float_as_int(u.z) : (((unsigned)__float_as_int(u.w) << SH) | (unsigned)i1));
            // (opaque to the optimiser: compare AND select then use the key's own register pair, so the distance is formed in
            // its high half instead of being copied there)
            asm("" : "+v"(kd_));
            asm("" : "+v"(ke_));
            lkey = kd_ < lkey ? kd_ : lkey;
            lkey = ke_ < lkey ? ke_ : lkey;
            i += 2;
            }
        }
    }
    const unsigned lo = (unsigned)(lkey & 0xffffffffull);
    if (act && lo != KeyFmt<SH>::NONE) { q.pbest = __uint_as_float((unsigned)(lkey >> 32)); q.pbi = (int)(lo & KeyFmt<SH>::POS_MASK); q.poi = SK ? (int)lo : (int)(lo >> SH); }
}

// Search the staged chunk for the queries of this wave whose bit is set in `todo` (wave-uniform); updates q in place.
__device__ __forceinline__ void search_chunk(const float4* s_tpl, const RunBoxes& bx, int c0, int cn, QueryRegs& q,
                                             unsigned long long todo) {
    const int lane = threadIdx.x & 63;
#ifdef CD_STATS
    const int nruns = (cn + ICP_SUB - 1) / ICP_SUB;
#endif
    while (todo) {
        const int k = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.px), k));
        const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.py), k));
        const float z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.pz), k));
        const float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.pbest), k));
        // Lanes start from (bound, no index): a lane can only be selected below if it beat `best`, and the
        // seed point itself always does in the chunk that holds it.  Only when `best` is a real distance
        // carried over from a lower chunk (c0 > 0) must ties against the carried neighbour be decided,
        // so only then is its original index needed.
        const int boi = c0 > 0 ? __builtin_amdgcn_readlane(q.poi, k) : 0x7fffffff;
        unsigned long long m0 = ballot64(box_lb(bx.L0, bx.H0, x, y, z) <= best);
        unsigned long long m1 = ballot64(box_lb(bx.L1, bx.H1, x, y, z) <= best);
#ifdef CD_STATS
        if (lane == 0) { atomicAdd(&g_icp_stats[1], (unsigned long long)(__popcll(m0) + __popcll(m1))); atomicAdd(&g_icp_stats[2], 1ull); }
#endif
        float lbest = best;
        int lbi = 0, loi = boi;
        // Visit the surviving runs (1.5 per query on average, so no unrolling/padding).  The template is stored
        // re-tiled into compact 64-point patches, so candidates are NOT met in original-index
        // order: the update is the lexicographic (d2, original index) comparison (rule C5).
#define CD_TAKE(dd, tt, rr)                                                                     \
        {                                                                                       \
            const int oi_ = __float_as_int(tt.w);                                               \
            const bool up_ = (dd < lbest) || (dd == lbest && oi_ < loi);                        \
            lbest = up_ ? dd : lbest; lbi = up_ ? c0 + rr * ICP_SUB + lane : lbi; loi = up_ ? oi_ : loi; \
        }
#define CD_VISIT2(mask, base)                                                                   \
        while (mask) {                                                                          \
            const int r0 = (base) + __ffsll((long long)mask) - 1; mask &= mask - 1;             \
            const float4 t0 = s_tpl[r0 * ICP_SUB + lane];                                       \
            const float d0 = dist2(x, y, z, t0.x, t0.y, t0.z);                                  \
            CD_TAKE(d0, t0, r0)                                                                 \
        }
        CD_VISIT2(m0, 0)
        CD_VISIT2(m1, 64)
#undef CD_VISIT2
#undef CD_TAKE
        // lexicographic (d2, original index) minimum over the wave.  Only lanes that beat the incoming
        // bound can hold it; when exactly one did (the usual case once seeds are tight) it IS the answer
        // and three v_readlane replace the reduction.  Otherwise: min distance by DPP, then the lowest
        // original index among the lanes that hold it.
        const unsigned long long imp = ballot64(lbest < best || (lbest == best && loi < boi));
        if (imp == 0) continue;   // nothing in this chunk beats the carried neighbour (multi-chunk templates only)
        float dmin;
        int rbi, roi;
        if (__popcll(imp) == 1) {
            const int l = __ffsll((long long)imp) - 1;
            dmin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbest), l));
            rbi = __builtin_amdgcn_readlane(lbi, l);
            roi = __builtin_amdgcn_readlane(loi, l);
        } else {
            dmin = wave_min_f32_nonneg(lbest);
            unsigned long long eq = ballot64(lbest == dmin);
            rbi = 0; roi = 0x7fffffff;
            while (eq) {
                const int l = __ffsll((long long)eq) - 1;
                eq &= eq - 1;
                const int oi_ = __builtin_amdgcn_readlane(loi, l);
                const int bi_ = __builtin_amdgcn_readlane(lbi, l);
                if (oi_ <= roi) { roi = oi_; rbi = bi_; }
            }
        }
        if (lane == k) { q.pbest = dmin; q.pbi = rbi; q.poi = roi; }
    }
}

// Wave-per-query search over k-d PATCHES of a cell-sorted template: patch r = the 64 stored positions s_kd[64 r ..], its box
// in bx.  The points stay where the grid walk wants them; this search gets the compact boxes it wants.  Lane l keeps the box
// of patch l of the LEFT half of the k-d root split and of patch l of the RIGHT half (psplit = patches in the left half);
// `need` says per query which halves its bound reaches at all (bit 0 / 1), so most queries test one box per lane, not two.
//
// Cost probes on the bench batch (the search run twice: +3.7 ms of 6.4; only the patch visits twice: +1.5 ms; only the box tests
// twice: +0.4 ms) showed where a far query's time goes: not into the arithmetic but into the control flow around it.  What
// paid: the lexicographic update as ONE 64-bit unsigned compare (the compiler turned "d < best || (d == best && oi < boi)"
// into two nested exec-mask regions with branches per patch), 6.4 -> 6.0 ms.  What did not: two queries interleaved per trip
// (same time: the SGPR pressure of two mask/coordinate sets spills to VGPR lanes).
struct FarQ { float x, y, z, best; unsigned long long m0, m1; unsigned long long lkey, seed; };   // lkey = d2 bits : (original index << 13 | stored position); seed: the query's seed key (wave-uniform)

// m with bit k cleared, in ONE scalar instruction (the compiler's m & (m - 1) is s_add_u32 + s_addc_u32 + s_and_b64).  The
// scalar unit is shared by the sixteen waves of the workgroup, and the far search issues 0.8 scalar instructions per vector
// instruction - three quarters of the kernel's scalar instructions (profiles/r04_valu_calibration.txt):
// every scalar instruction taken out of these loops is worth more than a vector one.
__device__ __forceinline__ unsigned long long clear_bit64(unsigned long long m, int k) {
#ifndef CD_NO_SALU_DIET
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(k));
    return m;
#else
    return m & (m - 1ull);   // (callers pass the lowest set bit)
#endif
}

// The query enters with ITS SEED'S KEY (round 4): q.pbest = d2(q, seed) exactly, q.poi = the seed's key word.  The box
// test `lb <= d2(q, seed)` keeps every patch that can hold a point as close as the seed (closer, or equally close with a lower
// original index); the lanes' running minima start at the seed key, so "some lane's key is below the seed key" says exactly
// "the neighbour changes" - and when it does not (the usual case once ICP has settled) the query is done after the visits:
// no reduction, no write-back.  Both halves' boxes are always tested: the per-query flag that skipped an unreachable half cost
// a readlane, two scalar bit tests and two branches to save eleven vector instructions - on a scalar unit that sixteen waves share.
__device__ __forceinline__ void far_begin(FarQ& f, const RunBoxes& bx, const QueryRegs& q, int k) {
    f.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.px), k));
    f.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.py), k));
    f.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.pz), k));
    f.best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.pbest), k));
    const unsigned kw = (unsigned)__builtin_amdgcn_readlane(q.poi, k);
    f.m0 = ballot64(box_lb(bx.L0, bx.H0, f.x, f.y, f.z) <= f.best);
    f.m1 = ballot64(box_lb(bx.L1, bx.H1, f.x, f.y, f.z) <= f.best);
    f.lkey = ((unsigned long long)__float_as_uint(f.best) << 32) | (unsigned long long)kw;
    f.seed = f.lkey;
}
__device__ __forceinline__ int far_next_patch(FarQ& f, int psplit) {   // wave-uniform; f.m0 | f.m1 != 0
    int r;
    if (f.m0) { r = __ffsll((long long)f.m0) - 1; f.m0 &= f.m0 - 1; }
    else { r = psplit + __ffsll((long long)f.m1) - 1; f.m1 &= f.m1 - 1; }
    return r;
}
template <bool PK = false>
__device__ __forceinline__ void far_take(FarQ& f, const float4& t, int pos) {
    // lexicographic (d2, original index) minimum, rule C5, as ONE unsigned 64-bit compare: squared distances are non-negative
    // floats (or +inf / NaN-free here), which order like their bit patterns; no branch, no tie special case
    // The low word carries the stored position below the original index (both < 2^13 for an LDS-resident template; the
    // position is a function of the index, so the order is still (d2, original index)): the key alone is the whole answer.
    const float d = dist2(f.x, f.y, f.z, t.x, t.y, PK ? t.w : t.z);   // (PK images are stored (x, y, key word, z))
    unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (PK ? (unsigned)__float_as_int(t.z) : (((unsigned)__float_as_int(t.w) << 13) | (unsigned)pos));
    asm("" : "+v"(key));   // (as in grid_search: keeps the key in the loaded quad's own registers)
    f.lkey = key < f.lkey ? key : f.lkey;
}
// Minimum over the wave WITHOUT a reduction: the query's owner lane puts the bound key into the wave's LDS word, every lane
// that beat the bound takes an LDS 64-bit atomic min on it (same address: the LDS serialises them, one per clock), and the
// word is read back - by all lanes, broadcast - and unpacked.  LDS operations of one wave execute in program order, so no
// fence is needed.  This replaced ballot + popcount + (DPP minimum + tie loop | three v_readlane) + a masked write-back,
// which cost more than the box tests (probe: the tail run twice = +1.05 ms of 5.9).
__device__ __forceinline__ void far_end(const FarQ& f, QueryRegs& q, int k, unsigned long long* slot) {
    const int lane = threadIdx.x & 63;
    const unsigned long long bound = f.seed;
    // lanes whose key beats the seed's.  None (the usual case once ICP has settled: the neighbour is still the seed): the
    // query keeps it, nothing to do.  Exactly one: that key IS the minimum, fetched with two readlanes.  Several: LDS 64-bit
    // min over them.
    const unsigned long long imp = ballot64(f.lkey < bound);
    if (imp == 0ull) return;
    unsigned long long res;
#ifndef CD_NO_SALU_DIET
    if (__popcll(imp) == 1) {
#else
    if ((imp & (imp - 1ull)) == 0ull) {
#endif
        const int src = __ffsll((long long)imp) - 1;
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(f.lkey >> 32), src);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)f.lkey, src);
        res = ((unsigned long long)hi << 32) | lo;
    } else {
        if (lane == k) __hip_atomic_store(slot, bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (f.lkey < bound) __hip_atomic_fetch_min(slot, f.lkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        res = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    const bool mine = lane == k;
    const unsigned lo = (unsigned)res;
    q.pbest = mine ? __uint_as_float((unsigned)(res >> 32)) : q.pbest;
    q.pbi = mine ? (int)(lo & 0x1fffu) : q.pbi;
    q.poi = mine ? (int)lo : q.poi;
}
template <bool PK = false>
__device__ __forceinline__ void search_patches(const float4* s_tpl, const unsigned short* s_kd, const RunBoxes& bx, int cn, QueryRegs& q,
                                               unsigned long long todo, int psplit, unsigned long long* slot, int* stat_acc = nullptr) {
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int k = __ffsll((long long)todo) - 1;
        todo = clear_bit64(todo, k);
        FarQ A;
        far_begin(A, bx, q, k);
#ifdef CD_STATS
        if (lane == 0) { atomicAdd(&g_icp_stats[1], (unsigned long long)(__popcll(A.m0) + __popcll(A.m1))); atomicAdd(&g_icp_stats[2], 1ull); }
#endif
#ifdef CD_ITSTATS
        if (stat_acc) { stat_acc[0] += 1; stat_acc[1] += __popcll(A.m0) + __popcll(A.m1); }
#endif
#ifndef CD_NO_SALU_DIET
        // the two halves' masks one after the other: no per-patch choice between them (three scalar instructions and a branch)
        while (A.m0) {
            const int r = __ffsll((long long)A.m0) - 1;
            A.m0 = clear_bit64(A.m0, r);
            const int pos = s_kd[r * ICP_SUB + lane];
            far_take<PK>(A, s_tpl[pos], pos);
        }
        const unsigned short* s_kd1 = s_kd + psplit * ICP_SUB;   // (the right half's table: no per-patch scalar add)
        while (A.m1) {
            const int r = __ffsll((long long)A.m1) - 1;
            A.m1 = clear_bit64(A.m1, r);
            const int pos = s_kd1[r * ICP_SUB + lane];
            far_take<PK>(A, s_tpl[pos], pos);
        }
#else
        while (A.m0 | A.m1) {
            const int r = far_next_patch(A, psplit);
            const int pos = s_kd[r * ICP_SUB + lane];
            far_take<PK>(A, s_tpl[pos], pos);
        }
#endif
        far_end(A, q, k, slot);
    }
}


// ---------------------------------------------------------------------------------------
// Wave-per-query search over a template that does NOT fit LDS (more than ICP_TPL_LDS points; up to 65535): the points stay in
// global memory (a few hundred KiB: L2-resident) in k-d patch order, so a patch is ONE coalesced 1 KiB read; the patch boxes
// (32 B each) are in LDS, and a second level above them - the boxes of k-d subtrees of at most 64 patches ("superpatches",
// IcpSuper; lane l keeps superpatch l's box in registers) - tells with one ballot which runs of <= 64 patch boxes need
// testing at all.  Same exactness argument as the run boxes: a box bound is a lower bound of the canonical float distance
// to everything inside it, and a superpatch box contains its patch boxes.
// ---------------------------------------------------------------------------------------
struct SuperRegs { float4 L, H; int first, cnt; };   // lane l: box, first patch and patch count of superpatch l (+inf box: none)

__device__ __forceinline__ void search_patches_big(const float4* __restrict__ tplk, const unsigned short* __restrict__ kdmap,
                                                   const float4* s_plo, const float4* s_phi, const SuperRegs& sp, QueryRegs& q,
                                                   unsigned long long todo, unsigned long long* slot) {
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int k = __ffsll((long long)todo) - 1;
        todo = clear_bit64(todo, k);
        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.px), k));
        const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.py), k));
        const float z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.pz), k));
        const float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.pbest), k));
        // the query's seed key, as in far_begin: d2(q, seed) exactly and the seed's key word - a query whose neighbour does not
        // change is done after its visits
        const unsigned long long bound = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane(q.poi, k);
        unsigned long long lkey = bound;
        unsigned long long ma = ballot64(box_lb(sp.L, sp.H, x, y, z) <= best);
#ifdef CD_STATS
        if (lane == 0) atomicAdd(&g_icp_stats[2], 1ull);
#endif
        while (ma) {
            const int sidx = __ffsll((long long)ma) - 1;
            ma = clear_bit64(ma, sidx);
            const int first = __builtin_amdgcn_readlane(sp.first, sidx), cnt = __builtin_amdgcn_readlane(sp.cnt, sidx);
            const int pl = first + min(lane, cnt - 1);
            unsigned long long mp = ballot64(lane < cnt && box_lb(s_plo[pl], s_phi[pl], x, y, z) <= best);
#ifdef CD_STATS
            if (lane == 0) atomicAdd(&g_icp_stats[1], (unsigned long long)__popcll(mp));
#endif
            while (mp) {   // two patches per trip: their loads are in flight together (an odd one out is read twice)
                const int b0 = __ffsll((long long)mp) - 1;
                mp = clear_bit64(mp, b0);
                const int b1 = mp ? __ffsll((long long)mp) - 1 : b0;
                mp = clear_bit64(mp, b1);          // (clearing a bit that is already clear changes nothing)
                const int r0 = first + b0, r1 = first + b1;
                const float4 t = tplk[(unsigned)(r0 * ICP_SUB + lane)];
                const float4 u = tplk[(unsigned)(r1 * ICP_SUB + lane)];
                const unsigned pt = kdmap[(unsigned)(r0 * ICP_SUB + lane)];
                const unsigned pu = kdmap[(unsigned)(r1 * ICP_SUB + lane)];
                const float d = dist2(x, y, z, t.x, t.y, t.z);
                const float e = dist2(x, y, z, u.x, u.y, u.z);
                const unsigned long long kd_ = ((unsigned long long)__float_as_uint(d) << 32) | (((unsigned)__float_as_int(t.w) << 16) | pt);
                const unsigned long long ke_ = ((unsigned long long)__float_as_uint(e) << 32) | (((unsigned)__float_as_int(u.w) << 16) | pu);
                lkey = kd_ < lkey ? kd_ : lkey;
                lkey = ke_ < lkey ? ke_ : lkey;
            }
        }
        // minimum over the wave: as far_end (none - the neighbour stays - / exactly one / several lanes beat the seed key)
        const unsigned long long imp = ballot64(lkey < bound);
        if (imp == 0ull) continue;
        unsigned long long res;
        if (__popcll(imp) == 1) {
            const int src = __ffsll((long long)imp) - 1;
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(lkey >> 32), src);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)lkey, src);
            res = ((unsigned long long)hi << 32) | lo;
        } else {
            if (lane == k) __hip_atomic_store(slot, bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (lkey < bound) __hip_atomic_fetch_min(slot, lkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            res = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        const bool mine = lane == k;
        const unsigned lo = (unsigned)res;
        q.pbest = mine ? __uint_as_float((unsigned)(res >> 32)) : q.pbest;
        q.pbi = mine ? (int)(lo & 0xffffu) : q.pbi;
        q.poi = mine ? (int)lo : q.poi;
    }
}


__device__ __forceinline__ void store_queries(const QueryRegs& q, int nk, int* nn, float* d2buf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < nk) {
        nn[wave + ICPT_WAVES * lane] = q.pbi;
        d2buf[wave + ICPT_WAVES * lane] = q.pbest;
    }
}

// One thread per cluster: TransformationEstimationSVD + final_transformation_ update +
// DefaultConvergenceCriteria for launch `it`, from the moments launch it-1 accumulated.
// Writes the state slot k_icp_iter(it) consumes, keeps the `done` flag in both parity slots and
// clears the moment buffer launch it+1 will accumulate into.
__global__ void __launch_bounds__(WAVE) k_icp_solve(int it, int ncl, const IcpCluster* __restrict__ cl,
                                                    IcpState* __restrict__ st, unsigned long long* __restrict__ acc,
                                                    int* __restrict__ queue, IcpParams prm) {
    const int k = blockIdx.x * WAVE + threadIdx.x;
    if (k == 0) *queue = 0;   // work queue of the k_icp_iter launch that follows
    if (k >= ncl) return;
    const IcpState* sin = st + (size_t)k * 2 + (it & 1);
    IcpState* sout = st + (size_t)k * 2 + ((it + 1) & 1);
    if (sin->done) { *sout = *sin; return; }
    IcpState so = *sin;
    if (it > 0) {
        const IcpCluster c = cl[k];
        const unsigned long long* A = acc + ((size_t)k * 3 + (it - 1) % 3) * 16;
        float T[16];
        umeyama_from_moments(A, c.n, T);
        // final_transformation_ = transformation_ * final_transformation_
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                so.Tfinal[4 * i + j] = ((T[4 * i] * sin->Tfinal[j] + T[4 * i + 1] * sin->Tfinal[4 + j]) +
                                        T[4 * i + 2] * sin->Tfinal[8 + j]) + T[4 * i + 3] * sin->Tfinal[12 + j];
        so.iters = sin->iters + 1;
        int done = 0;
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
        for (int i = 0; i < 16; ++i) so.T[i] = T[i];
    }
    *sout = so;
    unsigned long long* Z = acc + ((size_t)k * 3 + (it + 1) % 3) * 16;
    for (int i = 0; i < 16; ++i) Z[i] = 0ull;
}

// Persistent: one workgroup per CU.  The template image and this lane's run boxes stay resident
// while the workgroup pulls (cluster, slice) work items from an atomic queue (zeroed by k_icp_solve).
__global__ void __launch_bounds__(ICPT_THREADS) k_icp_iter(int it, int n_work, const IcpWork* __restrict__ work,
                                                           const IcpCluster* __restrict__ cl, const IcpState* __restrict__ st,
                                                           unsigned long long* __restrict__ acc,
                                                           const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                                           const float4* __restrict__ thi, const IcpGrid* __restrict__ grids,
                                                           float4* src, int* nn, float* d2buf, int qslice, int* queue) {
    __shared__ float4 s_tpl[ICPT_IMG];
    __shared__ unsigned long long s_scr[8 * ICP_QSLICE];   // moment scratch, 8 terms at a time (32 KiB)
    __shared__ int s_item[3];   // [0], [1]: work items (double-buffered), [2]: chunk_needed flag
    const int lane = threadIdx.x & 63;
    RunBoxes bx;
    int staged = -1;   // template offset whose (single-chunk) image is resident in LDS
    if (threadIdx.x == 0) s_item[0] = atomicAdd(queue, 1);
    for (int ph = 0;; ph ^= 1) {
        __syncthreads();                       // one barrier per item: publishes s_item[ph], retires the previous item
        const int item = s_item[ph];
        if (item >= n_work) break;
        if (threadIdx.x == 0) s_item[ph ^ 1] = atomicAdd(queue, 1);   // pop the next item while this one is processed
        const IcpWork wk = work[item];
        const IcpCluster c = cl[wk.cluster];
        const IcpState* sin = st + (size_t)wk.cluster * 2 + (it & 1);          // state before this launch
        const IcpState* snow = st + (size_t)wk.cluster * 2 + ((it + 1) & 1);   // written by k_icp_solve(it)
        if (sin->done) continue;
        const int done_now = snow->done;
        const int q0 = wk.tile * qslice;
        const int nq = min(qslice, c.n - q0);
        float4* pts = src + c.src_off + q0;
        int* nnq = nn + c.src_off + q0;
        float* d2q = d2buf + c.src_off + q0;
        if (it > 0) {   // X <- T * X, in place
            float T[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) T[k] = snow->T[k];
            for (int i = threadIdx.x; i < nq; i += ICPT_THREADS) {
                const float4 p = pts[i];
                float ox, oy, oz;
                xform(T, p.x, p.y, p.z, ox, oy, oz);
                pts[i] = make_float4(ox, oy, oz, p.w);
            }
        }
        if (done_now) continue;
        __syncthreads();   // the transformed points are read by other waves below
        const float4* tp = tpl + c.tpl_off;
        const float4* blo = tlo + c.tpl_off / ICP_SUB;
        const float4* bhi = thi + c.tpl_off / ICP_SUB;
        QueryRegs q;
        int nk;
        fetch_queries(tp, c.tpl_m, pts, nq, it > 0, it < 3, false, nullptr, nnq, q, nk);
        if (c.tpl_m <= ICPT_TPL_LDS) {
            if (staged != c.tpl_off) { stage_chunk(tp, blo, bhi, 0, c.tpl_m, s_tpl, bx); staged = c.tpl_off; }
            search_chunk(s_tpl, bx, 0, c.tpl_m, q, lanes_below(nk));
        } else {
            const IcpGrid& g = grids[c.slot];
            if (g.nchunk > 0) {
                for (int ci = 0; ci < g.nchunk; ++ci) {
                    bool any;
                    const unsigned long long todo = chunk_needed(g, ci, q, nk, &s_item[2], &any);
                    if (!any) continue;
                    stage_chunk(tp, blo, bhi, g.chunk_start[ci], g.chunk_n[ci], s_tpl, bx);
                    search_chunk(s_tpl, bx, g.chunk_start[ci], g.chunk_n[ci], q, todo);
                }
            } else {
                for (int c0 = 0; c0 < c.tpl_m; c0 += ICPT_TPL_LDS) {
                    const int cn = min(ICPT_TPL_LDS, c.tpl_m - c0);
                    stage_chunk(tp, blo, bhi, c0, cn, s_tpl, bx);
                    search_chunk(s_tpl, bx, c0, cn, q, lanes_below(nk));
                }
            }
            staged = -1;
        }
        store_queries(q, nk, nnq, d2q);
        __syncthreads();
        // 16 fixed-point moments of the correspondences of this slice: per-point terms go to LDS
        // (term-major, conflict-free), 8 terms at a time; wave k then sums term k.
        unsigned long long S[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) S[k] = 0ull;
        if (threadIdx.x < nq) {
            const int i = threadIdx.x;
            const float4 p = pts[i];
            const float4 qq = tp[nnq[i]];
            const float pv[3] = {p.x, p.y, p.z}, qv[3] = {qq.x, qq.y, qq.z};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                S[a] = (unsigned long long)fixq(pv[a], FIX_SHIFT);
                S[3 + a] = (unsigned long long)fixq(qv[a], FIX_SHIFT);
#pragma unroll
                for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] = (unsigned long long)fixq(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
            }
            S[15] = (unsigned long long)fixq(d2q[i], FIX_SHIFT_D2);
        }
        static_assert(ICP_QSLICE <= ICPT_THREADS, "one point per thread in the moment pass");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h) __syncthreads();
            if (threadIdx.x < ICP_QSLICE) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s_scr[k * ICP_QSLICE + threadIdx.x] = S[8 * h + k];
            }
            __syncthreads();
            const int k = threadIdx.x >> 6;   // waves 0..7 <-> the 8 moments of this half
            if (k < 8) {
                unsigned long long t = 0ull;
#pragma unroll
                for (int j = 0; j < ICP_QSLICE / 64; ++j) t += s_scr[k * ICP_QSLICE + j * 64 + lane];
                t = wave_sum_u64(t);
                if (lane == 0) atomicAdd(&acc[((size_t)wk.cluster * 3 + it % 3) * 16 + 8 * h + k], t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Persistent form of the sliced driver (few clusters: one frame, small batches).  The multi-launch form above pays two
// dependent kernel launches per PCL iteration (k_icp_solve, k_icp_iter: ~20 us together for one frame, most of it launch
// turnaround); here ONE launch runs every iteration and the fitness pass:
//   * work items (cluster, slice) are assigned statically, item i -> workgroup i % G, so a workgroup meets the same points
//     in every iteration (no other workgroup ever reads them) and stages its template once;
//   * the only data that crosses workgroups are the 16 moment sums of a cluster's correspondences: device-scope atomic adds
//     into acc[cluster][it % 3], read back with device-scope atomic loads after a grid barrier (atomic arrive counter; the
//     barrier inside a workgroup waits for its memory operations first, so no cache-flushing fence is needed);
//   * every workgroup solves (Umeyama + convergence tests) for the clusters of its own items from those sums - a few
//     redundant single-lane solves instead of a second barrier - and keeps their state in LDS; the workgroup that holds
//     slice 0 of a cluster publishes it.
// All G <= n_CU workgroups must be resident together (one per CU): they are, unless other kernels hold the CUs, in which
// case the ones already running wait at the barrier.  Every wait is bounded (2^15 polls, some tens of milliseconds): a
// barrier that does not complete raises the abort flag, every workgroup leaves, and the host runs the multi-launch form instead.
// Same arithmetic as k_icp_solve + k_icp_iter + k_icp_fitness (the moments are order-free integer sums): identical results.
// ---------------------------------------------------------------------------------------
constexpr int PERSIST_ITEMS = 8;   // work items per workgroup at most

// What crosses workgroups in k_icp_persist are agent-scope atomics only (moment adds, zero-stores, the counters below):
// they are performed at the device's coherence point, so no cache write-back is needed - but they must have been
// PERFORMED before this workgroup counts as arrived.  __syncthreads() alone does not give that on gfx950 (the backend
// emits s_waitcnt lgkmcnt(0) + s_barrier: a workgroup-scope fence does not wait for vector-memory operations outside
// tgsplit mode), so every thread first waits for its own outstanding vector-memory operations (gfx9 counts loads, stores
// and atomics without return in vmcnt alike).  The points / neighbour arrays a workgroup writes with plain stores are only
// ever read back by the same workgroup, hence no agent-scope release fence (which would write the L2 back every iteration).
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target, int* abort_flag, int* s_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        for (int spins = 0;; ++spins) {
            if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            if (spins > (1 << 15) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        *s_ok = ok;
    }
    __syncthreads();   // everything after it is issued after thread 0 saw the full count (loads are not hoisted over s_barrier)
    asm volatile("" ::: "memory");
    return *s_ok != 0;
}

// one PCL iteration's state update of a cluster from the moment sums of the previous iteration (what k_icp_solve does)
__device__ void persist_solve(IcpState& so, const unsigned long long* A, int n, const IcpParams& prm) {
    float T[16];
    umeyama_from_moments(A, n, T);
    float Tf[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            Tf[4 * i + j] = ((T[4 * i] * so.Tfinal[j] + T[4 * i + 1] * so.Tfinal[4 + j]) + T[4 * i + 2] * so.Tfinal[8 + j]) +
                            T[4 * i + 3] * so.Tfinal[12 + j];
    for (int i = 0; i < 16; ++i) so.Tfinal[i] = Tf[i];
    so.iters += 1;
    int done = 0;
    if (so.iters >= prm.max_iter) {
        done = 1;
    } else {
        const double cos_angle = 0.5 * (double)(((T[0] + T[5]) + T[10]) - 1.0f);
        const double translation_sqr = (double)((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
        if (cos_angle >= prm.rot_thr && translation_sqr <= prm.trans_eps) {
            done = 1;
        } else {
            const double mse = unfix(A[15], FIX_SHIFT_D2) / (double)n;
            if (fabs(mse - so.prev_mse) < prm.abs_mse) done = 1;
            else if (fabs(mse - so.prev_mse) / so.prev_mse < prm.rel_mse) done = 1;
            so.prev_mse = mse;
        }
    }
    so.done = done;
    so.converged = done;
    for (int i = 0; i < 16; ++i) so.T[i] = T[i];
}

#ifdef CD_PERSISTDBG
// time workgroup 0 spends per phase of an iteration (100 MHz ticks): solve, transform, fetch + search + store, moments, barrier
__device__ unsigned long long g_persist_dbg[8];
extern "C" int cd_debug_persist(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_persist_dbg), sizeof(g_persist_dbg)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_persist_dbg), z, sizeof(z)); }
    return 0;
}
#define PERSIST_PHASE(k) { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); g_persist_dbg[k] += t_ - tdbg_; tdbg_ = t_; } }
#else
#define PERSIST_PHASE(k)
#endif
__global__ void __launch_bounds__(ICPT_THREADS) k_icp_persist(int n_work, int max_it, const IcpWork* __restrict__ work,
                                                              const IcpCluster* __restrict__ cl, IcpState* st,
                                                              unsigned long long* acc, unsigned long long* __restrict__ accf,
                                                              const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                                              const float4* __restrict__ thi, const IcpGrid* __restrict__ grids,
                                                              float4* src, const float4* __restrict__ src0, int* nn, float* d2buf,
                                                              int qslice, unsigned* bar, int* abort_flag, int n_open, int* closed, IcpParams prm) {
    __shared__ float4 s_tpl[ICPT_IMG];
    __shared__ unsigned long long s_scr[8 * ICP_QSLICE];   // moment scratch, 8 terms at a time (32 KiB)
    __shared__ IcpState s_st[PERSIST_ITEMS];               // state of the clusters of this workgroup's items
    __shared__ unsigned long long s_A[16];
    __shared__ int s_flag, s_ok;
    const int lane = threadIdx.x & 63;
    const int G = gridDim.x;
    int n_items = 0;
    for (int i = blockIdx.x; i < n_work; i += G) ++n_items;   // <= PERSIST_ITEMS (host)
    for (int j = threadIdx.x; j < n_items; j += ICPT_THREADS) s_st[j] = st[2 * (size_t)work[blockIdx.x + j * G].cluster];
    __syncthreads();
    RunBoxes bx;
    int staged = -1;
    bool aborted = false;
#ifdef CD_PERSISTDBG
    unsigned long long tdbg_ = wall_clock64();
#endif
    int it = 0;
    int seen0 = 0, seen1 = 0, seen2 = 0, seen3 = 0;   // closed[0..3] as last read
    for (; it < max_it; ++it) {
        for (int j = 0; j < n_items; ++j) {
            const IcpWork wk = work[blockIdx.x + j * G];
            const IcpCluster c = cl[wk.cluster];
            const bool was_done = s_st[j].done != 0;   // uniform: LDS value written before the last barrier
            if (was_done) continue;
            if (it > 0) {   // this iteration's transformation from the previous iteration's correspondences
                if (threadIdx.x < 16)
                    s_A[threadIdx.x] = __hip_atomic_load(acc + ((size_t)wk.cluster * 3 + (it - 1) % 3) * 16 + threadIdx.x, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                if (threadIdx.x == 0) {
                    IcpState so = s_st[j];
                    persist_solve(so, s_A, c.n, prm);
                    s_st[j] = so;
                    // one count per cluster that closes in iteration `it`, into the slot of that iteration (see the exit test)
                    if (wk.tile == 0 && so.done) __hip_atomic_fetch_add(closed + (it & 3), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
            PERSIST_PHASE(0)
            if (wk.tile == 0 && threadIdx.x < 16)   // the sums of iteration it + 1 start from zero (slot last read in iteration it - 1)
                __hip_atomic_store(acc + ((size_t)wk.cluster * 3 + (it + 1) % 3) * 16 + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int done_now = s_st[j].done;
            const int q0 = wk.tile * qslice;
            const int nq = min(qslice, c.n - q0);
            float4* pts = src + c.src_off + q0;
            int* nnq = nn + c.src_off + q0;
            float* d2q = d2buf + c.src_off + q0;
            if (it > 0) {   // X <- T * X, in place
                float T[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) T[k] = s_st[j].T[k];
                for (int i = threadIdx.x; i < nq; i += ICPT_THREADS) {
                    const float4 p = pts[i];
                    float ox, oy, oz;
                    xform(T, p.x, p.y, p.z, ox, oy, oz);
                    pts[i] = make_float4(ox, oy, oz, p.w);
                }
            }
            if (done_now) continue;
            __syncthreads();   // the transformed points are read by other waves below
            PERSIST_PHASE(1)
            const float4* tp = tpl + c.tpl_off;
            const float4* blo = tlo + c.tpl_off / ICP_SUB;
            const float4* bhi = thi + c.tpl_off / ICP_SUB;
            QueryRegs q;
            int nk;
            fetch_queries(tp, c.tpl_m, pts, nq, it > 0, it < 3, false, nullptr, nnq, q, nk);
            if (c.tpl_m <= ICPT_TPL_LDS) {
                if (staged != c.tpl_off) { stage_chunk(tp, blo, bhi, 0, c.tpl_m, s_tpl, bx); staged = c.tpl_off; }
                search_chunk(s_tpl, bx, 0, c.tpl_m, q, lanes_below(nk));
            } else {
                const IcpGrid& g = grids[c.slot];
                if (g.nchunk > 0) {
                    for (int ci = 0; ci < g.nchunk; ++ci) {
                        bool any;
                        const unsigned long long todo = chunk_needed(g, ci, q, nk, &s_flag, &any);
                        if (!any) continue;
                        stage_chunk(tp, blo, bhi, g.chunk_start[ci], g.chunk_n[ci], s_tpl, bx);
                        search_chunk(s_tpl, bx, g.chunk_start[ci], g.chunk_n[ci], q, todo);
                    }
                } else {
                    for (int c0 = 0; c0 < c.tpl_m; c0 += ICPT_TPL_LDS) {
                        const int cn = min(ICPT_TPL_LDS, c.tpl_m - c0);
                        stage_chunk(tp, blo, bhi, c0, cn, s_tpl, bx);
                        search_chunk(s_tpl, bx, c0, cn, q, lanes_below(nk));
                    }
                }
                staged = -1;
            }
            store_queries(q, nk, nnq, d2q);
            __syncthreads();
            PERSIST_PHASE(2)
            unsigned long long S[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) S[k] = 0ull;
            if (threadIdx.x < nq) {
                const int i = threadIdx.x;
                const float4 p = pts[i];
                const float4 qq = tp[nnq[i]];
                const float pv[3] = {p.x, p.y, p.z}, qv[3] = {qq.x, qq.y, qq.z};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    S[a] = (unsigned long long)fixq(pv[a], FIX_SHIFT);
                    S[3 + a] = (unsigned long long)fixq(qv[a], FIX_SHIFT);
#pragma unroll
                    for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] = (unsigned long long)fixq(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
                }
                S[15] = (unsigned long long)fixq(d2q[i], FIX_SHIFT_D2);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h) __syncthreads();
                if (threadIdx.x < ICP_QSLICE) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) s_scr[k * ICP_QSLICE + threadIdx.x] = S[8 * h + k];
                }
                __syncthreads();
                const int k = threadIdx.x >> 6;   // waves 0..7 <-> the 8 moments of this half
                if (k < 8) {
                    unsigned long long t = 0ull;
#pragma unroll
                    for (int jj = 0; jj < ICP_QSLICE / 64; ++jj) t += s_scr[k * ICP_QSLICE + jj * 64 + lane];
                    t = wave_sum_u64(t);
                    if (lane == 0) atomicAdd(&acc[((size_t)wk.cluster * 3 + it % 3) * 16 + 8 * h + k], t);
                }
            }
            __syncthreads();   // s_scr is reused by the next item
            PERSIST_PHASE(3)
        }
        if (!grid_barrier(bar, (unsigned)(it + 1) * (unsigned)G, abort_flag, &s_ok)) { aborted = true; break; }
        PERSIST_PHASE(4)
        // Exit when every cluster has closed - decided from state that is FINAL at this barrier: closed[it & 3] only receives
        // the closes of iterations it, it - 4, ... (a workgroup that runs ahead adds to slot (it + 1) & 3, and slot it & 3 is
        // next written in iteration it + 4, which nobody enters before all have left barrier it + 3), and every workgroup
        // reads slot it & 3 between barrier it and its arrival at barrier it + 1.  All workgroups therefore see the same
        // running total and leave in the same iteration.  (A single shared count of open clusters was read "from the
        // future": a slower workgroup could see the zero a faster one produced one iteration later and leave early.)
        {
            const int v = __hip_atomic_load(closed + (it & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen0 = (it & 3) == 0 ? v : seen0; seen1 = (it & 3) == 1 ? v : seen1;
            seen2 = (it & 3) == 2 ? v : seen2; seen3 = (it & 3) == 3 ? v : seen3;
        }
        if (seen0 + seen1 + seen2 + seen3 >= n_open) { ++it; break; }
    }
    if (aborted) return;
    // fitness pass (what k_icp_fitness does) and publication of the states
    for (int j = 0; j < n_items; ++j) {
        const IcpWork wk = work[blockIdx.x + j * G];
        const IcpCluster c = cl[wk.cluster];
        if (wk.tile == 0 && threadIdx.x == 0) { st[2 * (size_t)wk.cluster] = s_st[j]; st[2 * (size_t)wk.cluster + 1] = s_st[j]; }
        if (s_st[j].status != CD_OK) continue;
        float T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = s_st[j].Tfinal[k];
        const int q0 = wk.tile * qslice;
        const int nq = min(qslice, c.n - q0);
        int* nnq = nn + c.src_off + q0;
        float* d2q = d2buf + c.src_off + q0;
        const float4* tp = tpl + c.tpl_off;
        const float4* blo = tlo + c.tpl_off / ICP_SUB;
        const float4* bhi = thi + c.tpl_off / ICP_SUB;
        __syncthreads();
        QueryRegs q;
        int nk;
        fetch_queries(tp, c.tpl_m, src0 + c.src_off + q0, nq, true, false, true, T, nnq, q, nk);
        if (c.tpl_m <= ICPT_TPL_LDS) {
            if (staged != c.tpl_off) { stage_chunk(tp, blo, bhi, 0, c.tpl_m, s_tpl, bx); staged = c.tpl_off; }
            search_chunk(s_tpl, bx, 0, c.tpl_m, q, lanes_below(nk));
        } else {
            const IcpGrid& g = grids[c.slot];
            if (g.nchunk > 0) {
                for (int ci = 0; ci < g.nchunk; ++ci) {
                    bool any;
                    const unsigned long long todo = chunk_needed(g, ci, q, nk, &s_flag, &any);
                    if (!any) continue;
                    stage_chunk(tp, blo, bhi, g.chunk_start[ci], g.chunk_n[ci], s_tpl, bx);
                    search_chunk(s_tpl, bx, g.chunk_start[ci], g.chunk_n[ci], q, todo);
                }
            } else {
                for (int c0 = 0; c0 < c.tpl_m; c0 += ICPT_TPL_LDS) {
                    const int cn = min(ICPT_TPL_LDS, c.tpl_m - c0);
                    stage_chunk(tp, blo, bhi, c0, cn, s_tpl, bx);
                    search_chunk(s_tpl, bx, c0, cn, q, lanes_below(nk));
                }
            }
            staged = -1;
        }
        store_queries(q, nk, nnq, d2q);
        __syncthreads();
        unsigned long long v = 0ull;
        for (int i = threadIdx.x; i < nq; i += ICPT_THREADS) v += (unsigned long long)fixq(d2q[i], FIX_SHIFT_D2);
        v = wave_sum_u64(v);
        if (lane == 0 && v) atomicAdd(&accf[wk.cluster], v);
    }
}

// ---------------------------------------------------------------------------------------
// Whole-cluster variant (batch mode): ONE persistent workgroup takes a cluster from the queue
// (largest first) and runs its complete ICP - every iteration's transform, search, moments,
// Umeyama/SVD solve and convergence test, then the fitness pass - without leaving the CU.
// Same arithmetic as k_icp_solve + k_icp_iter + k_icp_fitness (the moments are order-free
// integer sums), so the results are bit-identical; what disappears are the ~2 launches per
// iteration, the global moment atomics, the host completion polls and the under-filled tail.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum16(unsigned long long (&S)[16], unsigned long long (*s_part)[16],
                                            unsigned long long* s_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned long long t = wave_sum_u64(S[k]);
        if (lane == 0) s_part[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        unsigned long long t = 0ull;
#pragma unroll
        for (int w = 0; w < ICPT_WAVES; ++w) t += s_part[w][threadIdx.x];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(ICPT_THREADS) k_icp_cluster(int ncl, const int* __restrict__ order,
                                                              const IcpCluster* __restrict__ cl, IcpState* __restrict__ st,
                                                              unsigned long long* __restrict__ accf,
                                                              const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                                              const float4* __restrict__ thi,
                                                              const IcpGrid* __restrict__ grids,
                                                              const unsigned short* __restrict__ tcell, float4* src,
                                                              const float4* __restrict__ src0, int* nn, int* queue,
                                                              IcpParams prm) {
    __shared__ float4 s_tpl[ICPT_IMG];
    __shared__ unsigned short s_cs[ICP_MAX_CELLS + 8];   // cell start table of the staged template
    __shared__ unsigned long long s_part[ICPT_WAVES][16];
    __shared__ unsigned long long s_tot[16];
    __shared__ IcpState s_so;   // the cluster's ICP state (T = current transformation_, Tfinal = accumulated)
    __shared__ int s_flag[2];   // [0] queue item, [1] done
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RunBoxes bx;
    int staged = -1;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_flag[0] = atomicAdd(queue, 1);
        __syncthreads();
        const int item = s_flag[0];
        if (item >= ncl) break;
        const int k = order[item];
        const IcpCluster c = cl[k];
        if (st[2 * (size_t)k].done) continue;            // host pre-marked (too few points / no template)
        const float4* tp = tpl + c.tpl_off;
        const float4* blo = tlo + c.tpl_off / ICP_SUB;
        const float4* bhi = thi + c.tpl_off / ICP_SUB;
        const bool resident = c.tpl_m <= ICPT_TPL_LDS;
        const IcpGrid g = grids[c.slot];
        const bool grid_ok = resident && g.ncell > 0;
        const float rmax = __fmul_rn(prm.grid_rc, g.cell);
        const int gpad = (c.tpl_m + ICP_SUB - 1) / ICP_SUB * ICP_SUB;   // first point of the +inf pad run of the staged image
        if (resident && staged != c.tpl_off) {
            __syncthreads();
            if (grid_ok) for (int i = threadIdx.x; i <= g.ncell; i += ICPT_THREADS) s_cs[i] = tcell[g.cell_off + i];
            stage_chunk(tp, blo, bhi, 0, c.tpl_m, s_tpl, bx);
            staged = c.tpl_off;
        }
        float4* pts = src + c.src_off;
        const float4* pts0 = src0 + c.src_off;
        int* nnq = nn + c.src_off;
        if (threadIdx.x == 0) s_so = st[2 * (size_t)k];   // thread 0 owns and updates it
        int it = 0;
#ifdef CD_TIMERS
        long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
        for (;; ++it) {
            unsigned long long S[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = 0ull;
            for (int q0 = 0; q0 < c.n; q0 += ICPT_THREADS) {
                const int nq = min(ICPT_THREADS, c.n - q0);
                const int myq = q0 + wave * WAVE + lane;   // a wave takes 64 consecutive points: coalesced, and neighbours in
                QueryRegs q;                               // voxel order have neighbouring answers (less divergence per pass)
                int nk;
                // fetch (+ X <- T*X written back by the lane that owns the point)
                nk = min(WAVE, max(0, nq - wave * WAVE));
                q.px = q.py = q.pz = q.pbest = 0.f; q.pbi = 0; q.poi = 0x7fffffff;
                if (lane < nk) {
                    const float4 p = pts[myq];
                    q.px = p.x; q.py = p.y; q.pz = p.z;
                    if (it > 0) {
                        xform(s_so.T, p.x, p.y, p.z, q.px, q.py, q.pz);   // T read from LDS to keep registers free
                        pts[myq] = make_float4(q.px, q.py, q.pz, p.w);
                    }
                    q.pbest = 3.402823466e38f;
                    if (it > 0) {
                        q.pbi = nnq[myq];
                        const float4 q0p = tp[q.pbi];
                        q.pbest = dist2(q.px, q.py, q.pz, q0p.x, q0p.y, q0p.z);
                    }
                    if (it < 3) {
                        for (int j = 0; j < c.tpl_m; j += ICP_SUB) {
                            const float4 t = tp[j];
                            const float d = dist2(q.px, q.py, q.pz, t.x, t.y, t.z);
                            if (d < q.pbest) { q.pbest = d; q.pbi = j; }
                        }
                    }
                    q.pbest = seed_bound(q.pbest);
                    q.poi = __float_as_int(tp[q.pbi].w);
                }
                CD_PHASE(0)
                if (resident) {
                    // near queries: lane-per-query walk of the grid cells their seed ball touches; the rest: wave-per-query
                    float rr = 0.f;
                    bool near = false;
                    if (lane < nk && grid_ok) { rr = __fmul_rn(__fsqrt_rn(q.pbest), 1.0f + 2.0e-6f); near = rr <= rmax; }
                    if (ballot64(near)) grid_search(s_tpl, s_cs, g, near, rr, q, gpad);
                    CD_PHASE(1)
#ifdef CD_STATS
                    { const unsigned long long nb_ = ballot64(near); if (lane == 0) { atomicAdd(&g_icp_stats[0], (unsigned long long)nk); atomicAdd(&g_icp_stats[3], (unsigned long long)__popcll(nb_)); } }
#endif
                    search_chunk(s_tpl, bx, 0, c.tpl_m, q, ballot64(lane < nk && !near));
                    CD_PHASE(2)
                } else {
                    for (int c0 = 0; c0 < c.tpl_m; c0 += ICPT_TPL_LDS) {
                        const int cn = min(ICPT_TPL_LDS, c.tpl_m - c0);
                        stage_chunk(tp, blo, bhi, c0, cn, s_tpl, bx);
                        search_chunk(s_tpl, bx, c0, cn, q, lanes_below(nk));
                    }
                    staged = -1;
                }
                if (lane < nk) {
                    nnq[myq] = q.pbi;
                    const float4 qq = resident ? s_tpl[q.pbi] : tp[q.pbi];
                    const float pv[3] = {q.px, q.py, q.pz}, qv[3] = {qq.x, qq.y, qq.z};
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        S[a] += (unsigned long long)fixq(pv[a], FIX_SHIFT);
                        S[3 + a] += (unsigned long long)fixq(qv[a], FIX_SHIFT);
#pragma unroll
                        for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] += (unsigned long long)fixq(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
                    }
                    S[15] += (unsigned long long)fixq(q.pbest, FIX_SHIFT_D2);
                }
            }
            CD_PHASE(3)
            block_sum16(S, s_part, s_tot);
            CD_PHASE(4)
            if (threadIdx.x == 0) {   // solve for iteration it+1 (same code as k_icp_solve)
                float Tn[16];
                umeyama_from_moments(s_tot, c.n, Tn);
                float Tf[16];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        Tf[4 * i + j] = ((Tn[4 * i] * s_so.Tfinal[j] + Tn[4 * i + 1] * s_so.Tfinal[4 + j]) +
                                         Tn[4 * i + 2] * s_so.Tfinal[8 + j]) + Tn[4 * i + 3] * s_so.Tfinal[12 + j];
                for (int i = 0; i < 16; ++i) s_so.Tfinal[i] = Tf[i];
                s_so.iters += 1;
                int done = 0;
                if (s_so.iters >= prm.max_iter) {
                    done = 1;
                } else {
                    const double cos_angle = 0.5 * (double)(((Tn[0] + Tn[5]) + Tn[10]) - 1.0f);
                    const double translation_sqr = (double)((Tn[3] * Tn[3] + Tn[7] * Tn[7]) + Tn[11] * Tn[11]);
                    if (cos_angle >= prm.rot_thr && translation_sqr <= prm.trans_eps) {
                        done = 1;
                    } else {
                        const double mse = unfix(s_tot[15], FIX_SHIFT_D2) / (double)c.n;
                        if (fabs(mse - s_so.prev_mse) < prm.abs_mse) done = 1;
                        else if (fabs(mse - s_so.prev_mse) / s_so.prev_mse < prm.rel_mse) done = 1;
                        s_so.prev_mse = mse;
                    }
                }
                for (int i = 0; i < 16; ++i) s_so.T[i] = Tn[i];
                s_flag[1] = done;
            }
            __syncthreads();
            CD_PHASE(5)
            if (s_flag[1]) break;
        }
#ifdef CD_TIMERS
        if (lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(&g_icp_stats[8 + i], (unsigned long long)tph[i]);
#endif
        // final X <- T*X (PCL transforms before it tests convergence), then getFitnessScore()
        {
            unsigned long long S[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = 0ull;
            for (int q0 = 0; q0 < c.n; q0 += ICPT_THREADS) {
                const int nq = min(ICPT_THREADS, c.n - q0);
                const int myq = q0 + wave * WAVE + lane;
                const int nk = min(WAVE, max(0, nq - wave * WAVE));
                QueryRegs q;
                q.px = q.py = q.pz = q.pbest = 0.f; q.pbi = 0; q.poi = 0x7fffffff;
                if (lane < nk) {
                    const float4 p = pts[myq];
                    float ox, oy, oz;
                    xform(s_so.T, p.x, p.y, p.z, ox, oy, oz);
                    pts[myq] = make_float4(ox, oy, oz, p.w);
                    const float4 p0 = pts0[myq];
                    xform(s_so.Tfinal, p0.x, p0.y, p0.z, q.px, q.py, q.pz);
                    q.pbi = nnq[myq];
                    const float4 q0p = tp[q.pbi];
                    q.pbest = seed_bound(dist2(q.px, q.py, q.pz, q0p.x, q0p.y, q0p.z));
                    q.poi = __float_as_int(q0p.w);
                }
                if (resident) {
                    // near queries: lane-per-query walk of the grid cells their seed ball touches; the rest: wave-per-query
                    float rr = 0.f;
                    bool near = false;
                    if (lane < nk && grid_ok) { rr = __fmul_rn(__fsqrt_rn(q.pbest), 1.0f + 2.0e-6f); near = rr <= rmax; }
                    if (ballot64(near)) grid_search(s_tpl, s_cs, g, near, rr, q, gpad);
#ifdef CD_STATS
                    { const unsigned long long nb_ = ballot64(near); if (lane == 0) { atomicAdd(&g_icp_stats[0], (unsigned long long)nk); atomicAdd(&g_icp_stats[3], (unsigned long long)__popcll(nb_)); } }
#endif
                    search_chunk(s_tpl, bx, 0, c.tpl_m, q, ballot64(lane < nk && !near));
                } else {
                    for (int c0 = 0; c0 < c.tpl_m; c0 += ICPT_TPL_LDS) {
                        const int cn = min(ICPT_TPL_LDS, c.tpl_m - c0);
                        stage_chunk(tp, blo, bhi, c0, cn, s_tpl, bx);
                        search_chunk(s_tpl, bx, c0, cn, q, lanes_below(nk));
                    }
                    staged = -1;
                }
                if (lane < nk) S[0] += (unsigned long long)fixq(q.pbest, FIX_SHIFT_D2);
            }
            block_sum16(S, s_part, s_tot);
            if (threadIdx.x == 0) {
                s_so.done = 1;
                s_so.converged = 1;
                st[2 * (size_t)k] = s_so;
                st[2 * (size_t)k + 1] = s_so;
                accf[k] = s_tot[0];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// k_icp_pipe: whole-cluster ICP with TWO clusters in flight per workgroup and no workgroup
// barrier inside the iteration loop.
//
// k_icp_cluster spends about a fifth of its time in the per-iteration barrier (waves of one
// workgroup finish their share of an iteration at different times) and another tenth in the
// single-threaded Umeyama/SVD solve that 1023 threads wait for.  Here a workgroup owns up to
// CD_PIPE_SLOTS cluster slots (same template in LDS; the launch says how many it uses: two when it
// has the GPU to itself, four when the GPU is shared).  Every wave walks the slots round-robin:
//   wait until the slot's epoch says the previous step has been solved  ->  do its own share of
//   the slot's current step (a fixed set of source points per lane, as in k_icp_cluster)  ->
//   add its 16 fixed-point moment sums to the slot's LDS accumulators  ->  count itself arrived.
// The wave that arrives LAST runs the solve for that step (or the fitness hand-over and the
// refill of the slot from the global cluster queue) and bumps the epoch, while the other
// fifteen waves are already working on the other slots.  All waves visit the same sequence of
// (slot, epoch) steps and the slowest wave never waits for anything but a solve in progress, so
// there is no circular wait; a slot whose queue ran dry is published as exhausted through the
// same epoch mechanism and every wave leaves after seeing every slot exhausted.
// Arithmetic is that of k_icp_cluster / k_icp_iter + k_icp_solve + k_icp_fitness (order-free
// integer moment sums), so results are bit-identical.
// ---------------------------------------------------------------------------------------
constexpr int PIPE_SLOTS = CD_PIPE_SLOTS;   // most clusters in flight per workgroup (common.hpp); the launch says how many it uses
static_assert(PIPE_SLOTS >= 1 && PIPE_SLOTS <= 8, "one byte of the packed per-wave step counters per slot");   // (common.hpp allows 6: LDS)
enum { PH_ITER = 0, PH_FIT = 1, PH_EXHAUSTED = 2, PH_FILL = 3 };

struct PipeSlot {
    IcpState so;                    // T = transformation_ of the current iteration, Tfinal = accumulated
    unsigned long long acc[16];     // moment sums of the current step
    int src_off, n, k;              // the cluster
    int phase, it;
    int arrived;                    // waves that finished their share of the current step
    int epoch;                      // steps completed (solved) so far
    int next_pass;                  // next 64-point pass of the current step nobody has taken yet
    int give;                       // 1: the cluster is promised to a waiting workgroup - this step's stores of its points and neighbour indices go
                                    // through to memory (agent scope), then it leaves
    int take;                       // 1: the cluster has just arrived from another workgroup - this step reads them past L1 / L2 (agent scope)
};

// Umeyama + convergence test of one iteration (lane 0 of the finishing wave); same code as k_icp_solve.
__device__ __noinline__ int pipe_solve(PipeSlot* sl, const IcpParams& prm) {
    float Tn[16];
    umeyama_from_moments(sl->acc, sl->n, Tn);
    float Tf[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            Tf[4 * i + j] = ((Tn[4 * i] * sl->so.Tfinal[j] + Tn[4 * i + 1] * sl->so.Tfinal[4 + j]) +
                             Tn[4 * i + 2] * sl->so.Tfinal[8 + j]) + Tn[4 * i + 3] * sl->so.Tfinal[12 + j];
    for (int i = 0; i < 16; ++i) sl->so.Tfinal[i] = Tf[i];
    sl->so.iters += 1;
    int done = 0;
    if (sl->so.iters >= prm.max_iter) {
        done = 1;
    } else {
        const double cos_angle = 0.5 * (double)(((Tn[0] + Tn[5]) + Tn[10]) - 1.0f);
        const double translation_sqr = (double)((Tn[3] * Tn[3] + Tn[7] * Tn[7]) + Tn[11] * Tn[11]);
        if (cos_angle >= prm.rot_thr && translation_sqr <= prm.trans_eps) {
            done = 1;
        } else {
            const double mse = unfix(sl->acc[15], FIX_SHIFT_D2) / (double)sl->n;
            if (fabs(mse - sl->so.prev_mse) < prm.abs_mse) done = 1;
            else if (fabs(mse - sl->so.prev_mse) / sl->so.prev_mse < prm.rel_mse) done = 1;
            sl->so.prev_mse = mse;
        }
    }
    for (int i = 0; i < 16; ++i) sl->so.T[i] = Tn[i];
    return done;
}

// next cluster from the global queue into the slot (lane 0 of the finishing wave)
// (items [gbeg, gend) of `order`: the clusters that share the workgroup's template)
__device__ __forceinline__ void pipe_refill(PipeSlot* sl, int gbeg, int gend, const int* order, const IcpCluster* cl, const IcpState* st, int* queue,
                                            int* don, bool no_queue = false) {
    sl->give = 0; sl->take = 0;
    if (no_queue) { sl->phase = PH_EXHAUSTED; return; }   // (IcpParams::don_idle: this workgroup only ever works on hand-overs)
    for (;;) {
        const int item = gbeg + atomicAdd(queue, 1);
        if (item >= gend) { sl->phase = PH_EXHAUSTED; return; }
        const int k = order[item];
        if (st[2 * (size_t)k].done) {                    // host pre-marked (too few points / no template)
            if (don) __hip_atomic_fetch_add(don + DON_FINISHED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        const IcpCluster c = cl[k];
        sl->src_off = c.src_off; sl->n = c.n; sl->k = k;
        sl->so = st[2 * (size_t)k];
        sl->phase = PH_ITER; sl->it = 0;
        return;
    }
}

// ---- hand-over of running clusters (IcpParams::donate; launches that have the GPU to themselves) ---------------------------
// The bench batch is 522 clusters on 256 workgroups: two each, of 6 to 100 iterations that nothing predicts - the slowest
// workgroup runs 4.4 ms, the average one 3.5 (tools/probe_balance.py).  A workgroup that finds the queue empty with nothing
// left of its own therefore WAITS (one lane polls, the others sit at a barrier) instead of ending, and a workgroup that still
// has two or more clusters going gives one of them away:
//   waiter : DON_AVAIL += 1, then polls the mailbox (and DON_FINISHED == clusters of the launch -> ends)
//   donor  : at a step boundary sees DON_AVAIL > 0, takes one (DON_AVAIL -= 1: one promise per waiter) and sets the slot's `give`;
//            during the NEXT step every wave writes the cluster's points and neighbour indices with agent-scope (sc1) stores -
//            through to memory: the taker may sit on another XCD, whose L2 is not coherent with this one - and waits for them
//            (vmcnt) before it arrives; at the end of that step the finishing wave writes the slot's IcpState to st[] the same
//            way, waits, and publishes the cluster id in the next mailbox entry.  (A cluster that converged in that very step
//            is not handed over: the promise is returned, DON_AVAIL += 1.)
//   taker  : reads the entry and the state with agent-scope loads, sets the slot's `take`: in the cluster's FIRST step here every
//            wave reads its points and indices with agent-scope loads (past this CU's L1 and this XCD's L2, which may hold
//            lines from an earlier stay of the cluster); that step rewrites every point and index, so plain accesses are right
//            again from the second step on.  The arithmetic does not depend on who executes it: the records are bit-identical
//            with or without hand-overs (tests/test_gpu_timed_path.py).
// (First version: agent-scope FENCES - every wave of the give step wrote back its XCD's whole L2, the taker invalidated its own.
//  Same time, but the launch's HBM traffic went from 0.26 to 0.48 GB; profiles/r04_handover.txt.)
// Nobody ever waits for a donor or a taker: donors never block, and a waiter's poll ends when every cluster is finished,
// which the running workgroups reach on their own.  Both polls carry a bound all the same.
__device__ __forceinline__ int don_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// agent-scope (sc1) accesses of a cluster's points and neighbour indices: written through to memory / read past this CU's L1
// and this XCD's L2, dword by dword (relaxed atomics: the orderings come from s_waitcnt and the mailbox entry)
__device__ __forceinline__ void store_agent(float4* p, const float4& v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    __hip_atomic_store(u + 0, __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 1, __float_as_uint(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 2, __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 3, __float_as_uint(v.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 load_agent(const float4* p) {
    const unsigned* u = reinterpret_cast<const unsigned*>(p);
    const unsigned x = __hip_atomic_load(u + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), y = __hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned z = __hip_atomic_load(u + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w = __hip_atomic_load(u + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
}
// the slot's IcpState to / from st[] the same way (lane 0; 40 dwords)
__device__ __forceinline__ void state_store_agent(IcpState* dst, const IcpState& src) {
    static_assert(sizeof(IcpState) % 4 == 0, "IcpState is copied dword by dword");
    unsigned* d = reinterpret_cast<unsigned*>(dst);
    const unsigned* s = reinterpret_cast<const unsigned*>(&src);
    for (int i = 0; i < (int)(sizeof(IcpState) / 4); ++i) __hip_atomic_store(d + i, s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void state_load_agent(IcpState& dst, const IcpState* src) {
    unsigned* d = reinterpret_cast<unsigned*>(&dst);
    const unsigned* s = reinterpret_cast<const unsigned*>(src);
    for (int i = 0; i < (int)(sizeof(IcpState) / 4); ++i) d[i] = __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int pipe_active_slots(const PipeSlot* slots) {
    int a = 0;
    for (int i = 0; i < PIPE_SLOTS; ++i) a += (slots[i].phase == PH_ITER || slots[i].phase == PH_FIT) ? 1 : 0;
    return a;
}
// lane 0 of the finishing wave, cluster not converged: returns true when the slot's cluster went to the mailbox
__device__ __noinline__ bool pipe_give(PipeSlot* sl, const PipeSlot* slots, IcpState* st, int* don, bool fault) {
    bool gone = false;
    if (pipe_active_slots(slots) >= 2) {
        state_store_agent(&st[2 * (size_t)sl->k], sl->so);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the state is in memory before the entry says so
        const int t = __hip_atomic_fetch_add(don + DON_TAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < DON_CAP) {
            if (!(fault && t == 0))   // (fault injection, tests: the first entry is claimed and never written)
                __hip_atomic_store(don + DON_BOX + t, sl->k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gone = true;
        }
    }
    if (!gone) __hip_atomic_fetch_add(don + DON_AVAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // promise returned
    sl->give = 0;
    return gone;
}
// thread 0 of a workgroup with nothing left: the id of a cluster to carry on with, or -1 when the launch is finished
// Both bail-outs are REPORTED (DON_ERR; the host fails the call with CD_ERR_DEVICE): an entry that was claimed and never
// appeared means its cluster left its donor and reached nobody; running out of polls with clusters still open means the
// launch's bookkeeping is off.  Neither can happen in a healthy launch (a donor stores the entry right after claiming its
// index; every running workgroup finishes its clusters on its own) - tests/test_gpu_timed_path.py injects the first.
__device__ __noinline__ int pipe_wait_for_cluster(int* don, int total, int entry_spins) {
    __hip_atomic_fetch_add(don + DON_AVAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int polls = 0; polls < (1 << 22); ++polls) {
        const int head = don_load(don + DON_HEAD), tail = min(don_load(don + DON_TAIL), DON_CAP);
        if (head < tail) {
            int expect = head;
            if (__hip_atomic_compare_exchange_strong(don + DON_HEAD, &expect, head + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                int v = 0;
                for (int spins = 0; spins < entry_spins && v == 0; ++spins) v = don_load(don + DON_BOX + head);
                if (v == 0) __hip_atomic_store(don + DON_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // claimed, never published
                return v - 1;
            }
            continue;
        }
        if (don_load(don + DON_FINISHED) >= total) return -1;
        if (don_load(don + DON_ERR)) return -1;   // (somebody lost a cluster: FINISHED will never reach the total)
        __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_fetch_add(don + DON_AVAIL, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // this waiter is gone: no donor may promise it a cluster
    __hip_atomic_store(don + DON_ERR, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return -1;
}

// BIG = false: k_icp_pipe, the template (<= ICP_TPL_LDS points) and its k-d position table live in LDS.
// BIG = true : k_icp_pipe_big, a template of up to 65535 points stays in global memory (L2-resident: a few hundred KiB read by
//              every workgroup) - the grid walk reads the cell-sorted copy `tpl` point by point, the wave-per-query search the
//              k-d ordered copy `tplk` patch by patch (search_patches_big); LDS holds the cell start table and the patch boxes.
//              Same pipeline, same arithmetic, same tie rule: bit-identical results (keys carry 16-bit positions).
template <bool BIG>
__device__ __forceinline__ void icp_pipe_body(int ncl, const int* __restrict__ order,
                                              const IcpCluster* __restrict__ cl, IcpState* st,
                                              unsigned long long* __restrict__ accf,
                                              const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                              const float4* __restrict__ thi,
                                              const unsigned short* __restrict__ kdmap,
                                              const IcpGrid* __restrict__ grids,
                                              const unsigned short* __restrict__ tcell, float4* src,
                                              const float4* __restrict__ src0, int* nn, int* queue,
                                              const int* __restrict__ wgtab, IcpParams prm,
                                              const float4* __restrict__ tplk, const IcpSuper* __restrict__ supers) {
    __shared__ float4 s_tpl[BIG ? 1 : ICPT_IMG];
    __shared__ unsigned short s_cs[ICP_MAX_CELLS + 8];
    __shared__ PipeSlot s_slot[PIPE_SLOTS];
    __shared__ unsigned short s_kd[BIG ? 1 : ICPT_IMG];   // k-d patch order -> stored position (tlo/thi are the PATCH boxes)
    __shared__ unsigned long long s_far[ICPT_WAVES];   // one word per wave: the running minimum of the far query it is on
    __shared__ float4 s_plo[BIG ? ICP_BIG_PATCHES : 1], s_phi[BIG ? ICP_BIG_PATCHES : 1];   // BIG: boxes of all k-d patches
    constexpr int KSH = BIG ? 16 : 13;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // A workgroup keeps ONE (LDS-resident, gridded) template for its whole life.  With several templates in a launch
    // (every cluster against every template, opd flavour with template_slot = -1) the host groups the clusters by template,
    // gives every group a share of the workgroups and its own queue: wgtab[3 b] = {first item, end item, queue} of
    // workgroup b; without a table all clusters share one template and one queue.
    int gbeg = 0, gend = ncl;
    if (wgtab) { gbeg = wgtab[3 * blockIdx.x]; gend = wgtab[3 * blockIdx.x + 1]; queue += wgtab[3 * blockIdx.x + 2]; }
    int* const don = (prm.donate && !wgtab) ? prm.don : nullptr;   // (hand-overs: launches with one queue only)
    const bool no_queue = don && (int)blockIdx.x < prm.don_idle;   // (tests: this workgroup only ever gets work by hand-over)
    __shared__ int s_take;
    const IcpCluster c0 = cl[order[gbeg]];
    const IcpGrid g = grids[c0.slot];
    const float4* tp = tpl + c0.tpl_off;
    const int tpl_m = c0.tpl_m;
    const float rmax = __fmul_rn(prm.grid_rc, g.cell);
    // the point a lane without work reads: the first point of the +inf pad run of the staged image; from global memory the
    // template's last point (testing a real template point is always harmless: it can only be the answer if it is the answer)
    const int gpad = BIG ? tpl_m - 1 : (tpl_m + ICP_SUB - 1) / ICP_SUB * ICP_SUB;
    RunBoxes bx;
    SuperRegs sp;
    const float4* tk = BIG ? tplk + c0.tpl_off : nullptr;
    const unsigned short* km = kdmap + c0.tpl_off;
    for (int i = threadIdx.x; i <= g.ncell; i += ICPT_THREADS) s_cs[i] = tcell[g.cell_off + i];
    if constexpr (!BIG) {
        for (int i = threadIdx.x; i < (tpl_m + ICP_SUB - 1) / ICP_SUB * ICP_SUB; i += ICPT_THREADS) s_kd[i] = km[i];
        stage_chunk(tp, tlo + c0.tpl_off / ICP_SUB, thi + c0.tpl_off / ICP_SUB, 0, tpl_m, s_tpl, bx);
        // re-label the image: the low word of a search key, (original index << 13) | position (pads: all ones, and +inf
        // coordinates anyway), replaces the bare original index, which nothing in this kernel needs ...
        for (int i = threadIdx.x; i < (tpl_m + ICP_SUB - 1) / ICP_SUB * ICP_SUB + ICP_SUB; i += ICPT_THREADS) {
            // ... and stored as (x, y, key word, z): the key word then sits in the even register of the loaded quad and the
            // squared distance can be formed in the z register next to it - the 64-bit key (rule C5's compare) needs no move
            const float4 p = s_tpl[i];
            const unsigned oi = (unsigned)__float_as_int(p.w);
            s_tpl[i] = make_float4(p.x, p.y, __uint_as_float(i < tpl_m ? ((oi << 13) | (unsigned)i) : 0xffffffffu), p.z);
        }
        __syncthreads();
    } else {
        const int nruns = (tpl_m + ICP_SUB - 1) / ICP_SUB;
        for (int i = threadIdx.x; i < nruns; i += ICPT_THREADS) { s_plo[i] = tlo[c0.tpl_off / ICP_SUB + i]; s_phi[i] = thi[c0.tpl_off / ICP_SUB + i]; }
        const IcpSuper& su = supers[c0.slot];
        const float inf = __uint_as_float(0x7f800000u);
        sp.L = sp.H = make_float4(inf, inf, inf, 0.f);
        sp.first = 0; sp.cnt = 1;
        if (lane < su.n) {
            sp.L = make_float4(su.lo[lane][0], su.lo[lane][1], su.lo[lane][2], 0.f);
            sp.H = make_float4(su.hi[lane][0], su.hi[lane][1], su.hi[lane][2], 0.f);
            sp.first = su.first[lane]; sp.cnt = su.cnt[lane];
        }
        __syncthreads();
    }
    // lane l keeps the boxes of patch l of the LEFT half of the k-d root split and of patch l of the RIGHT half
    const int psplit = g.kd_split;
    if constexpr (!BIG) {
        const int nruns = (tpl_m + ICP_SUB - 1) / ICP_SUB;
        const float inf = __uint_as_float(0x7f800000u);
        const float4 none = make_float4(inf, inf, inf, 0.f);
        const float4* blo = tlo + c0.tpl_off / ICP_SUB;
        const float4* bhi = thi + c0.tpl_off / ICP_SUB;
        bx.L0 = bx.H0 = bx.L1 = bx.H1 = none;
        if (lane < psplit) { bx.L0 = blo[lane]; bx.H0 = bhi[lane]; }
        if (psplit + lane < nruns) { bx.L1 = blo[psplit + lane]; bx.H1 = bhi[psplit + lane]; }
    }
    if (threadIdx.x == 0) {
        for (int sidx = 0; sidx < PIPE_SLOTS; ++sidx) {
            PipeSlot* sl = &s_slot[sidx];
            for (int i = 0; i < 16; ++i) sl->acc[i] = 0ull;
            sl->arrived = 0; sl->epoch = 0; sl->it = 0; sl->n = 0; sl->src_off = 0; sl->k = 0; sl->next_pass = 0; sl->give = 0; sl->take = 0;
            sl->phase = sidx < prm.pipe_slots ? PH_FILL : PH_EXHAUSTED;   // (a slot the launch does not use is dropped at its first visit)
        }
        // slot 0 starts with a cluster; the other slots are filled at their first step (PH_FILL), after every workgroup took its first
        pipe_refill(&s_slot[0], gbeg, gend, order, cl, st, queue, don, no_queue);
        if (no_queue) for (int sidx = 0; sidx < PIPE_SLOTS; ++sidx) s_slot[sidx].phase = PH_EXHAUSTED;
    }
    __syncthreads();
#ifdef CD_TIMERS
    long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
    const long long wg_t0 = wall_clock64();
#endif
#ifdef CD_DONDBG
    const long long don_t0 = wall_clock64();
#endif
    // per wave: the steps it has completed on each slot (one byte per slot, mod 256: waves are never a whole step apart) and
    // the slots that still have work
    unsigned long long my_ep = 0ull;
    unsigned live = (1u << PIPE_SLOTS) - 1u;
    for (;;) {
    while (live) {
        for (int sidx = 0; sidx < PIPE_SLOTS; ++sidx) {
            if (!((live >> sidx) & 1u)) continue;
            PipeSlot* sl = &s_slot[sidx];
            const int want = (int)((my_ep >> (8 * sidx)) & 0xffu);
            while (__hip_atomic_load(&sl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != want) __builtin_amdgcn_s_sleep(2);
            __threadfence_block();
            CD_PHASE(0)
            const int phase = sl->phase;
            if (phase == PH_EXHAUSTED) { live &= ~(1u << sidx); continue; }
            const int give = sl->give;   // the cluster leaves this workgroup after this step: its stores must reach memory
            const int take = sl->take;   // it arrived before this step: its points and indices are read from memory
            if (phase == PH_ITER || phase == PH_FIT) {
                const int it = sl->it, n = sl->n;
                float4* pts = src + sl->src_off;
                const float4* pts0 = src0 + sl->src_off;
                int* nnq = nn + sl->src_off;
                // The step's queries are cut into passes of 64 consecutive points; waves PULL passes from the slot's
                // counter, so a wave that drew cheap passes (certainly-near queries) simply takes more of them and all
                // waves reach the end of the step together.  (A point is then touched by different waves in different
                // steps: same CU, and the step boundary is a workgroup-scope release/acquire.)
                const int npass = (n + 63) >> 6;
                for (;;) {
                    int pss = 0;
                    if (lane == 0) pss = atomicAdd(&sl->next_pass, 1);
                    pss = __builtin_amdgcn_readfirstlane(pss);
                    if (pss >= npass) break;
                    const int q0 = pss << 6;
                    const int nk = min(64, n - q0);
                    const int myq = q0 + lane;
#ifdef CD_ITSTATS
                    const long long tpass0 = clock64();
                    const int stat_it = phase == PH_ITER ? min(it, 30) : 31;
#endif
                    QueryRegs q;
                    q.px = q.py = q.pz = q.pbest = 0.f; q.pbi = 0; q.poi = 0x7fffffff;
                    // key word of the seed point: what the LDS image holds in .z, (original index << 16) | position for a
                    // template in global memory; "no index" while there is no seed
                    constexpr unsigned KW_NONE = BIG ? 0xffffffffu : 0x7fffffffu;
                    unsigned kw = KW_NONE;
#ifdef CD_TIMERS_FETCH
                    {   // how long a wave waits for its pass's points and neighbour indices (the loads below then hit L1)
                        float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
                        int nv = 0;
                        if (lane < nk) { pp = pts[myq]; nv = nnq[myq]; }
                        asm volatile("s_waitcnt vmcnt(0)" ::"v"(pp.x), "v"(nv) : "memory");
                        CD_PHASE(3)
                    }
#endif
                    if (lane < nk) {
                        const float4 p = take ? load_agent(&pts[myq]) : pts[myq];
                        if (phase == PH_ITER) {
                            q.px = p.x; q.py = p.y; q.pz = p.z;
                            if (it > 0) {   // X <- T*X, written back by the lane that owns the point
                                xform(sl->so.T, p.x, p.y, p.z, q.px, q.py, q.pz);
                                if (give) store_agent(&pts[myq], make_float4(q.px, q.py, q.pz, p.w));
                                else pts[myq] = make_float4(q.px, q.py, q.pz, p.w);
                            }
                            q.pbest = 3.402823466e38f;
                            if (it > 0) {
                                q.pbi = take ? don_load(&nnq[myq]) : nnq[myq];
                                const float4 q0p = BIG ? tp[(unsigned)q.pbi] : s_tpl[q.pbi];
                                q.pbest = dist2(q.px, q.py, q.pz, q0p.x, q0p.y, BIG ? q0p.z : q0p.w);
                                kw = BIG ? (((unsigned)__float_as_int(q0p.w) << 16) | (unsigned)q.pbi) : (unsigned)__float_as_int(q0p.z);
                            }
                            if (it < 3) {   // coarse seeds: first point of every run
                                for (int j = 0; j < tpl_m; j += ICP_SUB) {
                                    const float4 t = BIG ? tp[j] : s_tpl[j];
                                    const float d = dist2(q.px, q.py, q.pz, t.x, t.y, BIG ? t.z : t.w);
                                    if (d < q.pbest) { q.pbest = d; q.pbi = j; kw = BIG ? (((unsigned)__float_as_int(t.w) << 16) | (unsigned)j) : (unsigned)__float_as_int(t.z); }
                                }
                            }
                            q.pbest = seed_bound(q.pbest);
                            q.poi = 0;   // (BIG: only the chunked searches of the other kernels carry the seed's original index; LDS image: set below)
#ifdef CD_STATS
                            if (lane == 0) atomicAdd(&g_icp_stats[8], (unsigned long long)nk * (unsigned long long)((it > 0 ? 1 : 0) + (it < 3 ? (tpl_m + ICP_SUB - 1) / ICP_SUB : 0)));   // seed tests
#endif
                        } else {            // final X <- T*X, then getFitnessScore() of Tfinal * original source
                            float ox, oy, oz;
                            xform(sl->so.T, p.x, p.y, p.z, ox, oy, oz);
                            pts[myq] = make_float4(ox, oy, oz, p.w);
                            const float4 p0 = pts0[myq];
                            xform(sl->so.Tfinal, p0.x, p0.y, p0.z, q.px, q.py, q.pz);
                            q.pbi = nnq[myq];
                            const float4 q0p = BIG ? tp[(unsigned)q.pbi] : s_tpl[q.pbi];
                            q.pbest = seed_bound(dist2(q.px, q.py, q.pz, q0p.x, q0p.y, BIG ? q0p.z : q0p.w));
                            kw = BIG ? (((unsigned)__float_as_int(q0p.w) << 16) | (unsigned)q.pbi) : (unsigned)__float_as_int(q0p.z);
                        }
                    }
                    CD_PHASE(1)
                    float rr = 0.f;
                    bool near = false;
                    if (lane < nk) { rr = __fmul_rn(__fsqrt_rn(q.pbest), 1.0f + 2.0e-6f); near = rr <= rmax; }
                    {
                        // from the seed BOUND (next float above d2(q, seed): what the radius of the walk is derived from) to the
                        // seed's own KEY for both searches: d2 exactly (the bound's bit pattern minus one) and the key word of the
                        // seed point; a query without a finite seed distance keeps (+inf, no index)
                        const unsigned bb = __float_as_uint(q.pbest);
                        const bool fin = bb < 0x7f800000u && lane < nk;
                        q.pbest = __uint_as_float(fin ? bb - 1u : bb);
                        q.poi = (int)(fin ? kw : KW_NONE);
                    }
                    if (ballot64(near)) grid_search<KSH, !BIG, true>(BIG ? tp : s_tpl, s_cs, g, near, rr, q, gpad);
#ifdef CD_STATS
                    { const unsigned long long nb_ = ballot64(near); if (lane == 0) { atomicAdd(&g_icp_stats[0], (unsigned long long)nk); atomicAdd(&g_icp_stats[3], (unsigned long long)__popcll(nb_)); } }
                    if (lane < nk && !near && phase == PH_ITER)
                        atomicAdd(&g_icp_rhist[(it < 3 ? 0 : it < 16 ? 16 : 32) + min(15, (int)(rr * g.inv))], 1ull);
#endif
                    CD_PHASE(2)
                    if constexpr (BIG) {
                        search_patches_big(tk, km, s_plo, s_phi, sp, q, ballot64(lane < nk && !near), &s_far[wave]);
                    } else {
#ifdef CD_ITSTATS
                    int stat_acc[2] = {0, 0};
                    search_patches<true>(s_tpl, s_kd, bx, tpl_m, q, ballot64(lane < nk && !near), psplit, &s_far[wave], stat_acc);
                    if (lane == 0) {
                        atomicAdd(&g_icp_it[stat_it][0], (unsigned long long)(clock64() - tpass0)); atomicAdd(&g_icp_it[stat_it][1], 1ull);
                        atomicAdd(&g_icp_it[stat_it][2], (unsigned long long)stat_acc[0]); atomicAdd(&g_icp_it[stat_it][3], (unsigned long long)stat_acc[1]);
                    }
#else
                    search_patches<true>(s_tpl, s_kd, bx, tpl_m, q, ballot64(lane < nk && !near), psplit, &s_far[wave]);
#endif
                    }
#ifdef CD_STATS
                    {   // far queries whose TRUE neighbour lies within the grid walk's reach (a better seed would have made them near), by iteration class
                        const unsigned long long far_ = ballot64(lane < nk && !near && phase == PH_ITER);
                        const unsigned long long conv_ = ballot64(lane < nk && !near && phase == PH_ITER && __fmul_rn(__fsqrt_rn(q.pbest), 1.0f + 2.0e-6f) <= rmax);
                        const int cls_ = it < 3 ? 0 : it < 16 ? 1 : 2;
                        if (lane == 0) { atomicAdd(&g_icp_stats[9 + cls_], (unsigned long long)__popcll(conv_)); atomicAdd(&g_icp_stats[12 + cls_], (unsigned long long)__popcll(far_)); }
                    }
#endif
                    CD_PHASE(4)
                    // The pass's 16 moment terms go into the slot's accumulators right away: no sum is carried in registers
                    // across the searches (32 VGPRs that the search loops would otherwise spill around).
                    unsigned long long S[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) S[i] = 0ull;
                    // float -> fixed point (rule C4) in four instructions per term instead of eight (fixq_fast, common.hpp), valid
                    // while every term stays below 2^50 / 2^shift; a wave with a lane outside that range (coordinates beyond 256 m,
                    // neighbours more than 128 m away) takes the general conversion - same integers either way
                    float4 qq = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lane < nk && phase == PH_ITER) qq = BIG ? tp[(unsigned)q.pbi] : s_tpl[q.pbi];
                    const float pv[3] = {q.px, q.py, q.pz}, qv[3] = {qq.x, qq.y, BIG ? qq.z : qq.w};
                    const float big_c = fmaxf(fmaxf(fmaxf(fabsf(pv[0]), fabsf(pv[1])), fabsf(pv[2])), fmaxf(fmaxf(fabsf(qv[0]), fabsf(qv[1])), fabsf(qv[2])));
                    const bool fast = ballot64(lane < nk && !(big_c < 256.f && q.pbest < 16384.f)) == 0ull;
                    if (lane < nk) {
                        if (phase == PH_ITER) {
                            if (give) __hip_atomic_store(&nnq[myq], q.pbi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else nnq[myq] = q.pbi;
                            if (fast) {
#pragma unroll
                                for (int a = 0; a < 3; ++a) {
                                    S[a] = fixq_fast(pv[a], FIX_SHIFT);
                                    S[3 + a] = fixq_fast(qv[a], FIX_SHIFT);
#pragma unroll
                                    for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] = fixq_fast(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
                                }
                                S[15] = fixq_fast(q.pbest, FIX_SHIFT_D2);
                            } else {
#pragma unroll
                                for (int a = 0; a < 3; ++a) {
                                    S[a] = (unsigned long long)fixq(pv[a], FIX_SHIFT);
                                    S[3 + a] = (unsigned long long)fixq(qv[a], FIX_SHIFT);
#pragma unroll
                                    for (int b = 0; b < 3; ++b) S[6 + 3 * a + b] = (unsigned long long)fixq(__fmul_rn(qv[a], pv[b]), FIX_SHIFT);
                                }
                                S[15] = (unsigned long long)fixq(q.pbest, FIX_SHIFT_D2);
                            }
                        } else {
                            S[0] = fast ? fixq_fast(q.pbest, FIX_SHIFT_D2) : (unsigned long long)fixq(q.pbest, FIX_SHIFT_D2);
                        }
                    }
                    wave_fold_to_lds(S, phase == PH_ITER ? 16 : 1, sl->acc);
                    CD_PHASE(5)
                }
                CD_PHASE(5)
            }
            if (give) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the write-through stores of this wave have landed)
            __threadfence_block();
            int a = 0;
            if (lane == 0) a = atomicAdd(&sl->arrived, 1);
            a = __builtin_amdgcn_readfirstlane(a);
            if (a == ICPT_WAVES - 1) {   // last to arrive: finish the step for everybody
                __threadfence_block();
                if (lane == 0) {
                    if (phase == PH_ITER) {
                        // (the load is in flight while the step is solved)
                        const int waiting = don ? don_load(don + DON_AVAIL) : 0;
                        sl->take = 0;
                        if (pipe_solve(sl, prm)) {
                            sl->phase = PH_FIT;
                            if (sl->give) { sl->give = 0; __hip_atomic_fetch_add(don + DON_AVAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // converged: stays
                        } else {
                            sl->it += 1;
                            if (sl->give) {
                                if (pipe_give(sl, s_slot, st, don, prm.don_fault != 0)) { DON_DBG(24, don_t0); pipe_refill(sl, gbeg, gend, order, cl, st, queue, don, no_queue); }
                            } else if (waiting > 0 && pipe_active_slots(s_slot) >= 2) {
                                // promise this cluster to one of the waiting workgroups; it goes after the next step
                                if (__hip_atomic_fetch_add(don + DON_AVAIL, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) sl->give = 1;
                                else __hip_atomic_fetch_add(don + DON_AVAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    } else {
                        if (phase == PH_FIT) {
                            sl->so.done = 1;
                            sl->so.converged = 1;
                            st[2 * (size_t)sl->k] = sl->so;
                            st[2 * (size_t)sl->k + 1] = sl->so;
                            accf[sl->k] = sl->acc[0];
                            if (don) __hip_atomic_fetch_add(don + DON_FINISHED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        pipe_refill(sl, gbeg, gend, order, cl, st, queue, don, no_queue);
                    }
                    for (int i = 0; i < 16; ++i) sl->acc[i] = 0ull;
                    sl->arrived = 0;
                    sl->next_pass = 0;
                }
                __threadfence_block();
                if (lane == 0) __hip_atomic_store(&sl->epoch, (want + 1) & 0xff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                CD_PHASE(3)
            }
            my_ep = (my_ep & ~(0xffull << (8 * sidx))) | ((unsigned long long)((want + 1) & 0xff) << (8 * sidx));
        }
    }
    // nothing left here.  With hand-overs on, wait for a running cluster of a workgroup that still has several (see pipe_give):
    // it goes into slot 0, whose step counter every wave left at the slot's current epoch.
    if (!don) break;
    __syncthreads();
    if (threadIdx.x == 0) {
        DON_DBG(8, don_t0);
#ifdef CD_DONDBG
        const long long tw0 = wall_clock64();
#endif
        const int k = pipe_wait_for_cluster(don, gend - gbeg, prm.don_fault ? (1 << 14) : (1 << 24));
#ifdef CD_DONDBG
        atomicAdd(&g_don_dbg[2], (unsigned long long)(wall_clock64() - tw0));
        if (k >= 0) DON_DBG(40, don_t0); else atomicMax(&g_don_dbg[1], (unsigned long long)(wall_clock64() - don_t0));
#endif
        s_take = k;
        if (k >= 0) {
            PipeSlot* sl = &s_slot[0];
            const IcpCluster c = cl[k];
            sl->src_off = c.src_off; sl->n = c.n; sl->k = k;
            state_load_agent(sl->so, &st[2 * (size_t)k]);
            sl->phase = PH_ITER; sl->it = sl->so.iters; sl->give = 0; sl->take = 1;
            for (int i = 0; i < 16; ++i) sl->acc[i] = 0ull;
            sl->arrived = 0; sl->next_pass = 0;
        }
    }
    __syncthreads();
    if (s_take < 0) break;
    live = 1u;
    }
#ifdef CD_TIMERS
    if (lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(&g_icp_stats[8 + i], (unsigned long long)tph[i]);
    if (threadIdx.x == 0) {   // workgroup busy time (100 MHz wall clock): sum, max, count
        const unsigned long long dt = (unsigned long long)(wall_clock64() - wg_t0);
        atomicAdd(&g_icp_stats[14], dt); atomicMax(&g_icp_stats[15], dt); atomicAdd(&g_icp_stats[7], 1ull);
    }
#endif
}

__global__ void __launch_bounds__(ICPT_THREADS) k_icp_pipe(int ncl, const int* __restrict__ order,
                                                           const IcpCluster* __restrict__ cl, IcpState* st,
                                                           unsigned long long* __restrict__ accf,
                                                           const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                                           const float4* __restrict__ thi,
                                                           const unsigned short* __restrict__ kdmap,
                                                           const IcpGrid* __restrict__ grids,
                                                           const unsigned short* __restrict__ tcell, float4* src,
                                                           const float4* __restrict__ src0, int* nn, int* queue,
                                                           const int* __restrict__ wgtab, IcpParams prm) {
    icp_pipe_body<false>(ncl, order, cl, st, accf, tpl, tlo, thi, kdmap, grids, tcell, src, src0, nn, queue, wgtab, prm, nullptr, nullptr);
}
__global__ void __launch_bounds__(ICPT_THREADS) k_icp_pipe_big(int ncl, const int* __restrict__ order,
                                                               const IcpCluster* __restrict__ cl, IcpState* st,
                                                               unsigned long long* __restrict__ accf,
                                                               const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                                               const float4* __restrict__ thi,
                                                               const unsigned short* __restrict__ kdmap,
                                                               const IcpGrid* __restrict__ grids,
                                                               const unsigned short* __restrict__ tcell, float4* src,
                                                               const float4* __restrict__ src0, int* nn, int* queue,
                                                               const int* __restrict__ wgtab, IcpParams prm,
                                                               const float4* __restrict__ tplk, const IcpSuper* __restrict__ supers) {
    icp_pipe_body<true>(ncl, order, cl, st, accf, tpl, tlo, thi, kdmap, grids, tcell, src, src0, nn, queue, wgtab, prm, tplk, supers);
}

// getFitnessScore(): mean squared NN distance of T_final * (original source)
__global__ void __launch_bounds__(ICPT_THREADS) k_icp_fitness(const IcpWork* __restrict__ work,
                                                              const IcpCluster* __restrict__ cl,
                                                              const IcpState* __restrict__ st, int parity,
                                                              unsigned long long* __restrict__ accf,
                                                              const float4* __restrict__ tpl, const float4* __restrict__ tlo,
                                                              const float4* __restrict__ thi, const IcpGrid* __restrict__ grids,
                                                              const float4* src0, int* nn, float* d2buf, int qslice) {
    __shared__ float4 s_tpl[ICPT_IMG];
    __shared__ unsigned long long s_acc;
    __shared__ float s_T[16];
    __shared__ int s_flag;
    const IcpWork wk = work[blockIdx.x];
    const IcpCluster c = cl[wk.cluster];
    const IcpState* s = st + (size_t)wk.cluster * 2 + parity;
    if (s->status != CD_OK) return;
    if (threadIdx.x < 16) s_T[threadIdx.x] = s->Tfinal[threadIdx.x];
    if (threadIdx.x == 16) s_acc = 0ull;
    __syncthreads();
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = s_T[k];
    const int q0 = wk.tile * qslice;
    const int nq = min(qslice, c.n - q0);
    int* nnq = nn + c.src_off + q0;
    float* d2q = d2buf + c.src_off + q0;
    {
        const float4* tp = tpl + c.tpl_off;
        QueryRegs q;
        RunBoxes bx;
        int nk;
        fetch_queries(tp, c.tpl_m, src0 + c.src_off + q0, nq, true, false, true, T, nnq, q, nk);
        const IcpGrid& g = grids[c.slot];
        if (c.tpl_m > ICPT_TPL_LDS && g.nchunk > 0) {
            for (int ci = 0; ci < g.nchunk; ++ci) {
                bool any;
                const unsigned long long todo = chunk_needed(g, ci, q, nk, &s_flag, &any);
                if (!any) continue;
                stage_chunk(tp, tlo + c.tpl_off / ICP_SUB, thi + c.tpl_off / ICP_SUB, g.chunk_start[ci], g.chunk_n[ci], s_tpl, bx);
                search_chunk(s_tpl, bx, g.chunk_start[ci], g.chunk_n[ci], q, todo);
            }
        } else {
            for (int c0 = 0; c0 < c.tpl_m; c0 += ICPT_TPL_LDS) {
                const int cn = min(ICPT_TPL_LDS, c.tpl_m - c0);
                stage_chunk(tp, tlo + c.tpl_off / ICP_SUB, thi + c.tpl_off / ICP_SUB, c0, cn, s_tpl, bx);
                search_chunk(s_tpl, bx, c0, cn, q, lanes_below(nk));
            }
        }
        store_queries(q, nk, nnq, d2q);
    }
    __syncthreads();
    unsigned long long v = 0ull;
    for (int i = threadIdx.x; i < nq; i += ICPT_THREADS) v += (unsigned long long)fixq(d2q[i], FIX_SHIFT_D2);
    if ((threadIdx.x & ~63) < nq) {
        const unsigned long long t = wave_sum_u64(v);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_acc, t);
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&accf[wk.cluster], s_acc);
}

// pcl::Registration::align(output, guess): input_transformed = guess * source before the first iteration (the iterations
// then run on d_src as they do from the identity; final_transformation_ starts as the guess - the host puts it into the
// initial IcpState).  guesses: one row-major 4x4 per frame (per_frame != 0) or a single one.
__global__ void __launch_bounds__(BLOCK) k_icp_apply_guess(const IcpCluster* __restrict__ cl, const float* __restrict__ guesses,
                                                           int per_frame, const float4* __restrict__ src0, float4* __restrict__ src) {
    const IcpCluster c = cl[blockIdx.x];
    const float* G = guesses + (per_frame ? 16 * (size_t)c.frame : 0);
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = G[k];
    for (int i = blockIdx.y * BLOCK + threadIdx.x; i < c.n; i += gridDim.y * BLOCK) {
        const float4 p = src0[c.src_off + i];
        float ox, oy, oz;
        xform(T, p.x, p.y, p.z, ox, oy, oz);
        src[c.src_off + i] = make_float4(ox, oy, oz, p.w);
    }
}
void launch_icp_apply_guess(hipStream_t s, int ncl, int max_n, const IcpCluster* cl, const float* guesses, int per_frame,
                            const float4* src0, float4* src) {
    if (ncl <= 0 || max_n <= 0) return;
    const int gx = (max_n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_icp_apply_guess, dim3(ncl, gx < 64 ? gx : 64), dim3(BLOCK), 0, s, cl, guesses, per_frame, src0, src);
}

void launch_icp_iter(hipStream_t s, int it, int n_work, int ncl, const IcpWork* work, const IcpCluster* cl, IcpState* st,
                     unsigned long long* acc, const float4* tpl, const float4* tlo, const float4* thi, const IcpGrid* grids,
                     float4* src, int* nn, float* d2buf, int qslice, int* queue, int n_cu, IcpParams prm) {
    if (ncl <= 0) return;
    hipLaunchKernelGGL(k_icp_solve, dim3((ncl + WAVE - 1) / WAVE), dim3(WAVE), 0, s, it, ncl, cl, st, acc, queue, prm);
    if (n_work <= 0) return;   // every cluster converged: only the state bookkeeping above is needed
    hipLaunchKernelGGL(k_icp_iter, dim3(n_work < n_cu ? n_work : n_cu), dim3(ICPT_THREADS), 0, s, it, n_work, work, cl, st, acc, tpl, tlo,
                       thi, grids, src, nn, d2buf, qslice, queue);
}
void launch_icp_fitness(hipStream_t s, int n_work, const IcpWork* work, const IcpCluster* cl, const IcpState* st,
                        int parity, unsigned long long* accf, const float4* tpl, const float4* tlo, const float4* thi,
                        const IcpGrid* grids, const float4* src0, int* nn, float* d2buf, int qslice) {
    if (n_work <= 0) return;
    hipLaunchKernelGGL(k_icp_fitness, dim3(n_work), dim3(ICPT_THREADS), 0, s, work, cl, st, parity, accf, tpl, tlo, thi, grids, src0, nn, d2buf, qslice);
}

void launch_icp_persist(hipStream_t s, int n_work, int n_wg, int max_it, const IcpWork* work, const IcpCluster* cl, IcpState* st,
                         unsigned long long* acc, unsigned long long* accf, const float4* tpl, const float4* tlo, const float4* thi,
                         const IcpGrid* grids, float4* src, const float4* src0, int* nn, float* d2buf, int qslice, unsigned* bar,
                         int* abort_flag, int n_open, int* closed, IcpParams prm) {
    if (n_work <= 0 || n_wg <= 0) return;
    hipLaunchKernelGGL(k_icp_persist, dim3(n_wg), dim3(ICPT_THREADS), 0, s, n_work, max_it, work, cl, st, acc, accf, tpl, tlo, thi, grids,
                       src, src0, nn, d2buf, qslice, bar, abort_flag, n_open, closed, prm);
}

void launch_icp_cluster(hipStream_t s, int ncl, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                        const float4* tpl, const float4* tlo, const float4* thi, const IcpGrid* grids,
                        const unsigned short* tcell, float4* src, const float4* src0, int* nn,
                        int* queue, int n_cu, IcpParams prm) {
    if (ncl <= 0) return;
    hipLaunchKernelGGL(k_icp_cluster, dim3(ncl < n_cu ? ncl : n_cu), dim3(ICPT_THREADS), 0, s, ncl, order, cl, st, accf, tpl, tlo, thi,
                       grids, tcell, src, src0, nn, queue, prm);
}

void launch_icp_pipe(hipStream_t s, int ncl, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                     const float4* tpl, const float4* tlo, const float4* thi, const unsigned short* kdmap, const IcpGrid* grids,
                     const unsigned short* tcell, float4* src, const float4* src0, int* nn,
                     int* queue, int n_wg, const int* wgtab, IcpParams prm) {
    if (ncl <= 0 || n_wg <= 0) return;
    hipLaunchKernelGGL(k_icp_pipe, dim3(n_wg), dim3(ICPT_THREADS), 0, s, ncl, order, cl, st, accf, tpl, tlo, thi,
                       kdmap, grids, tcell, src, src0, nn, queue, wgtab, prm);
}

void launch_icp_pipe_big(hipStream_t s, int ncl, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                         const float4* tpl, const float4* tplk, const float4* tlok, const float4* thik, const unsigned short* kdmap,
                         const IcpGrid* grids, const IcpSuper* supers, const unsigned short* tcell, float4* src, const float4* src0,
                         int* nn, int* queue, int n_wg, const int* wgtab, IcpParams prm) {
    if (ncl <= 0 || n_wg <= 0) return;
    hipLaunchKernelGGL(k_icp_pipe_big, dim3(n_wg), dim3(ICPT_THREADS), 0, s, ncl, order, cl, st, accf, tpl, tlok, thik,
                       kdmap, grids, tcell, src, src0, nn, queue, wgtab, prm, tplk, supers);
}

}  // namespace cd
