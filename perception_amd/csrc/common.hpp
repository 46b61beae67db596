// common.hpp - shared definitions of libcuboid_hip (device + host).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cuboid_hip.h"

namespace cd {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;                 // 4 waves, one per SIMD
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int ITEMS = 8;                   // rows per wave in an ordered tile
constexpr int TILE = BLOCK * ITEMS;        // 2048 elements; wave w owns [w*512,(w+1)*512)
constexpr int WAVE_SPAN = WAVE * ITEMS;    // 512
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
#ifndef CD_SORT_BLOCK
#define CD_SORT_BLOCK 1024
#endif
constexpr int SORT_BLOCK = CD_SORT_BLOCK;           // the sort's tiles are 4x larger than the ordered tiles of the other stages: a
constexpr int SORT_TILE = SORT_BLOCK * ITEMS;   // (digit, tile) run of the scatter is then ~32 pairs = full 128-B lines
constexpr int KICP = CD_MAX_CLUSTERS_PER_FRAME;  // clusters per frame that get ICP
constexpr int FIX_SHIFT = 32;              // canonical rule C4
constexpr int FIX_SHIFT_D2 = 36;
constexpr int RND_TABLE = 1 << 15;         // precomputed mt19937(12345)>>1 draws
constexpr int SAMPLER_MAP = 8192;          // LDS sparse shuffle map (open addressing)
constexpr int MAX_HYP = 1024 + 16;         // plane_max_iterations+1 hypotheses at most kept per frame
constexpr int CELL_BUCKETS = 1 << 16;      // cluster spatial hash buckets per frame
constexpr int ICP_TPL_CHUNK = 2048;        // template points staged in LDS at a time (32 KiB)
constexpr int ICP_SUB = 64;                // template run length that carries one pruning box
#ifndef CD_PIPE_SLOTS
#define CD_PIPE_SLOTS 4                    // most clusters a workgroup of k_icp_pipe / k_icp_pipe_big can keep in flight (IcpParams::pipe_slots)
#endif
// template points resident in LDS: 119 runs (119 KiB) with two pipeline slots, which leave 288 bytes of the CU's 160 KiB;
// slots three to six (312 B each: state + moment sums) fit into ONE run of the image (64 x (16 + 2) B) less
static_assert(CD_PIPE_SLOTS >= 1 && CD_PIPE_SLOTS <= 6, "LDS budget of k_icp_pipe");
constexpr int ICP_TPL_LDS = CD_PIPE_SLOTS <= 2 ? 7616 : 7552;
constexpr int ICP_MAX_CELLS = 12288;       // cells of the template's uniform grid (uint16 start table, 24 KiB of LDS)
constexpr int ICP_CELL_STRIDE = ICP_MAX_CELLS + 8;   // table entries reserved per template slot
constexpr int ICP_MAX_CHUNKS = 12;         // k-d subtree chunks of a template that does not fit LDS
constexpr int ICP_BIG_MAX = 65535;          // largest template the persistent kernel searches in global memory (16-bit positions in the keys)
constexpr int ICP_BIG_PATCHES = 1024;      // its k-d patches of 64 points (boxes in LDS, 32 KiB)
constexpr int ICP_QSLICE = 512;            // max ICP source points (queries) per work item / workgroup

// Per-frame scalars that live on the device and are mirrored to pinned host memory.
struct FrameState {
    int32_t status;        // cd_status
    int32_t n_c;           // cropped points
    uint32_t mn[3], mx[3]; // order-preserving uint encoding of min/max of the cropped cloud
    int32_t min_b[3], div_b[3];
    int32_t key_bits;      // bits needed for the voxel index
    int32_t n_v;           // voxels
    int32_t n_plane;       // refined plane inliers
    int32_t n_o;           // extracted (object) points
    int32_t n_k;           // clusters found
    int32_t n_hyp;         // hypotheses generated so far by the sampler
    int32_t sampler_exhausted;
    int32_t ksize[KICP];   // sizes of the KICP largest clusters
    int32_t koff[KICP];    // their offsets in the per-frame ICP source segment
    float origin[3];       // decoded min of the cropped cloud
    int32_t n_cropped;     // N_c as reported (n_c is zeroed when the frame errors out)
    int32_t cl_done;       // S5: the frame was clustered by k_cluster_lds (0: left to the global-memory kernels)
    int32_t crop_overflow; // single-pass crop: a y cell index did not fit its bit field (the host redoes the batch in two passes)
    int32_t scan_stalled;  // a chained scan gave up waiting for a predecessor tile (reported as CD_ERR_DEVICE)
    int32_t n_runs;        // S1 by runs: runs of equal voxel index among the cropped points (k_voxel_runs; what the sort then moves)
    int32_t digit_vary;    // k_crop_runs' sort: bit d set when digit d of the packed cell keys takes more than one value in this frame
                           // (k_voxel_setup reads it off the frame's digit histograms) - the radix passes the sort needs
    int32_t pad_;
};

struct CropLimits {        // double limits folded to equivalent float compares (exact)
    float zlo, zhi, xlo, xhi;
};

// Single-pass crop (k_crop_fused): the voxel grid origin (min of the cropped cloud) is not known while the points are
// compacted, so a point's three floor(p / leaf) values are stored as absolute bit fields - x and z relative to the floors of
// the crop limits, y biased by half its field - and turned into PCL's idx = i + j dx + k dx dy by the first sort pass.
struct KeyPack {
    int32_t enabled;
    int32_t bi, bj;        // field widths of x and y (z takes the rest)
    int32_t ilo, jlo, klo; // value of field 0 on each axis
};

struct BBoxGate {          // bbox_filter.cpp within_bbox(), see cd_params.bbox_*
    double P[12];
    float rect[4];
    int32_t enable;
    int32_t pad;
};

struct IcpCluster {        // static description of one ICP problem (host-built)
    int32_t src_off;       // offset (points) into the ICP source buffers
    int32_t n;
    int32_t frame, k;
    int32_t tpl_off, tpl_m;
    int32_t tile0;         // first work item of this cluster
    int32_t slot;          // template slot (index of its IcpGrid)
};

// Uniform grid over a template's bounding box (built by cd_set_template).  Template points are stored
// sorted by cell id ((cz*ny + cy)*nx + cx, ties by original index), so the cells cx0..cx1 of one (cy,cz)
// row are ONE contiguous range [start[row+cx0], start[row+cx1+1]) of stored points.
// cell coordinate of a value v on axis a: clamp((int)floorf((v - o_a) * inv), 0, n_a - 1), float, no FMA.
struct IcpGrid {
    float ox, oy, oz, inv;
    float cell;            // edge length
    int32_t nx, ny, nz;
    int32_t ncell;         // 0: no table (template does not fit LDS)
    int32_t cell_off;      // offset of this template's start table in the table buffer
    // k-d patch layout: the first kd_split patches are the left half of the root split, the others the right half; the
    // tight boxes of the two halves let the wave-per-query search skip the box tests of a half a query cannot reach
    int32_t kd_split;
    int32_t pad;
    float half_lo[2][4], half_hi[2][4];
    // templates that do not fit LDS are searched chunk by chunk: the chunks are whole k-d subtrees of at most ICP_TPL_LDS
    // points (compact regions), with their boxes, so that a workgroup only stages the chunks its queries can reach
    int32_t nchunk;        // 0: fixed chunks of ICP_TPL_LDS points (more than ICP_MAX_CHUNKS subtrees)
    int32_t chunk_start[ICP_MAX_CHUNKS], chunk_n[ICP_MAX_CHUNKS];
    int32_t pad2[3];
    float chunk_lo[ICP_MAX_CHUNKS][4], chunk_hi[ICP_MAX_CHUNKS][4];
};

// Second level over the k-d patch boxes of a template that does not fit LDS (k_icp_pipe_big): the k-d subtrees of at most 64
// patches whose parent is larger ("superpatches": at most 64 of them for 1024 patches), each a run of consecutive patches.
struct IcpSuper {
    int32_t n;
    int32_t first[64], cnt[64];   // first patch and number of patches
    float lo[64][4], hi[64][4];   // box
};

// A template that is a union of axis-aligned LATTICES (every make_cuboid.py template: face k = the Cartesian product of two
// of three shared axis tables X, Y, Z at a constant third coordinate, written first-axis-fastest, mkc.py:38-55).  Detected
// and verified bit by bit against the uploaded points by cd_set_template (lattice_detect, cuboid_hip.hip); the nearest
// neighbour of a query is then a closed form over the axis tables (k_icp_lat.hip) - no search structure, no template image.
constexpr int LAT_MAX_FACES = 6;           // (a cuboid has six)
constexpr int LAT_MAX_TAB = 512;           // axis table entries of one template, the three axes together
struct IcpLattice {
    int32_t nface;                 // 0: not a lattice - the generic searches of k_icp.hip take the template
    int32_t ntab;                  // entries of tab in use
    int32_t n[3];                  // entries of the axis tables X, Y, Z (0: no face varies along that axis)
    int32_t toff[3];               // first entry of each table in tab
    float noi[3], inv[3];          // -T[0] / step and 1 / step of each table (the tables are uniform to 1/16 of a step): entry guess = q * inv + noi
    int32_t w[LAT_MAX_FACES];      // the face's constant axis
    int32_t fast[LAT_MAX_FACES];   // the axis whose index varies fastest: index = base + i_slow * n[fast] + i_fast
    int32_t base[LAT_MAX_FACES];   // original index of the face's first point (faces are consecutive, ascending)
    float c[LAT_MAX_FACES];        // its constant coordinate
    uint32_t m0[LAT_MAX_FACES], m1[LAT_MAX_FACES], m2[LAT_MAX_FACES];   // all ones when w == 0 / 1 / 2 (the face loop's selects are v_bfi_b32 with these
                                   // words); faces beyond nface: a constant z = NaN, so that their "distance" is NaN and never compares below anything
    float4 tab[LAT_MAX_TAB];       // entry i of a table: (T[i-1], T[i], T[i+1], unused) with T[-1] = -inf, T[n] = +inf
};

struct IcpState {          // dynamic ICP state, double-buffered by launch parity
    float Tfinal[16];
    float T[16];           // transformation_ of the current iteration
    double prev_mse;
    int32_t iters;
    int32_t done;
    int32_t converged;
    int32_t status;
};

struct IcpWork {
    int32_t cluster, tile;
};

struct IcpParams {
    int32_t max_iter;
    float grid_rc;         // lane-per-query grid search for seed balls up to grid_rc cells wide (tuning only)
    double trans_eps, rel_mse, rot_thr, abs_mse;
    int32_t pipe_slots;    // k_icp_pipe / k_icp_pipe_big: clusters a workgroup keeps in flight (1 .. CD_PIPE_SLOTS; scheduling only)
    int32_t donate;        // 1: workgroups that run out of clusters wait and take over running ones (scheduling only; `don` valid)
    int32_t* don;          // hand-over control block, zeroed before the launch: DON_* words, then DON_CAP mailbox entries
    int32_t don_idle;      // tests: workgroups 0 .. don_idle-1 never take from the queue - everything they run reaches them by hand-over
    int32_t don_fault;     // tests: 1 = the first donor claims a mailbox entry and never publishes it (the loss path must be REPORTED)
};
// k_icp_pipe's hand-over of RUNNING clusters (IcpParams::donate): words of the control block
constexpr int DON_AVAIL = 0;      // workgroups waiting for a cluster minus clusters promised to them
constexpr int DON_FINISHED = 1;   // clusters of the launch that are done (or were never started: pre-marked)
constexpr int DON_TAIL = 2;       // mailbox entries written (or being written)
constexpr int DON_HEAD = 3;       // mailbox entries taken
constexpr int DON_ERR = 4;        // a waiter gave up on an entry it had claimed, or ran out of polls with clusters still open: the host fails the call
constexpr int DON_BOX = 16;       // first mailbox entry: cluster id + 1 (0 = not written yet)
constexpr int DON_CAP = 2048;     // entries (a launch hands over a few hundred clusters at most; beyond the cap nothing is handed over)

// ---------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------
#ifdef __HIPCC__
// Wave priority of the FRONT-END kernels (crop .. clusters).  With batches in flight their waves share SIMDs with the ICP's, whose
// waves nearly always have a vector instruction ready: a SIMD picks among ready waves by priority first, so at equal priority
// every front-end instruction queues behind the ICP's (tools/probe_interference.py: a resident FMA chain slows the front end
// 3.5 x, parked waves of the same footprint 1.1 x).  The front end is latency-bound and short, the ICP a long chain that is not
// what bounds the pipeline: the front end goes first (s_setprio 3; the ICP kernels stay at 0).  -DCD_NO_FRONT_PRIO: A/B.
#ifndef CD_NO_FRONT_PRIO
#define CD_FRONT_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define CD_FRONT_PRIO()
#endif
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
// wave ballot of a BOOL: HIP's __ballot(int) first turns a lane mask that is already in SGPRs (e.g. a && of two compares)
// into a 0/1 vector and compares it again - two vector instructions per loop condition in the ICP search loops
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Debug build (-DCD_BOUNDS): CD_IN_RANGE(cond, code) is `cond`, and a false one is reported by a device printf (check number,
// workgroup, thread) - the access it guards is then skipped instead of faulting.  Without the flag: true.
#ifdef CD_BOUNDS
__device__ __forceinline__ bool cd_in_range(bool ok, unsigned code) {
    if (!ok) printf("cuboid_hip BOUNDS: check %u failed in workgroup %u thread %u\n", code, (unsigned)blockIdx.x, (unsigned)threadIdx.x);
    return ok;
}
#define CD_IN_RANGE(cond, code) cd::cd_in_range((cond), (code))
#else
#define CD_IN_RANGE(cond, code) true
#endif
// R rows of a wave's tile, ALL loaded before any is used: element first + j * 64 of an array of n >= 1 elements, the index
// clamped into the array instead of tested (callers still ignore rows past n).  A load inside `if (e < n)` is followed by
// s_waitcnt vmcnt(0) before the next row's load is even issued - one memory round trip per row, eight in a row
// (tools/isa_load_pattern.py shows which kernels have that shape).
template <int R, typename T>
__device__ __forceinline__ void load_rows_clamped(const T* __restrict__ a, int first, int n, T (&v)[R]) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        int e = first + j * WAVE;
        e = e < n ? e : n - 1;
        v[j] = a[e < 0 ? 0 : e];
    }
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    const uint32_t lane = threadIdx.x & 63;
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ long long fixq(float v, int shift) {
    // float->double exact, power-of-two scale exact, round to nearest even (C4)
    return __double2ll_rn(ldexp((double)v, shift));
}
// The same integer for |v| * 2^shift < 2^50, without the double -> int64 conversion (five float64 instructions): adding
// 1.5 * 2^52 to t = v * 2^shift rounds t to an integer (nearest, ties to even - the rounding of the addition itself) and
// leaves rint(t) + 2^51 in the sum's 52 mantissa bits; their low 51 bits are rint(t) in two's complement.
__device__ __forceinline__ unsigned long long fixq_fast(float v, int shift) {
    const double u = __dadd_rn(ldexp((double)v, shift), 6755399441055744.0);
    const unsigned long long b = (unsigned long long)__double_as_longlong(u);
    const int hi = __builtin_amdgcn_sbfe((int)(unsigned)(b >> 32), 0, 19);   // sign-extend bit 50 (bit 18 of the high word): v_bfe_i32
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)b;
}
// Adds the wave's per-lane 64-bit sums S[0..nsum) into dst[0..nsum) (LDS): an inclusive scan inside each row of 16 lanes - four
// steps of a 64-bit add whose first operand comes from the lane 1, 2, 4, 8 to the left through the DPP path
// (v_add_co_u32_dpp + v_addc_co_u32_dpp: no separate move, no LDS crossbar) - leaves the row totals in lanes 15, 31, 47, 63,
// which add them to dst with one LDS atomic per sum.  Two sums are interleaved per block so that every DPP read is >= 2 VALU
// instructions away from the write of its register, and every block starts with two wait states (the compiler schedules
// around inline asm without knowing that it reads through DPP).
#define CD_DPP_ADD2(a, b, ctrl)                                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                    \
                 "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"         \
                 "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"   \
                 "v_add_co_u32_dpp %2, vcc, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"         \
                 "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0"        \
                 : "+v"(a##lo), "+v"(a##hi), "+v"(b##lo), "+v"(b##hi)::"vcc")
__device__ __forceinline__ void wave_fold_to_lds(const unsigned long long (&S)[16], int nsum, unsigned long long* dst) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        if (k < nsum) {
            unsigned xlo = (unsigned)S[k], xhi = (unsigned)(S[k] >> 32), ylo = (unsigned)S[k + 1], yhi = (unsigned)(S[k + 1] >> 32);
            CD_DPP_ADD2(x, y, "row_shr:1");
            CD_DPP_ADD2(x, y, "row_shr:2");
            CD_DPP_ADD2(x, y, "row_shr:4");
            CD_DPP_ADD2(x, y, "row_shr:8");
            if ((lane & 15) == 15) {
                atomicAdd(&dst[k], ((unsigned long long)xhi << 32) | xlo);
                if (k + 1 < nsum) atomicAdd(&dst[k + 1], ((unsigned long long)yhi << 32) | ylo);
            }
        }
    }
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// canonical squared distance: (dx*dx + dy*dy) + dz*dz, no contraction
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
// A workgroup's TILE from an atomic ticket instead of blockIdx (round 4).  The chained scans below wait for tiles of the same
// frame with smaller ids; a workgroup that holds ticket t of a frame exists only after tickets 0..t-1 of that frame were
// drawn, by workgroups that are running or done - so every wait ends, whatever order the hardware starts a grid's
// workgroups in and whoever else shares the GPU (several contexts in flight: until round 3 the ids came from blockIdx and
// the argument rested on the dispatch order).  One counter PER FRAME, TICKET_PITCH ints apart: a single counter for the
// whole grid serialises 38 400 same-address atomics per launch (measured: crop + voxel 0.98 -> 1.37 ms); per frame it is
// ~150.  The workgroup that draws a frame's last ticket puts its counter back to zero for the next launch (the launches
// that share the counters are ordered on one stream).  One device-scope atomic round trip before the first load; ends
// with a workgroup barrier.  `s_ticket`: one int of LDS.
constexpr int TICKET_PITCH = 32;   // ints between the counters of two frames (128 bytes: one L2 line each)
__device__ __forceinline__ int take_ticket(int* counter, int tickets, int* s_ticket) {
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == tickets - 1) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ticket = CD_IN_RANGE(t >= 0 && t < tickets, 1u) ? t : 0;
    }
    __syncthreads();
    return *s_ticket;
}
// Exclusive prefix of `tot` over the tiles 0..tile-1 of one frame by a chained scan ("decoupled look-back"): a tile publishes
// its own total (flag 1), walks back over its predecessors adding totals until it meets an inclusive prefix (flag 2), and
// publishes its own inclusive prefix.  state words (zeroed before the launch): flag << 30 | value; tile q's word is
// state[q * stride] (one thread per workgroup for a scalar scan, one thread per bin for the radix scatter).
// A workgroup only ever waits for tiles with a smaller id, and the ids are tickets (take_ticket above): their holders are
// running or done, so the waits are finite by construction.  (The spin bound is a guard that cannot fire any more; it still
// turns a wait that does not end - a lost store, a faulted predecessor - into *gave_up = 1 instead of a hung GPU.)
__device__ __forceinline__ int chained_scan(int* state, int stride, int tile, int tot, int* gave_up) {
    unsigned* st = reinterpret_cast<unsigned*>(state);
    const unsigned FLAG_TOTAL = 1u << 30, FLAG_PREFIX = 2u << 30, VALUE = (1u << 30) - 1u;
    int excl = 0;
    if (tile > 0) {
        __hip_atomic_store(st + (size_t)tile * stride, FLAG_TOTAL | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = tile - 1; q >= 0; --q) {   // q == 0 always carries FLAG_PREFIX: the bound is a guard only
            unsigned v;
            int spins = 0;
            do {
                v = __hip_atomic_load(st + (size_t)q * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > (1 << 20)) { *gave_up = 1; v = FLAG_PREFIX; }
            } while ((v & (FLAG_TOTAL | FLAG_PREFIX)) == 0u);
            excl += (int)(v & VALUE);
            if (v & FLAG_PREFIX) break;
        }
    }
    __hip_atomic_store(st + (size_t)tile * stride, FLAG_PREFIX | (unsigned)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}
// The same for TWO counts at once (k_crop_runs: kept points and runs): a 64-bit word per tile, flag << 62 | second << 31 |
// first, both counts < 2^31.  Returns the exclusive prefixes through *excl_a / *excl_b.
__device__ __forceinline__ void chained_scan2(unsigned long long* state, int tile, int tot_a, int tot_b, int* excl_a, int* excl_b,
                                              int* gave_up) {
    const unsigned long long FLAG_TOTAL = 1ull << 62, FLAG_PREFIX = 2ull << 62, MASK = (1ull << 31) - 1ull;
    unsigned long long ea = 0ull, eb = 0ull;
    if (tile > 0) {
        __hip_atomic_store(state + tile, FLAG_TOTAL | ((unsigned long long)(unsigned)tot_b << 31) | (unsigned long long)(unsigned)tot_a,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = tile - 1; q >= 0; --q) {
            unsigned long long v;
            int spins = 0;
            do {
                v = __hip_atomic_load(state + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > (1 << 20)) { *gave_up = 1; v = FLAG_PREFIX; }
            } while ((v & (FLAG_TOTAL | FLAG_PREFIX)) == 0ull);
            ea += v & MASK;
            eb += (v >> 31) & MASK;
            if (v & FLAG_PREFIX) break;
        }
    }
    __hip_atomic_store(state + tile, FLAG_PREFIX | ((eb + (unsigned long long)(unsigned)tot_b) << 31) | (ea + (unsigned long long)(unsigned)tot_a),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *excl_a = (int)ea;
    *excl_b = (int)eb;
}
// canonical plane distance: |(a*x + b*y) + (c*z + d)|
__device__ __forceinline__ float plane_dist(float a, float b, float c, float d, float x, float y, float z) {
    return fabsf(__fadd_rn(__fadd_rn(__fmul_rn(a, x), __fmul_rn(b, y)), __fadd_rn(__fmul_rn(c, z), d)));
}
#endif

}  // namespace cd
