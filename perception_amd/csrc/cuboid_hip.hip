// cuboid_hip.hip - context, batch pipeline driver and the C-ABI of include/cuboid_hip.h.
//
// The whole chain runs on one HIP stream with every intermediate resident in HBM
// (frame-major arrays, pitch = points per frame).  The host only (a) sizes launches from a
// handful of per-frame scalars mirrored through pinned memory, (b) replays PCL's sequential
// RANSAC stop rule over the batched inlier counts, (c) solves the 3x3 plane-refit eigenproblem
// and (d) polls ICP completion.  There is no CPU compute fallback of any stage.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <condition_variable>
#include <random>
#include <vector>

#include "host_math.hpp"
#include "kernels.hpp"

using namespace cd;

struct cd_context {
    int device = 0;
    hipStream_t stream = nullptr;
    int N = 0, F = 0, T = 0;   // capacities: points per frame, frames, tiles per frame
    char err[512] = {0};
    // input staging (host-pointer API)
    char* d_in = nullptr;
    size_t d_in_bytes = 0;
    // per-frame scalars
    FrameState* d_fs = nullptr;
    FrameState* h_fs = nullptr;
    // ordered-compaction tile counters
    int *d_tileA = nullptr, *d_tileB = nullptr, *d_tileK = nullptr, *d_tileC = nullptr;   // (tileC: the centroid kernel's scan state - its own array, so that one launch can zero every array of a batch up front)
    bool batch_zeroed = false;      // a fused batch call has zeroed the scratch arrays of all its stages in one launch (zero_batch_arrays): the stages skip their own fills
    bool crop_two_pass = false;   // CUBOID_CROP_TWO_PASS=1: always the two-pass crop
    int icp_persist = 1;          // CUBOID_ICP_PERSIST=0: the sliced driver always in its multi-launch form; 2: the persistent
                                  // launch starts with its abort flag raised (tests the hand-over to the multi-launch form)
    // point buffers (float4 = x,y,z,rgb bits)
    float4 *d_cpt = nullptr, *d_vox = nullptr, *d_obj = nullptr, *d_src0 = nullptr, *d_src = nullptr;
    uint32_t *d_key[2] = {nullptr, nullptr}, *d_val[2] = {nullptr, nullptr}, *d_ghist = nullptr;
    int* d_sstate = nullptr;   // chained-scan state of the radix passes, [pass][F][tiles][256]
    unsigned long long* d_tile64 = nullptr;   // chained-scan state of k_crop_runs: (points, runs) per tile, [F][T]
    bool crop_runs = true;      // CUBOID_CROP_RUNS=0: the crop writes per-point keys and k_voxel_runs finds the runs (rounds 2-3)
    int* d_ticket = nullptr;   // ticket counters, one per frame (TICKET_PITCH ints apart), of the kernels that scan over tiles (take_ticket, common.hpp): zero between launches
    // RANSAC
    int* d_rnd = nullptr;
    float4* d_models = nullptr;
    int *d_valid = nullptr, *d_counts = nullptr, *d_active = nullptr;
    int *h_valid = nullptr, *h_counts = nullptr, *h_active = nullptr;
    float4 *d_model = nullptr, *h_model = nullptr, *h_models = nullptr;
    int *d_have = nullptr, *h_have = nullptr;
    unsigned long long *d_sums = nullptr, *h_sums = nullptr;
    // extract / cluster
    int *d_plane_idx = nullptr, *d_head = nullptr, *d_next = nullptr, *d_parent = nullptr, *d_csize = nullptr,
        *d_rank = nullptr, *d_cand = nullptr, *d_sizes = nullptr, *d_label = nullptr;
    // templates
    float4 *d_tpl = nullptr, *d_tlo = nullptr, *d_thi = nullptr;   // points + per-64-run boxes
    float4 *d_tplk = nullptr, *d_tlok = nullptr, *d_thik = nullptr;   // templates in k-d patch order (sliced path)
    IcpGrid* d_grid = nullptr;                                    // per template slot
    unsigned short* d_kdmap = nullptr;                            // k-d patch order -> cell-sorted position (resident templates)
    unsigned short* d_tcell = nullptr;                            // cell start tables, ICP_CELL_STRIDE entries per slot
    int* d_nn = nullptr;                                          // last NN index of every ICP source point
    float* d_d2 = nullptr;                                        // its squared distance
    int* d_queue = nullptr;                                       // ICP work queue heads (one per template group)
    int* d_don = nullptr;                                         // k_icp_pipe: hand-over control block + mailbox (common.hpp DON_*)
    int don_idle = 0, don_fault = 0;                              // CUBOID_ICP_DON_IDLE / CUBOID_ICP_DON_FAULT: tests of the hand-over path (IcpParams::don_idle / don_fault)
    int icp_donate = -1;                                          // CUBOID_ICP_DONATE: -1 auto (a call alone on the device), 0 never, 1 always
    int* d_wgtab = nullptr;                                       // k_icp_pipe: {first item, end item, queue} per workgroup
    int* h_wgtab = nullptr;                                       // its pinned staging copy (3 * 1024 ints)
    int* h_ctl = nullptr;                                         // pinned: control words of k_icp_persist going up [0..7], coming back [8]
    std::vector<IcpState> st_init;                                // initial ICP states of a persistent launch (kept in case it gives up)
    int force_stall = 0;                                          // CUBOID_FORCE_SCAN_STALL=n: the next n chained-scan checks report a stall (tests the retry)
    int scan_retries = 0;                                         // calls of this context that were redone because a chained scan stalled
    int persist_gave_up = 0;                                      // persistent launches of this context that handed over to the multi-launch loop
    int n_cu = 256;
    int icp_mode = 0;                                             // 0 auto, 1 sliced multi-launch, 2 whole-cluster kernel
    int icp_max_wg = 0;                                           // > 0: cap on the persistent ICP grid (CUBOID_ICP_MAX_WG; tests force slot refills with it)
    int icp_cpw = 0;                                              // CUBOID_ICP_CPW: clusters per workgroup a persistent ICP launch is sized for (0: by regime, stage_icp)
    bool fs_initialised = false;                                  // the device FrameStates hold their initial value (set by the zero launch of a fused batch call)
    int icp_direct = 1;                                           // CUBOID_ICP_DIRECT=0: an all-lattice ICP stage uploads its lists and reads its results back by copy launches
    int mirror_reads = 1;                                         // CUBOID_MIRROR_READS=0: active flags and chosen plane models are uploaded before the kernels that read them
    int mirror_writes = 1;                                        // CUBOID_MIRROR_WRITES=0: every host read-back of the FrameState array is a copy launch again
    int cluster_cells = 1;                                        // CUBOID_CLUSTER_CELLS=0: frames above 8192 object points straight to the point-graph kernels (rounds 1-5)
    int crop_direct = 1;                                          // CUBOID_CROP_DIRECT=0: the crop always copies the kept points (rounds 1-5)
    int centroid_lanes = 1;                                       // CUBOID_CENTROID_LANES=0: the quad-per-voxel centroid kernel (rounds 3-5) instead of a lane per voxel
    bool voxel_runs = true;                                       // CUBOID_VOXEL_RUNS=0: S1 sorts the cropped points instead of their runs of equal voxel index
    int icp_slots = 0;                                            // CUBOID_ICP_SLOTS: clusters in flight per workgroup, 1 .. CD_PIPE_SLOTS (0: by regime)
    int icp_big_weight = 0;                                       // workgroup share of a template in global memory, per point (CUBOID_ICP_BIG_WEIGHT; 0 = by the launch's regime, measured on config 5)
    int *d_order = nullptr, *h_order = nullptr;                   // clusters, largest first
    int tpl_cap = 0, tpl_used = 0;
    std::shared_ptr<const struct PreparedTemplate> tpl_prep[CD_MAX_TEMPLATES];   // host copies (shared across contexts)
    int tpl_off[CD_MAX_TEMPLATES] = {0}, tpl_m[CD_MAX_TEMPLATES] = {0};
    bool tpl_gridded[CD_MAX_TEMPLATES] = {false};                // slot has a cell start table
    bool tpl_big[CD_MAX_TEMPLATES] = {false};                    // slot does not fit LDS but has what k_icp_pipe_big needs (cell table, k-d map, superpatches)
    IcpSuper* d_super = nullptr;                                  // per template slot
    IcpLattice* d_lat = nullptr;                                  // per template slot: axis tables and faces of a lattice template (nface = 0: none)
    int tpl_faces[CD_MAX_TEMPLATES] = {0};                        // faces of the slot's lattice (0: the generic searches take it)
    int icp_lattice = 1;                                          // CUBOID_ICP_LATTICE=0: lattice templates take the generic searches too (A/B, fallback tests)
    int zero_once = 1;                                            // CUBOID_ZERO_ONCE=0: every stage of a fused batch call fills its scratch arrays itself (A/B)
    int copy_kernels = 1;                                         // CUBOID_COPY_KERNELS=0: the small pinned <-> device transfers go through hipMemcpyAsync (SDMA) again
    int lat_shape[3] = {0, 0, 0};                                 // CUBOID_LAT_SHAPE=cpw,wpc[,per_slot]: clusters per workgroup, waves per cluster, clusters per slot of k_icp_lat (0: by regime)
    hipStream_t stream2 = nullptr, stream3 = nullptr;             // streams of the persistent ICP launches: the second launch of a mixed-template batch runs beside the
                                                                  // first (stream2); with icp_lowprio both are low-priority streams, so that CUs that come free go to the
                                                                  // short front-end kernels of the other batches in flight before the next persistent workgroup
    int front_concurrent = 0;                                     // CUBOID_FRONT_CONCURRENT: at most that many fused batch calls of the device between crop and clusters (0 = no gate)
    int icp_concurrent = 0;                                       // CUBOID_ICP_CONCURRENT: admission gate of the whole-cluster ICP launches (0 = none)
    int icp_lowprio = 1;                                          // CUBOID_ICP_LOWPRIO: 0 never, 1 the launches of a mixed-template batch (measured: config 5 +30 %), 2 every
                                                                  // persistent ICP launch (config 3: -1 %)
    hipEvent_t ev2[3] = {nullptr, nullptr, nullptr};
    // ICP
    IcpCluster *d_cl = nullptr, *h_cl = nullptr;
    IcpWork *d_work = nullptr, *h_work = nullptr, *d_work2 = nullptr, *h_work2 = nullptr;
    int work_cap = 0;
    int cl_cap = 0;                                               // ICP problems the cluster arrays hold (grown on demand)
    int* d_koffx = nullptr;                                       // offsets of the clusters ranked >= KICP, [round][F][KICP]
    size_t koffx_cap = 0;
    std::vector<cd_cluster_result> last_clusters;                 // every cluster result of the last batch, frame-major
    std::vector<int> last_first;                                  // index of frame f's first cluster in it (F + 1 entries)
    // clouds of the last batch that stay resident for cd_get_frame_cloud / cd_get_cluster_points
    std::vector<int> last_nv, last_no;                            // voxels / object points per frame
    std::vector<long long> last_orig_off, last_al_off;            // per cluster: offset of its points in d_src0; of its aligned points in d_src (-1: not resident)
    bool last_clouds = false;
    // initial guesses (cd_set_frame_guesses), and their device copy (also used for the single guess of cd_params)
    std::vector<float> frame_guess;
    float* d_guess = nullptr;
    size_t guess_cap = 0;
    IcpState *d_st = nullptr, *h_st = nullptr;
    unsigned long long *d_acc = nullptr, *d_accf = nullptr, *h_accf = nullptr;
    hipEvent_t ev[8] = {nullptr};
    cd_timing timing;
};

namespace {

#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                                   \
            return CD_ERR_DEVICE;                                                                           \
        }                                                                                                   \
    } while (0)

// kernel launches report a bad configuration (grid, LDS size, arguments) through hipGetLastError only: name the kernel
#define LAUNCH(ctx, call)                                                                                     \
    do {                                                                                                      \
        call;                                                                                                 \
        hipError_t e_ = hipGetLastError();                                                                    \
        if (e_ != hipSuccess) {                                                                               \
            snprintf((ctx)->err, sizeof((ctx)->err), "launch failed: %s: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                                     \
            return CD_ERR_DEVICE;                                                                             \
        }                                                                                                     \
    } while (0)

// the per-batch read-backs (cd_get_cluster_results, cd_get_frame_cloud, cd_get_cluster_points) describe the last fused call;
// every other compute call reuses the same device buffers
void invalidate_last(cd_context* c) {
    c->last_clouds = false;
    c->last_first.clear();
}

int fail(cd_context* c, int code, const char* msg) {
    snprintf(c->err, sizeof(c->err), "%s", msg);
    return code;
}

// ---- chained scans and several contexts on one GPU ---------------------------------------------------------------------------
// A chained scan (crop, radix scatters, voxel heads, cd_extract) waits for tiles with smaller ids.  Since round 4 the ids are
// atomic tickets (take_ticket, common.hpp): the holder of a ticket has started, so a wait can only be for a workgroup that is
// running or done - finite by construction, whatever else shares the GPU.  (Rounds 1-3 took the ids from blockIdx and relied
// on the order in which an XCD starts a grid's workgroups; with several contexts in flight that argument had a hole.)  What is
// left of the old answer to that hole is a safety net that cannot trigger on its own: the waits are still bounded
// (common.hpp), a kernel that gives up reports scan_stalled, and the call is REDONE ALONE - every compute call holds this
// per-device lock shared, the redo exclusively.  CUBOID_FORCE_SCAN_STALL exercises the path.
// The redo must not starve: libstdc++'s shared_mutex is a reader-preferring pthread_rwlock, and with five contexts calling
// back to back some reader nearly always holds it.  So every call first passes a turnstile (a plain mutex, taken and
// released at once); a redo holds the turnstile while it waits for the exclusive lock - new calls queue behind it, the
// calls in flight drain, the redo runs, the queue moves on (tests/test_gpu_readback_guess.py, saturated pipeline).
constexpr int CD_INTERNAL_STALL = -100;   // never leaves the library
constexpr int MAX_DEVICES = 16;
std::shared_mutex g_scan_mu[MAX_DEVICES];
std::mutex g_turnstile[MAX_DEVICES];
std::atomic<int> g_calls_in_flight[MAX_DEVICES];

std::atomic<int> g_batches_in_flight[MAX_DEVICES];   // fused batch calls only: what the launch regime of the whole-cluster ICP kernel looks at
struct BatchGuard {  // (a cd_bbox_filter or cd_extract beside a batch must not change the shape of its ICP launch)
    int dev;
    explicit BatchGuard(int d) : dev(d & (MAX_DEVICES - 1)) { g_batches_in_flight[dev].fetch_add(1); }
    ~BatchGuard() { g_batches_in_flight[dev].fetch_sub(1); }
};
// Admission gate of the whole-cluster ICP launches (CUBOID_ICP_CONCURRENT = K; 0 = none): at most K contexts of a device are
// between the launch of their persistent ICP kernel and its completion.  The launches of several batches otherwise share the
// CUs workgroup by workgroup (processor sharing: five batches submitted together all finish late, together); through the gate
// they run K at a time, first come first served, so the first batches of a burst come back early and their contexts refill
// the pipeline.
struct IcpGate {
    std::mutex mu;
    std::condition_variable cv;
    int inside = 0;
    void enter(int k) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return inside < k; }); ++inside; }
    void leave() { { std::lock_guard<std::mutex> lk(mu); --inside; } cv.notify_one(); }
};
IcpGate g_icp_gate[MAX_DEVICES];
IcpGate g_front_gate[MAX_DEVICES];   // the same for the front end (crop .. clusters) of the fused batch calls: CUBOID_FRONT_CONCURRENT
struct GateHold {   // (released on every path out of its scope)
    IcpGate* g = nullptr;
    void enter(IcpGate* gate, int k) { g = gate; g->enter(k); }
    void release() { if (g) { g->leave(); g = nullptr; } }
    ~GateHold() { release(); }
};
struct CallGuard {   // one per compute call: counts the contexts at work on the device (k_icp_persist wants the chip to itself)
    int dev;
    explicit CallGuard(int d) : dev(d & (MAX_DEVICES - 1)) { g_calls_in_flight[dev].fetch_add(1); }
    ~CallGuard() { g_calls_in_flight[dev].fetch_sub(1); }
};

template <class Fn>
int with_scan_retry(cd_context* c, Fn&& fn) {
    const int dev = c->device & (MAX_DEVICES - 1);
    int st;
    {
        { std::lock_guard<std::mutex> pass(g_turnstile[dev]); }   // held by a pending redo: wait behind it
        std::shared_lock<std::shared_mutex> lk(g_scan_mu[dev]);
        CallGuard g(dev);
        st = fn();
    }
    if (st != CD_INTERNAL_STALL) return st;
    if (std::getenv("CUBOID_DEBUG")) std::fprintf(stderr, "cuboid_hip: %s - redoing the call with the device to itself\n", c->err);
    c->scan_retries += 1;
    hipStreamSynchronize(c->stream);
    {
        std::lock_guard<std::mutex> hold(g_turnstile[dev]);          // no new call starts until this redo is done
        std::unique_lock<std::shared_mutex> lk(g_scan_mu[dev]);      // ... and the calls in flight have drained
        CallGuard g(dev);
        st = fn();
    }
    c->timing.scan_retries = 1;
    if (st == CD_INTERNAL_STALL) return fail(c, CD_ERR_DEVICE, "a chained scan stalled twice, the second time with the device to itself");
    return st;
}

#ifdef CD_ALLOC_GUARD
// Debug build (make VARIANT=guard FLAGS_EXTRA=-DCD_ALLOC_GUARD=262144): every device allocation of the library gets that many
// bytes of 0xA5 behind it, checked when it is freed - a kernel that WRITES past the end of an array is named on stderr instead
// of corrupting its neighbour (or faulting only when the neighbour happens to be unmapped); a kernel that only READS past the
// end stops faulting and leaves the guards intact, which says as much.
struct GuardEntry { void* p; size_t bytes; std::string name; };
static std::mutex g_guard_mu;
static std::vector<GuardEntry> g_guards;
static hipError_t guard_malloc(void** p, size_t bytes, const char* name) {
    const hipError_t e = hipMalloc(p, bytes + (size_t)CD_ALLOC_GUARD);
    if (e != hipSuccess) return e;
    (void)hipMemset((char*)*p + bytes, 0xA5, (size_t)CD_ALLOC_GUARD);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guards.push_back(GuardEntry{*p, bytes, name});
    return hipSuccess;
}
static hipError_t guard_free(void* p) {
    GuardEntry ge{nullptr, 0, ""};
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        for (size_t i = 0; i < g_guards.size(); ++i)
            if (g_guards[i].p == p) { ge = g_guards[i]; g_guards.erase(g_guards.begin() + (long)i); break; }
    }
    if (ge.p) {
        std::vector<unsigned char> h((size_t)CD_ALLOC_GUARD);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(h.data(), (char*)p + ge.bytes, h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            size_t first = h.size(), last = 0, bad = 0;
            for (size_t i = 0; i < h.size(); ++i) if (h[i] != 0xA5) { if (first == h.size()) first = i; last = i; ++bad; }
            if (bad) std::fprintf(stderr, "cuboid_hip GUARD: %s (%zu bytes) was written past its end: %zu bytes between +%zu and +%zu\n", ge.name.c_str(), ge.bytes, bad, first, last);
        }
    }
    return (hipFree)(p);
}
#define hipFree(p) guard_free(p)
template <class Tp>
hipError_t dalloc_named(Tp** p, size_t n, const char* name) { return guard_malloc((void**)p, std::max<size_t>(n, 1) * sizeof(Tp), name); }
#define dalloc(p, n) dalloc_named(p, n, #p)
#else
template <class Tp>
hipError_t dalloc(Tp** p, size_t n) { return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(Tp)); }
#endif
template <class Tp>
hipError_t halloc(Tp** p, size_t n) { return hipHostMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(Tp), hipHostMallocDefault); }

const int FS_PITCH = (int)(sizeof(FrameState) / sizeof(int));
#define FS_FIELD(ctx, field) ((int*)((char*)(ctx)->d_fs + offsetof(FrameState, field)))

// blocking copy ordered on the context's own (non-blocking) stream: the NULL stream gives no ordering against it
static hipError_t copy_sync(cd_context* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(c->stream);
}

// a small transfer between a PINNED host mirror of the context and device memory, as a kernel on the context's stream
// (launch_copy_rows, k_plane.hip: why not hipMemcpyAsync).  bytes: a multiple of 4.  CUBOID_COPY_KERNELS=0: hipMemcpyAsync (A/B).
static hipError_t xfer(cd_context* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (!c->copy_kernels || (bytes & 3)) return hipMemcpyAsync(dst, src, bytes, kind, c->stream);
    launch_copy_rows(c->stream, dst, bytes, src, bytes, bytes, 1);
    return hipGetLastError();
}
static hipError_t xfer2d(cd_context* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int rows, hipMemcpyKind kind) {
    if (!c->copy_kernels || (width & 3) || (dpitch & 3) || (spitch & 3)) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, c->stream);
    launch_copy_rows(c->stream, dst, dpitch, src, spitch, width, rows);
    return hipGetLastError();
}

// a stage's zero-fill of one of its scratch arrays - skipped when the fused batch call has zeroed them all in one launch
#define ZERO_FILL(ctx, ptr, bytes)                                                            \
    do {                                                                                      \
        if (!(ctx)->batch_zeroed) HIPCHK(ctx, hipMemsetAsync((ptr), 0, (bytes), (ctx)->stream)); \
    } while (0)

// consecutive small transfers of a stage collected into one launch (xfer's fallback applies: without copy kernels, or for a
// width that is no multiple of 4, each goes through the runtime's copy as before)
struct XferBatch {
    cd_context* c;
    CopyList L;
    hipError_t err = hipSuccess;
    explicit XferBatch(cd_context* ctx) : c(ctx) { L.n = 0; }
    void add2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int rows, hipMemcpyKind kind) {
        if (err != hipSuccess || width == 0 || rows <= 0) return;
        if (!c->copy_kernels || (width & 3) || (dpitch & 3) || (spitch & 3)) { err = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, c->stream); return; }
        if (L.n == 8) flush();
        L.seg[L.n++] = CopySeg{(uint32_t*)dst, (const uint32_t*)src, dpitch / 4, spitch / 4, (int)(width / 4), rows};
    }
    void add(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) { add2d(dst, bytes, src, bytes, bytes, 1, kind); }
    hipError_t flush() {
        if (L.n > 0 && err == hipSuccess) { launch_copy_list(c->stream, L); err = hipGetLastError(); }
        L.n = 0;
        return err;
    }
};

int ensure_input(cd_context* c, size_t bytes) {
    if (bytes <= c->d_in_bytes) return CD_OK;
    if (c->d_in) hipFree(c->d_in);
    c->d_in = nullptr;
    c->d_in_bytes = 0;
#ifdef CD_ALLOC_GUARD
    HIPCHK(c, guard_malloc((void**)&c->d_in, bytes, "d_in"));
#else
    HIPCHK(c, hipMalloc((void**)&c->d_in, bytes));
#endif
    c->d_in_bytes = bytes;
    return CD_OK;
}

// cluster-indexed ICP arrays: sized for F * KICP problems at cd_create, re-allocated when a batch holds more
// (frames with more than KICP clusters) - no cluster is dropped
int ensure_clusters(cd_context* c, int ncl, long long points) {
    const long long work_need = points / 64 + (long long)ncl + 16;
    if (ncl <= c->cl_cap && work_need <= c->work_cap) return CD_OK;
    if (work_need > 0x7fffffffll) return fail(c, CD_ERR_CAPACITY, "ICP work list exceeds 2^31 items");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->batch_zeroed = false;   // (new arrays: stage_icp fills them itself)
    void* dev[] = {c->d_cl, c->d_order, c->d_work, c->d_work2, c->d_st, c->d_acc, c->d_accf};
    for (void* p : dev) if (p) hipFree(p);
    void* host[] = {c->h_cl, c->h_order, c->h_work, c->h_work2, c->h_st, c->h_accf};
    for (void* p : host) if (p) hipHostFree(p);
    c->d_cl = nullptr; c->d_order = nullptr; c->d_work = nullptr; c->d_work2 = nullptr; c->d_st = nullptr; c->d_acc = nullptr; c->d_accf = nullptr;
    c->h_cl = nullptr; c->h_order = nullptr; c->h_work = nullptr; c->h_work2 = nullptr; c->h_st = nullptr; c->h_accf = nullptr;
    // never shrink: the per-stage entry points (cd_icp) rely on the capacity cd_create gave them
    const size_t n = (size_t)std::max(std::max(ncl, c->F * KICP) + ncl / 4, c->cl_cap), w = (size_t)std::max<long long>(work_need + work_need / 4, c->work_cap);
    c->cl_cap = 0; c->work_cap = 0;
    HIPCHK(c, dalloc(&c->d_cl, n)); HIPCHK(c, halloc(&c->h_cl, n));
    HIPCHK(c, dalloc(&c->d_order, n)); HIPCHK(c, halloc(&c->h_order, n));
    HIPCHK(c, dalloc(&c->d_work, w)); HIPCHK(c, halloc(&c->h_work, w));
    HIPCHK(c, dalloc(&c->d_work2, w)); HIPCHK(c, halloc(&c->h_work2, w));
    HIPCHK(c, dalloc(&c->d_st, n * 2)); HIPCHK(c, halloc(&c->h_st, n * 2));
    HIPCHK(c, dalloc(&c->d_acc, n * 48)); HIPCHK(c, dalloc(&c->d_accf, n + 1)); HIPCHK(c, halloc(&c->h_accf, n + 1));   // (+ 1: k_icp_lat's wave-time word)
    c->cl_cap = (int)n;
    c->work_cap = (int)w;
    return CD_OK;
}

int sync_fs(cd_context* c, int F, bool copied = false) {   // copied: the caller has put the FrameState read-back on the stream already
    if (!copied) HIPCHK(c, xfer(c, c->h_fs, c->d_fs, sizeof(FrameState) * F, hipMemcpyDeviceToHost));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int f = 0; f < F; ++f)
        if (c->h_fs[f].scan_stalled) return fail(c, CD_INTERNAL_STALL, "a chained scan stalled (workgroups of a grid were not started in id order)");
    if (c->force_stall > 0) { c->force_stall -= 1; return fail(c, CD_INTERNAL_STALL, "a chained scan stalled (forced: CUBOID_FORCE_SCAN_STALL)"); }
    return CD_OK;
}

// ---- stage drivers (device-resident in/out; F frames) ------------------------------------

// S0+S1.  in: F*N records of `stride` bytes on the device.  out: d_vox / fs.n_v
int stage_crop_voxel(cd_context* c, const void* d_in, size_t stride, int N, int F, const cd_params* p, int* rounds_out) {
    const int T = c->T;
    CropLimits lim;
    lim.zlo = hm::fold_ge(p->crop_z_min);
    lim.zhi = hm::fold_le(p->crop_z_max);
    lim.xlo = hm::fold_ge(p->crop_x_min);
    lim.xhi = hm::fold_le(p->crop_x_max);
    const int rgb_off = (p->rgb_offset >= 0 && (size_t)p->rgb_offset + 4 <= stride) ? p->rgb_offset : -1;
    // FrameState init: status 0, mn = +max, mx = 0 (ordered-uint encoding)
    for (int f = 0; f < F; ++f) {
        FrameState& s = c->h_fs[f];
        std::memset(&s, 0, sizeof(s));
        for (int a = 0; a < 3; ++a) { s.mn[a] = 0xffffffffu; s.mx[a] = 0u; }
    }
    // Single pass (one read of the input) when the crop limits bound the x and z cell indices tightly enough to leave the
    // unbounded y index a wide bit field; a frame whose y does not fit anyway sends the batch through the two-pass path.
    KeyPack kp;
    std::memset(&kp, 0, sizeof(kp));
    {
        const float inv = 1.0f / p->leaf_size;
        const float fl[4] = {std::floor(lim.xlo * inv), std::floor(lim.xhi * inv), std::floor(lim.zlo * inv), std::floor(lim.zhi * inv)};
        bool ok = !c->crop_two_pass;
        for (float v : fl) ok = ok && std::fabs(v) < 1.0e9f;
        if (ok && fl[1] >= fl[0] && fl[3] >= fl[2]) {
            const long long ri = (long long)fl[1] - (long long)fl[0] + 1, rk = (long long)fl[3] - (long long)fl[2] + 1;
            int bi = 1, bk = 1;
            while ((1ll << bi) < ri) ++bi;
            while ((1ll << bk) < rk) ++bk;
            if (bi + bk <= 20) {
                kp.enabled = 1;
                kp.bi = bi; kp.bj = 32 - bi - bk;
                kp.ilo = (int)fl[0]; kp.klo = (int)fl[2]; kp.jlo = -(1 << (kp.bj - 1));
                // y = 0 sits 256 cells above an aligned block of 512 field values: a frame whose y cells stay within +-256 of the
                // optical axis (1.28 m at the 5 mm leaf) then has constant y bits above the ninth, and the sort of k_crop_runs,
                // which runs on these packed keys, skips the digit they fill (3 passes instead of 4 on the bench frames; with y = 0
                // ON a block boundary every frame straddles it)
                if (kp.bj >= 11) kp.jlo -= 256;
            }
        }
    }
    // Round 4: the single-pass crop also writes the RUNS (one record per run of equal cell key inside a row of 64 input points)
    // and their digit histograms, and the sort runs on the packed cell keys themselves (k_crop_runs, k_voxel.hip)
    bool crop_runs = kp.enabled && c->crop_runs && c->voxel_runs && c->N <= (1 << 20);
    // 16-byte records x y z rgb (the D435 driver's layout): k_crop_runs leaves the kept points where they are and the centroid
    // kernel reads the input (0.5 GB less written and the same bytes read per 256-frame batch); any other layout is copied
    const bool direct_pts = c->crop_direct && stride == 16 && (rgb_off == 12 || rgb_off < 0) && (reinterpret_cast<uintptr_t>(d_in) & 15u) == 0;
    int st = CD_OK;
    // (the ticket counters reset themselves at the end of every launch that uses them; zeroed here too so that a call that
    // failed half way can never leave the next one with a counter that is not zero)
    ZERO_FILL(c, c->d_ticket, sizeof(int) * (size_t)F * TICKET_PITCH);
    // (kernels write the pinned mirror of the FrameState array themselves where a whole copy launch used to follow them:
    // CUBOID_MIRROR_WRITES=0 restores the copies)
    FrameState* const fs_mirror = c->mirror_writes && c->copy_kernels ? c->h_fs : nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!(attempt == 0 && c->fs_initialised)) HIPCHK(c, xfer(c, c->d_fs, c->h_fs, sizeof(FrameState) * F, hipMemcpyHostToDevice));   // (a fused call's zero launch set them)
        c->fs_initialised = false;
        // (the rare redo, and the crops that use d_tileA themselves: from here on every stage fills its own arrays again)
        if (attempt > 0 || !(kp.enabled && crop_runs)) c->batch_zeroed = false;
        ZERO_FILL(c, c->d_tileA, sizeof(int) * (size_t)F * T);
        if (kp.enabled && crop_runs) {
            ZERO_FILL(c, c->d_tile64, sizeof(unsigned long long) * (size_t)F * T);
            ZERO_FILL(c, c->d_ghist, sizeof(uint32_t) * (size_t)F * SORT_MAX_PASSES_HOST * RADIX);
            LAUNCH(c, launch_crop_runs(c->stream, d_in, stride, N, c->N, F, rgb_off, lim, T, p->leaf_size, kp, c->d_fs, c->d_tile64, c->d_cpt, c->d_key[0], c->d_val[0],
                             c->d_ghist, c->d_ticket, direct_pts ? 1 : 0));
            LAUNCH(c, launch_voxel_setup(c->stream, c->d_fs, F, p->leaf_size, c->d_ghist, fs_mirror));
        } else if (kp.enabled) {
            LAUNCH(c, launch_crop_fused(c->stream, d_in, stride, N, c->N, F, rgb_off, lim, T, p->leaf_size, kp, c->d_fs, c->d_tileA, c->d_cpt, c->d_key[0], c->d_ticket));
            LAUNCH(c, launch_voxel_setup(c->stream, c->d_fs, F, p->leaf_size, nullptr, fs_mirror));
        } else {
            LAUNCH(c, launch_crop_count(c->stream, d_in, stride, N, F, rgb_off, lim, T, c->d_fs, c->d_tileA));
            LAUNCH(c, launch_scan_tiles(c->stream, c->d_tileA, F, T, FS_FIELD(c, n_c), FS_PITCH));
            LAUNCH(c, launch_voxel_setup(c->stream, c->d_fs, F, p->leaf_size, nullptr, fs_mirror));
            LAUNCH(c, launch_crop_compact(c->stream, d_in, stride, N, c->N, F, rgb_off, lim, T, p->leaf_size, c->d_fs, c->d_tileA, c->d_cpt, c->d_key[0]));
        }
        st = sync_fs(c, F, fs_mirror != nullptr);   // sync #1: n_c, key_bits (sort pass count); the mirror was written by k_voxel_setup
        if (st) return st;
        bool over = false;
        for (int f = 0; f < F; ++f) over = over || c->h_fs[f].crop_overflow != 0;
        if (!over) break;
        if (std::getenv("CUBOID_DEBUG")) std::fprintf(stderr, "cuboid_hip: single-pass crop overflowed, redoing the batch in two passes\n");
        kp.enabled = 0;       // rare: redo the crop in two passes (the FrameState init in h_fs was overwritten by the sync)
        crop_runs = false;
        for (int f = 0; f < F; ++f) {
            FrameState& s = c->h_fs[f];
            std::memset(&s, 0, sizeof(s));
            for (int a = 0; a < 3; ++a) { s.mn[a] = 0xffffffffu; s.mx[a] = 0u; }
            }
    }
    int max_nc = 0, max_bits = 0;
    for (int f = 0; f < F; ++f) { max_nc = std::max(max_nc, c->h_fs[f].n_c); max_bits = std::max(max_bits, c->h_fs[f].key_bits); }
    const int Tc = std::max(1, (max_nc + TILE - 1) / TILE);
    const int Tsc = std::max(1, (max_nc + SORT_TILE - 1) / SORT_TILE);
    const int npass = (max_bits + RADIX_BITS - 1) / RADIX_BITS;
    int cur = 0;
    // By runs (default): the sort moves one element per run of equal voxel index among the cropped points - they are in image
    // order, 2.4 points per run on the bench frames - and the centroid kernel reads the runs' points contiguously; the bound
    // on the tiles is the point count (the run count is only known on the device).  CUBOID_VOXEL_RUNS=0: sort the points.
    const bool by_runs = c->voxel_runs && npass > 0 && c->N <= (1 << 20);   // (a run's start takes 20 bits of its payload)
    int Tc_runs = Tc;
    if (crop_runs) {
        // the runs and their histograms are there already; which of the packed key's four digits vary in some frame?
        int vary = 0, max_runs = 0;
        for (int f = 0; f < F; ++f) {
            if (c->h_fs[f].n_c > 0) vary |= c->h_fs[f].digit_vary;
            max_runs = std::max(max_runs, c->h_fs[f].n_runs);
        }
        int digits[4], nd = 0;
        for (int d = 0; d < 4; ++d) if ((vary >> d) & 1) digits[nd++] = d;
        const int Tsr = std::max(1, (max_runs + SORT_TILE - 1) / SORT_TILE);   // (the run count is known here: tighter than the point count)
        Tc_runs = std::max(1, (max_runs + TILE - 1) / TILE);
        LAUNCH(c, cur = launch_radix_scatter_runs(c->stream, c->d_key, c->d_val, c->N, F, Tsr, digits, nd, c->d_fs, c->d_ghist, c->d_sstate, c->d_ticket));
    } else if (by_runs)
        LAUNCH(c, cur = launch_radix_sort_runs(c->stream, c->d_key, c->d_val, c->N, F, T, Tsc, npass, c->d_fs, c->d_ghist, c->d_sstate, c->d_tileA, kp, c->d_ticket));
    else
        LAUNCH(c, cur = launch_radix_sort(c->stream, c->d_key, c->d_val, c->N, F, Tsc, npass, c->d_fs, c->d_ghist, c->d_sstate, kp, c->d_ticket));
    if (cur < 0) return fail(c, CD_ERR_DEVICE, "radix sort: the scan state could not be zeroed");
    const uint32_t* vin = c->d_val[cur];   // zero passes (empty frames only): the permutation is never read
    // voxel heads + centroids in one kernel: n_v (0 from the FrameState init for empty frames) and every tile's output
    // offset come from a chained scan (state in d_tileA)
    ZERO_FILL(c, c->d_tileC, sizeof(int) * (size_t)F * T);
    if (crop_runs || by_runs)
        LAUNCH(c, launch_voxel_centroid_runs(c->stream, c->d_key[cur], vin, crop_runs && direct_pts ? reinterpret_cast<const float4*>(d_in) : c->d_cpt, c->N, F, T, crop_runs ? Tc_runs : Tc,
                                             rgb_off >= 0 ? 1 : 0, c->d_fs, c->d_tileC, c->d_vox, c->d_ticket, c->centroid_lanes, crop_runs && direct_pts ? N : c->N));
    else
        LAUNCH(c, launch_voxel_centroid(c->stream, c->d_key[cur], vin, c->d_cpt, c->N, F, T, Tc, rgb_off >= 0 ? 1 : 0, c->d_fs, c->d_tileC, c->d_vox, c->d_ticket));
    if (rounds_out) *rounds_out = 0;
    return CD_OK;
}

// S2.  in: d_vox + fs.n_v (host mirror h_fs[].n_v must be current).  out: h_model/h_have refined,
// fs.status updated for NO_MODEL, iterations per frame.
int stage_plane(cd_context* c, int F, const cd_params* p, std::vector<int>& iterations, int* rounds_out) {
    const int T = c->T;
    const bool mr = c->mirror_reads != 0;
    const float thr = hm::fold_ge(p->plane_distance_threshold);
    int max_nv = 0;
    for (int f = 0; f < F; ++f) max_nv = std::max(max_nv, c->h_fs[f].n_v);
    const int Tv = std::max(1, (max_nv + TILE - 1) / TILE);
    std::vector<hm::RansacReplay> rep(F);
    iterations.assign(F, 0);
    for (int f = 0; f < F; ++f) c->h_active[f] = (c->h_fs[f].status == CD_OK || c->h_fs[f].status == CD_ERR_NO_MODEL) ? 1 : 0;
    ZERO_FILL(c, c->d_counts, sizeof(int) * (size_t)F * MAX_HYP);
    const int targets[4] = {16, 64, 256, MAX_HYP};
    const int h_cap = std::min(MAX_HYP, p->plane_max_iterations + 40);   // max_iterations+1 plus 39 skipped models
    int h_prev = 0, rounds = 0;
    for (int r = 0; r < 4; ++r) {
        const int h_target = std::min(targets[r], h_cap);
        if (h_target <= h_prev) break;
        // (mirror reads: the kernels read the few per-frame words the host decides - active flags, chosen models - straight from
        // the pinned host arrays, which the device sees; CUBOID_MIRROR_READS=0 uploads them first as rounds 1-5 did)
        if (!mr) HIPCHK(c, xfer(c, c->d_active, c->h_active, sizeof(int) * F, hipMemcpyHostToDevice));
        const int* active_p = mr ? c->h_active : c->d_active;
        LAUNCH(c, launch_ransac_sample(c->stream, c->d_vox, c->N, F, c->d_fs, c->d_rnd, h_target, active_p, c->d_models, c->d_valid));
        LAUNCH(c, launch_ransac_count(c->stream, c->d_vox, c->N, F, Tv, c->d_fs, c->d_models, c->d_valid, active_p, h_prev, h_target, thr, c->d_counts));
        // only the first h_target columns of the [F][MAX_HYP] tables are live: one strided copy each
        {   // (one launch: the three tables and - sync_fs below finds it done - nothing else; the FrameState read-back follows)
            XferBatch xb(c);
            xb.add2d(c->h_counts, sizeof(int) * MAX_HYP, c->d_counts, sizeof(int) * MAX_HYP, sizeof(int) * h_target, F, hipMemcpyDeviceToHost);
            xb.add2d(c->h_valid, sizeof(int) * MAX_HYP, c->d_valid, sizeof(int) * MAX_HYP, sizeof(int) * h_target, F, hipMemcpyDeviceToHost);
            xb.add2d(c->h_models, sizeof(float4) * MAX_HYP, c->d_models, sizeof(float4) * MAX_HYP, sizeof(float4) * h_target, F, hipMemcpyDeviceToHost);
            xb.add(c->h_fs, c->d_fs, sizeof(FrameState) * F, hipMemcpyDeviceToHost);
            HIPCHK(c, xb.flush());
        }
        int st = sync_fs(c, F, /*copied=*/true);   // sync #2 (per round): counts + n_hyp
        if (st) return st;
        ++rounds;
        bool all_done = true;
        for (int f = 0; f < F; ++f) {
            if (!c->h_active[f]) continue;
            const FrameState& s = c->h_fs[f];
            if (p->plane_model != CD_PLANE) {   // constrained models: a hypothesis off the axis constraint scores 0 inliers
                for (int h = rep[f].pos; h < s.n_hyp; ++h) {
                    const float4 m = c->h_models[(size_t)f * MAX_HYP + h];
                    const float mm[4] = {m.x, m.y, m.z, m.w};
                    if (c->h_valid[(size_t)f * MAX_HYP + h] && !hm::plane_model_valid(p->plane_model, mm, p->plane_axis, p->plane_eps_angle))
                        c->h_counts[(size_t)f * MAX_HYP + h] = 0;
                }
            }
            const bool fin = rep[f].consume(c->h_counts + (size_t)f * MAX_HYP, c->h_valid + (size_t)f * MAX_HYP, s.n_hyp, std::max(1, s.n_v),
                                            p->plane_max_iterations, p->plane_probability, s.sampler_exhausted != 0 || h_target >= h_cap);
            if (fin) c->h_active[f] = 0; else all_done = false;
        }
        h_prev = h_target;
        if (all_done) break;
    }
    if (rounds_out) *rounds_out = rounds;
    // chosen models -> device; moments of their inliers -> host eigen33 -> refined models
    for (int f = 0; f < F; ++f) {
        iterations[f] = rep[f].iterations;
        c->h_have[f] = rep[f].best_h >= 0 ? 1 : 0;
        c->h_model[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c->h_have[f]) c->h_model[f] = c->h_models[(size_t)f * MAX_HYP + rep[f].best_h];
    }
    // selectWithinDistance of a constrained model returns nothing when the model violates the constraint: the
    // device then sees "no model" (h_active doubles as the pinned staging copy), the host keeps reporting it
    auto upload_have = [&](const float4* models_too = nullptr) -> int {
        for (int f = 0; f < F; ++f) {
            const float mm[4] = {c->h_model[f].x, c->h_model[f].y, c->h_model[f].z, c->h_model[f].w};
            c->h_active[f] = c->h_have[f] && hm::plane_model_valid(p->plane_model, mm, p->plane_axis, p->plane_eps_angle) ? 1 : 0;
        }
        if (mr) return CD_OK;   // (the kernels read h_model / h_active)
        XferBatch xb(c);
        if (models_too) xb.add(c->d_model, models_too, sizeof(float4) * F, hipMemcpyHostToDevice);
        xb.add(c->d_have, c->h_active, sizeof(int) * F, hipMemcpyHostToDevice);
        HIPCHK(c, xb.flush());
        return CD_OK;
    };
    if (int st = upload_have(&c->h_model[0])) return st;   // (the chosen models ride along: one launch)
    if (p->plane_optimize) {
        ZERO_FILL(c, c->d_sums, sizeof(unsigned long long) * 10 * F);
        LAUNCH(c, launch_plane_cov(c->stream, c->d_vox, c->N, F, Tv, c->d_fs, mr ? c->h_model : c->d_model, mr ? c->h_active : c->d_have, thr, c->d_sums));
        HIPCHK(c, xfer(c, c->h_sums, c->d_sums, sizeof(unsigned long long) * 10 * F, hipMemcpyDeviceToHost));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // sync #3
        for (int f = 0; f < F; ++f) {
            if (!c->h_have[f]) continue;
            float in[4] = {c->h_model[f].x, c->h_model[f].y, c->h_model[f].z, c->h_model[f].w}, out[4];
            hm::plane_refit_from_moments((const uint64_t*)(c->h_sums + 10 * (size_t)f), in, out);
            c->h_model[f] = make_float4(out[0], out[1], out[2], out[3]);
        }
        if (!mr) HIPCHK(c, xfer(c, c->d_model, c->h_model, sizeof(float4) * F, hipMemcpyHostToDevice));
        if (p->plane_model != CD_PLANE) {
            HIPCHK(c, hipStreamSynchronize(c->stream));   // h_active is about to be rewritten
            if (int st = upload_have()) return st;
        }
    }
    (void)T;
    return CD_OK;
}

// S3 (+S3b).  out: d_plane_idx, d_obj, fs.n_plane, fs.n_o
int stage_extract(cd_context* c, int F, const cd_params* p, int gate_mode = -1) {
    const int T = c->T;
    const float thr = hm::fold_ge(p->plane_distance_threshold);
    int max_nv = 0;
    for (int f = 0; f < F; ++f) max_nv = std::max(max_nv, c->h_fs[f].n_v);
    const int Tv = std::max(1, (max_nv + TILE - 1) / TILE);
    const float z2lo = hm::fold_ge(p->crop2_z_min), z2hi = hm::fold_le(p->crop2_z_max);
    BBoxGate gate;
    std::memset(&gate, 0, sizeof(gate));
    gate.enable = gate_mode >= 0 ? gate_mode : (p->bbox_enable ? 1 : 0);
    for (int i = 0; i < 12; ++i) gate.P[i] = p->bbox_P[i];
    for (int i = 0; i < 4; ++i) gate.rect[i] = (float)p->bbox_rect[i];
    const float4* model_p = c->mirror_reads ? c->h_model : c->d_model;   // (see stage_plane)
    const int* have_p = c->mirror_reads ? c->h_active : c->d_have;
    ZERO_FILL(c, c->d_tileA, sizeof(int) * (size_t)F * T);
    ZERO_FILL(c, c->d_tileB, sizeof(int) * (size_t)F * T);
    LAUNCH(c, launch_plane_flag_count(c->stream, c->d_vox, c->N, F, T, Tv, c->d_fs, model_p, have_p, thr, p->extract_negative, p->crop2_enable, z2lo, z2hi, gate, c->d_tileA, c->d_tileB));
    {   // both scans in one launch; the two totals also go straight into the host's FrameState mirror (nothing else of it changes here)
        const bool mirrored = c->mirror_writes && c->copy_kernels;
        const ScanJob ja{c->d_tileA, FS_FIELD(c, n_plane), mirrored ? (int*)((char*)c->h_fs + offsetof(FrameState, n_plane)) : nullptr};
        const ScanJob jb{c->d_tileB, FS_FIELD(c, n_o), mirrored ? (int*)((char*)c->h_fs + offsetof(FrameState, n_o)) : nullptr};
        LAUNCH(c, launch_scan_tiles2(c->stream, ja, jb, F, T, FS_PITCH));
    }
    LAUNCH(c, launch_extract_scatter(c->stream, c->d_vox, c->N, F, T, Tv, c->d_fs, model_p, have_p, thr, p->extract_negative, p->crop2_enable, z2lo, z2hi, gate, c->d_tileA, c->d_tileB, c->d_plane_idx, c->d_obj));
    return CD_OK;
}

// S5.  in: d_obj + fs.n_o (device).  out: d_label, d_sizes, fs.n_k/ksize/koff, d_src0/d_src
int stage_cluster(cd_context* c, int F, const cd_params* p, int max_no_hint, bool force_global = false) {
    const int T = c->T;
    const int To = std::max(1, (max_no_hint + TILE - 1) / TILE);
    const float cell = (float)(p->cluster_tolerance * (1.0 + 1.0 / 1024.0));
    const float inv_cell = 1.0f / cell;
    const float r2 = (float)(p->cluster_tolerance * p->cluster_tolerance);
    if (p->cluster_enable) {
        // frames with <= 8192 object points (the usual case) are clustered by one workgroup in LDS, over cells small
        // enough (edge just under tol / sqrt(3)) that the points of one cell are connected by construction ...
        const float small_cell = (float)(p->cluster_tolerance / std::sqrt(3.0) * (1.0 - 1.0 / 1024.0));
        LAUNCH(c, launch_cluster_lds(c->stream, c->d_obj, c->N, F, c->d_fs, 1.0f / small_cell, r2, c->d_parent, c->d_csize, c->d_rank));
    }
    const bool by_cells = c->cluster_cells && !force_global;
    if (p->cluster_enable && max_no_hint > 8192 && by_cells) {
        // ... larger ones by the same cell graph with the points in global memory (k_cluster_cells; d_src is free until the labels
        // are scattered) ...
        const float small_cell = (float)(p->cluster_tolerance / std::sqrt(3.0) * (1.0 - 1.0 / 1024.0));
        LAUNCH(c, launch_cluster_cells(c->stream, c->d_obj, c->N, F, c->d_fs, 1.0f / small_cell, r2, c->d_parent, c->d_csize, c->d_rank, c->d_src));
    }
    if (p->cluster_enable && ((max_no_hint > 8192 && !by_cells) || force_global)) {
        // ... and the rare frame whose cells do not fit the table (fs.cl_done == 0) by the point-graph kernels in global memory
        HIPCHK(c, hipMemsetAsync(c->d_head, 0xff, sizeof(int) * (size_t)F * CELL_BUCKETS, c->stream));
        LAUNCH(c, launch_cluster_build(c->stream, c->d_obj, c->N, F, To, c->d_fs, inv_cell, c->d_head, c->d_next, c->d_parent, c->d_csize, c->d_rank));
        LAUNCH(c, launch_cluster_hook(c->stream, c->d_obj, c->N, F, To, c->d_fs, inv_cell, r2, c->d_head, c->d_next, c->d_parent));
        LAUNCH(c, launch_cluster_flatten(c->stream, c->N, F, To, c->d_fs, c->d_parent, c->d_csize));
    }
    LAUNCH(c, launch_cluster_rank(c->stream, c->N, F, c->d_fs, p->cluster_enable, p->cluster_min_size, p->cluster_max_size, c->d_parent, c->d_csize, c->d_cand, c->d_rank, c->d_sizes, c->mirror_writes && c->copy_kernels ? c->h_fs : nullptr));
    ZERO_FILL(c, c->d_tileK, sizeof(int) * (size_t)F * KICP * T);
    LAUNCH(c, launch_label_count(c->stream, c->N, F, T, To, c->d_fs, p->cluster_enable, c->d_parent, c->d_rank, c->d_label, c->d_tileK, 0));
    LAUNCH(c, launch_scan_tiles(c->stream, c->d_tileK, F * KICP, T, nullptr, 0));
    LAUNCH(c, launch_label_scatter(c->stream, c->d_obj, c->N, F, T, To, c->d_fs, c->d_label, c->d_tileK, c->d_src0, c->d_src, 0, nullptr));
    return CD_OK;
}

// stage_cluster + sync; when the LDS kernel gave a frame up (too many cells / too wide a cloud for its table) and the
// global-memory kernels were not part of the launch, the stage is run again with them.
int stage_cluster_sync(cd_context* c, int F, const cd_params* p, int max_no) {
    const bool mirrored = c->mirror_writes && c->copy_kernels;   // (k_cluster_rank wrote the FrameState mirror)
    int st = stage_cluster(c, F, p, max_no);
    if (st) return st;
    st = sync_fs(c, F, mirrored);
    if (st || !p->cluster_enable || (max_no > 8192 && !c->cluster_cells)) return st;   // (the point-graph kernels were part of the launch)
    bool left = false;
    for (int f = 0; f < F; ++f) left = left || (c->h_fs[f].n_o > 0 && !c->h_fs[f].cl_done);
    if (!left) return CD_OK;
    if (std::getenv("CUBOID_DEBUG")) std::fprintf(stderr, "cuboid_hip: LDS clustering gave frames up, running the global-memory path\n");
    c->batch_zeroed = false;   // (d_tileK has been used: the redo fills it again)
    st = stage_cluster(c, F, p, max_no, true);
    if (st) return st;
    return sync_fs(c, F, mirrored);
}

// S6.  clusters described by h_cl[0..ncl) (src_off relative to d_src/d_src0).  Fills h_st / h_accf.
int stage_icp(cd_context* c, int ncl, const cd_params* p, long long* pair_tests) {
    c->timing.icp_kernel_launches = 0;
    c->timing.icp_kernel_ms = 0.f;
    if (pair_tests) *pair_tests = 0;
    if (ncl <= 0) return CD_OK;
    // initial guess (pcl::Registration::align(output, guess)); the default - and the reference's live path - is none
    const int guess_mode = p->icp_use_guess;
    int max_n = 0;
    if (guess_mode != CD_GUESS_NONE) {
        int max_frame = 0;
        for (int k = 0; k < ncl; ++k) { max_n = std::max(max_n, c->h_cl[k].n); max_frame = std::max(max_frame, c->h_cl[k].frame); }
        const size_t need = guess_mode == CD_GUESS_PER_FRAME ? 16 * ((size_t)max_frame + 1) : 16;
        if (guess_mode == CD_GUESS_PER_FRAME && c->frame_guess.size() < need) return fail(c, CD_ERR_INVALID_ARG, "icp_use_guess = CD_GUESS_PER_FRAME but cd_set_frame_guesses holds fewer frames than the batch");
        if (need > c->guess_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_guess) hipFree(c->d_guess);
            c->d_guess = nullptr; c->guess_cap = 0;
            HIPCHK(c, dalloc(&c->d_guess, need));
            c->guess_cap = need;
        }
        HIPCHK(c, copy_sync(c, c->d_guess, guess_mode == CD_GUESS_PER_FRAME ? c->frame_guess.data() : p->icp_guess, sizeof(float) * need, hipMemcpyHostToDevice));
    }
    auto guess_of = [&](const IcpCluster& cl) -> const float* {
        return guess_mode == CD_GUESS_PER_FRAME ? c->frame_guess.data() + 16 * (size_t)cl.frame : p->icp_guess;
    };
    // queries per workgroup: 512 when the batch fills the chip, smaller slices (more workgroups) otherwise
    long long qtot = 0;
    for (int k = 0; k < ncl; ++k) qtot += c->h_cl[k].n;
    int qslice = (int)((qtot / 512 + 15) / 16 * 16);
    qslice = std::max(64, std::min(ICP_QSLICE, qslice));
    // Template groups: runs of clusters that share a template (one run per template when every cluster is matched against
    // every template).  With several templates in the batch, the groups whose template is LDS-resident and gridded go through
    // ONE k_icp_pipe launch (each group gets a share of the workgroups and its own queue) when there are enough of them to
    // fill a fifth of the chip; the other clusters take the sliced driver below.
    // kind: 0 = no persistent kernel can take the template, 1 = k_icp_pipe (LDS-resident, gridded), 2 = k_icp_pipe_big (global memory)
    struct TplGroup { int beg, end, live; int kind; long long pts; };
    auto kind_of = [&](const IcpCluster& cl) -> int {
        if (cl.tpl_m <= 0 || !c->tpl_gridded[cl.slot]) return 0;
        if (cl.tpl_m <= ICP_TPL_LDS) return 1;
        return c->tpl_big[cl.slot] ? 2 : 0;
    };
    // Clusters whose template is a lattice (every make_cuboid.py template; IcpLattice, common.hpp) go to k_icp_lat - closed-form
    // nearest neighbour, one workgroup per cluster - and take no part in the grouping below.  CUBOID_ICP_LATTICE=0 and the
    // forced driver modes (CUBOID_ICP_MODE) keep them on the generic searches (A/B, and the GPU suite runs under each mode).
    const bool use_lat = c->icp_lattice != 0 && c->icp_mode == 0;
    std::vector<char> in_lat((size_t)ncl, 0);
    int n_lat = 0, n_live = 0, lat_max_n = 0;
    for (int k = 0; k < ncl; ++k) {
        const IcpCluster& cl = c->h_cl[k];
        const bool live = cl.n >= 3 && cl.tpl_m > 0;
        n_live += live ? 1 : 0;
        if (use_lat && live && c->tpl_faces[cl.slot] > 0) { in_lat[(size_t)k] = 1; ++n_lat; lat_max_n = std::max(lat_max_n, cl.n); }
    }
    c->timing.icp_search = n_lat == 0 ? 0 : (n_lat == n_live ? 1 : 2);
    std::vector<TplGroup> groups;
    for (int k = 0; k < ncl; ++k) {
        const IcpCluster& cl = c->h_cl[k];
        if (groups.empty() || c->h_cl[groups.back().beg].tpl_off != cl.tpl_off || c->h_cl[groups.back().beg].tpl_m != cl.tpl_m)
            groups.push_back(TplGroup{k, k, 0, use_lat && c->tpl_faces[cl.slot] > 0 ? 0 : kind_of(cl), 0});
        TplGroup& g = groups.back();
        g.end = k + 1;
        if (cl.n >= 3) { g.live += 1; g.pts += cl.n; }
    }
    int n_grouped = 0;
    for (const TplGroup& g : groups) if (g.kind) n_grouped += g.live;
    const bool grouped_pipe = groups.size() > 1 && groups.size() <= 16 && n_grouped > 0 &&
                              (c->icp_mode == 3 || (c->icp_mode == 0 && n_grouped * 5 >= c->n_cu));
    if (std::getenv("CUBOID_DEBUG")) {
        std::fprintf(stderr, "cuboid_hip: stage_icp %d clusters, %zu template groups, %d in pipe groups, grouped_pipe %d:", ncl, groups.size(), n_grouped, (int)grouped_pipe);
        for (const TplGroup& g : groups) std::fprintf(stderr, " [%d,%d) m=%d kind=%d live=%d", g.beg, g.end, c->h_cl[g.beg].tpl_m, g.kind, g.live);
        std::fprintf(stderr, "\n");
    }
    std::vector<char> in_pipe((size_t)ncl, 0);
    if (grouped_pipe)
        for (const TplGroup& g : groups) if (g.kind) for (int k = g.beg; k < g.end; ++k) in_pipe[(size_t)k] = 1;
    int nwork = 0;
    for (int k = 0; k < ncl; ++k) {
        IcpCluster& cl = c->h_cl[k];
        cl.tile0 = nwork;
        const int tiles = cl.n >= 3 && cl.tpl_m > 0 && !in_pipe[(size_t)k] && !in_lat[(size_t)k] ? (cl.n + qslice - 1) / qslice : 0;
        if (nwork + tiles > c->work_cap) return fail(c, CD_ERR_CAPACITY, "ICP work list overflow");
        for (int t = 0; t < tiles; ++t) c->h_work[nwork++] = IcpWork{k, t};
        for (int s = 0; s < 2; ++s) {
            IcpState& st = c->h_st[2 * k + s];
            std::memset(&st, 0, sizeof(st));
            for (int i = 0; i < 4; ++i) st.Tfinal[5 * i] = 1.f;
            if (guess_mode != CD_GUESS_NONE) std::memcpy(st.Tfinal, guess_of(cl), 64);   // final_transformation_ = guess
            st.prev_mse = std::numeric_limits<double>::max();
            if (cl.tpl_m <= 0) { st.done = 1; st.status = CD_ERR_NO_TEMPLATE; }
            else if (cl.n < 3) { st.done = 1; st.status = CD_ERR_FEW_CORRESPONDENCES; }
        }
    }
    // A stage whose clusters ALL take k_icp_lat (the default template, no initial guess): the kernel reads the cluster list, the
    // order and the initial states from the pinned host arrays and writes the final states and fitness sums there - a few
    // hundred bytes per cluster once at each end of its ICP - instead of three copy launches around it.
    const bool lat_direct = c->icp_direct && n_lat > 0 && n_lat == n_live && guess_mode == CD_GUESS_NONE && c->mirror_reads && c->mirror_writes && c->copy_kernels;
    if (!lat_direct) {
        XferBatch xb(c);   // (one launch)
        xb.add(c->d_cl, c->h_cl, sizeof(IcpCluster) * ncl, hipMemcpyHostToDevice);
        if (nwork > 0) xb.add(c->d_work, c->h_work, sizeof(IcpWork) * nwork, hipMemcpyHostToDevice);
        xb.add(c->d_st, c->h_st, sizeof(IcpState) * 2 * ncl, hipMemcpyHostToDevice);
        HIPCHK(c, xb.flush());
    }
    // (zeroed up front by the fused call for its FIRST ICP stage only: a batch that runs one stage per template fills them again)
    const bool pre_zeroed = c->batch_zeroed;
    c->batch_zeroed = false;
    if (!pre_zeroed) {
        HIPCHK(c, hipMemsetAsync(c->d_acc, 0, sizeof(unsigned long long) * 48 * (size_t)ncl, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_accf, 0, sizeof(unsigned long long) * ((size_t)ncl + 1), c->stream));   // (+ the wave-time word of k_icp_lat)
    }
    if (guess_mode != CD_GUESS_NONE)   // input_transformed = guess * source (d_src is a copy of d_src0 at this point)
        LAUNCH(c, launch_icp_apply_guess(c->stream, ncl, max_n, c->d_cl, c->d_guess, guess_mode == CD_GUESS_PER_FRAME ? 1 : 0, c->d_src0, c->d_src));
    IcpParams ip;
    ip.max_iter = p->icp_max_iterations;
    ip.grid_rc = 1.0f;
    if (const char* e = std::getenv("CUBOID_ICP_GRID_RC")) ip.grid_rc = (float)std::atof(e);
    ip.trans_eps = p->icp_transformation_epsilon;
    ip.rel_mse = p->icp_euclidean_fitness_epsilon;
    ip.rot_thr = 1.0 - p->icp_transformation_epsilon;
    ip.abs_mse = 1e-12;
    // Whole-cluster launches: how many clusters a workgroup keeps in flight and how many workgroups there are.  A call that has
    // the device to itself wants every CU at work: one workgroup per CU (or per cluster), two slots, the second filled as long
    // as clusters are left.  With other calls in flight (BatchPipeline) the chip is full anyway and what counts is the CU-time
    // a batch costs: FOUR clusters per workgroup keep its sixteen waves supplied while one of them solves a step (no wave waits
    // at a hand-over, no workgroup ends with one cluster alone), on a quarter as many workgroups - the other batches have the
    // other CUs (config 3, 522 clusters: 2 x 256 -> 4 x 128 is 48.4 -> 53.3 k frames/s; grids that are not a multiple of 32 -
    // 4 per XCD and shader engine - lose 2-4 %).
    // (crossover measured on config 3: with two calls in flight the two shapes tie, with three 2 x 256 wins 49.4 : 46.5 k, with
    // four 4 x 128 wins 52.9 : 49.7 k - and the last launches of a burst, which soon have the GPU to themselves, spread out)
    const bool crowded = g_batches_in_flight[c->device & (MAX_DEVICES - 1)].load() >= 4;
    ip.pipe_slots = c->icp_slots > 0 ? std::min(c->icp_slots, CD_PIPE_SLOTS) : (crowded ? CD_PIPE_SLOTS : std::min(2, CD_PIPE_SLOTS));
    ip.donate = 0;   // (set below for a whole-cluster launch that has the GPU to itself)
    ip.don = c->d_don;
    ip.don_idle = c->don_idle;
    ip.don_fault = c->don_fault;
    auto pipe_grid = [&](int n_items, int cap) {   // workgroups of a whole-cluster launch over n_items clusters
        if (c->icp_cpw > 0) return std::min((n_items + c->icp_cpw - 1) / c->icp_cpw, cap);
        if (!crowded || n_items <= cap) return std::min(n_items, cap);
        return std::min(cap, std::max(32, (n_items / ip.pipe_slots + 16) / 32 * 32));
    };
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    if (n_lat > 0) {
        // the clusters of the launch, largest first
        int no = 0;
        for (int k = 0; k < ncl; ++k) if (in_lat[(size_t)k]) c->h_order[no++] = k;
        std::stable_sort(c->h_order, c->h_order + no, [&](int a, int b) { return c->h_cl[a].n > c->h_cl[b].n; });
        if (!lat_direct) HIPCHK(c, xfer(c, c->d_order, c->h_order, sizeof(int) * (size_t)no, hipMemcpyHostToDevice));
        // Shape of the launch (k_icp_lat.hip): clusters per workgroup x waves per cluster.  A call that has the GPU to itself wants
        // the launch short: one cluster per workgroup, four waves each when there are clusters enough to fill the chip twice that way,
        // eight or sixteen for fewer or very large clusters (one frame; config 5's thousands of points).  With other batches in
        // flight the chip is full anyway: four clusters per workgroup, two waves each, pay the single-lane solve of a round once
        // for the four (measured on config 3, seven in flight, profiles/r05_lat_shapes.txt: 1x4 145-146 k, 4x2 149-151 k, 8x2 146 k,
        // 4x4 139 k, 4x1 131 k frames/s; alone 1x4 1.34 ms, 1x8 1.30, 4x2 2.2, 4x1 2.8).  CUBOID_LAT_SHAPE=cpw,wpc[,clusters per slot].
        const bool busy = g_batches_in_flight[c->device & (MAX_DEVICES - 1)].load() >= 3;
        int cpw = 1, wpc = n_lat >= 768 ? 4 : (n_lat >= 96 ? 8 : 16), per_slot = 1;   // (end of round 5, 522 clusters alone: 1x4 1.17 ms, 1x8 1.10, 1x16 1.53, 2x4 1.27)
        if (lat_max_n > 16384) wpc = 16;
        else if (lat_max_n > 4096) wpc = std::max(wpc, 8);
        if (busy && n_lat >= 256 && lat_max_n <= 8192) { cpw = 4; wpc = 2; }
        if (c->lat_shape[0] > 0) { cpw = c->lat_shape[0]; wpc = std::max(1, c->lat_shape[1]); per_slot = std::max(1, c->lat_shape[2]); }
        const int n_wg = std::max(1, (n_lat + cpw * per_slot - 1) / (cpw * per_slot));
        GateHold lat_hold;   // (CUBOID_ICP_CONCURRENT: at most that many contexts between this launch and its completion)
        if (c->icp_concurrent > 0) lat_hold.enter(&g_icp_gate[c->device & (MAX_DEVICES - 1)], c->icp_concurrent);
        if (!pre_zeroed) HIPCHK(c, hipMemsetAsync(c->d_queue, 0, 2 * sizeof(int), c->stream));   // head of the cluster queue, count of finished workgroups
        if (lat_direct) std::memset(c->h_accf, 0, sizeof(unsigned long long) * ((size_t)ncl + 1));   // (what the zeroed device array used to bring back for clusters the kernel skips)
        if (lat_direct)
            LAUNCH(c, launch_icp_lat(c->stream, n_lat, cpw, wpc, n_wg, c->h_order, c->h_cl, c->h_st, c->h_accf, c->d_lat, c->d_src, c->d_src0, c->d_queue, c->d_accf + ncl, c->h_accf + ncl, ip));
        else
            LAUNCH(c, launch_icp_lat(c->stream, n_lat, cpw, wpc, n_wg, c->d_order, c->d_cl, c->d_st, c->d_accf, c->d_lat, c->d_src, c->d_src0, c->d_queue, c->d_accf + ncl, nullptr, ip));
        c->timing.icp_kernel_launches = 1;
        c->timing.icp_regime = (cpw << 16) | std::min(n_wg, 0xffff);
        if (n_lat == n_live) {
            HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
            if (!lat_direct) {
                XferBatch xb(c);   // (one launch)
                xb.add(c->h_st, c->d_st, sizeof(IcpState) * 2 * ncl, hipMemcpyDeviceToHost);
                xb.add(c->h_accf, c->d_accf, sizeof(unsigned long long) * ((size_t)ncl + 1), hipMemcpyDeviceToHost);
                HIPCHK(c, xb.flush());
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float ms1 = 0.f;
            hipEventElapsedTime(&ms1, c->ev[5], c->ev[6]);
            c->timing.icp_kernel_ms = ms1;
            c->timing.icp_wave_ms = (float)((double)c->h_accf[ncl] / 1.0e5);   // 100 MHz ticks x waves -> wave-milliseconds
            if (pair_tests) {
                long long tot = 0;
                for (int k = 0; k < ncl; ++k)
                    if (c->h_st[2 * k].status == CD_OK) tot += (long long)c->h_cl[k].n * c->h_cl[k].tpl_m * (c->h_st[2 * k].iters + 1);
                *pair_tests = tot;
            }
            return CD_OK;
        }
        // mixed batch: the other clusters take the drivers below, which find these finished in d_st / h_st (the stream orders them)
        HIPCHK(c, xfer(c, c->h_st, c->d_st, sizeof(IcpState) * 2 * ncl, hipMemcpyDeviceToHost));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    // Batch mode: with at least ~n_cu/5 clusters every CU can own whole clusters, so each cluster runs its
    // complete ICP (all iterations + fitness) inside one persistent workgroup, one launch for the batch.
    // The pipelined variant (two to four clusters in flight per workgroup, no barrier in the iteration loop) needs one
    // LDS-resident gridded template shared by every cluster of the launch; otherwise k_icp_cluster runs.
    bool one_tpl = true;
    for (int k = 1; k < ncl && one_tpl; ++k) one_tpl = c->h_cl[k].tpl_off == c->h_cl[0].tpl_off && c->h_cl[k].tpl_m == c->h_cl[0].tpl_m;
    const bool pipe_ok = c->icp_mode != 2 && one_tpl && kind_of(c->h_cl[0]) == 1;
    // ... or, for a template that does not fit LDS, its twin that reads the template from global memory (k_icp_pipe_big)
    const bool big_ok = c->icp_mode != 2 && one_tpl && kind_of(c->h_cl[0]) == 2;
    // (auto mode: only the pipelined kernel beats the sliced driver; with mixed or non-resident templates k_icp_cluster's
    // barrier per iteration costs more than it saves, so those batches stay sliced unless the mode is forced)
    // (measured crossover on the bench frames: 33 clusters 3.3 ms sliced / 3.7 ms pipelined, 65 clusters 4.4 / 3.9, 130 clusters
    // 6.4 / 4.0: the pipelined kernel wins from about a fifth of the CUs' worth of clusters)
    const bool whole_cluster = !grouped_pipe && (c->icp_mode >= 2 || (c->icp_mode == 0 && ncl * 5 >= c->n_cu && (pipe_ok || big_ok)));
    if (grouped_pipe) {
        // items of `order`: the pipe groups one after the other (those of k_icp_pipe first, then those of k_icp_pipe_big),
        // each largest cluster first.  The two kernels are launched side by side (second stream): their workgroups share the
        // chip, every group gets workgroups in proportion to its points (a query against a template in global memory
        // counted BIG_WEIGHT times), at least one, no more than it has clusters.
        const int wg_cap = c->icp_max_wg > 0 ? std::min(c->icp_max_wg, c->n_cu) : c->n_cu;
        // (while every problem can have a workgroup of its own the launch lasts as long as its longest cluster, and those are the
        // global-memory ones: 4; with more problems than workgroups it is their throughput that counts, 1.6 x the LDS kernel's
        // cost per workgroup plus the longer tail: 2 - config 5 at 8 frames per batch 4 > 2 by 5 %, at 64 frames 2 > 4 by 6 %)
        int live_all = 0;
        for (const TplGroup& g : groups) if (g.kind) live_all += g.live;
        const long long BIG_WEIGHT = c->icp_big_weight > 0 ? c->icp_big_weight : (live_all <= wg_cap ? 4 : 2);
        const int cpw = std::max(1, c->icp_cpw);   // (0 = default: a group may have as many workgroups as clusters)
        // (template groups keep two clusters in flight per workgroup whatever the regime: every group has its share of the
        // workgroups and its own queue already, and four slots measure 3-7 % slower on config 5)
        IcpParams ipg = ip;
        if (c->icp_slots <= 0) ipg.pipe_slots = std::min(2, CD_PIPE_SLOTS);
        int* tab = c->h_wgtab;   // pinned: the copy below is asynchronous
        int ntab = 0;
        long long pts_all = 0;
        int npg = 0;
        for (const TplGroup& g : groups) if (g.kind && g.live > 0) { pts_all += g.pts * (g.kind == 2 ? BIG_WEIGHT : 1); ++npg; }
        int no = 0, gq = 0, n_wg = 0, wg_of[3] = {0, 0, 0}, tab_of[3] = {0, 0, 0};
        for (int kind = 1; kind <= 2; ++kind) {
            tab_of[kind] = ntab;
            for (const TplGroup& g : groups) {
                if (g.kind != kind || g.live == 0) continue;
                const int b = no;
                for (int k = g.beg; k < g.end; ++k) if (c->h_cl[k].n >= 3) c->h_order[no++] = k;
                std::stable_sort(c->h_order + b, c->h_order + no, [&](int x, int y) { return c->h_cl[x].n > c->h_cl[y].n; });
                int share = (int)((long long)wg_cap * g.pts * (kind == 2 ? BIG_WEIGHT : 1) / std::max(pts_all, 1ll));
                share = std::max(1, std::min(share, std::min((g.live + cpw - 1) / cpw, wg_cap - n_wg - (npg - 1 - gq))));
                for (int w = 0; w < share && ntab + 3 <= 3 * 1024; ++w) { tab[ntab++] = b; tab[ntab++] = no; tab[ntab++] = gq; }
                n_wg += share;
                wg_of[kind] += share;
                ++gq;
            }
        }
        if (!lat_direct) HIPCHK(c, xfer(c, c->d_order, c->h_order, sizeof(int) * (size_t)no, hipMemcpyHostToDevice));
        HIPCHK(c, xfer(c, c->d_wgtab, tab, sizeof(int) * (size_t)ntab, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemsetAsync(c->d_queue, 0, sizeof(int) * 16, c->stream));   // one queue head per group
        // the LDS-template launch goes to the context's stream (stream3 with icp_lowprio), the global-template launch beside it
        hipStream_t s1 = c->icp_lowprio ? c->stream3 : c->stream;
        hipStream_t s2 = (c->icp_lowprio || (wg_of[1] > 0 && wg_of[2] > 0)) ? c->stream2 : c->stream;
        if (s1 != c->stream || s2 != c->stream) {   // everything uploaded so far is visible to the side streams
            HIPCHK(c, hipEventRecord(c->ev2[0], c->stream));
            if (s1 != c->stream) HIPCHK(c, hipStreamWaitEvent(s1, c->ev2[0], 0));
            if (s2 != c->stream) HIPCHK(c, hipStreamWaitEvent(s2, c->ev2[0], 0));
        }
        if (wg_of[1] > 0)
            LAUNCH(c, launch_icp_pipe(s1, ncl, c->d_order, c->d_cl, c->d_st, c->d_accf, c->d_tpl, c->d_tlok, c->d_thik, c->d_kdmap, c->d_grid, c->d_tcell, c->d_src, c->d_src0, c->d_nn,
                            c->d_queue, wg_of[1], c->d_wgtab + tab_of[1], ipg));
        if (wg_of[2] > 0)
            LAUNCH(c, launch_icp_pipe_big(s2, ncl, c->d_order, c->d_cl, c->d_st, c->d_accf, c->d_tpl, c->d_tplk, c->d_tlok, c->d_thik, c->d_kdmap, c->d_grid,
                                c->d_super, c->d_tcell, c->d_src, c->d_src0, c->d_nn, c->d_queue, wg_of[2], c->d_wgtab + tab_of[2], ipg));
        if (s1 != c->stream && wg_of[1] > 0) {
            HIPCHK(c, hipEventRecord(c->ev2[2], s1));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev2[2], 0));
        }
        if (s2 != c->stream && wg_of[2] > 0) {
            HIPCHK(c, hipEventRecord(c->ev2[1], s2));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev2[1], 0));
        }
        c->timing.icp_kernel_launches = 1;
        c->timing.icp_regime = (ipg.pipe_slots << 16) | n_wg;
        if (nwork == 0) {
            HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
            HIPCHK(c, xfer(c, c->h_accf, c->d_accf, sizeof(unsigned long long) * ncl, hipMemcpyDeviceToHost));
            HIPCHK(c, xfer(c, c->h_st, c->d_st, sizeof(IcpState) * 2 * ncl, hipMemcpyDeviceToHost));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float ms1 = 0.f;
            hipEventElapsedTime(&ms1, c->ev[5], c->ev[6]);
            c->timing.icp_kernel_ms = ms1;
            if (pair_tests) {
                long long tot = 0;
                for (int k = 0; k < ncl; ++k)
                    if (c->h_st[2 * k].status == CD_OK) tot += (long long)c->h_cl[k].n * c->h_cl[k].tpl_m * (c->h_st[2 * k].iters + 1);
                *pair_tests = tot;
            }
            return CD_OK;
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (whole_cluster) {
        GateHold hold;   // (released on every path out of this block)
        if (c->icp_concurrent > 0) hold.enter(&g_icp_gate[c->device & (MAX_DEVICES - 1)], c->icp_concurrent);
        for (int k = 0; k < ncl; ++k) c->h_order[k] = k;
        std::stable_sort(c->h_order, c->h_order + ncl, [&](int a, int b) { return c->h_cl[a].n > c->h_cl[b].n; });
        HIPCHK(c, xfer(c, c->d_order, c->h_order, sizeof(int) * ncl, hipMemcpyHostToDevice));
        const int wg_cap = c->icp_max_wg > 0 ? std::min(c->icp_max_wg, c->n_cu) : c->n_cu;
        HIPCHK(c, hipMemsetAsync(c->d_queue, 0, sizeof(int), c->stream));   // head of the cluster queue
        hipStream_t si = c->icp_lowprio >= 2 ? c->stream3 : c->stream;
        if (pipe_ok || big_ok) {
            c->timing.icp_regime = (ip.pipe_slots << 16) | pipe_grid(ncl, wg_cap);
            // A launch that has the GPU to itself lets workgroups that run out of clusters wait and take over running ones
            // (k_icp.hip, "hand-over of running clusters"): the launch is as long as its slowest workgroup, and the bench batch's
            // slowest runs 25 % over the average.  With other calls in flight a waiting workgroup would sit on a CU their
            // kernels could use, and what counts there is the CU-time a batch costs, which hand-overs do not lower.
            const bool alone = g_batches_in_flight[c->device & (MAX_DEVICES - 1)].load() <= 1;
            ip.donate = (c->icp_donate < 0 ? (alone && ncl > pipe_grid(ncl, wg_cap)) : c->icp_donate != 0) ? 1 : 0;
            if (ip.donate) HIPCHK(c, hipMemsetAsync(c->d_don, 0, sizeof(int) * (size_t)(DON_BOX + DON_CAP), c->stream));
        }
        if (si != c->stream) {   // (after the control block was zeroed on c->stream: the launch on `si` must see it zeroed - ADVICE r4)
            HIPCHK(c, hipEventRecord(c->ev2[0], c->stream));
            HIPCHK(c, hipStreamWaitEvent(si, c->ev2[0], 0));
        }
        if (pipe_ok)
            LAUNCH(c, launch_icp_pipe(si, ncl, c->d_order, c->d_cl, c->d_st, c->d_accf, c->d_tpl, c->d_tlok, c->d_thik, c->d_kdmap, c->d_grid, c->d_tcell, c->d_src, c->d_src0, c->d_nn,
                            c->d_queue, pipe_grid(ncl, wg_cap), nullptr, ip));
        else if (big_ok)
            LAUNCH(c, launch_icp_pipe_big(si, ncl, c->d_order, c->d_cl, c->d_st, c->d_accf, c->d_tpl, c->d_tplk, c->d_tlok, c->d_thik, c->d_kdmap, c->d_grid, c->d_super, c->d_tcell,
                                c->d_src, c->d_src0, c->d_nn, c->d_queue, pipe_grid(ncl, wg_cap), nullptr, ip));
        else
            LAUNCH(c, launch_icp_cluster(si, ncl, c->d_order, c->d_cl, c->d_st, c->d_accf, c->d_tpl, c->d_tlo, c->d_thi, c->d_grid, c->d_tcell, c->d_src, c->d_src0, c->d_nn,
                               c->d_queue, wg_cap, ip));
        if (si != c->stream) {
            HIPCHK(c, hipEventRecord(c->ev2[2], si));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev2[2], 0));
        }
        HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
        c->timing.icp_kernel_launches = 1;
        HIPCHK(c, xfer(c, c->h_accf, c->d_accf, sizeof(unsigned long long) * ncl, hipMemcpyDeviceToHost));
        HIPCHK(c, xfer(c, c->h_st, c->d_st, sizeof(IcpState) * 2 * ncl, hipMemcpyDeviceToHost));
        if (ip.donate) HIPCHK(c, xfer(c, c->h_ctl + 8, c->d_don, sizeof(int) * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms1 = 0.f;
        hipEventElapsedTime(&ms1, c->ev[5], c->ev[6]);
        c->timing.icp_kernel_ms = ms1;
        if (ip.donate) {
            const int* w = c->h_ctl + 8;   // (copied with the records above)
            c->timing.icp_handovers = w[DON_HEAD];
            // every cluster of the launch was finished by somebody, and every mailbox entry that was written was taken
            if (w[DON_ERR] != 0 || w[DON_FINISHED] != ncl || w[DON_HEAD] != std::min(w[DON_TAIL], DON_CAP)) {
                c->timing.icp_handover_lost = 1;
                std::snprintf(c->err, sizeof(c->err), "ICP hand-over lost a cluster: %d of %d clusters finished, %d entries published, %d taken, error word %d",
                              w[DON_FINISHED], ncl, w[DON_TAIL], w[DON_HEAD], w[DON_ERR]);
                return CD_ERR_DEVICE;
            }
            if (std::getenv("CUBOID_DEBUG"))
                std::fprintf(stderr, "cuboid_hip: hand-overs: %d published, %d taken, %d clusters finished of %d, waiting balance %d\n", w[DON_TAIL], w[DON_HEAD], w[DON_FINISHED], ncl, w[DON_AVAIL]);
        }
        if (pair_tests) {
            long long tot = 0;
            for (int k = 0; k < ncl; ++k)
                if (c->h_st[2 * k].status == CD_OK) tot += (long long)c->h_cl[k].n * c->h_cl[k].tpl_m * (c->h_st[2 * k].iters + 1);
            *pair_tests = tot;
        }
        return CD_OK;
    }
    int it = 0;
    const int max_launch = p->icp_max_iterations + 3;
    // Few clusters: ONE persistent launch runs every iteration and the fitness pass (k_icp_persist: grid barrier per
    // iteration instead of two kernel launches).  Needs all its workgroups resident together; a barrier that does not
    // complete in time makes it give up, and the multi-launch loop below takes over from the initial state.
    // (with other contexts at work - BatchPipeline - their persistent kernels hold the CUs this launch's grid barrier needs:
    // it would wait, give up and hand over; go straight to the multi-launch loop then)
    const bool siblings = g_calls_in_flight[c->device & (MAX_DEVICES - 1)].load() > 1;
    if (nwork > 0 && c->icp_persist && (!siblings || c->icp_persist == 2)) {
        const int G = std::min(nwork, c->icp_max_wg > 0 ? std::min(c->icp_max_wg, c->n_cu) : c->n_cu);
        // measured against the multi-launch loop: 1 frame 1.50 / 1.78 ms, 4 frames 1.94 / 2.36, 8 frames 3.04 / 2.66 - with many
        // clusters the launch lasts as long as the slowest one while finished workgroups wait at the barriers, and a workgroup
        // with several items solves for each of them in turn
        if (ncl <= 8 && nwork <= G) {
            int n_open = 0;
            for (int k = 0; k < ncl; ++k) if (c->h_cl[k].tile0 < (k + 1 < ncl ? c->h_cl[k + 1].tile0 : nwork)) ++n_open;
            // d_queue[4] barrier counter, [5] abort flag, [6..9] clusters closed per iteration slot (k_icp_persist's exit test)
            int* ctl = c->h_ctl;
            for (int i = 0; i < 8; ++i) ctl[i] = 0;
            ctl[1] = c->icp_persist == 2 ? 1 : 0;
            HIPCHK(c, xfer(c, c->d_queue + 4, ctl, sizeof(int) * 6, hipMemcpyHostToDevice));
            LAUNCH(c, launch_icp_persist(c->stream, nwork, G, max_launch, c->d_work, c->d_cl, c->d_st, c->d_acc, c->d_accf, c->d_tplk, c->d_tlok, c->d_thik, c->d_grid,
                               c->d_src, c->d_src0, c->d_nn, c->d_d2, qslice, (unsigned*)(c->d_queue + 4), c->d_queue + 5, n_open, c->d_queue + 6, ip));
            HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
            HIPCHK(c, xfer(c, ctl + 8, c->d_queue + 5, sizeof(int), hipMemcpyDeviceToHost));
            HIPCHK(c, xfer(c, c->h_accf, c->d_accf, sizeof(unsigned long long) * ncl, hipMemcpyDeviceToHost));
            c->st_init.assign(c->h_st, c->h_st + 2 * (size_t)ncl);   // in case the launch gives up (no reallocation after the first call)
            std::vector<IcpState>& init = c->st_init;
            HIPCHK(c, xfer(c, c->h_st, c->d_st, sizeof(IcpState) * 2 * ncl, hipMemcpyDeviceToHost));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const int gave_up = ctl[8];
            if (!gave_up) {
                float ms1 = 0.f;
                hipEventElapsedTime(&ms1, c->ev[5], c->ev[6]);
                c->timing.icp_kernel_ms = ms1;
                c->timing.icp_kernel_launches = 1;
                if (pair_tests) {
                    long long tot = 0;
                    for (int k = 0; k < ncl; ++k)
                        if (c->h_st[2 * k].status == CD_OK) tot += (long long)c->h_cl[k].n * c->h_cl[k].tpl_m * (c->h_st[2 * k].iters + 1);
                    *pair_tests = tot;
                }
                return CD_OK;
            }
            c->persist_gave_up += 1;
            c->timing.icp_persist_gave_up += 1;
            if (std::getenv("CUBOID_DEBUG")) std::fprintf(stderr, "cuboid_hip: persistent ICP launch gave up at a grid barrier, running the multi-launch loop\n");
            // start over: initial states, zero sums, the source points as extracted (the launch transformed them in place)
            std::memcpy(c->h_st, init.data(), sizeof(IcpState) * 2 * (size_t)ncl);
            HIPCHK(c, xfer(c, c->d_st, c->h_st, sizeof(IcpState) * 2 * ncl, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemsetAsync(c->d_acc, 0, sizeof(unsigned long long) * 48 * (size_t)ncl, c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_accf, 0, sizeof(unsigned long long) * (size_t)ncl, c->stream));
            long long span = 0;
            for (int k = 0; k < ncl; ++k) span = std::max(span, (long long)c->h_cl[k].src_off + c->h_cl[k].n);
            HIPCHK(c, hipMemcpyAsync(c->d_src, c->d_src0, sizeof(float4) * (size_t)span, hipMemcpyDeviceToDevice, c->stream));
            if (guess_mode != CD_GUESS_NONE)
                LAUNCH(c, launch_icp_apply_guess(c->stream, ncl, max_n, c->d_cl, c->d_guess, guess_mode == CD_GUESS_PER_FRAME ? 1 : 0, c->d_src0, c->d_src));
        }
    }
    // The iteration kernel walks an ACTIVE work list (d_work2) that the host re-packs at every
    // completion poll, dropping the clusters that have converged; the full list (d_work) is kept for
    // the fitness pass.  A cluster whose state shows done in either parity slot has already had its
    // final transform applied, so it needs no further work items (k_icp_solve alone keeps its state).
    int nactive = nwork;
    std::memcpy(c->h_work2, c->h_work, sizeof(IcpWork) * (size_t)std::max(nwork, 1));
    HIPCHK(c, xfer(c, c->d_work2, c->h_work2, sizeof(IcpWork) * (size_t)std::max(nwork, 1), hipMemcpyHostToDevice));
    int group = 8;
    while (nwork > 0 && it < max_launch) {
        const int g = std::min(group, max_launch - it);
        for (int q = 0; q < g; ++q) LAUNCH(c, launch_icp_iter(c->stream, it++, nactive, ncl, c->d_work2, c->d_cl, c->d_st, c->d_acc, c->d_tplk, c->d_tlok, c->d_thik, c->d_grid, c->d_src, c->d_nn, c->d_d2, qslice, c->d_queue, c->n_cu, ip));
        HIPCHK(c, xfer(c, c->h_st, c->d_st, sizeof(IcpState) * 2 * ncl, hipMemcpyDeviceToHost));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bool all = true;
        int na = 0;
        for (int w = 0; w < nwork; ++w) {
            const int k = c->h_work[w].cluster;
            if (!(c->h_st[2 * k].done || c->h_st[2 * k + 1].done)) c->h_work2[na++] = c->h_work[w];
        }
        for (int k = 0; k < ncl && all; ++k) all = c->h_st[2 * k].done && c->h_st[2 * k + 1].done;
        if (all) break;
        if (na != nactive) {
            nactive = na;
            if (na > 0) HIPCHK(c, xfer(c, c->d_work2, c->h_work2, sizeof(IcpWork) * (size_t)na, hipMemcpyHostToDevice));
        }
        group = 16;
    }
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    c->timing.icp_kernel_launches = it;
    LAUNCH(c, launch_icp_fitness(c->stream, nwork, c->d_work, c->d_cl, c->d_st, 0, c->d_accf, c->d_tplk, c->d_tlok, c->d_thik, c->d_grid, c->d_src0, c->d_nn, c->d_d2, qslice));
    HIPCHK(c, xfer(c, c->h_accf, c->d_accf, sizeof(unsigned long long) * ncl, hipMemcpyDeviceToHost));
    HIPCHK(c, xfer(c, c->h_st, c->d_st, sizeof(IcpState) * 2 * ncl, hipMemcpyDeviceToHost));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev[5], c->ev[6]);
    c->timing.icp_kernel_ms = ms;
    if (pair_tests) {
        long long tot = 0;
        for (int k = 0; k < ncl; ++k)
            if (c->h_st[2 * k].status == CD_OK) tot += (long long)c->h_cl[k].n * c->h_cl[k].tpl_m * (c->h_st[2 * k].iters + 1);
        *pair_tests = tot;
    }
    return CD_OK;
}

void fill_cluster_result(const cd_context* c, int k, const cd_params* p, cd_cluster_result* r) {
    const IcpState& st = c->h_st[2 * k];
    const IcpCluster& cl = c->h_cl[k];
    std::memset(r, 0, sizeof(*r));
    r->size = cl.n;
    r->iterations = st.iters;
    r->converged = st.converged;
    std::memcpy(r->T, st.Tfinal, 64);
    r->fitness = st.status == CD_OK ? hm::unfix(c->h_accf[k], FIX_SHIFT_D2) / (double)cl.n : std::numeric_limits<double>::max();
    r->accepted = (r->converged && r->fitness < p->icp_accept_fitness) ? 1 : 0;
    double Td[16];
    for (int i = 0; i < 16; ++i) Td[i] = (double)st.Tfinal[i];
    if (!hm::mat4_inverse(Td, r->pose))
        for (int i = 0; i < 16; ++i) r->pose[i] = std::numeric_limits<double>::quiet_NaN();
}

int upload_points(cd_context* c, const void* pts, size_t stride, int n, float4* dst) {
    // host (stride) -> device float4 via the staging buffer
    if (n <= 0) return CD_OK;
    std::vector<float4> tmp((size_t)n);
    const char* b = (const char*)pts;
    for (int i = 0; i < n; ++i) {
        float v[3];
        std::memcpy(v, b + (size_t)i * stride, 12);
        tmp[i] = make_float4(v[0], v[1], v[2], 0.f);
    }
    HIPCHK(c, hipMemcpyAsync(dst, tmp.data(), sizeof(float4) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CD_OK;
}

int check_params(cd_context* c, const cd_params* p) {
    if (!p) return fail(c, CD_ERR_INVALID_ARG, "params is NULL");
    if (!(p->leaf_size > 0.f)) return fail(c, CD_ERR_INVALID_ARG, "leaf_size must be > 0");
    if (p->plane_max_iterations < 0 || p->plane_max_iterations > 1000) return fail(c, CD_ERR_INVALID_ARG, "plane_max_iterations must be in [0,1000]");
    if (p->template_slot < -1 || p->template_slot >= CD_MAX_TEMPLATES) return fail(c, CD_ERR_INVALID_ARG, "template_slot out of range");
    if (!(p->cluster_tolerance > 0.0)) return fail(c, CD_ERR_INVALID_ARG, "cluster_tolerance must be > 0");
    if (p->plane_model < CD_PLANE || p->plane_model > CD_PLANE_PARALLEL) return fail(c, CD_ERR_INVALID_ARG, "plane_model out of range");
    if (p->icp_use_guess < CD_GUESS_NONE || p->icp_use_guess > CD_GUESS_PER_FRAME) return fail(c, CD_ERR_INVALID_ARG, "icp_use_guess out of range");
    if (p->icp_use_guess == CD_GUESS_PARAMS)
        for (int i = 0; i < 16; ++i) if (!std::isfinite(p->icp_guess[i])) return fail(c, CD_ERR_INVALID_ARG, "icp_guess holds a non-finite value");
    return CD_OK;
}

int process_batch_impl(cd_context* c, const void* d_frames, size_t stride, int N, int F, const cd_params* p,
                       cd_frame_result* results, int32_t* plane_inliers, int32_t* labels) {
    int st = check_params(c, p);
    if (st) return st;
    if (!results || !d_frames) return fail(c, CD_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || F <= 0 || stride < 12 || (stride & 3)) return fail(c, CD_ERR_INVALID_ARG, "bad shape/stride");
    if (N > c->N || F > c->F) return fail(c, CD_ERR_CAPACITY, "batch larger than the context capacity");
    invalidate_last(c);
    std::memset(&c->timing, 0, sizeof(c->timing));
    BatchGuard in_flight(c->device);
    GateHold front;
    if (c->front_concurrent > 0) front.enter(&g_front_gate[c->device & (MAX_DEVICES - 1)], c->front_concurrent);
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    struct ZeroedScope {   // the stages skip their own fills for the duration of this call only
        cd_context* c;
        ~ZeroedScope() { c->batch_zeroed = false; c->fs_initialised = false; }
    } zeroed_scope{c};
    if (c->zero_once) {
        // every scratch array the stages want zeroed, in ONE launch (a batch had ~14 fill kernels, each a stream operation of its
        // own that queues behind the other contexts' kernels)
        ZeroRegions zr;
        zr.n = 0;
        zr.fs = c->mirror_writes && c->copy_kernels ? c->d_fs : nullptr;   // (the crop stage then skips its upload of the initial FrameStates)
        zr.nfs = F;
        c->fs_initialised = zr.fs != nullptr;
        auto add = [&](void* ptr, size_t bytes) { zr.ptr[zr.n] = (uint32_t*)ptr; zr.words[zr.n] = bytes / 4; ++zr.n; };
        const size_t FT = (size_t)F * c->T;
        add(c->d_ticket, sizeof(int) * (size_t)F * TICKET_PITCH);
        add(c->d_tileA, sizeof(int) * FT);
        add(c->d_tileB, sizeof(int) * FT);
        add(c->d_tileC, sizeof(int) * FT);
        add(c->d_tile64, sizeof(unsigned long long) * FT);
        add(c->d_ghist, sizeof(uint32_t) * (size_t)F * SORT_MAX_PASSES_HOST * RADIX);
        add(c->d_counts, sizeof(int) * (size_t)F * MAX_HYP);
        add(c->d_sums, sizeof(unsigned long long) * 10 * (size_t)F);
        add(c->d_tileK, sizeof(int) * FT * KICP);
        add(c->d_acc, sizeof(unsigned long long) * 48 * (size_t)c->cl_cap);
        add(c->d_accf, sizeof(unsigned long long) * ((size_t)c->cl_cap + 1));
        add(c->d_queue, sizeof(int) * 16);
        LAUNCH(c, launch_zero_regions(c->stream, zr));
        c->batch_zeroed = true;
    }
    int rounds = 0;
    st = stage_crop_voxel(c, d_frames, stride, N, F, p, nullptr);
    if (st) return st;
    st = sync_fs(c, F);   // n_v
    if (st) return st;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    std::vector<int> iterations;
    st = stage_plane(c, F, p, iterations, &rounds);
    if (st) return st;
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    st = stage_extract(c, F, p);
    if (st) return st;
    st = sync_fs(c, F, c->mirror_writes && c->copy_kernels);   // n_o per frame (written to the mirror by the scan): picks the clustering path and sizes the launches
    if (st) return st;
    int max_no = 0;
    for (int f = 0; f < F; ++f) max_no = std::max(max_no, c->h_fs[f].n_o);
    st = stage_cluster_sync(c, F, p, max_no);   // sync #4: n_plane, n_o, n_k, ksize, koff
    if (st) return st;
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    front.release();
    // Every cluster of every frame gets its ICP (opd.cpp:376-413).  The device extracts the ICP sources KICP clusters per
    // frame at a time; frames with more than KICP clusters (rare) need further rounds, and the host needs their sizes.
    int kmax = 0, ncl = 0;
    long long cl_points = 0;
    std::vector<int> first_cl((size_t)F + 1, 0);
    for (int f = 0; f < F; ++f) {
        first_cl[(size_t)f] = ncl;
        ncl += c->h_fs[f].n_k;
        kmax = std::max(kmax, c->h_fs[f].n_k);
    }
    first_cl[(size_t)F] = ncl;
    std::vector<int> csize((size_t)std::max(ncl, 1)), coff((size_t)std::max(ncl, 1));   // size / source offset of every cluster
    for (int f = 0; f < F; ++f) {
        const FrameState& s = c->h_fs[f];
        int* sz = csize.data() + first_cl[(size_t)f];
        if (s.n_k > KICP) HIPCHK(c, hipMemcpyAsync(sz, c->d_sizes + (size_t)f * c->N, sizeof(int) * (size_t)s.n_k, hipMemcpyDeviceToHost, c->stream));
        else for (int k = 0; k < s.n_k; ++k) sz[k] = s.ksize[k];
    }
    const int rounds_k = std::max(1, (kmax + KICP - 1) / KICP);
    if (rounds_k > 1) HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int f = 0; f < F; ++f) {
        int off = 0;
        for (int k = first_cl[(size_t)f]; k < first_cl[(size_t)f + 1]; ++k) { coff[(size_t)k] = off; off += csize[(size_t)k]; cl_points += csize[(size_t)k]; }
    }
    st = ensure_clusters(c, ncl, cl_points);
    if (st) return st;
    int max_no2 = 0;
    for (int f = 0; f < F; ++f) max_no2 = std::max(max_no2, c->h_fs[f].n_o);
    const int To2 = std::max(1, (max_no2 + TILE - 1) / TILE);
    if (rounds_k > 1) {   // offsets of the later rounds' clusters, [round][F][KICP]
        const size_t need = (size_t)rounds_k * F * KICP;
        if (need > c->koffx_cap) {
            if (c->d_koffx) hipFree(c->d_koffx);
            c->d_koffx = nullptr; c->koffx_cap = 0;
            HIPCHK(c, dalloc(&c->d_koffx, need));
            c->koffx_cap = need;
        }
        std::vector<int> tab(need, 0);
        for (int r = 1; r < rounds_k; ++r)
            for (int f = 0; f < F; ++f)
                for (int k = 0; k < KICP; ++k) {
                    const int q = first_cl[(size_t)f] + r * KICP + k;
                    if (q < first_cl[(size_t)f + 1]) tab[((size_t)r * F + f) * KICP + k] = coff[(size_t)q];
                }
        HIPCHK(c, copy_sync(c, c->d_koffx, tab.data(), sizeof(int) * need, hipMemcpyHostToDevice));
    }
    // (re)build the ICP sources d_src0 / d_src of every round.  Round 0 was extracted by stage_cluster; it is redone only
    // when d_src has been consumed by a previous template pass or d_tileK by a later round.
    auto extract_sources = [&](bool redo_round0) -> int {
        for (int r = redo_round0 ? 0 : 1; r < rounds_k; ++r) {
            if (r > 0 || rounds_k > 1) {
                HIPCHK(c, hipMemsetAsync(c->d_tileK, 0, sizeof(int) * (size_t)F * KICP * c->T, c->stream));
                LAUNCH(c, launch_label_count(c->stream, c->N, F, c->T, To2, c->d_fs, p->cluster_enable, c->d_parent, c->d_rank, c->d_label, c->d_tileK, r * KICP));
                LAUNCH(c, launch_scan_tiles(c->stream, c->d_tileK, F * KICP, c->T, nullptr, 0));
            }
            LAUNCH(c, launch_label_scatter(c->stream, c->d_obj, c->N, F, c->T, To2, c->d_fs, c->d_label, c->d_tileK, c->d_src0, c->d_src, r * KICP,
                                 r > 0 ? c->d_koffx + (size_t)r * F * KICP : nullptr));
        }
        return CD_OK;
    };
    // ICP problems.  template_slot >= 0: every cluster against that slot.  template_slot == -1: every cluster
    // against every loaded template, one ICP pass per slot; the result with the lowest fitness is kept
    // (ties -> lowest slot).  The sources are re-extracted between passes (ICP transforms d_src in place).
    std::vector<int> slots;
    if (p->template_slot >= 0) slots.push_back(p->template_slot);
    else for (int sidx = 0; sidx < CD_MAX_TEMPLATES; ++sidx) if (c->tpl_m[sidx] > 0) slots.push_back(sidx);
    if (slots.empty()) slots.push_back(0);
    // (the per-cluster results of the previous batch go away here: last_first stays empty until this batch has succeeded,
    // so a failure in between leaves cd_get_cluster_results with "no batch" rather than old offsets into new results)
    c->last_first.clear();
    std::vector<cd_cluster_result>& best = c->last_clusters;
    best.assign((size_t)std::max(ncl, 1), cd_cluster_result());
    std::vector<long long> orig_off((size_t)std::max(ncl, 1), -1), al_off((size_t)std::max(ncl, 1), -1);   // see cd_get_cluster_points
    {
        int q = 0;
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < c->h_fs[f].n_k; ++k, ++q) orig_off[(size_t)q] = (long long)f * c->N + coff[(size_t)q];
    }
    long long pairs = 0;
    // Several templates: when S copies of every frame's ICP sources fit its segment of the source buffers (they do unless a
    // frame is nearly all objects), every (cluster, template) pair becomes one ICP problem of ONE stage - the batch then
    // fills the chip with S x ncl problems instead of running S under-filled passes one after the other.
    const int S = (int)slots.size();
    bool one_stage = S > 1 && ncl > 0;
    std::vector<int> span((size_t)F, 0);
    for (int f = 0; f < F; ++f) {
        for (int k = first_cl[(size_t)f]; k < first_cl[(size_t)f + 1]; ++k) span[(size_t)f] += csize[(size_t)k];
        if ((long long)S * span[(size_t)f] > (long long)c->N) one_stage = false;
    }
    if ((long long)S * ncl > 0x3fffffffll) one_stage = false;
    if (one_stage) {
        const size_t need = (size_t)S * rounds_k * F * KICP;   // source offsets of copy t, round r: [t][r][F][KICP]
        if (need > c->koffx_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_koffx) hipFree(c->d_koffx);
            c->d_koffx = nullptr; c->koffx_cap = 0;
            HIPCHK(c, dalloc(&c->d_koffx, need));
            c->koffx_cap = need;
        }
        std::vector<int> tab(need, 0);
        for (int t = 0; t < S; ++t)
            for (int r = 0; r < rounds_k; ++r)
                for (int f = 0; f < F; ++f)
                    for (int k = 0; k < KICP; ++k) {
                        const int q = first_cl[(size_t)f] + r * KICP + k;
                        if (q < first_cl[(size_t)f + 1]) tab[(((size_t)t * rounds_k + r) * F + f) * KICP + k] = coff[(size_t)q] + t * span[(size_t)f];
                    }
        HIPCHK(c, copy_sync(c, c->d_koffx, tab.data(), sizeof(int) * need, hipMemcpyHostToDevice));
        for (int t = 0; t < S; ++t)
            for (int r = 0; r < rounds_k; ++r) {
                if (rounds_k > 1) {   // with a single round the tile counts of stage_cluster are still in d_tileK
                    HIPCHK(c, hipMemsetAsync(c->d_tileK, 0, sizeof(int) * (size_t)F * KICP * c->T, c->stream));
                    LAUNCH(c, launch_label_count(c->stream, c->N, F, c->T, To2, c->d_fs, p->cluster_enable, c->d_parent, c->d_rank, c->d_label, c->d_tileK, r * KICP));
                    LAUNCH(c, launch_scan_tiles(c->stream, c->d_tileK, F * KICP, c->T, nullptr, 0));
                }
                LAUNCH(c, launch_label_scatter(c->stream, c->d_obj, c->N, F, c->T, To2, c->d_fs, c->d_label, c->d_tileK, c->d_src0, c->d_src, r * KICP,
                                               c->d_koffx + (((size_t)t * rounds_k + r) * F) * KICP));
            }
        st = ensure_clusters(c, S * ncl, (long long)S * cl_points);
        if (st) return st;
        for (int t = 0; t < S; ++t) {
            int q = 0;
            for (int f = 0; f < F; ++f)
                for (int k = 0; k < c->h_fs[f].n_k; ++k, ++q) {
                    IcpCluster& cl = c->h_cl[(size_t)t * ncl + q];
                    cl.src_off = f * c->N + coff[(size_t)q] + t * span[(size_t)f];
                    cl.n = csize[(size_t)q];
                    cl.frame = f;
                    cl.k = k;
                    cl.tpl_off = c->tpl_off[slots[(size_t)t]];
                    cl.tpl_m = c->tpl_m[slots[(size_t)t]];
                    cl.tile0 = 0;
                    cl.slot = slots[(size_t)t];
                }
        }
        st = stage_icp(c, S * ncl, p, &pairs);
        if (st) return st;
        for (int t = 0; t < S; ++t)
            for (int k = 0; k < ncl; ++k) {
                cd_cluster_result r;
                fill_cluster_result(c, t * ncl + k, p, &r);
                r.template_slot = slots[(size_t)t];
                if (t == 0 || r.fitness < best[(size_t)k].fitness) { best[(size_t)k] = r; al_off[(size_t)k] = c->h_cl[(size_t)t * ncl + k].src_off; }
            }
    }
    for (size_t si = 0; si < slots.size() && !one_stage; ++si) {
        const int slot = slots[si];
        st = extract_sources(si > 0);
        if (st) return st;
        int q = 0;
        for (int f = 0; f < F; ++f) {
            for (int k = 0; k < c->h_fs[f].n_k; ++k, ++q) {
                IcpCluster& cl = c->h_cl[q];
                cl.src_off = f * c->N + coff[(size_t)q];
                cl.n = csize[(size_t)q];
                cl.frame = f;
                cl.k = k;
                cl.tpl_off = c->tpl_off[slot];
                cl.tpl_m = c->tpl_m[slot];
                cl.tile0 = 0;
                cl.slot = slot;
            }
        }
        long long pr = 0;
        st = stage_icp(c, ncl, p, &pr);
        if (st) return st;
        pairs += pr;
        for (int k = 0; k < ncl; ++k) {
            cd_cluster_result r;
            fill_cluster_result(c, k, p, &r);
            r.template_slot = slot;
            // (the aligned cloud of a pass stays in d_src only until the next pass re-extracts the sources)
            if (si == 0 || r.fitness < best[(size_t)k].fitness) { best[(size_t)k] = r; al_off[(size_t)k] = si + 1 == slots.size() ? (long long)c->h_cl[k].src_off : -1; }
        }
    }
    c->last_first = first_cl;
    c->last_orig_off.swap(orig_off);
    c->last_al_off.swap(al_off);
    c->last_nv.resize((size_t)F); c->last_no.resize((size_t)F);
    for (int f = 0; f < F; ++f) { c->last_nv[(size_t)f] = c->h_fs[f].n_v; c->last_no[(size_t)f] = c->h_fs[f].n_o; }
    c->last_clouds = true;
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    // records
    long long balg = 0;
    for (int f = 0; f < F; ++f) {
        const FrameState& s = c->h_fs[f];
        cd_frame_result& r = results[f];
        std::memset(&r, 0, sizeof(r));
        r.status = s.status;
        r.n_cropped = s.n_cropped;
        r.n_voxels = s.n_v;
        r.n_plane = s.n_plane;
        r.n_objects = s.n_o;
        r.n_clusters = s.n_k;
        r.flags = s.n_k > KICP ? CD_FRAME_MORE_CLUSTERS : 0;
        r.ransac_iterations = iterations[f];
        if (s.status == CD_OK && !c->h_have[f]) r.status = CD_ERR_NO_MODEL;
        if (c->h_have[f]) { r.plane[0] = c->h_model[f].x; r.plane[1] = c->h_model[f].y; r.plane[2] = c->h_model[f].z; r.plane[3] = c->h_model[f].w; }
        balg += 12ll * N + 12ll * s.n_v + 12ll * s.n_v * (rounds + 3) + 4ll * s.n_v + 16ll * s.n_o + 200ll * s.n_k;
        for (int k = 0; k < s.n_k; ++k) {
            const cd_cluster_result& cr = best[(size_t)(first_cl[(size_t)f] + k)];
            if (k < KICP) r.clusters[k] = cr;
            const long long b = 12ll * c->tpl_m[cr.template_slot] + 12ll * cr.size * (cr.iterations + 1);
            balg += b;
            c->timing.icp_algorithmic_bytes += b;
        }
    }
    if (plane_inliers || labels) {
        std::vector<int32_t> tmp((size_t)c->N);
        for (int f = 0; f < F; ++f) {
            const FrameState& s = c->h_fs[f];
            if (plane_inliers) {
                int32_t* dst = plane_inliers + (size_t)f * N;
                std::fill(dst, dst + N, -1);
                if (s.n_plane > 0) HIPCHK(c, copy_sync(c, dst, c->d_plane_idx + (size_t)f * c->N, sizeof(int) * s.n_plane, hipMemcpyDeviceToHost));
            }
            if (labels) {
                int32_t* dst = labels + (size_t)f * N;
                std::fill(dst, dst + N, -1);
                if (s.n_o > 0) HIPCHK(c, copy_sync(c, dst, c->d_label + (size_t)f * c->N, sizeof(int) * s.n_o, hipMemcpyDeviceToHost));
            }
        }
    }
    HIPCHK(c, hipEventSynchronize(c->ev[4]));
    for (int k = 0; k < 4; ++k) hipEventElapsedTime(&c->timing.stage_ms[k], c->ev[k], c->ev[k + 1]);
    hipEventElapsedTime(&c->timing.stage_ms[4], c->ev[0], c->ev[4]);
    c->timing.icp_pair_tests_lo = (int32_t)(pairs & 0xffffffffll);
    c->timing.icp_pair_tests_hi = (int32_t)(pairs >> 32);
    c->timing.algorithmic_bytes = balg;
    return CD_OK;
}

}  // namespace

// ==========================================================================================
extern "C" {

void cd_default_params(cd_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->crop_z_min = 0.0; p->crop_z_max = 0.9;            // gps.cpp:56
    p->crop_x_min = -0.2; p->crop_x_max = 0.2;           // gps.cpp:64
    p->leaf_size = 0.005f;                               // ground_plane_segmentation.launch:16
    p->rgb_offset = -1;
    p->plane_distance_threshold = 0.015;                 // ground_plane_segmentation.launch:18
    p->plane_max_iterations = 1000;                      // gps.cpp:88
    p->plane_optimize = 1;                               // gps.cpp:85
    p->plane_probability = 0.99;                         // PCL default
    p->extract_negative = 1;                             // launch: invert: true
    p->crop2_enable = 1; p->crop2_z_min = 0.0; p->crop2_z_max = 0.75;   // opd.cpp:335
    p->cluster_enable = 1;
    p->cluster_min_size = 200; p->cluster_max_size = 25000;             // opd.cpp:357-358
    p->cluster_tolerance = 0.02;                         // opd.cpp:356
    p->icp_max_iterations = 5000;                        // icp.cpp:173
    p->template_slot = 0;
    p->plane_model = CD_PLANE;
    p->plane_eps_angle = 0.0;
    p->icp_transformation_epsilon = 1e-9;                // icp.cpp:174
    p->icp_euclidean_fitness_epsilon = 0.0004;           // icp.cpp:176 + launch:42
    p->icp_accept_fitness = 0.0004;                      // icp.cpp:182
}

int cd_abi_version(void) { return CD_ABI_VERSION; }

int cd_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(cd_params);
        case 1: return (int)sizeof(cd_cluster_result);
        case 2: return (int)sizeof(cd_frame_result);
        case 3: return (int)sizeof(cd_timing);
        default: return -1;
    }
}

const char* cd_last_error(const cd_context* ctx) { return ctx ? ctx->err : "null context"; }

void cd_destroy(cd_context* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    void* dev[] = {c->d_in, c->d_fs, c->d_tileA, c->d_tileB, c->d_tileK, c->d_cpt, c->d_vox, c->d_obj, c->d_src0, c->d_src,
                   c->d_key[0], c->d_key[1], c->d_val[0], c->d_val[1], c->d_ghist, c->d_sstate, c->d_ticket, c->d_tile64, c->d_rnd, c->d_models, c->d_valid, c->d_counts,
                   c->d_active, c->d_model, c->d_have, c->d_sums, c->d_plane_idx, c->d_head, c->d_next, c->d_parent, c->d_csize,
                   c->d_rank, c->d_cand, c->d_sizes, c->d_label, c->d_tpl, c->d_tlo, c->d_thi, c->d_tplk, c->d_tlok, c->d_thik, c->d_kdmap, c->d_grid, c->d_tcell, c->d_nn, c->d_d2, c->d_queue, c->d_don, c->d_wgtab, c->d_order, c->d_cl, c->d_work, c->d_work2, c->d_st, c->d_acc, c->d_accf};
    for (void* p : dev) if (p) hipFree(p);
    if (c->d_koffx) hipFree(c->d_koffx);
    if (c->d_guess) hipFree(c->d_guess);
    if (c->d_super) hipFree(c->d_super);
    if (c->d_lat) hipFree(c->d_lat);
    if (c->d_tileC) hipFree(c->d_tileC);
    if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); }
    if (c->stream3) { hipStreamSynchronize(c->stream3); hipStreamDestroy(c->stream3); }
    for (auto& e : c->ev2) if (e) hipEventDestroy(e);
    void* host[] = {c->h_fs, c->h_valid, c->h_counts, c->h_active, c->h_model, c->h_models, c->h_have, c->h_sums, c->h_cl, c->h_order, c->h_work, c->h_work2, c->h_st, c->h_accf, c->h_wgtab, c->h_ctl};
    for (void* p : host) if (p) hipHostFree(p);
    for (auto& e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int cd_create(int device_id, int max_points, int max_frames, cd_context** out) {
    if (!out) return CD_ERR_INVALID_ARG;
    *out = nullptr;
    if (max_points <= 0 || max_frames <= 0) return CD_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return CD_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return CD_ERR_DEVICE;
    cd_context* c = new cd_context();
    c->device = device_id;
    c->N = max_points;
    c->F = max_frames;
    c->T = (max_points + TILE - 1) / TILE;
    const size_t N = (size_t)c->N, F = (size_t)c->F, T = (size_t)c->T, FN = F * N;
    // non-blocking: no implicit ordering against the NULL stream (torch ops, other contexts in flight)
    if (const char* m = std::getenv("CUBOID_ICP_LOWPRIO")) c->icp_lowprio = std::atoi(m);
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = prio_greatest = 0;
    bool ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, 0) == hipSuccess;
    for (auto& e : c->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && dalloc(&c->d_fs, F) == hipSuccess && halloc(&c->h_fs, F) == hipSuccess;
    ok = ok && dalloc(&c->d_tileA, F * T) == hipSuccess && dalloc(&c->d_tileB, F * T) == hipSuccess && dalloc(&c->d_tileK, F * KICP * T) == hipSuccess && dalloc(&c->d_tileC, F * T) == hipSuccess;
    ok = ok && dalloc(&c->d_cpt, FN) == hipSuccess && dalloc(&c->d_vox, FN) == hipSuccess && dalloc(&c->d_obj, FN) == hipSuccess;
    ok = ok && dalloc(&c->d_src0, FN) == hipSuccess && dalloc(&c->d_src, FN) == hipSuccess;
    for (int k = 0; k < 2; ++k) ok = ok && dalloc(&c->d_key[k], FN) == hipSuccess && dalloc(&c->d_val[k], FN) == hipSuccess;
    ok = ok && dalloc(&c->d_ghist, F * SORT_MAX_PASSES_HOST * RADIX) == hipSuccess;
    ok = ok && dalloc(&c->d_sstate, (size_t)SORT_MAX_PASSES_HOST * F * RADIX * ((N + SORT_TILE - 1) / SORT_TILE)) == hipSuccess;
    ok = ok && dalloc(&c->d_tile64, F * T) == hipSuccess;
    ok = ok && dalloc(&c->d_ticket, (size_t)F * TICKET_PITCH) == hipSuccess && hipMemset(c->d_ticket, 0, sizeof(int) * (size_t)F * TICKET_PITCH) == hipSuccess;
    ok = ok && dalloc(&c->d_rnd, (size_t)RND_TABLE) == hipSuccess;
    ok = ok && dalloc(&c->d_models, F * MAX_HYP) == hipSuccess && dalloc(&c->d_valid, F * MAX_HYP) == hipSuccess && dalloc(&c->d_counts, F * MAX_HYP) == hipSuccess;
    ok = ok && halloc(&c->h_valid, F * MAX_HYP) == hipSuccess && halloc(&c->h_counts, F * MAX_HYP) == hipSuccess && halloc(&c->h_models, F * MAX_HYP) == hipSuccess;
    ok = ok && dalloc(&c->d_active, F) == hipSuccess && halloc(&c->h_active, F) == hipSuccess;
    ok = ok && dalloc(&c->d_model, F) == hipSuccess && halloc(&c->h_model, F) == hipSuccess;
    ok = ok && dalloc(&c->d_have, F) == hipSuccess && halloc(&c->h_have, F) == hipSuccess;
    ok = ok && dalloc(&c->d_sums, F * 10) == hipSuccess && halloc(&c->h_sums, F * 10) == hipSuccess;
    ok = ok && dalloc(&c->d_plane_idx, FN) == hipSuccess && dalloc(&c->d_head, F * CELL_BUCKETS) == hipSuccess;
    ok = ok && dalloc(&c->d_next, FN) == hipSuccess && dalloc(&c->d_parent, FN) == hipSuccess && dalloc(&c->d_csize, FN) == hipSuccess;
    ok = ok && dalloc(&c->d_rank, FN) == hipSuccess && dalloc(&c->d_cand, FN) == hipSuccess && dalloc(&c->d_sizes, FN) == hipSuccess && dalloc(&c->d_label, FN) == hipSuccess;
    c->tpl_cap = 1 << 18;
    ok = ok && dalloc(&c->d_tpl, (size_t)c->tpl_cap) == hipSuccess;
    ok = ok && dalloc(&c->d_super, (size_t)CD_MAX_TEMPLATES) == hipSuccess;
    ok = ok && dalloc(&c->d_lat, (size_t)CD_MAX_TEMPLATES) == hipSuccess;
    {
        const int prio = c->icp_lowprio ? prio_least : 0;
        ok = ok && hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio) == hipSuccess;
    }
    for (auto& e : c->ev2) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && dalloc(&c->d_grid, (size_t)CD_MAX_TEMPLATES) == hipSuccess && dalloc(&c->d_tcell, (size_t)CD_MAX_TEMPLATES * ICP_CELL_STRIDE) == hipSuccess;
    ok = ok && dalloc(&c->d_tlo, (size_t)c->tpl_cap / ICP_SUB) == hipSuccess && dalloc(&c->d_thi, (size_t)c->tpl_cap / ICP_SUB) == hipSuccess;
    ok = ok && dalloc(&c->d_kdmap, (size_t)c->tpl_cap) == hipSuccess;
    ok = ok && dalloc(&c->d_tplk, (size_t)c->tpl_cap) == hipSuccess && dalloc(&c->d_tlok, (size_t)c->tpl_cap / ICP_SUB) == hipSuccess && dalloc(&c->d_thik, (size_t)c->tpl_cap / ICP_SUB) == hipSuccess;
    ok = ok && dalloc(&c->d_nn, FN) == hipSuccess && dalloc(&c->d_d2, FN) == hipSuccess && dalloc(&c->d_queue, (size_t)16) == hipSuccess && dalloc(&c->d_don, (size_t)(DON_BOX + DON_CAP)) == hipSuccess && dalloc(&c->d_wgtab, (size_t)3 * 1024) == hipSuccess;
    ok = ok && halloc(&c->h_wgtab, (size_t)3 * 1024) == hipSuccess && halloc(&c->h_ctl, (size_t)16) == hipSuccess;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    }
    const size_t ncl = F * KICP;
    c->cl_cap = (int)ncl;
    c->work_cap = (int)(F * (N / 64 + KICP + 1));
    ok = ok && dalloc(&c->d_cl, ncl) == hipSuccess && halloc(&c->h_cl, ncl) == hipSuccess;
    ok = ok && dalloc(&c->d_order, ncl) == hipSuccess && halloc(&c->h_order, ncl) == hipSuccess;
    if (const char* m = std::getenv("CUBOID_ICP_MAX_WG")) c->icp_max_wg = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_ICP_CPW")) c->icp_cpw = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_ICP_SLOTS")) c->icp_slots = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_ICP_DONATE")) c->icp_donate = std::atoi(m);
    if (const char* m = std::getenv("CUBOID_ICP_DON_IDLE")) c->don_idle = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_ICP_DON_FAULT")) c->don_fault = std::atoi(m);
    if (const char* m = std::getenv("CUBOID_ICP_LATTICE")) c->icp_lattice = std::atoi(m);
    if (const char* m = std::getenv("CUBOID_COPY_KERNELS")) c->copy_kernels = std::atoi(m);
    if (const char* m = std::getenv("CUBOID_ZERO_ONCE")) c->zero_once = std::atoi(m);
    if (const char* m = std::getenv("CUBOID_LAT_SHAPE")) std::sscanf(m, "%d,%d,%d", &c->lat_shape[0], &c->lat_shape[1], &c->lat_shape[2]);
    if (const char* m = std::getenv("CUBOID_VOXEL_RUNS")) c->voxel_runs = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_CENTROID_LANES")) c->centroid_lanes = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_CROP_DIRECT")) c->crop_direct = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_CLUSTER_CELLS")) c->cluster_cells = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_MIRROR_WRITES")) c->mirror_writes = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_MIRROR_READS")) c->mirror_reads = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_ICP_DIRECT")) c->icp_direct = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_ICP_BIG_WEIGHT")) c->icp_big_weight = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_CROP_TWO_PASS")) c->crop_two_pass = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_ICP_PERSIST")) c->icp_persist = std::atoi(m);
    if (const char* m = std::getenv("CUBOID_FORCE_SCAN_STALL")) c->force_stall = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_CROP_RUNS")) c->crop_runs = std::atoi(m) != 0;
    if (const char* m = std::getenv("CUBOID_ICP_CONCURRENT")) c->icp_concurrent = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_FRONT_CONCURRENT")) c->front_concurrent = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("CUBOID_ICP_MODE")) c->icp_mode = !std::strcmp(m, "sliced") ? 1 : (!std::strcmp(m, "cluster") ? 2 : (!std::strcmp(m, "pipe") ? 3 : 0));
    ok = ok && dalloc(&c->d_work, (size_t)c->work_cap) == hipSuccess && halloc(&c->h_work, (size_t)c->work_cap) == hipSuccess;
    ok = ok && dalloc(&c->d_work2, (size_t)c->work_cap) == hipSuccess && halloc(&c->h_work2, (size_t)c->work_cap) == hipSuccess;
    ok = ok && dalloc(&c->d_st, ncl * 2) == hipSuccess && halloc(&c->h_st, ncl * 2) == hipSuccess;
    ok = ok && dalloc(&c->d_acc, ncl * 48) == hipSuccess && dalloc(&c->d_accf, ncl + 1) == hipSuccess && halloc(&c->h_accf, ncl + 1) == hipSuccess;
    if (ok) {
        // PCL's SAC sampler: boost::mt19937 seeded 12345, uniform_int<>(0, INT_MAX) == mt() >> 1
        std::vector<int> tab((size_t)RND_TABLE);
        std::mt19937 gen(12345u);
        for (auto& v : tab) v = (int)(gen() >> 1);
        ok = copy_sync(c, c->d_rnd, tab.data(), sizeof(int) * RND_TABLE, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        cd_destroy(c);
        return CD_ERR_DEVICE;
    }
    *out = c;
    return CD_OK;
}

// ---- templates ------------------------------------------------------------------------------------------------------
// Everything cd_set_template derives from the template's points is device-independent host work (two sorted layouts, run
// boxes, the uniform grid): it is done once per distinct template and shared by every context of the process (bench.py
// keeps three contexts per GPU; the reference re-reads and re-indexes the template for every frame, icp.cpp:159).
struct PreparedTemplate {
    int m = 0, m_pad = 0;
    std::vector<float> xyz;                              // the caller's points (cache key check)
    std::vector<float4> cell_pts, cell_lo, cell_hi;      // layout 1: sorted by grid cell; boxes of its runs of 64
    std::vector<float4> kd_pts, kd_lo, kd_hi;            // layout 2: k-d patches of 64; their boxes
    std::vector<unsigned short> kdmap;                   // patch order -> cell-sorted position (LDS-resident templates)
    std::vector<unsigned short> cell_start;              // grid start table (LDS-resident templates)
    IcpGrid grid;                                        // cell_off is set per slot at upload
    IcpSuper super;                                      // second box level of a template that does not fit LDS (n = 0: none)
    bool big_ok = false;                                 // k_icp_pipe_big can search it
    IcpLattice lat;                                      // nface > 0: the template is a union of axis-aligned lattices (k_icp_lat.hip)
};

// Is the template what make_cuboid.py writes (mkc.py:38-55) - faces one after the other, each the Cartesian product of two of
// three shared, ascending, near-uniform axis tables at a constant third coordinate, first axis fastest?  Verified bit by bit
// against the points; anything else (a point moved, a row missing, the object templates) leaves nface = 0.
static void lattice_detect(const float* xyz, int m, IcpLattice* out) {
    std::memset(out, 0, sizeof(*out));
    struct Face { int w, u, v, base, nu, nv; float c; };
    std::vector<Face> faces;
    std::vector<float> tabs[3];
    auto P = [&](int i, int a) { return xyz[3 * (size_t)i + a]; };
    for (int i = 0; i < 3 * m; ++i) if (!std::isfinite(xyz[i])) return;
    int pos = 0;
    while (pos < m) {
        if ((int)faces.size() >= LAT_MAX_FACES || pos + 1 >= m) return;
        int u = -1;
        for (int a = 0; a < 3; ++a)
            if (P(pos + 1, a) != P(pos, a)) { if (u >= 0) return; u = a; }
        if (u < 0) return;
        int nu = 1;   // first row: only u moves, ascending
        while (pos + nu < m && P(pos + nu, u) > P(pos + nu - 1, u) && P(pos + nu, (u + 1) % 3) == P(pos, (u + 1) % 3) &&
               P(pos + nu, (u + 2) % 3) == P(pos, (u + 2) % 3)) ++nu;
        if (nu < 2 || pos + nu >= m) return;
        int v = -1;
        for (int a = 0; a < 3; ++a)
            if (P(pos + nu, a) != P(pos, a)) { if (v >= 0 || a == u) return; v = a; }
        if (v < 0) return;
        const int w = 3 - u - v;
        int nv = 1;   // further rows: the same u values, v constant within the row and ascending from row to row, w constant
        while (pos + (nv + 1) * nu <= m) {
            const int r = pos + nv * nu;
            bool ok = P(r, v) > P(r - nu, v);
            for (int i = 0; i < nu && ok; ++i) ok = P(r + i, u) == P(pos + i, u) && P(r + i, v) == P(r, v) && P(r + i, w) == P(pos, w);
            if (!ok) break;
            ++nv;
        }
        if (nv < 2) return;
        std::vector<float> U((size_t)nu), V((size_t)nv);
        for (int i = 0; i < nu; ++i) U[(size_t)i] = P(pos + i, u);
        for (int j = 0; j < nv; ++j) V[(size_t)j] = P(pos + j * nu, v);
        const std::pair<int, std::vector<float>*> both[2] = {{u, &U}, {v, &V}};
        for (const auto& av : both) {   // one table per axis, shared by every face that varies along it
            if (tabs[av.first].empty()) tabs[av.first] = *av.second;
            else if (tabs[av.first] != *av.second) return;
        }
        faces.push_back(Face{w, u, v, pos, nu, nv, P(pos, w)});
        pos += nu * nv;
    }
    int ntab = 0;
    for (int a = 0; a < 3; ++a) {
        const std::vector<float>& T = tabs[a];
        const int n = (int)T.size();
        out->toff[a] = ntab;
        if (n == 0) {   // no face varies along this axis: a one-entry table, so that the kernel treats every axis alike (never read by a face)
            if (ntab + 1 > LAT_MAX_TAB) { std::memset(out, 0, sizeof(*out)); return; }
            out->n[a] = 1; out->noi[a] = 0.f; out->inv[a] = 1.f;
            out->tab[ntab++] = make_float4(-INFINITY, 0.f, INFINITY, 0.f);
            continue;
        }
        out->n[a] = n;
        if (ntab + n > LAT_MAX_TAB) { std::memset(out, 0, sizeof(*out)); return; }
        const double step = ((double)T[(size_t)n - 1] - (double)T[0]) / (double)(n - 1);
        for (int i = 0; i < n; ++i)   // uniform to 1/16 of a step: the index guess of lat_axis is then at most one entry off
            if (!(std::fabs((double)T[(size_t)i] - ((double)T[0] + i * step)) <= step / 16.0)) { std::memset(out, 0, sizeof(*out)); return; }
        out->noi[a] = (float)(-(double)T[0] / step);
        out->inv[a] = (float)(1.0 / step);
        for (int i = 0; i < n; ++i)
            out->tab[ntab + i] = make_float4(i > 0 ? T[(size_t)i - 1] : -INFINITY, T[(size_t)i], i + 1 < n ? T[(size_t)i + 1] : INFINITY, 0.f);
        ntab += n;
    }
    out->ntab = ntab;
    for (size_t f = 0; f < faces.size(); ++f) {
        out->w[f] = faces[f].w; out->fast[f] = faces[f].u; out->base[f] = faces[f].base; out->c[f] = faces[f].c;
        out->m0[f] = faces[f].w == 0 ? ~0u : 0u; out->m1[f] = faces[f].w == 1 ? ~0u : 0u; out->m2[f] = faces[f].w == 2 ? ~0u : 0u;
    }
    for (size_t f = faces.size(); f < (size_t)LAT_MAX_FACES; ++f) { out->w[f] = 2; out->m2[f] = ~0u; out->c[f] = std::numeric_limits<float>::quiet_NaN(); }   // (see IcpLattice::m0)
    out->nface = (int)faces.size();
}

static std::shared_ptr<const PreparedTemplate> prepare_template(const void* xyz, size_t stride, int m) {
    std::vector<float> raw((size_t)m * 3);
    for (int i = 0; i < m; ++i) std::memcpy(&raw[3 * (size_t)i], (const char*)xyz + (size_t)i * stride, 12);
    static std::mutex mu;
    static std::vector<std::shared_ptr<const PreparedTemplate>> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const auto& e : cache)
            if (e->m == m && std::memcmp(e->xyz.data(), raw.data(), raw.size() * sizeof(float)) == 0) return e;
    }
    auto P = std::make_shared<PreparedTemplate>();
    P->m = m;
    const int m_pad = (m + ICP_SUB - 1) / ICP_SUB * ICP_SUB;   // slots start on a 64-point run boundary
    P->m_pad = m_pad;
    // Sort the template by the cells of a uniform grid over its bounding box (cell edge = 2 x the point
    // spacing, enlarged until the grid has at most ICP_MAX_CELLS cells).  The lane-per-query search of
    // k_icp.hip scans the few cell rows a query's seed ball touches; consecutive runs of 64 stored points
    // are still spatially compact (a strip of one cell row), which is what the run boxes of the
    // wave-per-query search feed on.  Each stored point keeps its ORIGINAL index in .w; the
    // nearest-neighbour tie rule (lowest original index) is evaluated on that.
    struct TP { float x, y, z; int oi; int cid; };
    std::vector<TP> tp((size_t)m);
    float gmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, gmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < m; ++i) {
        const float* v = &raw[3 * (size_t)i];
        tp[(size_t)i] = TP{v[0], v[1], v[2], i, 0};
        for (int a = 0; a < 3; ++a)
            if (std::isfinite(v[a])) { gmn[a] = std::fmin(gmn[a], v[a]); gmx[a] = std::fmax(gmx[a], v[a]); }
    }
    for (int a = 0; a < 3; ++a) if (!(gmn[a] <= gmx[a])) gmn[a] = gmx[a] = 0.f;
    IcpGrid& grid = P->grid;
    std::memset(&grid, 0, sizeof(grid));
    std::memset(&P->super, 0, sizeof(P->super));
    {
        // point spacing: median nearest-neighbour distance of a sample of (at most 64 of) the points
        std::vector<float> nn2;
        const int step = std::max(1, m / 64);
        for (int i = 0; i < m; i += step) {
            float best = FLT_MAX;
            for (int j = 0; j < m; ++j) {
                const float dx = tp[(size_t)i].x - tp[(size_t)j].x, dy = tp[(size_t)i].y - tp[(size_t)j].y, dz = tp[(size_t)i].z - tp[(size_t)j].z;
                const float d = dx * dx + dy * dy + dz * dz;
                if (d > 0.f && d < best) best = d;
            }
            if (best < FLT_MAX) nn2.push_back(best);
        }
        float pitch = 0.002f;
        if (!nn2.empty()) { std::nth_element(nn2.begin(), nn2.begin() + nn2.size() / 2, nn2.end()); pitch = std::sqrt(nn2[nn2.size() / 2]); }
        float cell_factor = 2.0f;
        if (const char* e = std::getenv("CUBOID_ICP_CELL_FACTOR")) cell_factor = (float)std::atof(e);   // tuning only
        float cell = std::fmax(cell_factor * pitch, 1.0e-4f);
        int nd[3];
        for (;;) {
            long long tot = 1;
            for (int a = 0; a < 3; ++a) {
                const double cnt = std::floor((double)(gmx[a] - gmn[a]) / cell) + 1.0;
                nd[a] = cnt > 1.0e6 ? 1000000 : (int)cnt;
                tot *= nd[a];
            }
            if (tot <= ICP_MAX_CELLS) break;
            cell *= 1.26f;
        }
        grid.ox = gmn[0]; grid.oy = gmn[1]; grid.oz = gmn[2];
        grid.cell = cell;
        grid.inv = 1.0f / cell;
        grid.nx = nd[0]; grid.ny = nd[1]; grid.nz = nd[2];
        const int ncell = nd[0] * nd[1] * nd[2];
        auto coord = [&](float v, float o, int n) {
            const float t = std::floor((v - o) * grid.inv);
            return t >= (float)(n - 1) ? n - 1 : (t > 0.f ? (int)t : 0);   // NaN -> 0
        };
        for (int i = 0; i < m; ++i) {
            TP& t = tp[(size_t)i];
            t.cid = (coord(t.z, grid.oz, grid.nz) * grid.ny + coord(t.y, grid.oy, grid.ny)) * grid.nx + coord(t.x, grid.ox, grid.nx);
        }
        std::sort(tp.begin(), tp.end(), [](const TP& a, const TP& bb) { return a.cid < bb.cid || (a.cid == bb.cid && a.oi < bb.oi); });
        if (m <= ICP_BIG_MAX) {   // (uint16 positions: templates the persistent kernels can walk)
            grid.ncell = ncell;
            P->cell_start.assign((size_t)ncell + 1, 0);
            int i = 0;
            for (int cid = 0; cid <= ncell; ++cid) {
                while (i < m && tp[(size_t)i].cid < cid) ++i;
                P->cell_start[(size_t)cid] = (unsigned short)i;
            }
        }
    }
    // one layout = the points (original index in .w) and the axis-aligned box of every run of 64 consecutive STORED
    // points (exact float min/max)
    auto layout = [&](std::vector<float4>& pts, std::vector<float4>& lo, std::vector<float4>& hi) {
        // (the last run is filled up with points at +inf, original index INT_MAX: kernels that read a whole run from global
        // memory - k_icp_pipe_big - meet them as candidates that can never win)
        const int imax = 0x7fffffff;
        float wpad;
        std::memcpy(&wpad, &imax, 4);
        pts.assign((size_t)m_pad, make_float4(INFINITY, INFINITY, INFINITY, wpad));
        for (int i = 0; i < m; ++i) {
            float w;
            std::memcpy(&w, &tp[(size_t)i].oi, 4);
            pts[(size_t)i] = make_float4(tp[(size_t)i].x, tp[(size_t)i].y, tp[(size_t)i].z, w);
        }
        const int nrun = m_pad / ICP_SUB;
        lo.resize((size_t)nrun);
        hi.resize((size_t)nrun);
        for (int r = 0; r < nrun; ++r) {
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int i = r * ICP_SUB; i < std::min(m, (r + 1) * ICP_SUB); ++i) {
                const float v[3] = {tp[(size_t)i].x, tp[(size_t)i].y, tp[(size_t)i].z};
                for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], v[a]); mx[a] = std::fmax(mx[a], v[a]); }
            }
            lo[(size_t)r] = make_float4(mn[0], mn[1], mn[2], 0.f);
            hi[(size_t)r] = make_float4(mx[0], mx[1], mx[2], 0.f);
        }
    };
    layout(P->cell_pts, P->cell_lo, P->cell_hi);                      // layout 1: cell-sorted (whole-cluster kernels)
    std::vector<int> pos_cell((size_t)m);                             // original index -> position in layout 1
    for (int i = 0; i < m; ++i) pos_cell[(size_t)tp[(size_t)i].oi] = i;
    std::vector<std::pair<int, int>> chunks;  // k-d subtrees of <= ICP_TPL_LDS points whose parent is larger (templates that do not fit LDS)
    std::vector<std::pair<int, int>> supers;  // k-d subtrees of <= 64 patches whose parent is larger (same templates: IcpSuper)
    {
        // Layout 2, for the wave-per-query search: compact patches of 64 points from k-d median splits whose left part is
        // a multiple of 64, so that consecutive runs of 64 stored points have the smallest boxes the run-box pruning can get.
        std::vector<std::pair<int, int>> stack;   // [lo, hi)
        stack.push_back({0, m});
        while (!stack.empty()) {
            const auto [lo, hi] = stack.back();
            stack.pop_back();
            const int n = hi - lo;
            if (m > ICP_TPL_LDS && n <= ICP_TPL_LDS) {
                bool inside = false;   // already inside a recorded chunk?
                for (const auto& ch : chunks) inside = inside || (lo >= ch.first && hi <= ch.second);
                if (!inside) chunks.push_back({lo, hi});
            }
            if (m > ICP_TPL_LDS && n <= 64 * ICP_SUB) {
                bool inside = false;
                for (const auto& su : supers) inside = inside || (lo >= su.first && hi <= su.second);
                if (!inside) supers.push_back({lo, hi});
            }
            if (n <= ICP_SUB) {
                std::sort(tp.begin() + lo, tp.begin() + hi, [](const TP& a, const TP& bb) { return a.oi < bb.oi; });
                continue;
            }
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int i = lo; i < hi; ++i) {
                const float v[3] = {tp[(size_t)i].x, tp[(size_t)i].y, tp[(size_t)i].z};
                for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], v[a]); mx[a] = std::fmax(mx[a], v[a]); }
            }
            int ax = 0;
            if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
            if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
            int k = ((n / 2 + ICP_SUB - 1) / ICP_SUB) * ICP_SUB;
            if (k >= n) k -= ICP_SUB;
            auto key = [ax](const TP& t) { return ax == 0 ? t.x : (ax == 1 ? t.y : t.z); };
            std::nth_element(tp.begin() + lo, tp.begin() + lo + k, tp.begin() + hi,
                             [&](const TP& a, const TP& bb) { return key(a) < key(bb) || (key(a) == key(bb) && a.oi < bb.oi); });
            stack.push_back({lo + k, hi});
            stack.push_back({lo, lo + k});
        }
    }
    layout(P->kd_pts, P->kd_lo, P->kd_hi);
    if (m > ICP_TPL_LDS && (int)chunks.size() <= ICP_MAX_CHUNKS) {   // chunk table of a template that does not fit LDS
        std::sort(chunks.begin(), chunks.end());
        grid.nchunk = (int)chunks.size();
        for (int ci = 0; ci < grid.nchunk; ++ci) {
            const int lo = chunks[(size_t)ci].first, hi = chunks[(size_t)ci].second;
            grid.chunk_start[ci] = lo;
            grid.chunk_n[ci] = hi - lo;
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int r = lo / ICP_SUB; r < (hi + ICP_SUB - 1) / ICP_SUB; ++r) {
                const float lo3[3] = {P->kd_lo[(size_t)r].x, P->kd_lo[(size_t)r].y, P->kd_lo[(size_t)r].z};
                const float hi3[3] = {P->kd_hi[(size_t)r].x, P->kd_hi[(size_t)r].y, P->kd_hi[(size_t)r].z};
                for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], lo3[a]); mx[a] = std::fmax(mx[a], hi3[a]); }
            }
            for (int a = 0; a < 3; ++a) { grid.chunk_lo[ci][a] = mn[a]; grid.chunk_hi[ci][a] = mx[a]; }
            grid.chunk_lo[ci][3] = grid.chunk_hi[ci][3] = 0.f;
        }
    }
    {   // the two halves of the root split (the stack above splits [0, m) at k0 first; patches are whole on either side)
        int k0 = m;
        if (m > ICP_SUB) {
            k0 = ((m / 2 + ICP_SUB - 1) / ICP_SUB) * ICP_SUB;
            if (k0 >= m) k0 -= ICP_SUB;
        }
        grid.kd_split = k0 >= m ? m_pad / ICP_SUB : k0 / ICP_SUB;
        const int nrun = m_pad / ICP_SUB;
        for (int h = 0; h < 2; ++h) {
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            const int r0 = h == 0 ? 0 : grid.kd_split, r1 = h == 0 ? grid.kd_split : nrun;
            for (int r = r0; r < r1; ++r) {
                const float lo3[3] = {P->kd_lo[(size_t)r].x, P->kd_lo[(size_t)r].y, P->kd_lo[(size_t)r].z};
                const float hi3[3] = {P->kd_hi[(size_t)r].x, P->kd_hi[(size_t)r].y, P->kd_hi[(size_t)r].z};
                for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], lo3[a]); mx[a] = std::fmax(mx[a], hi3[a]); }
            }
            for (int a = 0; a < 3; ++a) { grid.half_lo[h][a] = mn[a]; grid.half_hi[h][a] = mx[a]; }   // an empty half keeps +-FLT_MAX: never reached
            grid.half_lo[h][3] = grid.half_hi[h][3] = 0.f;
        }
    }
    if (m <= ICP_BIG_MAX) {
        // k-d patch r = the cell-sorted positions kdmap[64 r .. 64 r + 63]: the pipelined kernel searches far queries
        // patch by patch THROUGH this table (compact boxes) while the points themselves stay cell-sorted in LDS; for a
        // template in global memory (k_icp_pipe_big) it turns the position of a k-d ordered point into its cell-sorted one
        P->kdmap.assign((size_t)m_pad, (unsigned short)std::min(m_pad, 65535));   // padding -> the +inf pad run
        for (int i = 0; i < m; ++i) P->kdmap[(size_t)i] = (unsigned short)pos_cell[(size_t)tp[(size_t)i].oi];
    }
    if (m > ICP_TPL_LDS && m <= ICP_BIG_MAX && m_pad / ICP_SUB <= ICP_BIG_PATCHES && !supers.empty() && supers.size() <= 64 && grid.ncell > 0) {
        std::sort(supers.begin(), supers.end());
        IcpSuper& su = P->super;
        su.n = (int)supers.size();
        bool ok = true;
        int covered = 0;
        for (int k = 0; k < su.n; ++k) {
            const int lo = supers[(size_t)k].first, hi = supers[(size_t)k].second;
            ok = ok && lo % ICP_SUB == 0 && lo == covered;   // runs of whole patches that tile the template
            covered = hi;
            su.first[k] = lo / ICP_SUB;
            su.cnt[k] = (hi - lo + ICP_SUB - 1) / ICP_SUB;
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int r = su.first[k]; r < su.first[k] + su.cnt[k]; ++r) {
                const float lo3[3] = {P->kd_lo[(size_t)r].x, P->kd_lo[(size_t)r].y, P->kd_lo[(size_t)r].z};
                const float hi3[3] = {P->kd_hi[(size_t)r].x, P->kd_hi[(size_t)r].y, P->kd_hi[(size_t)r].z};
                for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], lo3[a]); mx[a] = std::fmax(mx[a], hi3[a]); }
            }
            for (int a = 0; a < 3; ++a) { su.lo[k][a] = mn[a]; su.hi[k][a] = mx[a]; }
            su.lo[k][3] = su.hi[k][3] = 0.f;
        }
        P->big_ok = ok && covered == m;
        if (!P->big_ok) su.n = 0;
    }
    lattice_detect(raw.data(), m, &P->lat);
    P->xyz.swap(raw);
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= 16) cache.erase(cache.begin());
    cache.push_back(P);
    return P;
}

// copies a prepared template into the context's template arena at point offset `off` (a multiple of 64)
static int upload_template(cd_context* c, int slot, int off, const PreparedTemplate& P) {
    const int nrun = P.m_pad / ICP_SUB;
    HIPCHK(c, copy_sync(c, c->d_tpl + off, P.cell_pts.data(), sizeof(float4) * (size_t)P.m_pad, hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->d_tlo + off / ICP_SUB, P.cell_lo.data(), sizeof(float4) * (size_t)nrun, hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->d_thi + off / ICP_SUB, P.cell_hi.data(), sizeof(float4) * (size_t)nrun, hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->d_tplk + off, P.kd_pts.data(), sizeof(float4) * (size_t)P.m_pad, hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->d_tlok + off / ICP_SUB, P.kd_lo.data(), sizeof(float4) * (size_t)nrun, hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->d_thik + off / ICP_SUB, P.kd_hi.data(), sizeof(float4) * (size_t)nrun, hipMemcpyHostToDevice));
    if (!P.kdmap.empty())
        HIPCHK(c, copy_sync(c, c->d_kdmap + off, P.kdmap.data(), sizeof(unsigned short) * P.kdmap.size(), hipMemcpyHostToDevice));
    IcpGrid grid = P.grid;
    grid.cell_off = slot * ICP_CELL_STRIDE;
    if (!P.cell_start.empty())
        HIPCHK(c, copy_sync(c, c->d_tcell + grid.cell_off, P.cell_start.data(), sizeof(unsigned short) * P.cell_start.size(), hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->d_grid + slot, &grid, sizeof(grid), hipMemcpyHostToDevice));
    c->tpl_off[slot] = off;
    c->tpl_m[slot] = P.m;
    c->tpl_gridded[slot] = grid.ncell > 0;
    c->tpl_big[slot] = P.big_ok;
    HIPCHK(c, copy_sync(c, c->d_super + slot, &P.super, sizeof(IcpSuper), hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->d_lat + slot, &P.lat, sizeof(IcpLattice), hipMemcpyHostToDevice));
    c->tpl_faces[slot] = P.lat.nface;
    return CD_OK;
}

int cd_set_template(cd_context* c, int slot, const void* xyz, size_t stride, int m) {
    if (!c) return CD_ERR_INVALID_ARG;
    if (slot < 0 || slot >= CD_MAX_TEMPLATES || !xyz || m <= 0 || stride < 12) return fail(c, CD_ERR_INVALID_ARG, "bad template arguments");
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::shared_ptr<const PreparedTemplate> P = prepare_template(xyz, stride, m);
    // a slot is re-used in place when the new template fits its space, appended otherwise; when the arena is full the
    // live slots are packed again (the space of replaced templates is reclaimed) before giving up
    const int old_pad = c->tpl_prep[slot] ? c->tpl_prep[slot]->m_pad : 0;
    if (old_pad >= P->m_pad) {
        c->tpl_prep[slot] = P;
        return upload_template(c, slot, c->tpl_off[slot], *P);
    }
    if (c->tpl_used + P->m_pad <= c->tpl_cap) {
        const int off = c->tpl_used;
        c->tpl_used += P->m_pad;
        c->tpl_prep[slot] = P;
        return upload_template(c, slot, off, *P);
    }
    long long total = P->m_pad;
    for (int k = 0; k < CD_MAX_TEMPLATES; ++k) if (k != slot && c->tpl_prep[k]) total += c->tpl_prep[k]->m_pad;
    if (total > c->tpl_cap) return fail(c, CD_ERR_CAPACITY, "template storage exhausted");
    c->tpl_prep[slot] = P;
    int off = 0;
    for (int k = 0; k < CD_MAX_TEMPLATES; ++k) {
        if (!c->tpl_prep[k]) continue;
        if (int ust = upload_template(c, k, off, *c->tpl_prep[k])) return ust;
        off += c->tpl_prep[k]->m_pad;
    }
    c->tpl_used = off;
    return CD_OK;
}

int cd_template_lattice_faces(const cd_context* c, int slot) {
    if (!c || slot < 0 || slot >= CD_MAX_TEMPLATES) return CD_ERR_INVALID_ARG;
    if (c->tpl_m[slot] <= 0) return CD_ERR_NO_TEMPLATE;
    return c->tpl_faces[slot];
}

int cd_lattice_detect(const void* xyz, size_t stride, int m, int32_t* out) {
    if (!xyz || m <= 0 || stride < 12) return CD_ERR_INVALID_ARG;
    std::vector<float> raw((size_t)m * 3);
    for (int i = 0; i < m; ++i) std::memcpy(&raw[3 * (size_t)i], (const char*)xyz + (size_t)i * stride, 12);
    auto L = std::make_unique<IcpLattice>();
    lattice_detect(raw.data(), m, L.get());
    for (int f = 0; out && f < L->nface; ++f) {
        const int w = L->w[f], u = L->fast[f], v = 3 - w - u;
        const int32_t row[5] = {w, u, L->base[f], L->n[u], L->n[v]};
        std::memcpy(out + 5 * f, row, sizeof(row));
    }
    return L->nface;
}

int cd_template_nearest(cd_context* c, int slot, const void* queries, size_t stride, int n, int32_t* out_index, float* out_d2) {
    if (!c) return CD_ERR_INVALID_ARG;
    if (slot < 0 || slot >= CD_MAX_TEMPLATES || !queries || n <= 0 || stride < 12 || !out_index || !out_d2) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    if (c->tpl_m[slot] <= 0) return fail(c, CD_ERR_NO_TEMPLATE, "template slot empty");
    if (c->tpl_faces[slot] <= 0) return fail(c, CD_ERR_INVALID_ARG, "the slot's template is not a lattice: its nearest-neighbour searches only exist inside the ICP kernels");
    if ((long long)n > (long long)c->N * c->F) return fail(c, CD_ERR_CAPACITY, "more queries than the context holds points");
    hipSetDevice(c->device);
    invalidate_last(c);
    int st = upload_points(c, queries, stride, n, c->d_src0);
    if (st) return st;
    LAUNCH(c, launch_lat_nn(c->stream, c->d_lat + slot, c->d_src0, n, c->d_nn, c->d_d2));
    HIPCHK(c, copy_sync(c, out_index, c->d_nn, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(c, copy_sync(c, out_d2, c->d_d2, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return CD_OK;
}

static int cd_crop_voxel_impl(cd_context* c, const void* points, size_t stride, int n, const cd_params* p, float* out_xyz,
                  uint32_t* out_rgb, int capacity, int* out_n_cropped, int* out_n_voxels) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    int st = check_params(c, p);
    if (st) return st;
    if (!points || !out_xyz || n < 0 || stride < 12 || (stride & 3)) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    if (n > c->N) return fail(c, CD_ERR_CAPACITY, "more points than the context capacity");
    if (out_n_cropped) *out_n_cropped = 0;
    if (out_n_voxels) *out_n_voxels = 0;
    if (n == 0) return CD_OK;
    st = ensure_input(c, (size_t)n * stride);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->d_in, points, (size_t)n * stride, hipMemcpyHostToDevice, c->stream));
    st = stage_crop_voxel(c, c->d_in, stride, n, 1, p, nullptr);
    if (st) return st;
    st = sync_fs(c, 1);
    if (st) return st;
    const FrameState& s = c->h_fs[0];
    if (out_n_cropped) *out_n_cropped = s.n_cropped;
    if (s.status != CD_OK) return fail(c, s.status, "voxel grid: leaf size too small for the input extent");
    if (s.n_v > capacity) return fail(c, CD_ERR_CAPACITY, "output capacity too small");
    std::vector<float4> tmp((size_t)std::max(s.n_v, 1));
    HIPCHK(c, copy_sync(c, tmp.data(), c->d_vox, sizeof(float4) * s.n_v, hipMemcpyDeviceToHost));
    for (int i = 0; i < s.n_v; ++i) {
        out_xyz[3 * i] = tmp[i].x; out_xyz[3 * i + 1] = tmp[i].y; out_xyz[3 * i + 2] = tmp[i].z;
        if (out_rgb) std::memcpy(&out_rgb[i], &tmp[i].w, 4);
    }
    if (out_n_voxels) *out_n_voxels = s.n_v;
    return CD_OK;
}

// helper: load a caller cloud as the (single-frame) voxel cloud / object cloud
static int load_as(cd_context* c, const void* xyz, size_t stride, int n, float4* dst, int32_t FrameState::*count) {
    if (n > c->N) return fail(c, CD_ERR_CAPACITY, "more points than the context capacity");
    int st = upload_points(c, xyz, stride, n, dst);
    if (st) return st;
    std::memset(&c->h_fs[0], 0, sizeof(FrameState));
    c->h_fs[0].*count = n;
    // origin for the cluster hash: min of the cloud
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const char* b = (const char*)xyz;
    for (int i = 0; i < n; ++i) {
        float v[3];
        std::memcpy(v, b + (size_t)i * stride, 12);
        for (int a = 0; a < 3; ++a) if (v[a] < mn[a]) mn[a] = v[a];
    }
    for (int a = 0; a < 3; ++a) c->h_fs[0].origin[a] = n > 0 ? mn[a] : 0.f;
    HIPCHK(c, xfer(c, c->d_fs, c->h_fs, sizeof(FrameState), hipMemcpyHostToDevice));
    return CD_OK;
}

static int cd_segment_plane_impl(cd_context* c, const void* xyz, size_t stride, int n, const cd_params* p, float coeff[4],
                     int32_t* inliers, int capacity, int* out_n_inliers, int* out_iterations) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    int st = check_params(c, p);
    if (st) return st;
    if (!xyz || !coeff || !inliers || n < 0 || stride < 12) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    if (out_n_inliers) *out_n_inliers = 0;
    if (out_iterations) *out_iterations = 0;
    st = load_as(c, xyz, stride, n, c->d_vox, &FrameState::n_v);
    if (st) return st;
    std::vector<int> iters;
    st = stage_plane(c, 1, p, iters, nullptr);
    if (st) return st;
    if (out_iterations) *out_iterations = iters[0];
    if (!c->h_have[0]) return CD_ERR_NO_MODEL;
    cd_params q = *p;
    q.extract_negative = 1;
    q.crop2_enable = 0;
    st = stage_extract(c, 1, &q);
    if (st) return st;
    st = sync_fs(c, 1);
    if (st) return st;
    const int ni = c->h_fs[0].n_plane;
    if (ni > capacity) return fail(c, CD_ERR_CAPACITY, "inlier capacity too small");
    if (ni > 0) HIPCHK(c, copy_sync(c, inliers, c->d_plane_idx, sizeof(int) * ni, hipMemcpyDeviceToHost));
    coeff[0] = c->h_model[0].x; coeff[1] = c->h_model[0].y; coeff[2] = c->h_model[0].z; coeff[3] = c->h_model[0].w;
    if (out_n_inliers) *out_n_inliers = ni;
    return CD_OK;
}

// surface_normal_estimation.cpp:167-234.  The three constrained fits run on the device (cd_segment_plane's
// stages); the bookkeeping between them (ExtractIndices, pcl::compute3DCentroid - a sequential float32 sum -,
// the size sort, the handedness flip and the pose assembly) is the callback's own host code.
static int cd_surface_frame_impl(cd_context* c, const void* xyz, size_t stride, int n, const float table_normal[3], int invert,
                     const cd_params* p, cd_surface_frame_result* out) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    int st = check_params(c, p);
    if (st) return st;
    if ((!xyz && n > 0) || !table_normal || !out || n < 0 || stride < 12) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    std::memset(out, 0, sizeof(*out));
    struct P3 { float x, y, z; };
    std::vector<P3> cloud((size_t)n);
    for (int i = 0; i < n; ++i) std::memcpy(&cloud[(size_t)i], (const char*)xyz + (size_t)i * stride, 12);
    float normals[3][4], mids[3][4];
    int counts[3];
    for (int i = 0; i < 3; ++i) {   // sne.cpp:183-197
        cd_params q = *p;
        q.plane_model = i == 0 ? CD_PLANE_PERPENDICULAR : CD_PLANE_PARALLEL;
        for (int a = 0; a < 3; ++a) q.plane_axis[a] = table_normal[a];
        q.plane_eps_angle = 0.1;                       // sne.cpp:123
        q.plane_optimize = 1;                          // sne.cpp:118
        q.plane_max_iterations = 1000;                 // sne.cpp:125
        q.extract_negative = 1;
        q.crop2_enable = 0;
        q.bbox_enable = 0;
        const int m = (int)cloud.size();
        if (m > c->N) return fail(c, CD_ERR_CAPACITY, "more points than the context capacity");
        st = load_as(c, cloud.data(), sizeof(P3), m, c->d_vox, &FrameState::n_v);
        if (st) return st;
        std::vector<int> iters;
        st = stage_plane(c, 1, &q, iters, nullptr);
        if (st) return st;
        out->iterations[i] = iters[0];
        if (!c->h_have[0]) return CD_ERR_NO_MODEL;
        st = stage_extract(c, 1, &q);
        if (st) return st;
        st = sync_fs(c, 1);
        if (st) return st;
        const int ni = c->h_fs[0].n_plane;
        std::vector<int> inl((size_t)std::max(ni, 1));
        if (ni > 0) HIPCHK(c, copy_sync(c, inl.data(), c->d_plane_idx, sizeof(int) * ni, hipMemcpyDeviceToHost));
        // getNormal(): plane_pc = ExtractIndices(negative = !invert), leftover = ExtractIndices(negative = invert)
        std::vector<char> is_inl((size_t)std::max(m, 1), 0);
        for (int k = 0; k < ni; ++k) is_inl[(size_t)inl[(size_t)k]] = 1;
        std::vector<P3> plane_pc, leftover;
        for (int k = 0; k < m; ++k) {
            const bool in = is_inl[(size_t)k] != 0;
            if (in == (invert != 0)) plane_pc.push_back(cloud[(size_t)k]); else leftover.push_back(cloud[(size_t)k]);
        }
        // pcl::compute3DCentroid: sequential float32 sums, then one division per component
        float cs[3] = {0.f, 0.f, 0.f};
        for (const P3& q3 : plane_pc) { cs[0] += q3.x; cs[1] += q3.y; cs[2] += q3.z; }
        const float cnt = (float)plane_pc.size();
        for (int a = 0; a < 3; ++a) mids[i][a] = cs[a] / cnt;
        mids[i][3] = 0.f;
        normals[i][0] = c->h_model[0].x; normals[i][1] = c->h_model[0].y; normals[i][2] = c->h_model[0].z; normals[i][3] = c->h_model[0].w;
        counts[i] = (int)plane_pc.size();
        cloud.swap(leftover);
    }
    // sne.cpp:199-212: order by size (the callback's own exchange loops), largest first
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j)
            if (counts[i] < counts[j]) {
                std::swap(counts[i], counts[j]);
                for (int a = 0; a < 4; ++a) { std::swap(normals[i][a], normals[j][a]); std::swap(mids[i][a], mids[j][a]); }
            }
    hm::surface_frame(normals, mids, out->Rt);
    for (int i = 0; i < 3; ++i) {
        out->n_points[i] = counts[i];
        for (int a = 0; a < 4; ++a) { out->coeff[i][a] = normals[i][a]; out->midpoint[i][a] = mids[i][a]; }
    }
    return CD_OK;
}

static int cd_bbox_filter_impl(cd_context* c, const void* xyz, size_t stride, int n, const double P[12], const int32_t rect[4],
                   int32_t* out_indices, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    if ((!xyz && n > 0) || !P || !rect || !out_indices || !out_n || n < 0 || stride < 12) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    *out_n = 0;
    if (n == 0) return CD_OK;
    int st = load_as(c, xyz, stride, n, c->d_vox, &FrameState::n_v);
    if (st) return st;
    cd_params q;
    cd_default_params(&q);
    for (int i = 0; i < 12; ++i) q.bbox_P[i] = P[i];
    for (int i = 0; i < 4; ++i) q.bbox_rect[i] = rect[i];
    HIPCHK(c, hipMemsetAsync(c->d_have, 0, sizeof(int), c->stream));
    c->h_active[0] = 0;   // (no plane: with mirror reads the kernels look here; no kernel of this context is in flight)
    st = stage_extract(c, 1, &q, 2);
    if (st) return st;
    st = sync_fs(c, 1);
    if (st) return st;
    const int ni = c->h_fs[0].n_plane;
    if (ni > capacity) return fail(c, CD_ERR_CAPACITY, "index capacity too small");
    if (ni > 0) HIPCHK(c, copy_sync(c, out_indices, c->d_plane_idx, sizeof(int) * ni, hipMemcpyDeviceToHost));
    *out_n = ni;
    return CD_OK;
}

static int cd_extract_impl(cd_context* c, const void* points, size_t stride, int n, const int32_t* indices, int n_indices, int negative,
               void* out_points, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    if ((!points && n > 0) || (!indices && n_indices > 0) || !out_n || n < 0 || n_indices < 0 || capacity < 0 || stride < 4 || (stride & 3) ||
        (!out_points && capacity > 0))
        return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    *out_n = 0;
    if (n > c->N || (!negative && n_indices > c->N)) return fail(c, CD_ERR_CAPACITY, "more points than the context capacity");
    if (n == 0) return CD_OK;
    const int words = (int)(stride / 4);
    // staging buffer: the input records, then room for the records that are kept
    const size_t in_bytes = ((size_t)n * stride + 255) & ~(size_t)255;
    int st = ensure_input(c, in_bytes + (size_t)(negative ? n : n_indices) * stride);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->d_in, points, (size_t)n * stride, hipMemcpyHostToDevice, c->stream));
    // index list -> d_label (upload), kept indices -> d_plane_idx
    const int mi = std::min(n_indices, c->N);
    if (mi > 0) HIPCHK(c, hipMemcpyAsync(c->d_label, indices, sizeof(int) * (size_t)mi, hipMemcpyHostToDevice, c->stream));
    int kept = 0;
    const int* d_keep = c->d_label;
    if (negative) {
        if (n_indices > c->N) return fail(c, CD_ERR_CAPACITY, "index list longer than the context capacity");
        std::memset(&c->h_fs[0], 0, sizeof(FrameState));
        HIPCHK(c, xfer(c, c->d_fs, c->h_fs, sizeof(FrameState), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemsetAsync(c->d_rank, 0, sizeof(int) * (size_t)n, c->stream));            // marks
        HIPCHK(c, hipMemsetAsync(c->d_tileA, 0, sizeof(int) * (size_t)c->T, c->stream));          // chained-scan state
        HIPCHK(c, hipMemsetAsync(c->d_ticket, 0, sizeof(int), c->stream));
        LAUNCH(c, launch_mark_indices(c->stream, c->d_label, mi, n, c->d_rank));
        LAUNCH(c, launch_select_unmarked(c->stream, c->d_rank, n, c->d_tileA, c->d_fs, c->d_plane_idx, c->d_ticket));
        st = sync_fs(c, 1);
        if (st) return st;
        kept = c->h_fs[0].n_plane;
        d_keep = c->d_plane_idx;
    } else {
        for (int i = 0; i < n_indices; ++i)
            if (indices[i] < 0 || indices[i] >= n) return fail(c, CD_ERR_INVALID_ARG, "index out of range");
        kept = n_indices;
    }
    if (kept > capacity) return fail(c, CD_ERR_CAPACITY, "output capacity too small");
    if (kept > 0) {
        char* d_out = (char*)c->d_in + in_bytes;
        LAUNCH(c, launch_gather_records(c->stream, c->d_in, words, d_keep, kept, d_out));
        HIPCHK(c, copy_sync(c, out_points, d_out, (size_t)kept * stride, hipMemcpyDeviceToHost));
    } else {
        HIPCHK(c, hipStreamSynchronize(c->stream));   // the uploads read caller memory
    }
    *out_n = kept;
    return CD_OK;
}

static int cd_passthrough_impl(cd_context* c, const void* points, size_t stride, int n, int field, double lo, double hi, int negative,
                               void* out_points, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    if ((!points && n > 0) || !out_n || n < 0 || capacity < 0 || stride < 12 || (stride & 3) || (!out_points && capacity > 0) || field < -1 || field > 2)
        return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    *out_n = 0;
    if (n > c->N) return fail(c, CD_ERR_CAPACITY, "more points than the context capacity");
    if (n == 0) return CD_OK;
    const int words = (int)(stride / 4);
    const size_t in_bytes = ((size_t)n * stride + 255) & ~(size_t)255;   // staging: the input records, then the records that are kept
    int st = ensure_input(c, in_bytes + (size_t)n * stride);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->d_in, points, (size_t)n * stride, hipMemcpyHostToDevice, c->stream));
    std::memset(&c->h_fs[0], 0, sizeof(FrameState));
    HIPCHK(c, xfer(c, c->d_fs, c->h_fs, sizeof(FrameState), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemsetAsync(c->d_tileA, 0, sizeof(int) * (size_t)c->T, c->stream));          // chained-scan state
    HIPCHK(c, hipMemsetAsync(c->d_ticket, 0, sizeof(int), c->stream));
    LAUNCH(c, launch_passthrough_mark(c->stream, c->d_in, stride, n, field < 0 ? -1 : 4 * field, lo, hi, negative ? 1 : 0, c->d_rank));
    LAUNCH(c, launch_select_unmarked(c->stream, c->d_rank, n, c->d_tileA, c->d_fs, c->d_plane_idx, c->d_ticket));
    st = sync_fs(c, 1);
    if (st) return st;
    const int kept = c->h_fs[0].n_plane;
    if (kept > capacity) return fail(c, CD_ERR_CAPACITY, "output capacity too small");
    if (kept > 0) {
        char* d_out = (char*)c->d_in + in_bytes;
        LAUNCH(c, launch_gather_records(c->stream, c->d_in, words, c->d_plane_idx, kept, d_out));
        HIPCHK(c, copy_sync(c, out_points, d_out, (size_t)kept * stride, hipMemcpyDeviceToHost));
    }
    *out_n = kept;
    return CD_OK;
}

static int cd_cluster_impl(cd_context* c, const void* xyz, size_t stride, int n, const cd_params* p, int32_t* labels,
               int32_t* sizes, int sizes_capacity, int* out_k) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    int st = check_params(c, p);
    if (st) return st;
    if (!xyz || !labels || n < 0 || stride < 12 || (sizes_capacity > 0 && !sizes)) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    if (out_k) *out_k = 0;
    if (n == 0) return CD_OK;
    st = load_as(c, xyz, stride, n, c->d_obj, &FrameState::n_o);
    if (st) return st;
    st = stage_cluster_sync(c, 1, p, n);
    if (st) return st;
    HIPCHK(c, copy_sync(c, labels, c->d_label, sizeof(int) * n, hipMemcpyDeviceToHost));
    const int K = c->h_fs[0].n_k;
    const int ks = std::min(K, sizes_capacity);
    if (ks > 0) HIPCHK(c, copy_sync(c, sizes, c->d_sizes, sizeof(int) * ks, hipMemcpyDeviceToHost));
    if (out_k) *out_k = K;
    return CD_OK;
}

static int cd_icp_impl(cd_context* c, int slot, const void* src_xyz, size_t stride, int n, const cd_params* p,
           cd_cluster_result* out, float* aligned) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    int st = check_params(c, p);
    if (st) return st;
    if (!src_xyz || !out || n < 0 || stride < 12 || slot < 0 || slot >= CD_MAX_TEMPLATES) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    if (n > c->N) return fail(c, CD_ERR_CAPACITY, "more points than the context capacity");
    if (c->tpl_m[slot] <= 0) return fail(c, CD_ERR_NO_TEMPLATE, "template slot is empty");
    st = upload_points(c, src_xyz, stride, n, c->d_src0);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->d_src, c->d_src0, sizeof(float4) * (size_t)std::max(n, 1), hipMemcpyDeviceToDevice, c->stream));
    IcpCluster& cl = c->h_cl[0];
    cl.src_off = 0; cl.n = n; cl.frame = 0; cl.k = 0; cl.tpl_off = c->tpl_off[slot]; cl.tpl_m = c->tpl_m[slot]; cl.tile0 = 0; cl.slot = slot;
    st = stage_icp(c, 1, p, nullptr);
    if (st) return st;
    fill_cluster_result(c, 0, p, out);
    out->template_slot = slot;
    if (aligned && n > 0) {
        std::vector<float4> tmp((size_t)n);
        HIPCHK(c, copy_sync(c, tmp.data(), c->d_src, sizeof(float4) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) { aligned[3 * i] = tmp[i].x; aligned[3 * i + 1] = tmp[i].y; aligned[3 * i + 2] = tmp[i].z; }
    }
    return c->h_st[0].status;
}

static int cd_process_batch_device_impl(cd_context* c, const void* d_frames, size_t stride, int points_per_frame, int n_frames,
                            const cd_params* p, cd_frame_result* results, int32_t* plane_inliers, int32_t* labels) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    return process_batch_impl(c, d_frames, stride, points_per_frame, n_frames, p, results, plane_inliers, labels);
}

static int cd_process_batch_impl(cd_context* c, const void* frames, size_t stride, int points_per_frame, int n_frames,
                     const cd_params* p, cd_frame_result* results, int32_t* plane_inliers, int32_t* labels) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    if (!frames || points_per_frame <= 0 || n_frames <= 0) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    const size_t bytes = (size_t)points_per_frame * n_frames * stride;
    int st = ensure_input(c, bytes);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->d_in, frames, bytes, hipMemcpyHostToDevice, c->stream));
    return process_batch_impl(c, c->d_in, stride, points_per_frame, n_frames, p, results, plane_inliers, labels);
}

int cd_process_frame(cd_context* c, const void* points, size_t stride, int n, const cd_params* p, cd_frame_result* result,
                     int32_t* plane_inliers, int32_t* labels) {
    return cd_process_batch(c, points, stride, n, 1, p, result, plane_inliers, labels);
}

int cd_get_cluster_results(const cd_context* c, int frame, int first, int capacity, cd_cluster_result* out, int* out_total) {
    if (!c) return CD_ERR_INVALID_ARG;
    if (out_total) *out_total = 0;
    if (frame < 0 || first < 0 || capacity < 0 || (capacity > 0 && !out)) return CD_ERR_INVALID_ARG;
    if ((size_t)frame + 1 >= c->last_first.size()) return CD_ERR_INVALID_ARG;   // not a frame of the last batch
    const int lo = c->last_first[(size_t)frame], hi = c->last_first[(size_t)frame + 1];
    if (lo < 0 || hi < lo || (size_t)hi > c->last_clusters.size()) return CD_ERR_INVALID_ARG;
    if (out_total) *out_total = hi - lo;
    int n = 0;
    for (int k = lo + first; k < hi && n < capacity; ++k) out[n++] = c->last_clusters[(size_t)k];
    return n;
}

// float4 points on the device -> `m` records of `stride` bytes in caller memory (k_pack_records, then ONE download).
// The staging area is the context's input buffer (its contents - the frames of a host-pointer call - are dead by now).
static int download_records(cd_context* c, const float4* d_pts, int m, size_t stride, int rgb_offset, uint32_t pad3, void* out,
                            size_t staging_skip = 0) {
    if (m <= 0) return CD_OK;
    const size_t skip = (staging_skip + 255) & ~(size_t)255;
    int st = CD_OK;
    if (skip + (size_t)m * stride > c->d_in_bytes) {
        if (skip) return fail(c, CD_ERR_CAPACITY, "record staging area too small");   // (callers that keep the input size the buffer beforehand)
        st = ensure_input(c, (size_t)m * stride);
        if (st) return st;
    }
    char* d_out = c->d_in + skip;
    LAUNCH(c, launch_pack_records(c->stream, d_pts, m, (int)(stride / 4), rgb_offset >= 0 ? rgb_offset / 4 : -1, pad3, d_out));
    HIPCHK(c, copy_sync(c, out, d_out, (size_t)m * stride, hipMemcpyDeviceToHost));
    return CD_OK;
}

int cd_get_frame_cloud(cd_context* c, int frame, int which, void* out_records, size_t stride, int rgb_offset, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    if (!out_n || capacity < 0 || (capacity > 0 && !out_records) || stride < 12 || (stride & 3) || (which != CD_CLOUD_VOXELS && which != CD_CLOUD_OBJECTS) ||
        (rgb_offset >= 0 && (rgb_offset < 12 || (rgb_offset & 3) || (size_t)rgb_offset + 4 > stride)))
        return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    *out_n = 0;
    if (!c->last_clouds || frame < 0 || (size_t)frame >= c->last_nv.size()) return fail(c, CD_ERR_INVALID_ARG, "not a frame of the last fused call of this context");
    const int m = which == CD_CLOUD_VOXELS ? c->last_nv[(size_t)frame] : c->last_no[(size_t)frame];
    if (m > capacity) { *out_n = m; return fail(c, CD_ERR_CAPACITY, "output capacity too small"); }
    const float4* src = (which == CD_CLOUD_VOXELS ? c->d_vox : c->d_obj) + (size_t)frame * c->N;
    const int st = download_records(c, src, m, stride, rgb_offset, 0u, out_records);
    if (st) return st;
    *out_n = m;
    return CD_OK;
}

int cd_get_cluster_points(cd_context* c, int frame, int k, int aligned, void* out_points, size_t stride, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    if (!out_n || capacity < 0 || (capacity > 0 && !out_points) || stride < 12 || (stride & 3)) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    *out_n = 0;
    if (!c->last_clouds || frame < 0 || (size_t)frame + 1 >= c->last_first.size()) return fail(c, CD_ERR_INVALID_ARG, "not a frame of the last fused call of this context");
    const int lo = c->last_first[(size_t)frame], hi = c->last_first[(size_t)frame + 1];
    if (k < 0 || lo + k >= hi || (size_t)(lo + k) >= c->last_clusters.size()) return fail(c, CD_ERR_INVALID_ARG, "no such cluster");
    const cd_cluster_result& r = c->last_clusters[(size_t)(lo + k)];
    const int m = r.size;
    if (m > capacity) { *out_n = m; return fail(c, CD_ERR_CAPACITY, "output capacity too small"); }
    const long long o0 = c->last_orig_off[(size_t)(lo + k)], o1 = c->last_al_off[(size_t)(lo + k)];
    const uint32_t one = 0x3f800000u;   // pcl::PointXYZ::data[3]
    int st;
    if (!aligned || o1 >= 0) {
        st = download_records(c, (aligned ? c->d_src : c->d_src0) + (aligned ? o1 : o0), m, stride, -1, one, out_points);
        if (st) return st;
    } else {
        // the pass that produced the best result has been overwritten by a later template pass: final_transformation * cluster
        st = download_records(c, c->d_src0 + o0, m, stride, -1, one, out_points);
        if (st) return st;
        for (int i = 0; i < m; ++i) {
            float v[3];
            char* rec = (char*)out_points + (size_t)i * stride;
            std::memcpy(v, rec, 12);
            const float* T = r.T;
            const float o[3] = {((T[0] * v[0] + T[1] * v[1]) + T[2] * v[2]) + T[3], ((T[4] * v[0] + T[5] * v[1]) + T[6] * v[2]) + T[7],
                                ((T[8] * v[0] + T[9] * v[1]) + T[10] * v[2]) + T[11]};
            std::memcpy(rec, o, 12);
        }
    }
    *out_n = m;
    return CD_OK;
}

static int cd_ground_plane_impl(cd_context* c, const void* points, size_t stride, int n, const cd_params* p, float coeff[4], void* out_records,
                    int capacity, int* out_n, int* out_n_inliers) {
    if (!c) return CD_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    invalidate_last(c);
    int st = check_params(c, p);
    if (st) return st;
    if ((!points && n > 0) || !coeff || !out_n || n < 0 || capacity < 0 || (capacity > 0 && !out_records) || stride < 12 || (stride & 3))
        return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    *out_n = 0;
    if (out_n_inliers) *out_n_inliers = 0;
    if (n > c->N) return fail(c, CD_ERR_CAPACITY, "more points than the context capacity");
    if (n == 0) return CD_ERR_NO_MODEL;
    // one buffer: the input blob, then (256-byte aligned) room for the records that go back
    const size_t in_bytes = (size_t)n * stride;
    st = ensure_input(c, ((in_bytes + 255) & ~(size_t)255) + in_bytes);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->d_in, points, in_bytes, hipMemcpyHostToDevice, c->stream));   // the ONE upload
    st = stage_crop_voxel(c, c->d_in, stride, n, 1, p, nullptr);
    if (st) return st;
    st = sync_fs(c, 1);
    if (st) return st;
    if (c->h_fs[0].status != CD_OK) return fail(c, c->h_fs[0].status, "voxel grid: leaf size too small for the input extent");
    std::vector<int> iters;
    st = stage_plane(c, 1, p, iters, nullptr);
    if (st) return st;
    st = stage_extract(c, 1, p);
    if (st) return st;
    st = sync_fs(c, 1);
    if (st) return st;
    const int no = c->h_fs[0].n_o;
    if (out_n_inliers) *out_n_inliers = c->h_fs[0].n_plane;
    if (no > capacity) { *out_n = no; return fail(c, CD_ERR_CAPACITY, "output capacity too small"); }
    const int rgb_off = (p->rgb_offset >= 12 && !(p->rgb_offset & 3) && (size_t)p->rgb_offset + 4 <= stride) ? p->rgb_offset : -1;
    st = download_records(c, c->d_obj, no, stride, rgb_off, 0u, out_records, in_bytes);               // the ONE download
    if (st) return st;
    *out_n = no;
    if (!c->h_have[0]) return CD_ERR_NO_MODEL;
    coeff[0] = c->h_model[0].x; coeff[1] = c->h_model[0].y; coeff[2] = c->h_model[0].z; coeff[3] = c->h_model[0].w;
    return CD_OK;
}

int cd_set_frame_guesses(cd_context* c, const float* guesses, int n_frames) {
    if (!c) return CD_ERR_INVALID_ARG;
    if (n_frames < 0 || (n_frames > 0 && !guesses)) return fail(c, CD_ERR_INVALID_ARG, "bad arguments");
    for (size_t i = 0; i < 16 * (size_t)n_frames; ++i)
        if (!std::isfinite(guesses[i])) return fail(c, CD_ERR_INVALID_ARG, "a guess holds a non-finite value");
    c->frame_guess.assign(guesses, guesses + 16 * (size_t)n_frames);
    return CD_OK;
}

// the compute entry points proper: each call runs under the per-device scan lock (with_scan_retry)
int cd_crop_voxel(cd_context* c, const void* points, size_t stride, int n, const cd_params* p, float* out_xyz, uint32_t* out_rgb, int capacity, int* out_n_cropped, int* out_n_voxels) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_crop_voxel_impl(c, points, stride, n, p, out_xyz, out_rgb, capacity, out_n_cropped, out_n_voxels); });
}
int cd_segment_plane(cd_context* c, const void* xyz, size_t stride, int n, const cd_params* p, float coeff[4], int32_t* inliers, int capacity, int* out_n_inliers, int* out_iterations) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_segment_plane_impl(c, xyz, stride, n, p, coeff, inliers, capacity, out_n_inliers, out_iterations); });
}
int cd_surface_frame(cd_context* c, const void* xyz, size_t stride, int n, const float table_normal[3], int invert, const cd_params* p, cd_surface_frame_result* out) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_surface_frame_impl(c, xyz, stride, n, table_normal, invert, p, out); });
}
int cd_bbox_filter(cd_context* c, const void* xyz, size_t stride, int n, const double P[12], const int32_t rect[4], int32_t* out_indices, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_bbox_filter_impl(c, xyz, stride, n, P, rect, out_indices, capacity, out_n); });
}
int cd_extract(cd_context* c, const void* points, size_t stride, int n, const int32_t* indices, int n_indices, int negative, void* out_points, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_extract_impl(c, points, stride, n, indices, n_indices, negative, out_points, capacity, out_n); });
}
int cd_passthrough(cd_context* c, const void* points, size_t stride, int n, int field, double limit_min, double limit_max, int negative, void* out_points, int capacity, int* out_n) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&] { return cd_passthrough_impl(c, points, stride, n, field, limit_min, limit_max, negative, out_points, capacity, out_n); });
}
int cd_cluster(cd_context* c, const void* xyz, size_t stride, int n, const cd_params* p, int32_t* labels, int32_t* sizes, int sizes_capacity, int* out_k) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_cluster_impl(c, xyz, stride, n, p, labels, sizes, sizes_capacity, out_k); });
}
int cd_icp(cd_context* c, int slot, const void* src_xyz, size_t stride, int n, const cd_params* p, cd_cluster_result* out, float* aligned) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_icp_impl(c, slot, src_xyz, stride, n, p, out, aligned); });
}
int cd_process_batch_device(cd_context* c, const void* d_frames, size_t stride, int points_per_frame, int n_frames, const cd_params* p, cd_frame_result* results, int32_t* plane_inliers, int32_t* labels) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_process_batch_device_impl(c, d_frames, stride, points_per_frame, n_frames, p, results, plane_inliers, labels); });
}
int cd_process_batch(cd_context* c, const void* frames, size_t stride, int points_per_frame, int n_frames, const cd_params* p, cd_frame_result* results, int32_t* plane_inliers, int32_t* labels) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_process_batch_impl(c, frames, stride, points_per_frame, n_frames, p, results, plane_inliers, labels); });
}
int cd_ground_plane(cd_context* c, const void* points, size_t stride, int n, const cd_params* p, float coeff[4], void* out_records, int capacity, int* out_n, int* out_n_inliers) {
    if (!c) return CD_ERR_INVALID_ARG;
    return with_scan_retry(c, [&]() { return cd_ground_plane_impl(c, points, stride, n, p, coeff, out_records, capacity, out_n, out_n_inliers); });
}

int cd_get_timing(const cd_context* c, cd_timing* out) {
    if (!c || !out) return CD_ERR_INVALID_ARG;
    *out = c->timing;
    return CD_OK;
}

void cd_pose_to_position_quaternion(const double H[16], double pos[3], double q[4]) {
    pos[0] = H[3]; pos[1] = H[7]; pos[2] = H[11];
    const double m[3][3] = {{H[0], H[1], H[2]}, {H[4], H[5], H[6]}, {H[8], H[9], H[10]}};
    const double trace = m[0][0] + m[1][1] + m[2][2];
    double t[4];
    if (trace > 0.0) {
        double s = std::sqrt(trace + 1.0);
        t[3] = s * 0.5;
        s = 0.5 / s;
        t[0] = (m[2][1] - m[1][2]) * s;
        t[1] = (m[0][2] - m[2][0]) * s;
        t[2] = (m[1][0] - m[0][1]) * s;
    } else {
        const int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        t[i] = s * 0.5;
        s = 0.5 / s;
        t[3] = (m[k][j] - m[j][k]) * s;
        t[j] = (m[j][i] + m[i][j]) * s;
        t[k] = (m[k][i] + m[i][k]) * s;
    }
    q[0] = t[0]; q[1] = t[1]; q[2] = t[2]; q[3] = t[3];
}

void cd_bbox_corners(const double H[16], double l, double w, double h, float out[24]) {
    float Hf[16];
    for (int i = 0; i < 16; ++i) Hf[i] = (float)H[i];
    const double sx[8] = {-1, -1, -1, -1, 1, 1, 1, 1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1}, sz[8] = {-1, 1, -1, 1, -1, 1, -1, 1};
    for (int k = 0; k < 8; ++k) {
        const float x = (float)(sx[k] * l / 2), y = (float)(sy[k] * w / 2), z = (float)(sz[k] * h / 2);
        out[3 * k] = ((Hf[0] * x + Hf[1] * y) + Hf[2] * z) + Hf[3];
        out[3 * k + 1] = ((Hf[4] * x + Hf[5] * y) + Hf[6] * z) + Hf[7];
        out[3 * k + 2] = ((Hf[8] * x + Hf[9] * y) + Hf[10] * z) + Hf[11];
    }
}

}  // extern "C"
