// cuboid_multi_gpu.cpp - native frame-per-GPU batch driver.
//
// One process, one host thread + one cd_context per GPU, one RCCL communicator per GPU from ncclCommInitAll (no launcher,
// no MPI), and ONE collective per batch: ncclAllGather of the fixed-size per-frame pose records (cd_frame_result).
// Frames are independent in the reference - ground_plane_segmentation.cpp:146,153 is a queue-1 subscriber on one spinner,
// RANSAC is re-seeded per frame - so a batch of F frames is cut into contiguous slices (rank g owns
// [g F/G + min(g, F%G), ...), sizes differ by at most one: the same rule as perception_amd/batch.py shard_range) and there is
// no data-path collective; the gather moves ~1.8 KB per frame and is latency-bound.
//
// usage: cuboid_multi_gpu --frames frames.bin --points N --template t.pcd --out records.bin
//                         [--gpus G] [--devices 0,1,...] [--gather rccl|host] [--steps K] [--warmup W] [--voxel_size v] ...
//   frames.bin : F x N little-endian records x,y,z,rgb (float32 x 4), as perception_amd.synth writes them
//   --devices  : device of every rank (default 0..G-1).  RCCL refuses two ranks on one device, so a list with repeats
//                (the one-GPU rehearsal of the threading and slicing: --devices 0,0,0) needs --gather host, where rank 0
//                assembles the slices through host memory instead
//   records.bin: the gathered F records as rank 0 holds them; every rank's copy is compared with it (exit code 6 if one differs)
// Prints one JSON line with frames/s over the timed steps (barrier on both sides, maximum over the ranks).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pcl_compat.hpp"

namespace {

struct Barrier {   // (std::barrier is C++20; the rest of the host code is C++17)
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0, phase = 0;
    bool broken = false;    // a rank failed: nobody waits any more
    explicit Barrier(int n_) : n(n_) {}
    bool wait() {           // false: the barrier was broken (before or while waiting)
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const int ph = phase;
        if (++waiting == n) { waiting = 0; ++phase; cv.notify_all(); }
        else cv.wait(lk, [&] { return phase != ph || broken; });
        return !broken;
    }
    void abort() { { std::lock_guard<std::mutex> lk(mu); broken = true; } cv.notify_all(); }
};

void shard_range(int n_frames, int rank, int world, int* lo, int* hi) {
    const int base = n_frames / world, rem = n_frames % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

struct Shared {
    int G = 1, F = 0, N = 0, steps = 1, warmup = 0;
    bool use_rccl = true;
    std::vector<int> devices;
    const char* frames = nullptr;          // F x N x 16 bytes, host
    std::vector<float> tpl;                // M x 3
    cd_params prm;
    std::vector<ncclComm_t> comms;
    std::vector<std::vector<cd_frame_result>> gathered;   // per rank: F records
    std::vector<std::vector<cd_frame_result>> local;      // per rank: its slice (host gather)
    std::vector<double> step_s;
    std::vector<double> hbm_used_gb;      // per rank: device memory in use on its device after the last step (hipMemGetInfo)
    std::atomic<int> failed{0};
    std::string err[64];
};

#define RANK_CHECK(cond, msg)                                                                       \
    do {                                                                                            \
        if (!(cond)) { sh->err[g] = std::string(msg); sh->failed.store(1); bar->abort(); goto done; } \
    } while (0)

void rank_main(Shared* sh, Barrier* bar, int g) {
    const int per_max = (sh->F + sh->G - 1) / sh->G;
    int lo = 0, hi = 0;
    shard_range(sh->F, g, sh->G, &lo, &hi);
    const int nloc = hi - lo;
    cd_context* ctx = nullptr;
    char* d_frames = nullptr;
    cd_frame_result *d_send = nullptr, *d_recv = nullptr;
    hipStream_t stream = nullptr;
    std::vector<cd_frame_result> res((size_t)per_max), all((size_t)per_max * sh->G);
    const size_t rec = sizeof(cd_frame_result);
    bool ok_dev = hipSetDevice(sh->devices[g]) == hipSuccess;
    {
        RANK_CHECK(ok_dev, "hipSetDevice failed");
        RANK_CHECK(cd_create(sh->devices[g], sh->N, per_max > 0 ? per_max : 1, &ctx) == CD_OK, "cd_create failed");
        RANK_CHECK(cd_set_template(ctx, 0, sh->tpl.data(), 12, (int)(sh->tpl.size() / 3)) == CD_OK, cd_last_error(ctx));
        RANK_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess, "hipStreamCreate failed");
        RANK_CHECK(hipMalloc((void**)&d_frames, (size_t)(nloc > 0 ? nloc : 1) * sh->N * 16) == hipSuccess, "hipMalloc(frames) failed");
        RANK_CHECK(hipMalloc((void**)&d_send, rec * per_max) == hipSuccess && hipMalloc((void**)&d_recv, rec * per_max * sh->G) == hipSuccess,
                   "hipMalloc(records) failed");
        if (nloc > 0)   // the rank's slice is resident in HBM before the timed region starts
            RANK_CHECK(hipMemcpy(d_frames, sh->frames + (size_t)lo * sh->N * 16, (size_t)nloc * sh->N * 16, hipMemcpyHostToDevice) == hipSuccess,
                       "hipMemcpy(frames) failed");
    }
    for (int s = 0; s < sh->warmup + sh->steps; ++s) {
        if (sh->failed.load()) goto done;          // another rank failed: leave at the step boundary
        if (s == sh->warmup) { (void)hipDeviceSynchronize(); if (!bar->wait()) goto done; }
        const auto t0 = std::chrono::steady_clock::now();
        std::memset(res.data(), 0, rec * res.size());
        if (nloc > 0)
            RANK_CHECK(cd_process_batch_device(ctx, d_frames, 16, sh->N, nloc, &sh->prm, res.data(), nullptr, nullptr) == CD_OK, cd_last_error(ctx));
        if (sh->use_rccl) {   // ONE collective per batch: every rank contributes per_max records (its slice, zero padded)
            RANK_CHECK(hipMemcpyAsync(d_send, res.data(), rec * per_max, hipMemcpyHostToDevice, stream) == hipSuccess, "H2D(records) failed");
            RANK_CHECK(ncclAllGather(d_send, d_recv, rec * per_max, ncclUint8, sh->comms[g], stream) == ncclSuccess, "ncclAllGather failed");
            RANK_CHECK(hipMemcpyAsync(all.data(), d_recv, rec * per_max * sh->G, hipMemcpyDeviceToHost, stream) == hipSuccess, "D2H(records) failed");
            RANK_CHECK(hipStreamSynchronize(stream) == hipSuccess, "stream sync failed");
        } else {              // one-GPU rehearsal: the slices meet in host memory
            sh->local[g] = res;
            if (!bar->wait()) goto done;
            for (int r = 0; r < sh->G; ++r) std::memcpy(all.data() + (size_t)r * per_max, sh->local[r].data(), rec * per_max);
            if (!bar->wait()) goto done;
        }
        if (s >= sh->warmup) {
            (void)hipDeviceSynchronize();
            sh->step_s[g] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    }
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) sh->hbm_used_gb[g] = (double)(tot - fr) / 1e9;
    }
    if (!bar->wait()) goto done;
    sh->gathered[g].resize((size_t)sh->F);
    for (int r = 0; r < sh->G; ++r) {   // drop the padding: frame order
        int rlo, rhi;
        shard_range(sh->F, r, sh->G, &rlo, &rhi);
        if (rhi > rlo) std::memcpy(sh->gathered[g].data() + rlo, all.data() + (size_t)r * per_max, rec * (size_t)(rhi - rlo));
    }
done:
    // A failed rank (or one that saw another fail) must not leave the others inside a collective: the barrier is broken (nobody
    // waits at it any more), every rank leaves at its next step boundary, and an RCCL communicator is aborted so that a
    // collective already entered by the others returns.  main() reports and exits with 5.
    if (sh->failed.load()) {
        if (!sh->err[g].empty()) std::fprintf(stderr, "rank %d: %s\n", g, sh->err[g].c_str());
        if (sh->use_rccl && sh->comms[g]) { (void)ncclCommAbort(sh->comms[g]); sh->comms[g] = nullptr; }
    }
    if (d_frames) (void)hipFree(d_frames);
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (stream) (void)hipStreamDestroy(stream);
    if (ctx) cd_destroy(ctx);
}

}  // namespace

int main(int argc, char** argv) {
    Shared sh;
    cd_default_params(&sh.prm);
    sh.prm.rgb_offset = 12;
    std::string frames_path, tpl_path, out_path, devices, gather = "rccl";
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--frames") frames_path = v;
        else if (k == "--points") sh.N = std::atoi(v.c_str());
        else if (k == "--template") tpl_path = v;
        else if (k == "--out") out_path = v;
        else if (k == "--gpus") sh.G = std::atoi(v.c_str());
        else if (k == "--devices") devices = v;
        else if (k == "--gather") gather = v;
        else if (k == "--steps") sh.steps = std::atoi(v.c_str());
        else if (k == "--warmup") sh.warmup = std::atoi(v.c_str());
        else if (k == "--voxel_size") sh.prm.leaf_size = (float)std::atof(v.c_str());
        else if (k == "--distance_threshold") sh.prm.plane_distance_threshold = std::atof(v.c_str());
        else if (k == "--icp_fitness_score") { sh.prm.icp_euclidean_fitness_epsilon = std::atof(v.c_str()); sh.prm.icp_accept_fitness = std::atof(v.c_str()); }
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (frames_path.empty() || tpl_path.empty() || sh.N <= 0 || sh.G <= 0 || sh.G > 64 || sh.steps <= 0) {
        std::fprintf(stderr, "usage: cuboid_multi_gpu --frames f.bin --points N --template t.pcd [--out r.bin] [--gpus G] [--devices 0,1,..] "
                             "[--gather rccl|host] [--steps K] [--warmup W]\n");
        return 2;
    }
    sh.use_rccl = gather == "rccl";
    for (int g = 0; g < sh.G; ++g) sh.devices.push_back(g);
    if (!devices.empty()) {
        sh.devices.clear();
        for (size_t p = 0; p <= devices.size();) {
            const size_t e = devices.find(',', p);
            sh.devices.push_back(std::atoi(devices.substr(p, e == std::string::npos ? e : e - p).c_str()));
            if (e == std::string::npos) break;
            p = e + 1;
        }
        if ((int)sh.devices.size() != sh.G) { std::fprintf(stderr, "--devices names %zu devices for %d ranks\n", sh.devices.size(), sh.G); return 2; }
    }
    for (int a = 0; a < sh.G && sh.use_rccl; ++a)
        for (int b = a + 1; b < sh.G; ++b)
            if (sh.devices[a] == sh.devices[b]) { std::fprintf(stderr, "two ranks on device %d: RCCL needs one device per rank (use --gather host)\n", sh.devices[a]); return 2; }
    // frames: mapped, not read (BASELINE config 4 is 2048 frames = 10 GB; every rank uploads its slice straight from the mapping)
    {
        const int fd = open(frames_path.c_str(), O_RDONLY);
        struct stat stt;
        if (fd < 0 || fstat(fd, &stt) != 0) { std::perror("frames"); return 2; }
        const long long bytes = (long long)stt.st_size;
        if (bytes <= 0 || bytes % ((long long)sh.N * 16) != 0) { std::fprintf(stderr, "%s is not a whole number of %d-point frames\n", frames_path.c_str(), sh.N); return 2; }
        void* m = mmap(nullptr, (size_t)bytes, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { std::perror("mmap(frames)"); return 2; }
        sh.F = (int)(bytes / ((long long)sh.N * 16));
        sh.frames = (const char*)m;
    }
    // template (iterative_closest_point.cpp:159)
    {
        pclhip::PointCloud<pclhip::PointXYZ> t;
        if (pclhip::io::loadPCDFile(tpl_path, t) == -1) { std::fprintf(stderr, "Couldn't read the template PCL file\n"); return 2; }
        for (const auto& p : t.points) { sh.tpl.push_back(p.x); sh.tpl.push_back(p.y); sh.tpl.push_back(p.z); }
    }
    sh.comms.assign((size_t)sh.G, nullptr);
    if (sh.use_rccl) {
        const ncclResult_t r = ncclCommInitAll(sh.comms.data(), sh.G, sh.devices.data());
        if (r != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(r)); return 3; }
    }
    sh.gathered.resize((size_t)sh.G);
    sh.local.resize((size_t)sh.G);
    sh.step_s.assign((size_t)sh.G, 0.0);
    sh.hbm_used_gb.assign((size_t)sh.G, 0.0);
    Barrier bar(sh.G);
    std::vector<std::thread> th;
    for (int g = 0; g < sh.G; ++g) th.emplace_back(rank_main, &sh, &bar, g);
    for (auto& t : th) t.join();
    if (sh.failed.load()) { std::fprintf(stderr, "cuboid_multi_gpu: a rank failed\n"); return 5; }
    if (sh.use_rccl) for (auto c : sh.comms) if (c) ncclCommDestroy(c);
    int differ = 0;
    for (int g = 1; g < sh.G; ++g)
        if (std::memcmp(sh.gathered[g].data(), sh.gathered[0].data(), sizeof(cd_frame_result) * (size_t)sh.F) != 0) ++differ;
    if (!out_path.empty()) {
        FILE* f = std::fopen(out_path.c_str(), "wb");
        if (!f) { std::perror("out"); return 2; }
        std::fwrite(sh.gathered[0].data(), sizeof(cd_frame_result), (size_t)sh.F, f);
        std::fclose(f);
    }
    double t = 0.0, hbm = 0.0;
    for (double s : sh.step_s) t = s > t ? s : t;
    for (double v : sh.hbm_used_gb) hbm = v > hbm ? v : hbm;
    // (with --gather host the step time of a rank includes two host barriers: a rehearsal figure, not the RCCL path's)
    std::printf("{\"driver\": \"cuboid_multi_gpu\", \"n_gpus\": %d, \"gather\": \"%s\", \"frames\": %d, \"steps\": %d, \"warmup\": %d, "
                "\"ms_per_step\": %.4f, \"frames_per_s\": %.1f, \"record_bytes\": %zu, \"ranks_identical\": %s, \"hbm_used_gb_max\": %.2f}\n",
                sh.G, sh.use_rccl ? "rccl" : "host", sh.F, sh.steps, sh.warmup, 1e3 * t / sh.steps, sh.F * sh.steps / (t > 0 ? t : 1e-9),
                sizeof(cd_frame_result), differ ? "false" : "true", hbm);
    return differ ? 6 : 0;
}
