#!/usr/bin/env python3
"""bench.py - frames/s of the full cuboid_detection point-cloud chain (crop -> voxel -> RANSAC
plane -> extract -> Euclidean clusters -> per-cluster ICP) on synthetic 640x480 D435 frames.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of `--frames` frames per GPU, already
resident in HBM when the timed region starts, followed by the gather of the fixed-size pose
records of all ranks (one all_gather per batch; RCCL over xGMI for N > 1).  Weak scaling:
per-GPU work is fixed, value = all frames of all ranks / time (max over ranks).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues per priority level (default 4); streams that share a queue run
# strictly one after the other.  Five batches in flight need five queues (4 -> 8: 42 k -> 46 k frames/s, tools/ab_hwq.sh);
# config 5's six batches have two low-priority ICP streams each, and with 8 queues four of those twelve streams share
# one (rocprofv3 kernel trace: a batch's k_icp_pipe starts when another batch's k_icp_pipe_big ends) - 16 there.  Must be
# set before the HIP runtime initialises: main() does it once the arguments are known.
# But the queues of a PROCESS are a budget too (round 4): every context has one normal-priority stream and two low-priority
# ones, each class capped at GPU_MAX_HW_QUEUES, queues of closed contexts stay with the process, and from ~24 hardware queues
# on the GPU time-slices them - with seven contexts for the headline and 16 queues per class the legs that create their own
# contexts afterwards lost a quarter (config-5 leg 1.85 k instead of 2.5 k frames/s, big template 12.0 instead of 11.0 ms;
# eight contexts in the headline itself: 50 k instead of 61 k).  10 per class keeps the whole default run below the limit
# (headline 61.0 k, legs 2.51 k / 11.1 ms; with 12: 61.7 k / 2.42 k / 11.0; with 8: 59.1 k) - profiles/r04_sweep_inflight.txt.
DEFAULT_HW_QUEUES = {3: "10", 5: "16"}

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
FP32_VALU_PEAK_TFLOPS = 157.3  # vector fp32 peak
# Counter files of the current round under profiles/ (written by tools/profile_round.sh and tools/probe_icp_work.py on the
# GPU box, copied into profiles/ and committed); the previous round's are the fallback until this round's exist.
PMC_TRAFFIC_FILES = ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")   # HBM bytes per dispatch: separate --pmc FETCH_SIZE / WRITE_SIZE passes
PMC_SQ_FILES = ("r05_pmc_icp.txt", "r04_pmc_icp.txt", "r03_pmc_icp.txt", "r02_pmc_icp.txt")                   # SQ counters per kernel (separate --pmc passes)
ICP_WORK_FILES = ("r04_icp_work.json", "r03_icp_work.json")                                # executed distance tests of the dominant kernel (-DCD_STATS build)
PMC_SQ_INFLIGHT_SHAPE_FILES = ("r05_pmc_icp_inflight_shape.txt",)   # the same passes with the launch shape of the timed region forced (CUBOID_LAT_SHAPE=4,2)
VALU_CALIBRATION_FILES = ("r04_valu_calibration.json",)   # what the SQ counters read on a SATURATED vector pipe (tools/valu_calib.hip, same launch shape)
WAVES_PER_SIMD_ICP = 4          # k_icp_pipe: one 1024-thread workgroup per CU = 16 waves = 4 per SIMD
SIMDS = 1024                    # 256 CUs x 4


def _first_profile(names):
    for n in names:
        p = os.path.join(ROOT, "profiles", n)
        if os.path.exists(p):
            return n, p
    return None, None


def sq_counters(kernel, files=None):
    """SQ counters of `kernel` from the committed PMC summary (lines "<kernel> {...} dispatches n" per pass)."""
    import ast
    name, path = _first_profile(files or PMC_SQ_FILES)
    out = {}
    if not path:
        return None, out
    for ln in open(path):
        if ln.startswith(kernel + " {"):
            try:
                d = ast.literal_eval(ln[len(kernel) + 1:ln.rindex("}") + 1])
                out.update({k: float(v) for k, v in d.items()})
            except (ValueError, SyntaxError):
                pass
    return name, out


def _render(i):
    from perception_amd import synth
    return synth.frame(i)


def make_frames(start, count, config=3):
    """Synthetic frames [start, start+count), rendered on host threads (numpy releases the
    GIL in its array loops).  No fork/exec: under rocprofv3 the GPU runtime is already
    initialised when this runs, and forking such a process hangs."""
    from concurrent.futures import ThreadPoolExecutor
    from perception_amd import synth
    nthr = max(1, min(16, (os.cpu_count() or 2), count))
    npts = synth.CONFIG5_SENSOR ** 2 if config == 5 else 640 * 480
    out = np.empty((count, npts, 4), np.float32)

    def work(i):
        out[i] = synth.frame_config5(start + i) if config == 5 else _render(start + i)

    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(work, range(count)))
    return out


def _same_as_oracle(rg, ro):
    """counts, plane bits, and per cluster size / iterations / flags / T bits / fitness of a GPU record vs the oracle's"""
    if (rg.status, rg.n_cropped, rg.n_voxels, rg.n_plane, rg.n_objects, rg.n_clusters, rg.ransac_iterations) != \
       (ro.status, ro.n_cropped, ro.n_voxels, ro.n_plane, ro.n_objects, ro.n_clusters, ro.ransac_iterations):
        return False
    if bytes(rg.plane) != bytes(ro.plane):
        return False
    for k in range(min(ro.n_clusters, len(ro.clusters))):
        a, b = rg.clusters[k], ro.clusters[k]
        if (a.size, a.iterations, a.converged, a.accepted, bytes(a.T), a.fitness) != (b.size, b.iterations, b.converged, b.accepted, bytes(b.T), b.fitness):
            return False
        if float(np.linalg.norm(np.array(a.pose) - np.array(b.pose))) >= 1e-4:
            return False
    return True


def _same_as_oracle_best_of(rg, per):
    """template_slot = -1: `per` = the oracle's record of the frame against each template alone; per cluster the lowest fitness wins"""
    ro = per[0]
    if (rg.status, rg.n_cropped, rg.n_voxels, rg.n_plane, rg.n_objects, rg.n_clusters, rg.ransac_iterations) != \
       (ro.status, ro.n_cropped, ro.n_voxels, ro.n_plane, ro.n_objects, ro.n_clusters, ro.ransac_iterations) or bytes(rg.plane) != bytes(ro.plane):
        return False
    for k in range(min(ro.n_clusters, len(ro.clusters))):
        want = min(range(len(per)), key=lambda s_: (per[s_].clusters[k].fitness, s_))
        a, b = rg.clusters[k], per[want].clusters[k]
        if a.template_slot != want or (a.size, a.iterations, a.converged, a.accepted, bytes(a.T), a.fitness) != (b.size, b.iterations, b.converged, b.accepted, bytes(b.T), b.fitness):
            return False
    return True


def cpu_baseline(frames, prm, tpl, budget_s=20.0, max_frames=96, gpu_records=None):
    """The CPU oracle (kind 'port': a restatement of the PCL chain, kd-tree NN, grid clustering)
    timed on one host thread over a bounded sample of the same frames.  The oracle is the CHECKER here as well: the
    records it produces for the sampled frames are compared with the GPU records of the last timed step."""
    from oracle import oracle_py as O      # allowed: bench.py's cpu_baseline leg
    from perception_amd import capi
    O.lib()
    O.process_frame(frames[0], prm, tpl, nn_mode=1)   # warm
    gpu = capi.results_from_array(gpu_records) if gpu_records is not None else None
    n, t0, bad = 0, time.perf_counter(), []
    orec = []
    while n < min(max_frames, len(frames)) and (time.perf_counter() - t0) < budget_s:
        orec.append(O.process_frame(frames[n], prm, tpl, nn_mode=1)["result"])
        n += 1
    dt = time.perf_counter() - t0
    if gpu is not None:
        bad = [i for i in range(n) if not _same_as_oracle(gpu[i], orec[i])]
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "first %d frames of the bench batch, oracle/liboracle.so (g++ -O2), 1 thread, %.1f s" % (n, dt),
           "host_cpus": os.cpu_count()}
    if gpu is not None:
        out["oracle_check"] = {"frames": n, "ok": not bad, "mismatching_frames": bad[:8],
                               "what": "GPU records of the last timed step vs the oracle: counts, plane bits, per-cluster size/"
                                       "iterations/converged/accepted/T bits/fitness identical, pose Frobenius < 1e-4"}
    # SURVEY 8(d) also asks for the frame-parallel figure: one frame per thread (ctypes releases the GIL), on the
    # box's CPU share for one GPU
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(16, os.cpu_count() or 1))
    m = min(len(frames), max(threads, int(out["value"] * threads * 8)))   # about 8 s of work
    t1 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda i: O.process_frame(frames[i], prm, tpl, nn_mode=1)["status"], range(m)))
    dt1 = time.perf_counter() - t1
    out["frame_parallel"] = {"value": m / dt1, "unit": "frames/s", "cores": threads,
                             "sample": "first %d frames, one frame per thread, %.1f s" % (m, dt1)}
    return out


def launch_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's JSON line.
    Runs BEFORE this process imports torch or touches HIP - a process that has initialised the GPU must never be
    re-executed or forked - and the parent only waits.  Children get RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* exactly as
    torch.distributed.run would set them."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # wait for all; if one rank fails the others would sit in a collective until its timeout: end them (these exact
        # children, by handle)
        code = 0
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and code == 0:
                    code = abs(rc) or 1
                    for q in live:
                        q.terminate()
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
    return code


class stdout_to_stderr:
    """RCCL (and gloo) print a banner (host name, library path, peer count) on stdout when a communicator comes up; stdout
    is reserved for the one JSON line, so file descriptor 1 points at stderr while the process group initialises."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def dry_run(rank, local_rank, world):
    import torch
    import torch.distributed as dist
    ids = [(rank, local_rank)]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with stdout_to_stderr():
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(t, torch.tensor([rank, local_rank], dtype=torch.int64))
            ids = [tuple(int(v) for v in x) for x in t]
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": [i[0] for i in ids], "local_ranks": [i[1] for i in ids]}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node to use, one rank process per GPU (default: WORLD_SIZE or 1)")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 300, about 2.5 s of timed region)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before the clock starts (default: one per batch in flight, at least 4, so that every context has run once)")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (BASELINE config 3: 256; config 5: 64 - the config names no batch size; 8 was round 2's choice, tools/sweep_c5_frames.sh)")
    ap.add_argument("--config", type=int, default=3, choices=(3, 5),
                    help="3: the headline workload (256 D435 frames per GPU, one template); 5: the multi-template stress of "
                         "BASELINE config 5 (1 M-point frames, five cuboids, five templates, every cluster x every template)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency measurement (BASELINE config 2; runs after the timed region)")
    ap.add_argument("--no-verify", action="store_true", help="skip the self-check of the timed path's records (runs after the timed region)")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs of the default N = 1 line (config 5, the 21 400-point template, host-fed): they run after the timed region and are never part of `value`")
    ap.add_argument("--legs-config5-frames", type=int, default=64, help="frames per batch of the config-5 leg")
    ap.add_argument("--guess", choices=("none", "sne", "track"), default="none",
                    help="after the headline measurement, a separately labelled leg with per-frame initial guesses (opt-in ICP guess; "
                         "reported under guess_leg, never part of value).  sne: the inverse of the cuboid frame of "
                         "surface_normal_estimation, taken literally; track: the previous estimate of the frame's largest cluster, "
                         "perturbed by 1 cm / 2 degrees (frame-to-frame tracking)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: the ranks rendezvous over gloo, gather their rank ids and rank 0 "
                         "prints them (tests/test_bench_launcher.py)")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight per GPU: each has its own context (stream + device arena) and host thread, so the "
                         "front end of batch i+1 fills the CUs that the tail of batch i's ICP leaves idle (1 = strictly serial; "
                         "default for config 3: 7 for runs of 60 steps or more (59.7 / 60.6 / 61.4-62.2 k frames/s with 5 / 6 / 7, 50 k with "
                         "8 and more: profiles/r04_sweep_inflight.txt), 5 for shorter runs, whose clock is mostly fill and drain (the driver's "
                         "20-step shape reads the same with 5 and 7); config 5: 4 - measured, DESIGN.md section 6)")
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 256 if args.config == 3 else 64
    if args.steps is None:
        args.steps = 300 if args.config == 3 else 12
    if args.inflight is None:
        args.inflight = (7 if args.steps >= 60 else 5) if args.config == 3 else 4
    if args.warmup is None:
        args.warmup = max(4, args.inflight)
    if args.steps < 1 or args.warmup < 0 or args.frames < 1:
        raise SystemExit("bench.py: --steps/--frames must be >= 1, --warmup >= 0")

    env_world = int(os.environ["WORLD_SIZE"]) if "WORLD_SIZE" in os.environ else None
    if env_world is None and (args.gpus or 1) > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # parent: no torch, no HIP
    if env_world is not None and args.gpus is not None and args.gpus != env_world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, env_world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = env_world or 1
    F = args.frames
    if args.dry_run:
        return dry_run(rank, local_rank, world)

    os.environ.setdefault("GPU_MAX_HW_QUEUES", DEFAULT_HW_QUEUES[args.config])
    # host-side inputs first (fork pool must not follow GPU init)
    frames = make_frames(rank * F, F, args.config)
    # the extra legs of the default N = 1 line (after the timed region, never part of `value`): config 5 and the reference's
    # 21 400-point template, driver-visible, and the host-fed rate (item 5)
    legs = world == 1 and args.config == 3 and not args.no_legs
    frames_c5 = make_frames(0, args.legs_config5_frames, 5) if legs else None

    import torch
    import torch.distributed as dist
    from perception_amd import batch, capi, templates

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but this node shows %d" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("CUBOID_BENCH_FORCE_DIST") == "1"   # force: rehearse the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with stdout_to_stderr():          # until the first collective has run
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()

    tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
    prm = capi.default_params()
    prm.rgb_offset = 12
    if os.environ.get("CUBOID_BENCH_ICP_ITERS"):      # experiment only (how fast is the chain WITHOUT its ICP iterations?): the line is marked
        prm.icp_max_iterations = int(os.environ["CUBOID_BENCH_ICP_ITERS"])
    tpl_by_slot = {0: tpl}
    if args.config == 5:
        from perception_amd import synth
        tpl_by_slot = {k: templates.template_xyz32(L, W, H, d) for k, (L, W, H, d) in enumerate(synth.CONFIG5_DIMS)}
        prm.template_slot = -1
        prm.crop_x_min, prm.crop_x_max = -synth.CONFIG5_CROP_X, synth.CONFIG5_CROP_X
        prm.crop_z_max = prm.crop2_z_max = 1.2
    N = frames.shape[1]
    M = max(1, args.inflight)
    pipe = batch.BatchPipeline(N, F, tpl_by_slot, device_id=local_rank, inflight=M)
    ctx = pipe.contexts[0]
    d_frames = torch.from_numpy(frames).to(dev)           # resident in HBM before timing
    torch.cuda.synchronize()

    side = torch.cuda.Stream(device=dev)

    def run_steps(k):
        """k steps with up to M batches in flight.  A gather thread takes the finished batches in step order and
        all-gathers their records, so a gather never delays the submission of the next batch."""
        import queue
        import threading
        q, tims, last = queue.Queue(), [], [None]
        trace = [] if os.environ.get("CUBOID_BENCH_TRACE") else None   # debug: completion time of every step (stderr)
        t_begin = time.perf_counter()

        def gatherer():
            torch.cuda.set_device(local_rank)
            while True:
                fut = q.get()
                if fut is None:
                    return
                rec, t = fut.result()
                tims.append(t)
                if trace is not None:
                    trace.append(time.perf_counter())
                with torch.cuda.stream(side):   # not the NULL stream: its copies would queue behind the persistent ICP launches
                    last[0] = batch.gather_records(rec, F * world, dist if use_dist else None, dev)

        th = threading.Thread(target=gatherer)
        th.start()
        for _ in range(k):
            q.put(pipe.submit(d_frames.data_ptr(), 16, N, F, prm))   # one pass of the hot path over one batch
        q.put(None)
        th.join()
        if trace:
            print("bench.py trace: %d steps, completions at ms %s" % (k, " ".join("%.1f" % ((x - t_begin) * 1e3) for x in trace)), file=sys.stderr)
        return last[0], tims

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    fence()
    dbg_lib = capi.load_library()
    dbg_stats = None
    if hasattr(dbg_lib, "cd_debug_icp_stats"):       # only a -DCD_TIMERS build (CUBOID_HIP_LIB=.../libtimers.so) exports it
        import ctypes
        dbg_stats = (ctypes.c_ulonglong * 16)()
        dbg_lib.cd_debug_icp_stats(dbg_stats, 1)
    t0 = time.perf_counter()
    allrec, timings = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    cu_fill = None
    if dbg_stats is not None:
        dbg_lib.cd_debug_icp_stats(dbg_stats, 1)
        o = list(dbg_stats)
        if o[7]:   # persistent ICP workgroups hold a whole CU each: their busy time (100 MHz clock) over wall time x 256 CUs
            cu_fill = {"icp_workgroups": o[7], "busy_ms_mean": o[14] / o[7] / 1e5, "busy_ms_max": o[15] / 1e5,
                       "cu_time_frac": o[14] / 1e8 / (elapsed * 256.0)}
    icp_ms = icp_launches = 0.0
    stage = np.zeros(5)
    for t in timings:
        icp_ms += t.icp_kernel_ms
        icp_launches += t.icp_kernel_launches
        stage += np.array(list(t.stage_ms))
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # single-frame latency (BASELINE config 2), not part of the timed region
    t = ctx.timing()
    pairs = (t.icp_pair_tests_hi << 32) | (t.icp_pair_tests_lo & 0xffffffff)
    balg, icp_balg = t.algorithmic_bytes, t.icp_algorithmic_bytes
    one = (capi.CdFrameResult * 1)()
    lat = [float("nan")]
    for _ in range(0 if args.no_latency else 12):
        torch.cuda.synchronize()
        a = time.perf_counter()
        ctx.process_batch_device(d_frames.data_ptr(), 16, N, 1, prm, results=one)
        lat.append((time.perf_counter() - a) * 1e3)

    # Host-fed single frame (what a ROS callback gets: one H2D of the 4.9 MB blob, the chain, the results back), after the
    # timed region: cd_process_frame with a host pointer, and ground_plane_segmentation's callback as ONE call (cd_ground_plane).
    lat_host, lat_gp = [], []
    if not args.no_latency and args.config == 3:
        f0 = np.ascontiguousarray(frames[0])
        gp = capi.default_params()
        gp.rgb_offset = 12
        gp.crop2_enable = 0
        for _ in range(10):
            a = time.perf_counter()
            ctx.process_frame(f0, prm)
            lat_host.append((time.perf_counter() - a) * 1e3)
        for _ in range(10):
            a = time.perf_counter()
            ctx.ground_plane(f0, gp)
            lat_gp.append((time.perf_counter() - a) * 1e3)

    # Separately labelled leg, NOT part of `value`: the same batch with the registration started from a per-frame initial
    # guess (cd_params.icp_use_guess, opt-in; the reference's authors meant to feed surface_normal_estimation's pose to
    # ICP, icp.cpp:130-134,165-167).  --guess sne: guess of frame f = inverse of the cuboid frame cd_surface_frame finds
    # in the frame's object cloud (three axis-constrained planes, sne.cpp:167-234), identity where it finds none.
    guess_leg = None
    if args.guess != "none" and args.config == 3:
        fence()
        res0 = (capi.CdFrameResult * F)()
        ctx.process_batch_device(d_frames.data_ptr(), 16, N, F, prm, results=res0)
        clouds = [(ctx.frame_cloud(f, capi.CD_CLOUD_OBJECTS, 16, -1)[:, :3].copy().view(np.float32), np.array(res0[f].plane[:3], np.float32)) for f in range(F)]
        base_it = [res0[f].clusters[k].iterations for f in range(F) for k in range(min(res0[f].n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME))]
        base_acc = sum(res0[f].clusters[k].accepted for f in range(F) for k in range(min(res0[f].n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)))
        guesses = np.tile(np.eye(4, dtype=np.float32), (F, 1, 1))
        found = 0
        sp = capi.default_params()
        sp.plane_distance_threshold = 0.003      # the faces of a 30 mm cuboid must come apart (the launch value 0.015 is the table's)
        t_sne = time.perf_counter()
        if args.guess == "track":
            c_, s_ = np.cos, np.sin
            rx, ry, rz = 0.02, -0.015, 0.03
            Rp = (np.array([[c_(rz), -s_(rz), 0], [s_(rz), c_(rz), 0], [0, 0, 1]]) @ np.array([[c_(ry), 0, s_(ry)], [0, 1, 0], [-s_(ry), 0, c_(ry)]])
                  @ np.array([[1, 0, 0], [0, c_(rx), -s_(rx)], [0, s_(rx), c_(rx)]]))
            P = np.eye(4)
            P[:3, :3], P[:3, 3] = Rp, [0.008, -0.006, 0.005]
            for f in range(F):
                if res0[f].n_clusters > 0:
                    guesses[f] = (P @ np.array(res0[f].clusters[0].T, np.float64).reshape(4, 4)).astype(np.float32)
                    found += 1
            clouds = []
        for f, (obj, nrm) in enumerate(clouds):
            if len(obj) < 10:
                continue
            st_s, rs = ctx.surface_frame(obj, nrm, sp)
            if st_s != capi.CD_OK:
                continue
            Rt = np.array(rs.Rt, np.float64).reshape(4, 4)
            if not np.isfinite(Rt).all() or abs(np.linalg.det(Rt[:3, :3])) < 1e-6:
                continue
            guesses[f] = np.linalg.inv(Rt).astype(np.float32)
            found += 1
        t_sne = time.perf_counter() - t_sne
        gprm = capi.default_params()
        gprm.rgb_offset = 12
        gprm.icp_use_guess = capi.CD_GUESS_PER_FRAME
        for cx in pipe.contexts:
            cx.set_frame_guesses(guesses)
        prm_saved, prm = prm, gprm
        gsteps = max(10, args.steps // 3)
        run_steps(2)
        fence()
        tg = time.perf_counter()
        grec, _ = run_steps(gsteps)
        fence()
        tg = time.perf_counter() - tg
        prm = prm_saved
        for cx in pipe.contexts:
            cx.set_frame_guesses(None)
        gr = capi.results_from_array(grec)[rank * F:(rank + 1) * F]
        g_it = [gr[f].clusters[k].iterations for f in range(F) for k in range(min(gr[f].n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME))]
        g_acc = sum(gr[f].clusters[k].accepted for f in range(F) for k in range(min(gr[f].n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)))
        guess_leg = {"guess": args.guess, "frames_per_s": F * gsteps / tg, "ms_per_step": tg / gsteps * 1e3, "steps": gsteps,
                     "frames_with_a_guess": found, "frames": F, "guess_ms_per_frame_host_loop": t_sne / F * 1e3,
                     "mean_iterations": float(np.mean(g_it)) if g_it else 0.0, "mean_iterations_identity_guess": float(np.mean(base_it)) if base_it else 0.0,
                     "accepted": int(g_acc), "accepted_identity_guess": int(base_acc), "clusters": len(g_it),
                     "note": ("opt-in path, outside the headline value: per-frame guess = inverse of cd_surface_frame's cuboid frame, taken literally "
                              "(surface_normal_estimation.cpp:212-225 -> iterative_closest_point.cpp:130-134,165-167; plane threshold 3 mm); the "
                              "frame it finds has its origin on a face and normals of arbitrary sign, which is not the template's frame: the "
                              "registrations stop early in a wrong minimum (see accepted) - the reference leaves this path disabled"
                              if args.guess == "sne" else
                              "opt-in path, outside the headline value: per-frame guess = the final transformation of the frame's largest cluster "
                              "from the identity-guess run, moved by 1 cm / 2 degrees (what a tracker hands over from the previous frame); a "
                              "frame's other clusters start from the same matrix") +
                             "; the guesses are computed before this leg's timed region"}

    # Self-check of the timed path, outside the timed region: the records of the LAST timed step (the in-flight launch shape of
    # the ICP kernel with refilled slots, other batches in flight, gathered over all ranks) must be byte-identical to a strictly serial pass of this
    # rank's batch on an otherwise idle GPU.
    verified = None
    serial_rec = None
    serial_t = None
    serial_ts = []
    if not args.no_verify:
        fence()
        res = (capi.CdFrameResult * F)()
        ctx.process_batch_device(d_frames.data_ptr(), 16, N, F, prm, results=res)
        serial_rec = capi.results_to_array(res).copy()
        serial_t = ctx.timing()
        serial_ts = [serial_t] if serial_t.icp_kernel_launches == 1 and serial_t.icp_kernel_ms > 0 else []
        for _ in range(4 if M > 1 else 0):   # (a few more for the timing: the first pass after the pipelined region still finds the
            # other contexts' data in L2, and a lone launch varies by ~3 % from one to the next; the MEDIAN of the five is reported)
            ctx.process_batch_device(d_frames.data_ptr(), 16, N, F, prm, results=res)
            t2 = ctx.timing()
            if t2.icp_kernel_launches == 1 and t2.icp_kernel_ms > 0:
                serial_ts.append(t2)
        ok = bool(np.array_equal(serial_rec, allrec[rank * F:(rank + 1) * F]))
        if use_dist:
            tv = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(tv, op=dist.ReduceOp.MIN)
            ok = bool(tv.item())
        verified = ok

    # ---- extra legs of the default N = 1 line: after the timed region, separately labelled, never part of `value` ----------
    legs_out = {}
    if legs:
        def pump(pl, ptr, n_pts, n_fr, pr, k, host=False):
            """k batches through pipeline `pl`, all waited for; returns (records of the last one, seconds)"""
            a = time.perf_counter()
            futs = [pl.submit(ptr, 16, n_pts, n_fr, pr, host=host) for _ in range(k)]
            rec_last = [f.result()[0] for f in futs][-1]
            torch.cuda.synchronize()
            return rec_last, time.perf_counter() - a

        fence()
        # (1) host-fed: the frames come from HOST memory through cd_process_batch (the upload is part of every call, on the
        # context's own stream, so it overlaps the other contexts' kernels) - what a ROS callback has (gps.cpp:43-49).  Pinned
        # memory for the PCIe rate; pageable (what roscpp hands over) beside it.
        try:
            mb = frames.nbytes / 1e6
            pinned = torch.from_numpy(frames).pin_memory()
            pump(pipe, pinned.data_ptr(), N, F, prm, M, host=True)                       # staging buffers, first touch
            hrec, hs = pump(pipe, pinned.data_ptr(), N, F, prm, 4 * M, host=True)
            pump(pipe, frames.ctypes.data, N, F, prm, 2, host=True)
            prec, ps = pump(pipe, frames.ctypes.data, N, F, prm, M, host=True)
            same = bool(np.array_equal(hrec, allrec[rank * F:(rank + 1) * F]) and np.array_equal(prec, hrec))
            legs_out["host_fed"] = {
                "frames_per_s": F * 4 * M / hs, "GBps": mb * 4 * M / hs / 1e3, "frac_of_63_GBps_pcie": mb * 4 * M / hs / 1e3 / 63.0,
                "batches": 4 * M, "batches_in_flight": M, "pageable_frames_per_s": F * M / ps, "pageable_GBps": mb * M / ps / 1e3,
                "records_identical_to_the_timed_path": same,
                "note": "cd_process_batch from host memory: %.1f MB per batch over PCIe Gen5 x16 (63 GB/s spec), pinned source, the H2D copy of a "
                        "batch on its context's stream beside the other contexts' kernels; pageable = the same from ordinary "
                        "memory (the runtime stages it through its own pinned buffers); outside `value`" % mb}
            del pinned
        except Exception as e:   # a leg must not take the headline down
            legs_out["host_fed"] = {"error": repr(e)}
        # (2) the reference's own 21 400-point six-face template (template_cuboid_L200_W100_H75.pcd: not LDS-resident ->
        # k_icp_pipe_big), same 256 frames, strictly serial
        try:
            from perception_amd import pcd
            big = pcd.read_xyz(os.path.join(ROOT, "tests", "golden", "template_cuboid_L200_W100_H75.pcd")).astype(np.float32)
            cb = capi.Context(max_points=N, max_frames=F, device_id=local_rank)
            cb.set_template(0, big)
            resb = (capi.CdFrameResult * F)()
            tb = []
            for _ in range(4):
                torch.cuda.synchronize()
                a = time.perf_counter()
                cb.process_batch_device(d_frames.data_ptr(), 16, N, F, prm, results=resb)
                tb.append((time.perf_counter() - a) * 1e3)
            tmb = cb.timing()
            itb = [resb[f].clusters[k].iterations for f in range(F) for k in range(min(resb[f].n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME))]
            legs_out["big_template_ms"] = {"batch_ms": float(min(tb[1:])), "icp_kernel_ms": float(tmb.icp_kernel_ms), "frames": F,
                                           "template_points": int(len(big)), "clusters": len(itb), "mean_iterations": float(np.mean(itb)) if itb else 0.0,
                                           "icp_regime": {"slots": tmb.icp_regime >> 16, "workgroups": tmb.icp_regime & 0xffff},
                                           "note": "one batch of the same frames against the reference's 21 400-point template, one context, idle GPU"}
            cb.close()
        except Exception as e:
            legs_out["big_template_ms"] = {"error": repr(e)}
        # (2b) the generic (pruned) search on the headline's own batch and template: the same pipeline with CUBOID_ICP_LATTICE=0 -
        # what a template that is NOT a lattice of this size costs, and the A/B of the closed form
        pipe.close()
        pipe = None
        try:
            os.environ["CUBOID_ICP_LATTICE"] = "0"
            pg = batch.BatchPipeline(N, F, tpl_by_slot, device_id=local_rank, inflight=M)
            os.environ.pop("CUBOID_ICP_LATTICE")
            pump(pg, d_frames.data_ptr(), N, F, prm, 2 * M)
            grec, gs = pump(pg, d_frames.data_ptr(), N, F, prm, 60)
            legs_out["generic_search"] = {"frames_per_s": F * 60 / gs, "ms_per_step": gs / 60 * 1e3, "steps": 60, "batches_in_flight": M,
                                          "records_identical_to_the_timed_path": bool(np.array_equal(grec, allrec[rank * F:(rank + 1) * F])),
                                          "icp_search": int(pg.contexts[0].timing().icp_search),
                                          "note": "the headline's batch with CUBOID_ICP_LATTICE=0: k_icp_pipe's pruned search (grid walk + k-d patches) over the "
                                                  "same 7250-point template; outside `value`"}
            pg.close()
        except Exception as e:
            os.environ.pop("CUBOID_ICP_LATTICE", None)
            legs_out["generic_search"] = {"error": repr(e)}
        # (2c) the object_detection flavour (object_detection.launch:30-38: voxel_size 0.001, distance_threshold 0.01; opd.cpp:331-336
        # second z crop; opd.cpp:376-413 every cluster against every object template): the reference's four scanned templates
        # (tests/golden/*_ascii_tf.pcd - arbitrary clouds, so the pruned searches run), 64 frames per batch
        try:
            from perception_amd import pcd
            names = ("eraser", "clamp", "screwdriver", "marker")
            tplo = {k: pcd.read_xyz(os.path.join(ROOT, "tests", "golden", nm + "_ascii_tf.pcd")).astype(np.float32) for k, nm in enumerate(names)}
            prmo = capi.default_params()
            prmo.rgb_offset = 12
            prmo.leaf_size = 0.001
            prmo.plane_distance_threshold = 0.01
            prmo.template_slot = -1
            Fo, Mo, Ko = min(64, F), 4, 24      # (the first frames of the batch that is resident: never more than it holds)
            po = batch.BatchPipeline(N, Fo, tplo, device_id=local_rank, inflight=Mo)
            pump(po, d_frames.data_ptr(), N, Fo, prmo, 2 * Mo)
            orec, osec = pump(po, d_frames.data_ptr(), N, Fo, prmo, Ko)
            ro_ = (capi.CdFrameResult * Fo)()
            po.contexts[0].process_batch_device(d_frames.data_ptr(), 16, N, Fo, prmo, results=ro_)      # strictly serial pass
            to_ = po.contexts[0].timing()
            ok_serial = bool(np.array_equal(capi.results_to_array(ro_)[:Fo], orec))
            ok_oracle = None
            if not args.no_cpu_baseline:
                from oracle import oracle_py as O      # allowed: the checker of a bench leg
                from concurrent.futures import ThreadPoolExecutor
                O.lib()
                sample = list(range(0, Fo, max(1, Fo // 16)))
                tl = [tplo[k] for k in range(len(names))]
                import copy

                def oracle_frame(f):   # every template on its own, as the oracle runs them; the lowest fitness wins (ties: lowest slot)
                    per = []
                    for slot_, t_ in enumerate(tl):
                        pr_ = copy.copy(prmo)
                        pr_.template_slot = slot_
                        per.append(O.process_frame(frames[f], pr_, t_)["result"])
                    return per
                with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
                    want = list(ex.map(oracle_frame, sample))
                ok_oracle = all(_same_as_oracle_best_of(ro_[f], w) for f, w in zip(sample, want))
            legs_out["object_launch"] = {
                "frames_per_s": Fo * Ko / osec, "ms_per_step": osec / Ko * 1e3, "frames_per_batch": Fo, "steps": Ko, "batches_in_flight": Mo,
                "n_voxels_mean": float(np.mean([ro_[f].n_voxels for f in range(Fo)])), "clusters": int(sum(ro_[f].n_clusters for f in range(Fo))),
                "templates": {nm: int(len(tplo[k])) for k, nm in enumerate(names)}, "icp_search": int(to_.icp_search),
                "serial_stage_ms": {"crop_voxel": to_.stage_ms[0], "plane": to_.stage_ms[1], "extract_cluster": to_.stage_ms[2], "icp": to_.stage_ms[3], "total": to_.stage_ms[4]},
                "verified": ok_serial, "oracle_sample_ok": ok_oracle, "oracle_sample_frames": len(range(0, Fo, max(1, Fo // 16))) if ok_oracle is not None else 0,
                "note": "object_detection.launch parameters (leaf 0.001, plane threshold 0.01, second z crop) on the bench frames, every cluster "
                        "against the reference's four scanned object templates, lowest fitness wins; verified = the last pipelined batch equals a "
                        "strictly serial pass; oracle_sample_ok = a sample of 16 frames equals the CPU oracle (counts, plane bits, T bits, fitness); "
                        "outside `value`"}
            po.close()
        except Exception as e:
            legs_out["object_launch"] = {"error": repr(e)}
        # (3) BASELINE config 5: 1 M-point frames, five cuboids, five templates, every cluster against every template.  The headline's
        # pipeline is closed first (its arenas are not needed any more; what slowed this leg down was the process's hardware-queue
        # budget, see DEFAULT_HW_QUEUES).
        if pipe is not None:
            pipe.close()
        pipe = None
        try:
            from perception_amd import synth
            F5, M5, K5 = len(frames_c5), 4, 64   # (64 steps of ~25 ms: with 16 the fill and drain of the four-deep pipeline were a quarter of the clock)
            tpl5 = {k: templates.template_xyz32(L, W, H, dd) for k, (L, W, H, dd) in enumerate(synth.CONFIG5_DIMS)}
            prm5 = capi.default_params()
            prm5.rgb_offset = 12
            prm5.template_slot = -1
            prm5.crop_x_min, prm5.crop_x_max = -synth.CONFIG5_CROP_X, synth.CONFIG5_CROP_X
            prm5.crop_z_max = prm5.crop2_z_max = 1.2
            N5 = frames_c5.shape[1]
            pipe5 = batch.BatchPipeline(N5, F5, tpl5, device_id=local_rank, inflight=M5)
            d5 = torch.from_numpy(frames_c5).to(dev)
            torch.cuda.synchronize()
            pump(pipe5, d5.data_ptr(), N5, F5, prm5, 2 * M5)      # every context has had its first batches (lazy allocations) before the clock starts
            rec5, s5 = pump(pipe5, d5.data_ptr(), N5, F5, prm5, K5)
            r5 = (capi.CdFrameResult * F5)()
            pipe5.contexts[0].process_batch_device(d5.data_ptr(), 16, N5, F5, prm5, results=r5)      # strictly serial pass
            ok5 = bool(np.array_equal(capi.results_to_array(r5)[:F5], rec5))
            legs_out["config5"] = {"frames_per_s": F5 * K5 / s5, "ms_per_step": s5 / K5 * 1e3, "frames_per_batch": F5, "steps": K5,
                                   "batches_in_flight": M5, "verified": ok5, "points_per_frame": int(N5),
                                   "templates": [len(t) for t in tpl5.values()],
                                   "note": "BASELINE config 5 on one GPU after the headline measurement: every cluster against every template, "
                                           "lowest fitness wins; verified = the last pipelined batch's records equal a strictly serial pass"}
            pipe5.close()
            del d5
        except Exception as e:
            legs_out["config5"] = {"error": repr(e)}

    exit_code = 0
    if rank == 0:
        recs = capi.results_from_array(allrec)
        nfr = len(recs)
        ncl = sum(min(r.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME) for r in recs)
        acc = sum(r.clusters[k].accepted for r in recs for k in range(min(r.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)))
        iters = [r.clusters[k].iterations for r in recs for k in range(min(r.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME))]
        ms_per_step = elapsed / args.steps * 1e3
        value = world * F * args.steps / elapsed
        avg_launch_ms = icp_ms / max(icp_launches, 1)
        per_launch_bytes = icp_balg / max(icp_launches / args.steps, 1)
        achieved = per_launch_bytes / (avg_launch_ms * 1e-3) / 1e9
        # HBM bytes per ICP kernel launch from the PMC passes (profiles/r01_pmc_traffic.json: separate
        # FETCH_SIZE / WRITE_SIZE runs of this same workload, gfx950 x2 FETCH correction applied)
        traffic = None
        # The dominant kernel: k_icp_lat when the template is a lattice (closed-form nearest neighbour, cd_timing.icp_search = 1);
        # otherwise the pruned searches - whole-cluster mode: ONE persistent launch per batch (k_icp_pipe; k_icp_cluster when
        # forced or when the template does not fit LDS); sliced mode: one k_icp_iter per iteration
        whole = icp_launches == args.steps
        lattice = bool(timings) and all(t.icp_search == 1 for t in timings)
        icp_kernel = "k_icp_lat" if lattice else (("k_icp_cluster" if os.environ.get("CUBOID_ICP_MODE") == "cluster" else "k_icp_pipe") if whole else "k_icp_iter")
        traffic_file, tpath = _first_profile(PMC_TRAFFIC_FILES)
        try:
            pmc = json.load(open(tpath))
            if F == 256 and N == 307200:
                traffic = pmc["kernels"][icp_kernel]["hbm_bytes_per_dispatch"]
        except (OSError, KeyError, ValueError, TypeError):
            traffic, traffic_file = None, None
        # `achieved` / `frac` / `avg_launch_ms` have ONE definition (the contract's): algorithmic bytes per launch over the average
        # duration of the kernel's launches INSIDE the timed region (HIP events on the launch's stream).  With M batches in flight
        # those launches share the chip with the other batches' kernels (`launches_in_flight` of them at any time), so the
        # duration is queueing plus work; the same kernel with the GPU to itself is under the fixed key `exclusive` (median of
        # the serial passes after the timed region; its launch shape differs: `regime`).  (no key changes meaning.)
        overlap = (icp_ms * 1e-3) / elapsed if elapsed > 0 else None

        def _regime(t):
            return {"clusters_per_workgroup": t.icp_regime >> 16, "workgroups": t.icp_regime & 0xffff, "handovers": int(t.icp_handovers),
                    "search": {0: "pruned (generic template)", 1: "lattice closed form", 2: "both"}.get(int(t.icp_search), "?")}
        timed_regime = None
        if timings and any(t.icp_regime for t in timings):   # the shape most launches of the timed region had
            import collections
            reg, cnt = collections.Counter(t.icp_regime for t in timings if t.icp_regime).most_common(1)[0]
            timed_regime = dict(_regime(next(t for t in timings if t.icp_regime == reg)), launches=cnt, of=len(timings))
        exclusive = None
        if serial_ts:
            ex_ms = float(np.median([t.icp_kernel_ms for t in serial_ts]))
            exclusive = {"avg_launch_ms": ex_ms, "avg_launch_ms_is": "median of %d launches alone on an idle GPU (serial passes after the timed region)" % len(serial_ts),
                         "achieved": per_launch_bytes / (ex_ms * 1e-3) / 1e9, "frac": per_launch_bytes / (ex_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "regime": _regime(serial_ts[0]),
                         "wave_ms_per_launch": float(np.median([t.icp_wave_ms for t in serial_ts])) if lattice else None}
        # What a batch's ICP costs with batches in flight is not the launch's duration but the WAVE-TIME it holds: sum over the
        # launch's workgroups of lifetime x waves (cd_timing.icp_wave_ms, measured by the kernel itself), against the chip's wave
        # slots at this kernel's register count (256 CUs x 16).
        wave_time = None
        if lattice and timings:
            wm = float(np.mean([t.icp_wave_ms for t in timings]))
            wave_time = {"wave_ms_per_batch": wm, "chip_wave_slots": 256 * 16, "chip_ms_per_batch": wm / (256 * 16),
                         "frac_of_step": wm / (256 * 16) / ms_per_step,
                         "note": "sum over the ICP launch's workgroups of (lifetime x waves), measured in the kernel (100 MHz clock): the share "
                                 "of the chip's wave slots (256 CUs x 16 waves at this kernel's 128 registers) a batch's ICP holds for one step"}
        # What bounds the dominant kernel: neither HBM (traffic: a tenth of the algorithmic bytes, the clusters stay in L2) nor vector
        # issue - the committed SQ counters (tools/profile_round5.sh: one launch alone, for both launch shapes) against the
        # saturated vector rate of the same chip (profiles/r04_valu_calibration.json, tools/valu_calib.hip).
        valu = None
        bound = "hbm"
        cname, cpath = _first_profile(VALU_CALIBRATION_FILES)
        cal = json.load(open(cpath)) if cpath else None

        def _valu_block(files, launch_ms, shape):
            sq_file, sq = sq_counters(icp_kernel, files)
            if not (sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_INSTS_VALU") and launch_ms):
                return None
            out_ = {"counters_file": "profiles/" + sq_file, "launch_shape_of_the_counters": shape, "launch_ms_used": launch_ms,
                    "SQ_INSTS_VALU": sq["SQ_INSTS_VALU"], "SQ_INSTS_SALU": sq.get("SQ_INSTS_SALU"), "SQ_INSTS_LDS": sq.get("SQ_INSTS_LDS"),
                    "SQ_INSTS_VMEM": sq.get("SQ_INSTS_VMEM"), "SQ_WAVE_CYCLES": sq["SQ_WAVE_CYCLES"], "SQ_WAVES": sq.get("SQ_WAVES"),
                    "valu_active_frac_of_wave_life": (sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_ACTIVE_INST_VALU") else None,
                    "wait_any_frac": (sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAIT_ANY") else None,
                    "wait_inst_any_frac": (sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAIT_INST_ANY") else None,
                    "lds_bank_conflict_frac_of_lds_active": (sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]) if sq.get("SQ_LDS_IDX_ACTIVE") and sq.get("SQ_LDS_BANK_CONFLICT") else None,
                    "vector_wave_instructions_per_64_query_pass": (sq["SQ_INSTS_VALU"] * 64.0 / (icp_balg / max(icp_launches / args.steps, 1) / 12.0)) if icp_balg else None}
            if cal:
                sat = cal["saturation"]["valu_per_ns_per_simd"]
                out_["pipe_frac"] = sq["SQ_INSTS_VALU"] / (SIMDS * launch_ms * 1e6 * sat)
                out_["saturated_valu_per_ns_per_simd"] = sat
                out_["calibration_file"] = "profiles/" + cname
            return out_
        if args.config == 3 and lattice:
            v_in = _valu_block(PMC_SQ_INFLIGHT_SHAPE_FILES, avg_launch_ms, "4 clusters x 2 waves per workgroup (forced with CUBOID_LAT_SHAPE=4,2: the shape of the timed region), one launch alone - counters serialise kernels")
            v_ex = _valu_block(PMC_SQ_FILES, exclusive["avg_launch_ms"] if exclusive else None, "1 cluster per workgroup x 8 waves (522 clusters): the shape a call with the GPU to itself picks")
            if v_in or v_ex:
                valu = {"in_flight_shape": v_in, "exclusive_shape": v_ex,
                        "note": "pipe_frac = vector wave-instructions of the launch / (1024 SIMDs x launch duration x the saturated vector rate of "
                                "profiles/r04_valu_calibration.json): for in_flight_shape the duration is the timed region's (the chip is shared), for "
                                "exclusive_shape the lone launch's.  The closed-form search issues ~5x fewer vector instructions than round 4's pruned "
                                "search (3.8e8 against 1.95e9 per launch); what is left is a chain: per round of a cluster ~19 us of passes on one wave "
                                "and ~6.5 us of single-lane solve (tools/probe_lat_phases.py, profiles/r05_lat_phases.txt)"}
                pf = (v_in or v_ex).get("pipe_frac")
                bound = "valu" if (pf or 0) >= 0.8 else "latency"
        elif args.config == 3:
            sq_file, sq = sq_counters(icp_kernel, ("r04_pmc_icp.txt",))
            if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_INSTS_VALU"):
                valu = {"counters_file": "profiles/" + sq_file, "SQ_INSTS_VALU": sq["SQ_INSTS_VALU"], "SQ_WAVE_CYCLES": sq["SQ_WAVE_CYCLES"],
                        "wait_any_frac": (sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAIT_ANY") else None,
                        "note": "generic (pruned) search: round 4's counters of one launch alone, see profiles/r04_pmc_icp.txt and DESIGN.md"}
                bound = "latency"
        if os.environ.get("CUBOID_BENCH_ICP_ITERS"):
            print("bench.py: CUBOID_BENCH_ICP_ITERS is set - an EXPERIMENT with the ICP cut short, not the metric", file=sys.stderr)
        out = {
            "metric": "frames/sec (640x480 D435 cloud, plane+cluster+ICP)" if not os.environ.get("CUBOID_BENCH_ICP_ITERS") else
                      "EXPERIMENT (ICP capped at %s iterations): NOT the metric" % os.environ["CUBOID_BENCH_ICP_ITERS"],
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("batch of %d synthetic 1 M-point frames per GPU (BASELINE config 5: 1000x1000 virtual sensor, five "
                                    "cuboids, five templates %s, every cluster against every template, lowest fitness wins)"
                                    % (F, [len(t) for t in tpl_by_slot.values()])) if args.config == 5 else
                                   ("batch of %d synthetic 640x480 D435 frames per GPU (BASELINE config 3), cuboid launch "
                                    "parameters, 7250-point template, full chain S0-S6 + pose-record gather" % F) if world == 1 else
                                   ("batch of %d synthetic 640x480 D435 frames sharded frame-per-GPU over %d GPUs, %d per GPU "
                                    "(BASELINE config 4%s), RCCL all_gather of the pose records per batch; per GPU the config-3 "
                                    "workload" % (F * world, world, F, "" if F * world == 2048 else " shape")),
                       "rccl_ranks": (dist.get_world_size() if use_dist else 0),
                       "frames_per_gpu": F, "points_per_frame": int(N), "template_points": int(len(tpl)),
                       "batches_in_flight": M,
                       "sharding": "frame-per-GPU, one all_gather of %d-byte records per batch" % capi.FRAME_RESULT_BYTES},
            "roofline": {"kernel": icp_kernel, "bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_file": ("profiles/" + traffic_file) if traffic is not None else None,
                         "traffic_note": "HBM bytes of one launch from the committed counter file (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                         "this workload, gfx950 x2 FETCH correction; counters cannot be collected inside this run)",
                         "avg_launch_ms": avg_launch_ms,
                         "avg_launch_ms_is": "the kernel's launches INSIDE the timed region (HIP events on the launch's stream), batches in flight",
                         "launches_in_flight": overlap, "launches_per_step": icp_launches / args.steps,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "regime": timed_regime, "exclusive": exclusive, "wave_time": wave_time,
                         "valu": valu,
                         "note": "achieved / frac: algorithmic bytes of the dominant kernel (SURVEY 8(d): 12 M + 12 N_s (I + 1) per cluster) over the average "
                                 "duration of its launches in the timed region, against the HBM peak as the contract defines the block; `exclusive` = the same "
                                 "for a launch that has the GPU to itself.  `bound`: HBM is not this kernel's roof (traffic is a fraction of the algorithmic "
                                 "bytes: the clusters stay in L2, the template is a few hundred table entries in LDS) and neither is vector issue "
                                 "(valu.*.pipe_frac): every cluster is a chain of ~70 rounds, each a pass over its points and a single-lane 3x3 SVD "
                                 "(DESIGN.md section 4).  With batches in flight a batch's ICP costs the chip `wave_time.chip_ms_per_batch`"},
            "icp_search": {"kernel": icp_kernel, "search": "lattice closed form (k_icp_lat.hip)" if lattice else "pruned search over an arbitrary template (k_icp.hip)",
                           "bruteforce_equivalent_pair_tests_per_step": pairs,
                           "bruteforce_equivalent_pair_tests_per_s": pairs / (icp_ms / args.steps * 1e-3) if icp_ms else None,
                           "note": "the nearest neighbour of every query is exact in both searches (same neighbour, same lowest-index tie rule as a brute "
                                   "force); the brute-force-equivalent rate is NOT executed work.  CUBOID_ICP_LATTICE=0 runs the pruned search on the same "
                                   "template: `generic_search` below",
                           "fp32_valu_peak_tflops": FP32_VALU_PEAK_TFLOPS},
            "pipeline_hbm": {"algorithmic_bytes_per_frame": balg / F, "achieved_GBps": balg / F * value / world / 1e9,
                             "frac_of_peak": balg / F * value / world / 1e9 / HBM_PEAK_GBS},
            "stage_ms_note": "per-batch stage latencies from HIP events on the batch's own stream; with more than one batch in "
                             "flight they overlap with the other batch's kernels and do not add up to ms_per_step",
            "stage_ms_per_step": {"crop_voxel": stage[0] / args.steps, "plane": stage[1] / args.steps,
                                  "extract_cluster": stage[2] / args.steps, "icp": stage[3] / args.steps,
                                  "device_total": stage[4] / args.steps},
            "single_frame_ms": ({"median": float(np.median(lat[2:])), "min": float(np.min(lat[2:])),
                                 "note": "BASELINE config 2: one frame, full chain, device-resident input, host wall clock of the "
                                         "synchronous C-ABI call; measured after the timed region"} if len(lat) > 2 else None),
            "single_frame_ms_host": ({"process_frame_median": float(np.median(lat_host[2:])), "process_frame_min": float(np.min(lat_host[2:])),
                                      "ground_plane_median": float(np.median(lat_gp[2:])), "ground_plane_min": float(np.min(lat_gp[2:])),
                                      "note": "host-fed: pageable host buffer in (4.9 MB H2D), results out; cd_process_frame = the whole chain, "
                                              "cd_ground_plane = ground_plane_segmentation's callback (gps.cpp:43-112) as one call incl. the download of "
                                              "the kept records; host wall clock of the synchronous C-ABI call"} if len(lat_host) > 2 else None),
            "guess_leg": guess_leg,
            "host_fed": legs_out.get("host_fed"), "big_template_ms": legs_out.get("big_template_ms"), "config5": legs_out.get("config5"),
            "generic_search": legs_out.get("generic_search"), "object_launch": legs_out.get("object_launch"),
            "cu_fill_debug": cu_fill,
            "verified": verified,
            "verified_note": "records of the last timed step (batches in flight, the ICP kernel in its in-flight launch shape with refilled slots, gathered) are "
                             "byte-identical to a strictly serial pass run after the timed region",
            "icp": {"clusters": ncl, "accepted": int(acc), "mean_iterations": float(np.mean(iters)) if iters else 0.0,
                    "max_iterations": int(max(iters)) if iters else 0, "frames": nfr},
        }
        if not args.no_cpu_baseline and world == 1 and args.config == 3:   # the contract: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(frames, prm, tpl, gpu_records=allrec)
            out["speedup_vs_cpu_1thread"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
        if verified is False or out.get("cpu_baseline", {}).get("oracle_check", {}).get("ok") is False:
            sys.stdout.flush()
            print("bench.py: the timed path's records FAILED verification", file=sys.stderr)
            exit_code = 1
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if pipe is not None:
        pipe.close()
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
