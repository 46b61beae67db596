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
PMC_TRAFFIC_FILES = ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")   # HBM bytes per dispatch: separate --pmc FETCH_SIZE / WRITE_SIZE passes
PMC_SQ_FILES = ("r04_pmc_icp.txt", "r03_pmc_icp.txt", "r02_pmc_icp.txt")                   # SQ counters per kernel (separate --pmc passes)
ICP_WORK_FILES = ("r04_icp_work.json", "r03_icp_work.json")                                # executed distance tests of the dominant kernel (-DCD_STATS build)
VALU_CALIBRATION_FILES = ("r04_valu_calibration.json",)   # what the SQ counters read on a SATURATED vector pipe (tools/valu_calib.hip, same launch shape)
WAVES_PER_SIMD_ICP = 4          # k_icp_pipe: one 1024-thread workgroup per CU = 16 waves = 4 per SIMD
SIMDS = 1024                    # 256 CUs x 4


def _first_profile(names):
    for n in names:
        p = os.path.join(ROOT, "profiles", n)
        if os.path.exists(p):
            return n, p
    return None, None


def sq_counters(kernel):
    """SQ counters of `kernel` from the committed PMC summary (lines "<kernel> {...} dispatches n" per pass)."""
    import ast
    name, path = _first_profile(PMC_SQ_FILES)
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

    # Self-check of the timed path, outside the timed region: the records of the LAST timed step (k_icp_pipe with refilled
    # slots, other batches in flight, gathered over all ranks) must be byte-identical to a strictly serial pass of this
    # rank's batch on an otherwise idle GPU.
    verified = None
    serial_rec = None
    serial_t = None
    if not args.no_verify:
        fence()
        res = (capi.CdFrameResult * F)()
        ctx.process_batch_device(d_frames.data_ptr(), 16, N, F, prm, results=res)
        serial_rec = capi.results_to_array(res).copy()
        serial_t = ctx.timing()
        for _ in range(4 if M > 1 else 0):   # (a few more for the timing: the first pass after the pipelined region still finds the
            # other contexts' data in L2, and a lone launch varies by ~3 % from one to the next; the fastest of five is kept)
            ctx.process_batch_device(d_frames.data_ptr(), 16, N, F, prm, results=res)
            t2 = ctx.timing()
            if 0 < t2.icp_kernel_ms < serial_t.icp_kernel_ms:
                serial_t = t2
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
        # (3) BASELINE config 5: 1 M-point frames, five cuboids, five templates, every cluster against every template.  The headline's
        # pipeline is closed first (its arenas are not needed any more; what slowed this leg down was the process's hardware-queue
        # budget, see DEFAULT_HW_QUEUES).
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
        # whole-cluster mode: ONE persistent launch per batch (k_icp_pipe; k_icp_cluster when forced or when the
        # template does not fit LDS); sliced mode: one k_icp_iter per iteration
        whole = icp_launches == args.steps
        icp_kernel = ("k_icp_cluster" if os.environ.get("CUBOID_ICP_MODE") == "cluster" else "k_icp_pipe") if whole else "k_icp_iter"
        traffic_file, tpath = _first_profile(PMC_TRAFFIC_FILES)
        try:
            pmc = json.load(open(tpath))
            if F == 256 and N == 307200:
                traffic = pmc["kernels"][icp_kernel]["hbm_bytes_per_dispatch"]
        except (OSError, KeyError, ValueError, TypeError):
            pass
        # With M batches in flight the persistent launches of several batches share the chip: the per-launch duration of the
        # timed region is then longer than the kernel's cost (8.89 ms per launch against 5.48 ms per step).  The
        # headline achieved / frac therefore use the duration of ONE launch on an otherwise idle GPU - the strictly serial pass of
        # the self-check after the timed region, HIP events on the context's stream like the timed launches, <= ms_per_step -
        # and the overlapped figure is the secondary one (`in_flight`).
        overlap = (icp_ms * 1e-3) / elapsed if elapsed > 0 else None
        in_flight = {"avg_launch_ms": avg_launch_ms, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "launches_in_flight": overlap,
                     "note": "the same kernel's launches inside the timed region (HIP events): `launches_in_flight` of them share the chip at any "
                             "time, so this duration is queueing plus work"}
        excl_ms = None
        regime = None
        if serial_t is not None and serial_t.icp_kernel_launches == 1 and serial_t.icp_kernel_ms > 0:
            excl_ms = float(serial_t.icp_kernel_ms)
            regime = {"slots": serial_t.icp_regime >> 16, "workgroups": serial_t.icp_regime & 0xffff, "handovers": int(getattr(serial_t, "icp_handovers", 0))}
        head_ms = excl_ms if excl_ms is not None else avg_launch_ms
        head_achieved = per_launch_bytes / (head_ms * 1e-3) / 1e9
        timed_regime = None
        if timings:   # the shape most launches of the timed region had (the last ones, with the pipeline draining, fall back to 2 x 256)
            import collections
            reg, cnt = collections.Counter(t.icp_regime for t in timings if t.icp_regime).most_common(1)[0] if any(t.icp_regime for t in timings) else (0, 0)
            if reg:
                timed_regime = {"slots": reg >> 16, "workgroups": reg & 0xffff, "launches": cnt, "of": len(timings)}
        # What bounds the dominant kernel.  Its working set is LDS/L2-resident (traffic = 0.37 x algorithmic bytes), so HBM is not
        # its roof; the SQ counters (committed, collected by tools/profile_round4.sh on ONE launch alone: the regime of `achieved`)
        # are read against what the same counters show on a SATURATED vector pipe at the same launch shape
        # (profiles/r04_valu_calibration.json, tools/valu_calib.hip): both round 3's formula and the first-principles one are
        # reported, raw and as a fraction of their saturation value.
        sq_file, sq = sq_counters(icp_kernel)
        valu = None
        bound = "hbm"
        if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_ACTIVE_INST_VALU") and sq.get("SQ_INSTS_VALU") and args.config == 3:
            cname, cpath = _first_profile(VALU_CALIBRATION_FILES)
            cal = json.load(open(cpath)) if cpath else None
            wave_cycles = sq["SQ_WAVE_CYCLES"] * 4.0 / (256 * 16)       # quad-cycles summed over 4096 waves -> cycles of one wave's life
            A = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"] * WAVES_PER_SIMD_ICP
            B = sq["SQ_INSTS_VALU"] * 2.0 / (SIMDS * wave_cycles)
            valu = {"counters_file": "profiles/" + sq_file, "calibration_file": ("profiles/" + cname) if cname else None,
                    "regime_of_the_counters": "one launch alone (bench.py --steps 1 --warmup 0): 2 slots x 256 workgroups, the regime of roofline.achieved",
                    "SQ_INSTS_VALU": sq["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": sq["SQ_ACTIVE_INST_VALU"], "SQ_WAVE_CYCLES": sq["SQ_WAVE_CYCLES"],
                    "waves_per_simd": WAVES_PER_SIMD_ICP,
                    "A_active_over_wave_cycles_x_waves": A, "B_insts_x2_over_simd_cycles": B,
                    "salu_per_valu": (sq.get("SQ_INSTS_SALU", 0.0) / sq["SQ_INSTS_VALU"]),
                    "wait_any_frac": (sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAIT_ANY") else None,
                    "wait_inst_any_frac": (sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAIT_INST_ANY") else None,
                    "active_inst_any_frac": (sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_ACTIVE_INST_ANY") else None}
            if cal:
                sat, mix = cal["saturation"], cal["saturation_search_mix"]
                valu.update({"A_at_saturation": sat["A"], "B_at_saturation": sat["B"],
                             "pipe_frac": B / sat["B"], "pipe_frac_vs_search_mix": B / mix["B"],
                             "note": "pipe_frac = B / B_at_saturation = A / A_at_saturation: the share of the saturated f32 vector rate (independent "
                                     "v_fma_f32, four waves per SIMD) this kernel issues; against a stream of its own instruction mix (LDS reads, 64-bit "
                                     "key minima) pipe_frac_vs_search_mix.  Round 3 printed A as 'issue_frac 0.86': A reads 2.36, not 1.0, when the "
                                     "pipe is full.  The counters are of ONE launch alone on the GPU, which since the second half of round 4 keeps "
                                     "the workgroups that ran out of clusters waiting inside the launch (hand-overs): their parked waves count in "
                                     "SQ_WAVE_CYCLES (8.2e9 -> 1.0e10 for the same 1.95e9 vector instructions), so the per-wave-cycle rates read a "
                                     "fifth lower than the 0.40 of a launch whose idle workgroups end"})
                pf = valu["pipe_frac"]
                bound = "valu" if pf >= 0.8 else "latency"
            wname, wpath = _first_profile(ICP_WORK_FILES)
            if wpath:
                try:
                    w = json.load(open(wpath))
                    fl = float(w["executed_flops"])
                    valu.update({"work_file": "profiles/" + wname, "executed_pair_tests": w["executed_pair_tests"]["total"],
                                 "flops_per_test": w["flops_per_pair_test"], "executed_flops_per_launch": fl,
                                 "achieved_tflops": fl / (head_ms * 1e-3) / 1e12,
                                 "frac_of_157.3_TFLOPs": fl / (head_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS})
                except (OSError, KeyError, ValueError):
                    pass
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
            "roofline": {"kernel": icp_kernel, "bound": bound, "achieved": head_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": head_achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_file": ("profiles/" + traffic_file) if traffic is not None else None,
                         "traffic_note": "HBM bytes of one launch from the committed counter file (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                         "this workload, gfx950 x2 FETCH correction; counters cannot be collected inside this run)",
                         "avg_launch_ms": head_ms, "avg_launch_ms_is": "one launch alone on an idle GPU (fastest of five serial passes after the timed region)" if excl_ms is not None else
                                                                        "launches of the timed region (no serial pass: --no-verify)",
                         "launches_per_step": icp_launches / args.steps,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "regime": regime, "in_flight": in_flight, "in_flight_regime": timed_regime,
                         "valu": valu,
                         "note": "achieved/frac: algorithmic bytes of the dominant kernel (SURVEY 8(d): 12 M + 12 N_s (I + 1) per cluster) over the "
                                 "duration of ONE launch with the GPU to itself, against the HBM peak as the contract defines the block - a duration "
                                 "that is the kernel's cost (<= ms_per_step); `in_flight` = the same over the launches of the timed region, which "
                                 "overlap.  `bound`: HBM is not this kernel's roof (traffic < algorithmic bytes: the working set is LDS/L2-resident) "
                                 "and, measured against a saturated pipe, neither is vector issue (valu.pipe_frac): its waves sit at s_waitcnt / "
                                 "s_sleep for valu.wait_any_frac of their life - dependent LDS round trips and scalar/vector hand-overs (DESIGN.md "
                                 "section 4)"},
            "icp_search": {"kernel": icp_kernel, "bruteforce_equivalent_pair_tests_per_step": pairs,
                           "bruteforce_equivalent_pair_tests_per_s": pairs / (icp_ms / args.steps * 1e-3) if icp_ms else None,
                           "note": "exact search by pruning (lane-per-query grid walk for near queries, wave-per-query k-d patch search for far ones): "
                                   "the brute-force-equivalent rate is NOT executed work; executed tests and the issue-slot occupancy are in roofline.valu",
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
            "cu_fill_debug": cu_fill,
            "verified": verified,
            "verified_note": "records of the last timed step (batches in flight, k_icp_pipe with refilled slots, gathered) are "
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
