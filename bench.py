"""bench.py -- frames/sec of the hot path (Brox flow + EKF update) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one video frame: upload of frame k+1 and its mask (copy stream, frame ring), Brox
optical flow of the frame pair (k, k+1), then IteratedMSKalmanFilter.compute on frame k+1 with
that flow (predict, mask projection, iterated measurement update), flow handed over in
device memory.  Workload at every N: BASELINE.json config "1024x1024 video, ~200-vertex
mesh" -- a textured disk advected by an analytic field, synthetic, one independent video
per GPU (weak scaling: the EKF is a recurrence over the frames of one video, videos are
the shardable unit; the only collective is the final gather of the tracked states).

Prints ONE JSON line (rank 0).  The frames and masks of the timed region are read from host memory and
uploaded inside it (the pipeline's frame ring: one frame + one mask per step over a copy stream), as
SURVEY.md 8d asks.  `roofline` is the SOR kernel: kernel start/stop events of every SOR launch of one
full flow series inside the timed region.
  frac / achieved    PHYSICAL: HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE x 2 +
                     WRITE_SIZE, profiles/r04_sor_pmc.json, valid for the kernel sources it names) over the
                     launch time measured here, against 8 TB/s.  When the PMC file was taken for other
                     kernel sources the lower bound 52 B/px per LAUNCH stands in (traffic: null).
  *_contract         the contract's figure, 52 B per pixel per red-black ITERATION.  k_sor
                     runs K iterations per pass over memory, so this is not a bound and exceeds 1.
  limited_by         what the SQ counters show (profiles/r03_sor_sq_counters.csv).
  finest_level       the same figures for the launches of pyramid level 0 alone (the 40 % target).
`steady_state` is the rate over the last 60 % of the timed frames (the first flow series of the region
has nothing to hide behind; `value` includes it).  `cpu_baseline` is the oracle (C/OpenMP restatement of
the reference's CPU path) timed on this host: whole Brox pairs and one whole KFState.update at the
bench's size.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SOR_BYTES_PER_PIXEL_ITERATION = 52.0      # 11 f32 fields read + 2 written
HBM_PEAK_GBPS = 8000.0                    # MI355X_MICROARCH.md: 8.0 TB/s spec


SOR_PMC = os.path.join(ROOT, "profiles", "r04_sor_pmc.json")
SOR_LIMITED_BY = ("tile load phase (seven coefficient planes per tile through the L2 -> CU path) not overlapped with the "
                  "sweeps of the same workgroup, then vector issue and barrier waits: profiles/r03_sor_sq_counters.csv")


def kernel_revision():
    """sha256 of the SOR kernel's sources: the PMC file names the revision it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for fn in ("brox_kernels.h", "brox.hip"):
        with open(os.path.join(ROOT, "kalman-hydra_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per SOR launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of the same 8-pair 1024^2 flow series, tools/run_pmc.sh + tools/sor_pmc_json.py;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes and as the k_add_out calibration in that file
    confirms) -> dict with traffic_bytes_per_launch (series average), finest_level.traffic_bytes_per_launch.
    None unless the file was taken for the kernel sources as they are now."""
    try:
        with open(SOR_PMC) as f:
            d = json.load(f)
        if d.get("kernel_revision") != kernel_revision():
            return None
        float(d["traffic_bytes_per_launch"])
        return d
    except (OSError, KeyError, ValueError, TypeError):
        return None


def _sor_figures(ms, launches, pxit, px, traffic_per_launch):
    """One set of roofline figures for `launches` SOR launches that took `ms` in total."""
    t = ms * 1e-3
    n = max(1, launches)
    alg = SOR_BYTES_PER_PIXEL_ITERATION * pxit
    minb = SOR_BYTES_PER_PIXEL_ITERATION * px
    phys = traffic_per_launch * n if traffic_per_launch else minb
    gbps = (lambda b: b / t / 1e9) if t > 0 else (lambda b: 0.0)
    return {"achieved": gbps(phys), "frac": gbps(phys) / HBM_PEAK_GBPS,
            "traffic": traffic_per_launch, "traffic_source": "pmc" if traffic_per_launch else "model: 52 B/px per launch",
            "achieved_contract": gbps(alg), "frac_contract": gbps(alg) / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_launch": alg / n, "min_bytes_per_launch": minb / n,
            "frac_min_traffic": gbps(minb) / HBM_PEAK_GBPS,
            "launches": launches, "avg_launch_us": 1e3 * ms / n}


def sor_roofline(sor_ms, sor_launches, sor_pxit, sor_px, pmc, profiled, levels=None, pairs=8):
    """The roofline block of the JSON line.  sor_pxit = sum over the timed launches of pixels x fused
    iterations, sor_px = sum of pixels (one pass over memory each); levels = hm_brox_profile_levels."""
    out = {"bound": "hbm", "kernel": "k_sor", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "bytes_per_pixel_iteration": SOR_BYTES_PER_PIXEL_ITERATION, "limited_by": SOR_LIMITED_BY,
           "profiled": profiled}
    # the PMC passes were taken on series of 8 pairs; every launch's traffic is proportional to the pairs it covers
    scale = pairs / 8.0
    # (the series average only for the same launches: a series of fewer pairs takes more levels in one launch each)
    same_mix = pmc is not None and 2 * sor_launches == int(pmc.get("launches", 0))
    out.update(_sor_figures(sor_ms, sor_launches, sor_pxit, sor_px, scale * float(pmc["traffic_bytes_per_launch"]) if same_mix else None))
    if levels and levels[0]["launches"] > 0:
        l0 = levels[0]
        tl = None
        if pmc and scale * pmc.get("finest_level", {}).get("pixels_per_launch", -1) == l0["pixels"] / l0["launches"]:
            tl = scale * float(pmc["finest_level"]["traffic_bytes_per_launch"])
        f = _sor_figures(l0["ms"], l0["launches"], l0["pixel_iterations"], l0["pixels"], tl)
        f["level"] = "%dx%d" % (l0["w"], l0["h"])
        out["finest_level"] = f
    out["note"] = ("frac = HBM bytes actually moved (PMC) / launch time / 8 TB/s; frac_contract follows the contract's "
                   "per-iteration byte model and exceeds 1 because k_sor fuses K red-black iterations per pass over memory")
    return out


def make_video(n, frames, seed):
    from hydra_mi import synth
    video, masks, centre, radius = synth.disk_video(n, frames, "translate_leftup", seed)
    return video, masks, centre, radius


def cpu_baseline(n, video, masks, dm, iters_per_frame, budget_s=30.0):
    """The oracle on the host cores, BASELINE.md section 2 protocol where the budget allows:

    * Brox: whole 1024^2 pairs with the C oracle, 3 warm + 5 timed, median, with all cores and with
      one thread;
    * EKF: ONE whole KFState.update of the reference's CPU path (kalman.py:491-518, 583-606: 2*4N jz
      evaluations + one j per non-zero of the upper triangle of J, every one of them full-frame renders,
      cuda.py:972-1010) with the C/OpenMP twin of the NumPy oracle (oracle/ekf_ref_c.c), all cores, timed
      once -- it is tens of seconds; with one thread a 1/32 strided sample of the same evaluations is timed
      and scaled.
    frames/sec = 1 / (t_brox + iterations_per_frame * t_update)."""
    from oracle import brox_oracle, ekf_c, ekf_ref
    cores = os.cpu_count() or 1
    threads = max(1, min(64, cores))

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    brox_oracle.set_threads(threads)
    t_brox = timed(lambda: brox_oracle.calc(video[0], video[1]), 3, 5)
    brox_oracle.set_threads(1)
    t_brox1 = timed(lambda: brox_oracle.calc(video[0], video[1]), 0, 1)
    brox_oracle.set_threads(threads)
    u, v = brox_oracle.calc(video[0], video[1])
    brox_oracle.set_threads(1)
    N = dm.size()
    meas = ekf_c.Measurement(N, dm.t, dm.p, video[0], 1e-3, 1.0, 1.0, threads=threads)
    X = np.concatenate((dm.p.reshape(-1), np.zeros(2 * N)))
    flow = np.dstack((u, v)).astype(np.float32)
    _, J = ekf_ref.adjacency(N, dm.t)
    pi, pj = np.nonzero(np.triu(J == 1))
    t0 = time.perf_counter()
    meas.jacobian_all(X, video[1], flow, masks[1])
    t_jac = time.perf_counter() - t0
    # the Hessian pass within the budget: everything if it fits, else a strided sample scaled up
    t0 = time.perf_counter()
    meas.hessian_pairs(pi[::64], pj[::64])
    t_probe = time.perf_counter() - t0
    est = t_probe * 64
    stride = 1 if est <= budget_s else int(np.ceil(est / budget_s))
    t0 = time.perf_counter()
    meas.hessian_pairs(pi[::stride], pj[::stride])
    t_hess = (time.perf_counter() - t0) * (len(pi) / len(pi[::stride]))
    t_update = t_jac + t_hess
    # one thread: a 1/32 sample of both passes
    meas.set_threads(1)
    idx = np.arange(0, 4 * N, 32, dtype=np.int32)
    t0 = time.perf_counter()
    meas.jacobian_all(X, video[1], flow, masks[1], 2.0, idx)
    t1 = (time.perf_counter() - t0) * (4 * N / len(idx))
    t0 = time.perf_counter()
    meas.hessian_pairs(pi[::128], pj[::128])
    t1 += (time.perf_counter() - t0) * (len(pi) / len(pi[::128]))
    t_frame = t_brox + iters_per_frame * t_update
    t_frame1 = t_brox1 + iters_per_frame * t1
    return {"value": 1.0 / t_frame, "unit": "frames/sec", "cores": threads, "kind": "port",
            "value_one_thread": 1.0 / t_frame1, "os_cpu_count": cores,
            "seconds": {"brox_pair": t_brox, "brox_pair_one_thread": t_brox1, "kfstate_update": t_update,
                        "kfstate_update_one_thread_scaled": t1},
            "sample": "C/OpenMP restatement of the reference CPU path (oracle/): Brox on a %dx%d pair, 3 warm + 5 timed, "
                      "median (%d threads: %.3f s; 1 thread, 1 run: %.2f s); one whole KFState.update = %d jz (central "
                      "differences, 2 renders each) + %s of the %d j evaluations (2 renders each%s), %d threads: %.1f s; "
                      "1 thread from a 1/32 (jz) and 1/128 (j) strided sample scaled: %.0f s; x %.1f IEKF iterations per "
                      "frame" % (n, n, threads, t_brox, t_brox1, 4 * N, "all" if stride == 1 else "every %d-th" % stride,
                                 len(pi), "" if stride == 1 else ", scaled", threads, t_update, t1, iters_per_frame)}


class StubFlow:
    """Stands in for brox.BroxOpticalFlow when the sharding / gather / reporting code is rehearsed without a
    GPU (tests/test_host_cpu.py drives `bench.py --workload flowbatch --backend gloo --stub-flow` with two
    ranks): u = f1 - f0, v = f0 + f1 as floats, on the CPU.  Never used on a measured run."""

    def __init__(self, n):
        self.n = n

    def run(self, F0, F1, U, V, B):
        import torch
        U.copy_(F1.to(torch.float32) - F0.to(torch.float32))
        V.copy_(F0.to(torch.float32) + F1.to(torch.float32))

    def profile(self, on):
        pass

    def profile_read(self):
        return 0.0, 0, 0.0, 0.0

    def profile_levels(self):
        return []


class DeviceFlow:
    """The product path: hm_brox_calc_dev in launch series of B pairs on frames resident in HBM."""

    def __init__(self, n, B, dev):
        from hydra_mi import brox
        self.bf = brox.BroxOpticalFlow(n, n, max_batch=B, device=dev)

    def run(self, F0, F1, U, V, B):
        P = F0.shape[0]
        for s in range(0, P, B):
            nb = min(B, P - s)
            self.bf.calc_dev(nb, F0[s].data_ptr(), F1[s].data_ptr(), U[s].data_ptr(), V[s].data_ptr())
        self.bf.sync()

    def profile(self, on):
        self.bf.profile(on)

    def profile_read(self):
        return self.bf.profile_read()

    def profile_levels(self):
        return self.bf.profile_levels()


def dist_info(args, world, dev):
    """What a SCALE record needs to show that the collective backend saw N ranks: world size, backend as
    torch.distributed reports it, and the device of every rank (all_gather_object)."""
    import torch.distributed as dist
    mine = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
            "device": ("cpu (stub)" if args.stub_flow else "cuda:%d" % dev), "pid": os.getpid()}
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        return {"world": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks}
    return {"world": 1, "backend": None, "ranks": [mine]}


def flowbatch(args, rank, world, dev, coll_dev):
    """BASELINE config 5: world x `pairs-per-gpu` independent 1024^2 pairs (pair i = seed i of the
    `translate_leftup_stretch` warp, SURVEY.md 8d: 256 pairs on 8 GPUs), cut into contiguous blocks by
    hydra_mi.batch.shard, one block per rank; one step = the flow of the whole block in launch series of
    --flow-batch pairs; nothing is exchanged while computing, and the timed region ends with the batch path's
    only collective, hydra_mi.batch.gather_to_root of all flow planes (RCCL over xGMI with --backend nccl)."""
    import torch
    import torch.distributed as dist
    from hydra_mi import synth, batch
    n, P, B = args.size, args.pairs_per_gpu, max(1, args.flow_batch)
    total = P * world
    mine = batch.shard(total, rank, world)
    pairs = [synth.warp_pair(n, "translate_leftup_stretch", batch.pair_seed(i))[:2] for i in mine]
    place = (lambda t: t) if args.stub_flow else (lambda t: t.cuda())
    F0 = place(torch.from_numpy(np.stack([p[0] for p in pairs])))
    F1 = place(torch.from_numpy(np.stack([p[1] for p in pairs])))
    U = torch.empty((len(mine), n, n), dtype=torch.float32, device=F0.device)
    V = torch.empty_like(U)
    sync = (lambda: None) if args.stub_flow else torch.cuda.synchronize
    sync()
    eng = StubFlow(n) if args.stub_flow else DeviceFlow(n, B, dev)
    info = dist_info(args, world, dev)

    for _ in range(args.warmup):
        eng.run(F0, F1, U, V, B)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k == args.steps - 1:
            eng.profile(True)
        eng.run(F0, F1, U, V, B)
    levels = eng.profile_levels()
    sor_ms, sor_launches, sor_pxit, sor_px = eng.profile_read()
    flows = torch.stack((U, V), dim=1).to(coll_dev)                  # (pairs of this rank, 2, n, n)
    gathered = batch.gather_to_root(flows, total, dst=0)
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        assert gathered.shape[0] == total
        print(json.dumps({
            "metric": "frame pairs/sec (Brox flow) at %d^2" % n, "value": total * args.steps / elapsed,
            "unit": "pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d independent %dx%d frame pairs (seeds 0..%d) in blocks of %d per GPU, Brox defaults, "
                                   "flows gathered to rank 0" % (total, n, n, total - 1, P),
                       "flow_batch": B, "parallelism": "pairs x%d" % world},
            "distributed": dict(info, gathered_bytes=int(gathered.numel() * 4), collective="gather (hydra_mi.batch.gather_to_root)"),
            "stub_flow": bool(args.stub_flow),
            "roofline": None if args.stub_flow else sor_roofline(
                sor_ms, sor_launches, sor_pxit, sor_px, pmc_traffic() if (n == 1024 and B == 8) else None,
                "last step of the timed region", levels)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--h0", type=float, default=0.047, help="mesh edge length as a fraction of the frame size")
    ap.add_argument("--flow-batch", type=int, default=8, help="consecutive frame pairs per Brox launch series")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-flow-handle", action="store_true", help="experiment: one flow series in flight at a time (default: two, on two handles)")
    ap.add_argument("--split-start", action="store_true", help="experiment: a phase starts with two flow series side by side (pipeline.split_start)")
    ap.add_argument("--first-series", type=int, default=0, help="experiment: fixed size of the first flow series of a phase (default: sized from measurements)")
    ap.add_argument("--cu-reserve", type=int, default=None,
                    help="compute units the flow stream leaves to the filter (default: the pipeline's)")
    ap.add_argument("--workload", default="video", choices=["video", "flowbatch"],
                    help="video = the headline metric (flow + EKF per frame); flowbatch = BASELINE config 5: "
                         "independent frame pairs sharded over the GPUs, one gather of the flows at the end")
    ap.add_argument("--pairs-per-gpu", type=int, default=32)
    ap.add_argument("--videos-per-gpu", type=int, default=1,
                    help="video workload: independent videos tracked concurrently on each GPU, one thread each "
                         "(default 1: the configuration the metric is quoted on)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real run); gloo only to rehearse several ranks on one GPU")
    ap.add_argument("--resident", action="store_true",
                    help="video workload: upload the whole video before the clock starts (round 2's protocol) instead "
                         "of streaming every frame and mask through the frame ring inside the timed region")
    ap.add_argument("--stub-flow", action="store_true",
                    help="flowbatch only: a CPU stand-in for the flow, to rehearse sharding + gather without a GPU "
                         "(with --backend gloo); the line it prints is marked stub_flow and is not a measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Asked for N ranks without a launcher: start them -- torch.distributed.run as a CHILD process, before this one
        # has imported anything that touches the GPU (never an exec of a process that has) -- and leave with its exit code.
        # A run that printed `n_gpus: 1` for --gpus 8 would be a wrong record.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d (python -m torch.distributed.run --nproc-per-node %d ... "
                         "bench.py --gpus %d ...)" % (args.gpus, world, args.gpus, args.gpus))

    import torch
    import torch.distributed as dist
    import hydra_mi  # noqa: F401

    if args.stub_flow:
        if args.workload != "flowbatch" or args.backend != "gloo":
            raise SystemExit("--stub-flow is for --workload flowbatch --backend gloo")
        if world > 1:
            dist.init_process_group("gloo")
        return flowbatch(args, rank, world, -1, "cpu")
    from hydra_mi import kalman, mesh, batch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    ndev = torch.cuda.device_count()
    dev = local_rank if args.backend == "nccl" else local_rank % max(1, ndev)
    if dev >= ndev:
        raise SystemExit("rank %d wants cuda:%d but only %d device(s) are visible" % (rank, dev, ndev))
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"

    n = args.size
    K, Wm = args.steps, args.warmup
    if args.workload == "flowbatch":
        return flowbatch(args, rank, world, dev, coll_dev)
    import threading
    from hydra_mi.pipeline import FlowEKFPipeline
    B = max(1, args.flow_batch)
    V = max(1, args.videos_per_gpu)
    info = dist_info(args, world, dev)
    # the videos of the job are the shardable unit: video i has seed i, contiguous blocks per rank (batch.shard)
    my_videos = batch.shard(world * V, rank, world)

    class Track:
        """One video: a filter and the package's streaming pipeline (hydra_mi.pipeline.FlowEKFPipeline: every frame
        and mask read from host memory and uploaded over a copy stream into a ring of frame slots, the flow of the
        next frames computed on the flow handle's stream while the filter works on the current ones)."""

        def __init__(self, seed):
            frames = K + Wm + 1
            self.video, self.masks, centre, radius = make_video(n, frames, seed=seed)
            self.dm = mesh.disk_mesh(centre[0], centre[1], radius - 1.0, args.h0 * n)
            self.kf = kalman.IteratedMSKalmanFilter(self.dm, self.video[0], np.zeros((n, n, 2), np.float32), True, device=dev)
            extra = {} if args.cu_reserve is None else {"cu_reserve": args.cu_reserve}
            if args.one_flow_handle or args.split_start:     # (split_start is an experiment of the one-series mode)
                extra["concurrent_series"] = False
            self.pipe = FlowEKFPipeline(self.kf, self.video, self.masks, flow_batch=B, device=dev, resident=args.resident, **extra)
            self.bf = self.pipe.bf
            if args.split_start:
                self.pipe.split_start = True
                if len(self.pipe.bfs) < 2:               # (not inside the timed region)
                    self.pipe.bfs.append(self.pipe._make_handle())
            if args.first_series:
                self.pipe.first_series = args.first_series
            if os.environ.get("HYDRA_MI_BENCH_TRACE"):
                self.pipe.trace = lambda msg: print(msg, file=sys.stderr)

        def warmup(self):
            for bf in self.pipe.bfs:
                bf.profile(True)
            self.pipe.run(0, Wm)
            self.pipe.flow_sync()
            for bf in self.pipe.bfs:
                bf.profile_read()
                bf.profile(False)
            self.pipe.t_flow = self.pipe.t_ekf = 0.0
            self.pipe.iters = 0
            self.pipe.frame_done = []
            self.pipe.ring.bytes_uploaded = 0
            self.kf.predtime = self.kf.updatetime = self.kf.projecttime = 0.0

        def timed(self):
            self.pipe.run(Wm, Wm + K)

        def flow_sync(self):
            self.pipe.flow_sync()

        def close(self):
            """helper threads joined, copy stream / flow handles / filter handle (its streams, page-locked blocks and
            worker thread) released: nothing native is alive when the interpreter and the HIP runtime finalise"""
            self.pipe.close()
            self.kf.close()

    tracks = [Track(seed) for seed in my_videos]
    try:
        run_video(args, tracks, rank, world, dev, coll_dev, info, V, K, Wm, B, n)
    finally:
        for tr in tracks:
            tr.close()
        from hydra_mi import _lib
        _lib.close_all()
    if world > 1:
        dist.destroy_process_group()


def run_video(args, tracks, rank, world, dev, coll_dev, info, V, K, Wm, B, n):
    import threading
    import torch
    import torch.distributed as dist
    from hydra_mi import batch
    for tr in tracks:
        tr.warmup()
    # the flow series of the timed region start small and grow to B pairs (pipeline.py: sized from what series and
    # frames have taken so far); the first one of B pairs is profiled (kernel start/stop events for every SOR launch)
    tracks[0].pipe.profile_full = True
    # (with two series in flight, each sized from measurements, a timed region of 20 frames has series of 1, 2, 2, 3, 3, 5
    # and 8 pairs: the first one of at least 3 that runs beside the filter is the one profiled; the PMC traffic per
    # launch scales with the pairs)
    tracks[0].pipe.profile_min_pairs = min(B, 3) if K < 40 else B      # (a full series only comes after ~30 frames)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if V == 1:
        tracks[0].timed()
    else:                               # independent trackers, one thread each (the long calls drop the GIL)
        failures = []

        def run(tr):
            try:
                tr.timed()
            except Exception as exc:    # noqa: BLE001 -- reported below: the bench must not hang on a dead thread
                failures.append(repr(exc))
        threads = [threading.Thread(target=run, args=(tr,)) for tr in tracks]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if failures:
            raise SystemExit("a tracker failed: %s" % failures[0])
    # the batch path's only exchange: the tracked states of all videos to rank 0 (hydra_mi.batch.gather_to_root)
    state = torch.from_numpy(np.stack([tr.kf.state.X.reshape(-1) for tr in tracks])).to(coll_dev)
    gathered = batch.gather_to_root(state, world * V, dst=0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    elapsed = t_end - t0
    for tr in tracks:
        tr.flow_sync()
    prof_pairs, prof_where = tracks[0].pipe.profiled_pairs, "of the timed region"
    if prof_pairs == 0:                 # a timed region too short for a series of B pairs: one more, after the clock
        tr = tracks[0]
        prof_pairs, prof_where = min(B, len(tr.video) - 1), "after the timed region"
        tr.pipe.begin(0, prof_pairs)
        tr.pipe.ring.ensure(prof_pairs + 1, 0)
        tr.pipe.ring.sync()
        tr.bf.profile(True)
        tr.bf.calc_dev(prof_pairs, tr.pipe.ring.run_ptr(0), tr.pipe.ring.run_ptr(0) + n * n, tr.pipe.d_u.ptr, tr.pipe.d_v.ptr)
        tr.bf.sync()
    prof_bf = tracks[0].pipe.profiled_handle or tracks[0].bf
    levels = prof_bf.profile_levels()
    sor_ms, sor_launches, sor_pxit, sor_px = prof_bf.profile_read()
    # the same series once more with nothing beside it (after the clock): what the kernel does when it has the chip
    tr = tracks[0]
    alone_pairs = min(B, len(tr.video) - 1)
    tr.pipe.begin(0, alone_pairs)
    tr.pipe.ring.ensure(alone_pairs + 1, 0)
    tr.pipe.ring.sync()
    torch.cuda.synchronize()
    tr.bf.profile(True)
    tr.bf.calc_dev(alone_pairs, tr.pipe.ring.run_ptr(0), tr.pipe.ring.run_ptr(0) + n * n, tr.pipe.d_u.ptr, tr.pipe.d_v.ptr)
    tr.bf.sync()
    alone_levels = tr.bf.profile_levels()
    alone = tr.bf.profile_read()
    tr.bf.profile(False)
    kf, video, masks, dm = tracks[0].kf, tracks[0].video, tracks[0].masks, tracks[0].dm
    N = kf.N
    t_flow = sum(tr.pipe.t_flow for tr in tracks) / V
    t_ekf = sum(tr.pipe.t_ekf for tr in tracks) / V
    iters = sum(tr.pipe.iters for tr in tracks) / V
    predtime = sum(tr.kf.predtime for tr in tracks) / V
    updatetime = sum(tr.kf.updatetime for tr in tracks) / V
    # steady state: the last 60 % of the frames of this rank's first video (the first flow series of the timed
    # region has nothing to hide behind; `value` includes it, this figure leaves it out)
    done = tracks[0].pipe.frame_done
    steady = None
    if len(done) >= 5:
        skip = len(done) - max(2, int(round(0.6 * len(done))))
        steady = {"frames": len(done) - skip, "skipped": skip,
                  "value": world * V * (len(done) - skip) / (done[-1] - done[skip - 1]), "unit": "frames/sec",
                  "note": "frames %d..%d of the timed region of rank 0's first video, scaled by the number of videos" % (skip, len(done) - 1)}

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        assert gathered.shape[0] == world * V
        h2d = sum(tr.pipe.ring.bytes_uploaded for tr in tracks)
        out = {
            "metric": "frames/sec (Brox flow + EKF update) at 1024^2" if n == 1024 else
                      "frames/sec (Brox flow + EKF update) at %d^2" % n,
            "value": world * V * K / elapsed, "unit": "frames/sec", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (flow, renders) / f64 (EKF sums and state)", "data": "synthetic",
            "config": {"workload": "%dx%d video, %d-vertex mesh (%d triangles), %s per GPU; Brox defaults "
                                   "alpha .197 gamma 50 scale .8 inner 10 outer 77 solver 10; IteratedMSKalmanFilter "
                                   "defaults" % (n, n, N, kf.state.NT, "one video" if V == 1 else "%d concurrent videos" % V),
                       "frames_per_gpu": K * V, "videos_per_gpu": V, "flow_batch": B, "parallelism": "videos x%d" % (world * V),
                       "h2d": ("whole video resident before the clock" if args.resident else
                               "streamed inside the timed region: %d bytes (frame + mask per step, copy stream, "
                               "ring of %d slots)" % (h2d, tracks[0].pipe.ring.R))},
            "steady_state": steady,
            "distributed": dict(info, collective="gather of the tracked states (hydra_mi.batch.gather_to_root)"),
            "breakdown_ms_per_step": {"brox_flow": 1e3 * t_flow / K, "ekf_compute": 1e3 * t_ekf / K,
                                      "ekf_predict": 1e3 * predtime / K, "ekf_update": 1e3 * updatetime / K,
                                      "iekf_iterations": iters / K},
            "roofline": sor_roofline(sor_ms, sor_launches, sor_pxit, sor_px,
                                     pmc_traffic() if n == 1024 else None,
                                     "one flow series (%d pairs) %s, %s" % (prof_pairs, prof_where, "the filter's kernels running beside it"
                                                                             if prof_where == "of the timed region" else "nothing beside it"),
                                     levels, prof_pairs),
        }
        # ... and with the chip to itself (the figure the 40 % target of north_star is about)
        ra = sor_roofline(alone[0], alone[1], alone[2], alone[3], pmc_traffic() if n == 1024 else None,
                          "one flow series (%d pairs) after the timed region, nothing beside it" % alone_pairs, alone_levels, alone_pairs)
        out["roofline"]["alone"] = {k: ra[k] for k in ("achieved", "frac", "traffic", "traffic_source", "achieved_contract",
                                                         "frac_contract", "launches", "avg_launch_us", "profiled") if k in ra}
        if "finest_level" in ra:
            out["roofline"]["alone"]["finest_level"] = ra["finest_level"]
        out["cpu_baseline"] = None                           # a reported baseline, timed at N = 1 only
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, video, masks, dm, max(1.0, iters / K))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
