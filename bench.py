"""bench.py -- frames/sec of the hot path (Brox flow + EKF update) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one video frame: Brox optical flow of the frame pair (k, k+1), both frames
already resident in HBM, followed by IteratedMSKalmanFilter.compute on frame k+1 with
that flow (predict, mask projection, iterated measurement update), flow handed over in
device memory.  Workload at every N: BASELINE.json config "1024x1024 video, ~200-vertex
mesh" -- a textured disk advected by an analytic field, synthetic, one independent video
per GPU (weak scaling: the EKF is a recurrence over the frames of one video, videos are
the shardable unit; the only collective is the final gather of the tracked states).

Prints ONE JSON line (rank 0).  `roofline` is the SOR kernel: kernel start/stop events of every
SOR launch of one full flow series inside the timed region.  `achieved` / `frac` follow the contract's
figure, 52 B per pixel per red-black iteration -- but k_sor runs K iterations per pass
over memory, so that figure is not a bound; the numbers that are:  `min_bytes_per_launch` = 52 B/px per
LAUNCH (every field crosses HBM once per launch at best), `frac_min_traffic` = that over time over
8 TB/s (cannot exceed 1), and `traffic` / `frac_hbm` = HBM bytes per launch from the rocprofv3 PMC passes
(FETCH_SIZE x 2 + WRITE_SIZE, profiles/r02_sor_pmc.json, null when that file was taken for another
kernel revision).  `cpu_baseline` is the oracle (C/OpenMP restatement of the reference's CPU path)
timed on this host: whole Brox pairs and one whole KFState.update at the bench's size.
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


SOR_PMC = os.path.join(ROOT, "profiles", "r02_sor_pmc.json")


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
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes and as the k_add calibration in that file
    confirms).  None unless the file was taken for the kernel sources as they are now."""
    try:
        with open(SOR_PMC) as f:
            d = json.load(f)
        if d.get("kernel_revision") != kernel_revision():
            return None
        return float(d["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def sor_roofline(sor_ms, sor_launches, sor_pxit, sor_px, traffic, profiled):
    """The roofline block of the JSON line.  sor_pxit = sum over the timed launches of pixels x fused
    iterations, sor_px = sum of pixels (one pass over memory each)."""
    t = sor_ms * 1e-3
    alg = SOR_BYTES_PER_PIXEL_ITERATION * sor_pxit
    achieved = alg / t / 1e9 if t > 0 else 0.0
    minb = SOR_BYTES_PER_PIXEL_ITERATION * sor_px
    n = max(1, sor_launches)
    out = {"bound": "hbm", "kernel": "k_sor", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
           "algorithmic_bytes_per_launch": alg / n, "bytes_per_pixel_iteration": SOR_BYTES_PER_PIXEL_ITERATION,
           "min_bytes_per_launch": minb / n,
           "frac_min_traffic": (minb / t / 1e9 / HBM_PEAK_GBPS) if t > 0 else 0.0,
           "frac_hbm": (traffic * n / t / 1e9 / HBM_PEAK_GBPS) if (traffic and t > 0) else None,
           "launches": sor_launches, "avg_launch_us": 1e3 * sor_ms / n, "profiled": profiled,
           "note": "frac follows the contract's per-iteration byte model and exceeds 1 because k_sor fuses several "
                   "red-black iterations per pass over memory; frac_min_traffic (52 B/px per launch) and frac_hbm "
                   "(PMC bytes) are physical bounds; profiles/r02_sor_sq_counters.csv: the kernel is bound by vector "
                   "instruction issue and barrier waits, not by HBM"}
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


def flowbatch(args, rank, world, dev, coll_dev):
    """BASELINE config 5: `pairs-per-gpu` independent 1024^2 pairs per rank (seeds differ per pair),
    one step = the flow of all of them in launch series of --flow-batch pairs; no communication while
    computing, one gather of all flow planes to rank 0 at the end of the timed region."""
    import torch
    import torch.distributed as dist
    from hydra_mi import brox, synth, batch
    n, P, B = args.size, args.pairs_per_gpu, max(1, args.flow_batch)
    mine = batch.shard(P * world, rank, world)
    base = [synth.warp_pair(n, "translate_leftup_stretch", seed)[:2] for seed in range(4)]   # 4 distinct pairs, cycled
    F0 = torch.from_numpy(np.stack([base[i % 4][0] for i in mine])).cuda()
    F1 = torch.from_numpy(np.stack([base[i % 4][1] for i in mine])).cuda()
    U = torch.empty((P, n, n), dtype=torch.float32, device="cuda")
    V = torch.empty_like(U)
    torch.cuda.synchronize()
    bf = brox.BroxOpticalFlow(n, n, max_batch=B, device=dev)

    def step():
        for s in range(0, P, B):
            nb = min(B, P - s)
            bf.calc_dev(nb, F0[s].data_ptr(), F1[s].data_ptr(), U[s].data_ptr(), V[s].data_ptr())
        bf.sync()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k == args.steps - 1:
            bf.profile(True)
        step()
    sor_ms, sor_launches, sor_pxit, sor_px = bf.profile_read()
    if world > 1:
        flows = torch.stack((U, V)).to(coll_dev)
        parts = [torch.empty_like(flows) for _ in range(world)] if rank == 0 else None
        dist.gather(flows, parts, dst=0)
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "frame pairs/sec (Brox flow) at %d^2" % n, "value": world * P * args.steps / elapsed,
            "unit": "pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d independent %dx%d frame pairs per GPU, Brox defaults, flows gathered to rank 0"
                                   % (P, n, n), "flow_batch": B, "parallelism": "pairs x%d" % world},
            "roofline": sor_roofline(sor_ms, sor_launches, sor_pxit, sor_px,
                                     pmc_traffic() if (n == 1024 and B == 8) else None,
                                     "last step of the timed region")}), flush=True)
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
    ap.add_argument("--two-flow-handles", action="store_true", help="two flow series in flight (default: one at a time)")
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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    import hydra_mi  # noqa: F401
    from hydra_mi import kalman, mesh

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

    class Track:
        """One video: a filter and the package's streaming pipeline (hydra_mi.pipeline.FlowEKFPipeline: frames
        and flow planes in HBM, the flow of the next frames computed on the flow handle's stream while the
        filter works on the current ones)."""

        def __init__(self, seed):
            frames = K + Wm + 1
            self.video, self.masks, centre, radius = make_video(n, frames, seed=seed)
            self.dm = mesh.disk_mesh(centre[0], centre[1], radius - 1.0, args.h0 * n)
            self.kf = kalman.IteratedMSKalmanFilter(self.dm, self.video[0], np.zeros((n, n, 2), np.float32), True, device=dev)
            extra = {} if args.cu_reserve is None else {"cu_reserve": args.cu_reserve}
            if args.two_flow_handles:
                extra["concurrent_series"] = True
            self.pipe = FlowEKFPipeline(self.kf, self.video, self.masks, flow_batch=B, device=dev, **extra)
            self.bf = self.pipe.bf
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
            self.kf.predtime = self.kf.updatetime = self.kf.projecttime = 0.0

        def timed(self):
            self.pipe.run(Wm, Wm + K)

        def flow_sync(self):
            self.pipe.flow_sync()

    # the videos of this rank: seeds rank * V .. rank * V + V - 1 (one video per GPU unless --videos-per-gpu)
    tracks = [Track(rank * V + i) for i in range(V)]
    for tr in tracks:
        tr.warmup()
    # the flow series of the timed region start small and grow to B pairs (pipeline.py: sized from what series and
    # frames have taken so far); the first one of B pairs is profiled (kernel start/stop events for every SOR launch)
    tracks[0].pipe.profile_full = True

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
    state = torch.from_numpy(np.concatenate([tr.kf.state.X.reshape(-1) for tr in tracks])).to(coll_dev)
    if world > 1:                       # the batch path's only exchange: gather the tracked states
        gathered = [torch.empty_like(state) for _ in range(world)]
        dist.all_gather(gathered, state)
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for tr in tracks:
        tr.flow_sync()
    prof_pairs, prof_where = tracks[0].pipe.profiled_pairs, "of the timed region"
    if prof_pairs == 0:                 # a timed region too short for a series of B pairs: one more, after the clock
        tr = tracks[0]
        prof_pairs, prof_where = min(B, len(tr.video) - 1), "after the timed region"
        tr.bf.profile(True)
        tr.bf.calc_dev(prof_pairs, tr.pipe.d_video.ptr, tr.pipe.d_video.ptr + n * n, tr.pipe.d_u.ptr, tr.pipe.d_v.ptr)
        tr.bf.sync()
    sor_ms, sor_launches, sor_pxit, sor_px = (tracks[0].pipe.profiled_handle or tracks[0].bf).profile_read()
    kf, video, masks, dm = tracks[0].kf, tracks[0].video, tracks[0].masks, tracks[0].dm
    N = kf.N
    t_flow = sum(tr.pipe.t_flow for tr in tracks) / V
    t_ekf = sum(tr.pipe.t_ekf for tr in tracks) / V
    iters = sum(tr.pipe.iters for tr in tracks) / V
    predtime = sum(tr.kf.predtime for tr in tracks) / V
    updatetime = sum(tr.kf.updatetime for tr in tracks) / V

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        out = {
            "metric": "frames/sec (Brox flow + EKF update) at 1024^2" if n == 1024 else
                      "frames/sec (Brox flow + EKF update) at %d^2" % n,
            "value": world * V * K / elapsed, "unit": "frames/sec", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (flow, renders) / f64 (EKF sums and state)", "data": "synthetic",
            "config": {"workload": "%dx%d video, %d-vertex mesh (%d triangles), %s per GPU; Brox defaults "
                                   "alpha .197 gamma 50 scale .8 inner 10 outer 77 solver 10; IteratedMSKalmanFilter "
                                   "defaults" % (n, n, N, kf.state.NT, "one video" if V == 1 else "%d concurrent videos" % V),
                       "frames_per_gpu": K * V, "videos_per_gpu": V, "flow_batch": B, "parallelism": "videos x%d" % (world * V)},
            "breakdown_ms_per_step": {"brox_flow": 1e3 * t_flow / K, "ekf_compute": 1e3 * t_ekf / K,
                                      "ekf_predict": 1e3 * predtime / K, "ekf_update": 1e3 * updatetime / K,
                                      "iekf_iterations": iters / K},
            "roofline": sor_roofline(sor_ms, sor_launches, sor_pxit, sor_px,
                                     pmc_traffic() if (n == 1024 and prof_pairs == 8) else None,
                                     "one flow series (%d pairs) %s" % (prof_pairs, prof_where)),
        }
        out["cpu_baseline"] = None                           # a reported baseline, timed at N = 1 only
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, video, masks, dm, max(1.0, iters / K))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
