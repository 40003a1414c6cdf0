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

Prints ONE JSON line (rank 0).  `roofline` is the SOR kernel: HIP-event time of every
SOR launch inside the timed region against 52 B per pixel per red-black iteration
(SURVEY.md 8d).  `cpu_baseline` is the oracle (NumPy/C restatement of the reference's
CPU path) timed on this host on a bounded sample.
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


def pmc_traffic():
    """HBM bytes per SOR launch from the committed PMC passes (profiles/r01_sor_pmc.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the same 8-pair 1024^2 flow batch, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes and as the k_add calibration in that file confirms)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_sor_pmc.json")) as f:
            return float(json.load(f)["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def make_video(n, frames, seed):
    from hydra_mi import synth
    video, masks, centre, radius = synth.disk_video(n, frames, "translate_leftup", seed)
    return video, masks, centre, radius


def cpu_baseline(n, video, masks, dm, iters_per_frame, budget_s=25.0):
    """Oracle on the host cores: one Brox pair + a timed sample of EKF perturbation evaluations,
    scaled to the evaluations one frame needs (2*4N jz + nzj j per IEKF iteration, kalman.py:397,498-515,595-598)."""
    from oracle import brox_oracle, ekf_ref
    threads = max(1, min(16, os.cpu_count() or 1))
    brox_oracle.set_threads(threads)
    npairs = min(3, len(video) - 1)
    t0 = time.perf_counter()
    for k in range(npairs):
        u, v = brox_oracle.calc(video[k], video[k + 1])
    t_brox = (time.perf_counter() - t0) / npairs
    u, v = brox_oracle.calc(video[0], video[1])
    brox_oracle.set_threads(1)
    N = dm.size()
    meas = ekf_ref.Measurement(N, dm.t, dm.p, video[0], 1e-3, 1.0, 1.0)
    X = np.concatenate((dm.p.reshape(-1), np.zeros(2 * N)))
    flow = np.dstack((u, v)).astype(np.float32)
    t0 = time.perf_counter()
    meas.initjacobian(X, video[1], flow, masks[1])
    t_init = time.perf_counter() - t0
    n_jz = n_j = 0
    t_jz = t_j = 0.0
    k = 0
    while t_jz + t_j < budget_s and k < 4 * N:
        Xp = X.copy()
        Xp[k] += 2.0
        t0 = time.perf_counter(); meas.jz(Xp); t_jz += time.perf_counter() - t0; n_jz += 1
        t0 = time.perf_counter(); meas.j(2.0, k, k); t_j += time.perf_counter() - t0; n_j += 1
        k += max(1, (4 * N) // 96)
    _, J = ekf_ref.adjacency(N, dm.t)
    nzj = float(np.sum(np.triu(J)))
    per_iter = t_init + 2 * 4 * N * (t_jz / n_jz) + nzj * (t_j / n_j)
    t_frame = t_brox + iters_per_frame * per_iter
    return {"value": 1.0 / t_frame, "unit": "frames/sec", "cores": threads, "kind": "port",
            "sample": "oracle Brox on %d pairs (%d OpenMP threads, %.2f s each) + %d jz and %d j evaluations of the NumPy "
                      "EKF twin (1 thread) scaled to 2*4N=%d jz + %d j per IEKF iteration x %.1f iterations/frame"
                      % (npairs, threads, t_brox, n_jz, n_j, 8 * N, int(nzj), iters_per_frame)}


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
    bf.tune("sor_threads", 512)

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
    sor_ms, sor_launches, sor_pxit = bf.profile_read()
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
        achieved = SOR_BYTES_PER_PIXEL_ITERATION * sor_pxit / (sor_ms * 1e-3) / 1e9 if sor_ms > 0 else 0.0
        print(json.dumps({
            "metric": "frame pairs/sec (Brox flow) at %d^2" % n, "value": world * P * args.steps / elapsed,
            "unit": "pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d independent %dx%d frame pairs per GPU, Brox defaults, flows gathered to rank 0"
                                   % (P, n, n), "flow_batch": B, "parallelism": "pairs x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_sor", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic() if (n == 1024 and B == 8) else None,
                         "launches": sor_launches, "profiled": "last step of the timed region"}}), flush=True)
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
    from hydra_mi import brox, kalman, mesh
    from hydra_mi.renderer import DeviceObservation

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
    B = max(1, args.flow_batch)
    V = max(1, args.videos_per_gpu)

    class Track:
        """One video: its frames and flow planes in HBM, a flow handle, a filter, and the schedule that
        computes the flow of the next frames while the filter works on the current ones."""

        def __init__(self, seed):
            frames = K + Wm + 1
            self.video, self.masks, centre, radius = make_video(n, frames, seed=seed)
            self.dm = mesh.disk_mesh(centre[0], centre[1], radius - 1.0, args.h0 * n)
            self.d_video = torch.from_numpy(self.video).cuda()
            self.d_masks = torch.from_numpy(self.masks).cuda()
            self.d_u = torch.empty((2, B, n, n), dtype=torch.float32, device="cuda")      # double-buffered flow planes
            self.d_v = torch.empty_like(self.d_u)
            torch.cuda.synchronize()
            self.bf = brox.BroxOpticalFlow(n, n, max_batch=B, device=dev)
            self.bf.tune("sor_threads", 512)
            self.kf = kalman.IteratedMSKalmanFilter(self.dm, self.video[0], np.zeros((n, n, 2), np.float32), True, device=dev)
            self.t_flow = self.t_ekf = 0.0
            self.iters = 0
            # frame pairs [lo, hi) of `ready` have their flow in buffer `buf`; `pending` is being computed
            # on the flow handle's own stream while the filter works on `ready`
            self.sched = {"ready": (0, 0), "buf": 0, "pending": None, "profile_from": None, "thread": None}

        def launch(self, k, phase_end, buf, most=None):
            """Queue the flow of the pairs [k, k+nb) on the flow handle's stream.  The calls are made from a
            helper thread (ctypes drops the GIL): a series is ~900 launches (the profiled one with an event
            pair for every SOR launch on top), several milliseconds of host time that the filter's thread
            does not have to spend."""
            nb = min(B if most is None else most, phase_end - k)
            bf, sched = self.bf, self.sched

            def work():
                if sched["profile_from"] == k:
                    bf.profile(True)
                elif sched["profile_from"] is not None and k > sched["profile_from"]:
                    bf.profile(False)              # totals stay readable (hm_brox_profile_read)
                bf.calc_dev(nb, self.d_video[k].data_ptr(), self.d_video[k + 1].data_ptr(), self.d_u[buf].data_ptr(),
                            self.d_v[buf].data_ptr())
            th = threading.Thread(target=work)
            th.start()
            sched["thread"] = th
            return (k, k + nb)

        def flow_sync(self):
            if self.sched["thread"] is not None:
                self.sched["thread"].join()
                self.sched["thread"] = None
            self.bf.sync()

        def step(self, k, phase_end):
            """Frame k+1: flow of (k, k+1) -- computed for up to B consecutive pairs per launch series (they do
            not depend on the filter), the next series running on the GPU while the filter works through
            this one (series of 1, 2, 4, ... pairs at the start of a phase: nothing to overlap the first with) --
            then the EKF on frame k+1."""
            sched = self.sched
            t0 = time.perf_counter()
            if k >= sched["ready"][1]:
                if sched["pending"] is not None and sched["pending"][0] == k:
                    sched["buf"] ^= 1
                else:                      # start of a phase: one pair only, so that the filter can start
                    sched["pending"] = self.launch(k, phase_end, sched["buf"], most=1)
                self.flow_sync()
                sched["ready"], sched["pending"] = sched["pending"], None
                lo, nxt = sched["ready"]
                if nxt < phase_end:        # ramp: what the GPU gets done beside the frames just made ready
                    sched["pending"] = self.launch(nxt, phase_end, sched["buf"] ^ 1, most=min(B, 2 * (nxt - lo)))
            i = k - sched["ready"][0]
            cur = sched["buf"]
            t1 = time.perf_counter()
            obs = DeviceObservation(self.d_video[k + 1].data_ptr(), self.d_u[cur, i].data_ptr(), self.d_v[cur, i].data_ptr(),
                                    self.d_masks[k + 1].data_ptr(), y_m_host=self.masks[k + 1])
            self.kf.compute(obs, None, None)
            t2 = time.perf_counter()
            self.t_flow += t1 - t0
            self.t_ekf += t2 - t1
            if os.environ.get("HYDRA_MI_BENCH_TRACE"):
                print("step %d: flow wait %.2f ms, filter %.2f ms (%d iterations), series ready %s pending %s"
                      % (k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), self.kf.niter, sched["ready"], sched["pending"]), file=sys.stderr)
            self.iters += self.kf.niter

        def warmup(self):
            self.bf.profile(True)
            for k in range(Wm):
                self.step(k, Wm)
            self.bf.profile_read()
            self.bf.profile(False)
            self.t_flow = self.t_ekf = 0.0
            self.iters = 0
            self.kf.predtime = self.kf.updatetime = self.kf.projecttime = 0.0

        def timed(self):
            for k in range(Wm, Wm + K):
                self.step(k, Wm + K)

    # the videos of this rank: seeds rank * V .. rank * V + V - 1 (one video per GPU unless --videos-per-gpu)
    tracks = [Track(rank * V + i) for i in range(V)]
    for tr in tracks:
        tr.warmup()
    # series of the timed region: 1 pair, then 2, 4, ... up to B at a time (a series of n pairs takes
    # about 7.5 + 0.65 (n - 1) ms here, a frame of the filter about 8 ms); one of them is profiled (kernel
    # start/stop events for every SOR launch): the first full one, or the last if there is no full one
    starts, k_ = [], Wm
    while k_ < Wm + K:
        size = 1 if not starts else min(B, 2 * starts[-1][1], Wm + K - k_)
        starts.append((k_, size))
        k_ += size
    full = [st for st in starts if st[1] == B]
    tracks[0].sched["profile_from"], prof_pairs = full[0] if full else starts[-1]

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
    sor_ms, sor_launches, sor_pxit = tracks[0].bf.profile_read()
    kf, video, masks, dm = tracks[0].kf, tracks[0].video, tracks[0].masks, tracks[0].dm
    N = kf.N
    t_flow = sum(tr.t_flow for tr in tracks) / V
    t_ekf = sum(tr.t_ekf for tr in tracks) / V
    iters = sum(tr.iters for tr in tracks) / V
    predtime = sum(tr.kf.predtime for tr in tracks) / V
    updatetime = sum(tr.kf.updatetime for tr in tracks) / V

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        achieved = SOR_BYTES_PER_PIXEL_ITERATION * sor_pxit / (sor_ms * 1e-3) / 1e9 if sor_ms > 0 else 0.0
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
            "roofline": {"bound": "hbm", "kernel": "k_sor", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic() if (n == 1024 and prof_pairs == 8) else None,
                         "algorithmic_bytes_per_launch": SOR_BYTES_PER_PIXEL_ITERATION * sor_pxit / max(1, sor_launches),
                         "launches": sor_launches, "profiled": "one flow series (%d pairs) of the timed region" % prof_pairs, "avg_launch_us": 1e3 * sor_ms / max(1, sor_launches),
                         "bytes_per_pixel_iteration": SOR_BYTES_PER_PIXEL_ITERATION},
        }
        out["cpu_baseline"] = None                           # a reported baseline, timed at N = 1 only
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, video, masks, dm, max(1.0, iters / K))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
