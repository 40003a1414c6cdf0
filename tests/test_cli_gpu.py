"""The two command-line tools end to end on a small synthetic video (reference CLIs:
src/optical_flow_ext.cpp:441-507 and run_kalmanfilter.py:9-93)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flow_tool_then_tracker(hm, tmp_path, oracle_brox):
    from hydra_mi import matio, synth
    sys.path.insert(0, ROOT)
    import optical_flow_ext
    import run_kalmanfilter
    n, F = 96, 5
    video, masks, c, r = synth.disk_video(n, F, "translate_leftup", 0)
    vid = str(tmp_path / "video.npy")
    np.save(vid, video)
    prefix = str(tmp_path / "flow")
    # positional parameters as in the reference: alpha gamma scale inner outer solver
    assert optical_flow_ext.main(["optical_flow_ext", vid, prefix, "0.197", "50", "0.8", "5", "77", "5"]) == 0
    for k in range(F - 1):
        fx = matio.read_mat(prefix + "_%03d_x.mat" % k)
        fy = matio.read_mat(prefix + "_%03d_y.mat" % k)
        ru, rv = oracle_brox.calc(video[k], video[k + 1], inner=5, solver=5)
        assert fx.dtype == np.float32 and np.array_equal(fx, ru) and np.array_equal(fy, rv)
    assert not os.path.exists(prefix + "_%03d_x.mat" % (F - 1))
    out = str(tmp_path / "states.npz")
    assert run_kalmanfilter.main([vid, prefix, out, "-s", "14", "-t", "9"]) == 0
    res = np.load(out)
    assert res["X"].shape[0] == F - 1 and np.all(np.isfinite(res["X"]))
    N = res["p"].shape[0]
    disp = res["X"][-1][:2 * N].reshape(-1, 2).mean(0) - res["p"].mean(0)
    true = np.array(synth.scaled_field("translate_leftup", n)(0.0, 0.0)) * (F - 1)
    assert np.linalg.norm(disp - true) < 1.0           # the mesh follows the object
    # without flow files the tracker computes the flow itself
    out2 = str(tmp_path / "states2.npz")
    assert run_kalmanfilter.main([vid, str(tmp_path / "nothing"), out2, "-s", "14"]) == 0
    res2 = np.load(out2)
    assert res2["X"].shape == res["X"].shape and np.all(np.isfinite(res2["X"]))
    assert np.abs(res2["X"][-1][:2 * N] - res["X"][-1][:2 * N]).max() < 1.5   # flows with 10/10 vs 5/5 iterations


def test_native_flow_tool_writes_the_files_of_the_python_tool(hm, tmp_path):
    """The reference's flow tool is a C++ program (src/optical_flow_ext.cpp:333-507); csrc/optical_flow_ext.cpp is its
    counterpart over the C-ABI (hm_brox_create / hm_brox_calc_batch): same positional arguments, same .mat files --
    byte for byte those of optical_flow_ext.py, for a gray and for a BGR video, with non-default parameters and a series
    size that does not divide the number of pairs."""
    import subprocess
    from hydra_mi import synth
    sys.path.insert(0, ROOT)
    import optical_flow_ext
    exe = os.path.join(ROOT, "kalman-hydra_amd", "optical_flow_ext")
    n, F = 80, 6
    video, _, _, _ = synth.disk_video(n, F, "rotate", 3)
    rng = np.random.default_rng(0)
    colour = np.clip(video[..., None].astype(np.int32) + rng.integers(-20, 21, video.shape + (3,)), 0, 255).astype(np.uint8)
    for name, arr, extra in (("gray", video, []), ("bgr", colour, ["0.3", "40", "0.75", "4", "50", "6"])):
        fn = str(tmp_path / (name + ".npy"))
        np.save(fn, arr)
        assert optical_flow_ext.main(["optical_flow_ext.py", fn, str(tmp_path / (name + "_py"))] + extra) == 0
        env = dict(os.environ, HYDRA_MI_FLOW_BATCH="2")
        r = subprocess.run([exe, fn, str(tmp_path / (name + "_cc"))] + extra, capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0 and "Finished." in r.stdout, r.stderr
        for k in range(F - 1):
            for c in "xy":
                a = open(str(tmp_path / ("%s_py_%03d_%s.mat" % (name, k, c))), "rb").read()
                b = open(str(tmp_path / ("%s_cc_%03d_%s.mat" % (name, k, c))), "rb").read()
                assert a == b and len(a) == 12 + 4 * n * n, (name, k, c)
        assert not os.path.exists(str(tmp_path / ("%s_cc_%03d_x.mat" % (name, F - 1))))


def test_pipeline_equals_sequential_calls(hm):
    """hydra_mi.pipeline.FlowEKFPipeline (flow series of 1, 2, 4 pairs on the flow handle's stream, overlapped
    with the filter, flow handed over in device memory) gives bit for bit the states of the sequential
    loop bf.calc(frame k, frame k+1) -> kf.compute (reference run_kalmanfilter.py:78-89 with the flow of
    README.md:26-31 computed in process)."""
    from hydra_mi import brox, kalman, mesh, synth
    from hydra_mi.pipeline import FlowEKFPipeline
    n, F = 96, 7
    video, masks, c, r = synth.disk_video(n, F, "warp", 1)
    zero = np.zeros((n, n, 2), np.float32)

    def new_filter():
        return kalman.IteratedMSKalmanFilter(mesh.disk_mesh(c[0], c[1], r - 1.0, 14.0), video[0], zero, True, nI=4)

    kf_a = new_filter()
    bf = brox.BroxOpticalFlow(n, n)
    seq, seq_err, flows = [], [], []
    for k in range(F - 1):
        u, v = bf.calc(video[k], video[k + 1])
        flows.append((u, v))
        e = kf_a.compute(video[k + 1], np.dstack((u, v)), masks[k + 1])
        seq.append(kf_a.state.X.copy())
        seq_err.append(e[:4])
    kf_b = new_filter()
    pipe = FlowEKFPipeline(kf_b, video, masks, flow_batch=4)
    got = []
    pipe.run(on_frame=lambda k, e: got.append((k, kf_b.state.X.copy(), e[:4])))
    assert [g[0] for g in got] == list(range(F - 1))
    for k, X, e in got:
        assert np.array_equal(X, seq[k]), k
        assert e[0] == seq_err[k][0] and e[3] == seq_err[k][3] and e[1] == seq_err[k][1] and e[2] == seq_err[k][2]
    assert np.array_equal(kf_a.state.W, kf_b.state.W)
    # random access restarts the series; the flow of a pair does not depend on the series it is computed in
    f3 = pipe.flow_host(3)
    assert np.array_equal(f3[:, :, 0], flows[3][0]) and np.array_equal(f3[:, :, 1], flows[3][1])
    f0 = pipe.flow_host(0)
    assert np.array_equal(f0[:, :, 0], flows[0][0]) and np.array_equal(f0[:, :, 1], flows[0][1])
    with pytest.raises(IndexError):
        pipe.step(F - 1)
    pipe.close()


def test_two_series_in_flight_at_the_edges(hm):
    """The default mode (two flow series in flight on two handles) where there is hardly room for it: series of one pair,
    a video of two and of three frames, several phases through one pipeline with the series sized from the measurements
    of the earlier ones, random access in between -- the flow of every pair the bits of a single call, nothing hangs."""
    from hydra_mi import brox, kalman, mesh, synth
    from hydra_mi.pipeline import FlowEKFPipeline
    n, F = 64, 9
    video, masks, c, r = synth.disk_video(n, F, "rotate", 2)
    zero = np.zeros((n, n, 2), np.float32)
    bf = brox.BroxOpticalFlow(n, n)
    flows = [bf.calc(video[k], video[k + 1]) for k in range(F - 1)]

    def new_filter():
        return kalman.IteratedMSKalmanFilter(mesh.disk_mesh(c[0], c[1], r - 1.0, 12.0), video[0], zero, True, nI=2)

    def same(pipe, k):
        f = pipe.flow_host(k)
        return np.array_equal(f[:, :, 0], flows[k][0]) and np.array_equal(f[:, :, 1], flows[k][1])

    for B in (1, 2, 3):
        for frames in (2, 3, F):
            with FlowEKFPipeline(new_filter(), video[:frames], masks[:frames], flow_batch=B) as pipe:
                assert pipe.concurrent_series and len(pipe.bfs) == 2
                seen = []
                pipe.run(on_frame=lambda k, e: seen.append(k))
                assert seen == list(range(frames - 1)), (B, frames)
                assert all(same(pipe, k) for k in range(frames - 1)), (B, frames)
    with FlowEKFPipeline(new_filter(), video, masks, flow_batch=3) as pipe:
        pipe.run(0, 2)                                  # a first phase: calibrates, measures a frame of the filter
        assert pipe._series_s and pipe._frame_s is not None
        pipe.run(2, 5)                                  # the second one sizes its series from that (_next_concurrent)
        assert same(pipe, 7) and same(pipe, 1)          # random access, backwards too
        pipe.run(5, F - 1)
        assert pipe.kf.state.X.shape[0] > 0
        assert all(same(pipe, k) for k in range(F - 1))


def test_streaming_frame_ring_equals_resident_video(hm):
    """the frame loop reads one frame per iteration (reference run_kalmanfilter.py:78-89).  The
    pipeline's default -- frames and masks read from the source one by one and uploaded over a copy stream into a ring
    of 3 B + 3 frame slots, the frames of the next flow series going up while the current one runs -- gives the bits
    of the whole video uploaded at once (resident=True), through several laps of the ring (16 frames, 9 slots), and a
    VideoStream source (mask and background-subtracted frame derived per frame) the bits of the arrays it stands for."""
    from hydra_mi import kalman, mesh, synth
    from hydra_mi.pipeline import FlowEKFPipeline, VideoStream, threshold_mask
    n, F = 96, 16
    video, _, c, r = synth.disk_video(n, F, "warp", 2)
    masks = np.stack([threshold_mask(f, 9) for f in video])
    zero = np.zeros((n, n, 2), np.float32)

    def new_filter():
        return kalman.IteratedMSKalmanFilter(mesh.disk_mesh(c[0], c[1], r - 1.0, 14.0), video[0] * masks[0], zero, True, nI=3)

    def track(make):
        kf = new_filter()
        pipe = make(kf)
        got = []
        pipe.run(on_frame=lambda k, e: got.append((kf.state.X.copy(), e[:4], kf.niter)))
        return pipe, kf, got

    pr, kfr, ref = track(lambda kf: FlowEKFPipeline(kf, video, masks, observed=video * masks, flow_batch=2, resident=True))
    assert pr.ring.R == F and pr.ring.bytes_uploaded == 3 * F * n * n
    ps, kfs, got = track(lambda kf: FlowEKFPipeline(kf, video, masks, observed=video * masks, flow_batch=2))
    assert ps.ring.R == 9 and ps.ring.extra == 2 and not ps.resident
    assert ps.ring.bytes_uploaded == 3 * F * n * n            # every frame, mask and observed frame exactly once
    pv, kfv, gotv = track(lambda kf: FlowEKFPipeline(kf, VideoStream(video, 9), flow_batch=2))
    assert len(ref) == len(got) == len(gotv) == F - 1
    for k in range(F - 1):
        for g in (got[k], gotv[k]):
            assert np.array_equal(g[0], ref[k][0]) and g[1] == ref[k][1] and g[2] == ref[k][2], k
    assert np.array_equal(kfs.state.W, kfr.state.W)
    # a new phase uploads its frames again (nothing of an earlier phase is assumed to be in the ring), random access
    # and steps past the end of the phase run() announced work
    f5 = ps.flow_host(5)
    r5 = pr.flow_host(5)
    assert np.array_equal(f5, r5)
    kf2 = new_filter()
    p2 = FlowEKFPipeline(kf2, video, masks, observed=video * masks, flow_batch=2)
    p2.run(0, 3)
    for k in (3, 4, 5):
        p2.step(k)                                            # past the end of the phase: goes on pair by pair
    assert np.array_equal(kf2.state.X, ref[5][0])
    with pytest.raises(IndexError):
        p2.flow_ready(F - 1)
    # with split_start a phase that follows a measured one starts with two series side by side on two handles (a short
    # one so that the filter starts soon, the one sized from the measurements behind it): the same bits
    kf3 = new_filter()
    p3 = FlowEKFPipeline(kf3, video, masks, observed=video * masks, flow_batch=4, concurrent_series=False)
    p3.split_start = True                                     # (off by default: measured slower, pipeline.py)
    got3 = []
    p3.run(0, 2, on_frame=lambda k, e: got3.append((kf3.state.X.copy(), e[:4], kf3.niter)))
    assert len(p3.bfs) == 1 and p3._series_s and p3._frame_s is not None
    p3.run(2, F - 1, on_frame=lambda k, e: got3.append((kf3.state.X.copy(), e[:4], kf3.niter)))
    assert len(p3.bfs) == 2                                   # the second handle was needed
    assert len(got3) == F - 1
    for k in range(F - 1):
        assert np.array_equal(got3[k][0], ref[k][0]) and got3[k][1] == ref[k][1] and got3[k][2] == ref[k][2], k
    # one series in flight at a time (the default is two, on two handles): the same bits
    p4, kf4, got4 = track(lambda kf: FlowEKFPipeline(kf, video, masks, observed=video * masks, flow_batch=3, concurrent_series=False))
    assert len(p4.bfs) == 1 and len(ps.bfs) == 2
    for k in range(F - 1):
        assert np.array_equal(got4[k][0], ref[k][0]) and got4[k][1] == ref[k][1] and got4[k][2] == ref[k][2], k
    for p in (pr, ps, pv, p2, p3, p4):
        p.close()


def _short_track(hm, tune=(), attrs=(), F=8, n=128):
    """The whole frame loop on a small video -- flow series beside the filter, state prediction as a launch started by the
    update, covariance queued ahead, speculative measurement, predict -> projectmask -> update chained on the device --
    with hm_ctx_tune knobs / filter attributes set; -> (per frame: state, iterations, Newton iterations, the four error
    terms, predicted state, projected state), final covariance, seconds."""
    import time
    from hydra_mi import kalman, mesh, synth
    from hydra_mi.pipeline import FlowEKFPipeline, threshold_mask
    video, _, c, r = synth.disk_video(n, F, "translate_leftup", 1)
    masks = np.stack([threshold_mask(f, 9) for f in video])
    masks[3:, 2:6, 2:9] = 1                                     # (a second object far from the mesh from frame 3 on)
    zero = np.zeros((n, n, 2), np.float32)
    kf = kalman.IteratedMSKalmanFilter(mesh.disk_mesh(c[0] + 2.5, c[1], r + 1.5, 12.0), video[0], zero, True)
    for key, value in tune:
        kf.state.renderer.tune(key, value)
    for key, value in attrs:
        setattr(kf, key, value)
    out = []
    with FlowEKFPipeline(kf, video, masks, flow_batch=4) as pipe:
        t0 = time.perf_counter()
        pipe.run(on_frame=lambda k, e: out.append((kf.state.X.copy(), kf.niter, kf.newton_iterations, tuple(e[:4]),
                                                   np.array(kf.pred_x).reshape(-1), np.array(getattr(kf, "proj_x", kf.pred_x)).reshape(-1))))
        dt = time.perf_counter() - t0
        W = np.array(kf.state.W)
    kf.close()
    return out, W, dt


def _same_track(a, b, upto=4):
    assert len(a[0]) == len(b[0])
    for k, (x, y) in enumerate(zip(a[0], b[0])):
        for i in range(upto):
            assert np.array_equal(x[i], y[i]) if isinstance(x[i], np.ndarray) else x[i] == y[i], (k, i)
    assert np.array_equal(a[1], b[1])


def test_result_blocks_do_not_depend_on_the_order_their_words_arrive_in(hm):
    """Result blocks in page-locked host memory (csrc/host_block.h: the IEKF iterations' step + error sums that drive
    reference kalman.py:792-822, the gains, the state prediction, projectmask).  Round 3 trusted a block once its ticket
    was visible and one track in a hundred took half of a block from the launch before.  Now every word carries its
    launch's stamp and the host takes a block when all of it is there -- so a kernel that publishes the block's LAST word
    first and everything else 1000 us later (hm_ctx_tune "result_delay") must give exactly the track of an undelayed
    run, only slower.  One run, deterministic: the delayed path is taken for every block of every frame."""
    ref = _short_track(hm)
    assert all(o[1] >= 1 for o in ref[0])
    late = _short_track(hm, tune=[("result_delay", 1000)])
    _same_track(ref, late, upto=6)
    blocks = sum(o[1] + 1 for o in ref[0])                        # one per iteration, one with the gains per frame
    assert late[2] - ref[2] >= 0.5 * blocks * 1000e-6, (ref[2], late[2], blocks)        # (the knob was in effect)


def test_chained_state_path_equals_the_three_calls(hm):
    """compute() = predict -> projectmask -> update (reference kalman.py:676-700).  Chained on the device
    (hm_chain_project: the projection queued behind the prediction's kernel, the update starting from the projected state
    in device memory) it gives the bits of the three separate calls -- states, iteration counts, error terms, the
    predicted and the projected state (vertices start outside the object here, so projectmask does move some) -- and so
    does the update's tail on the handle's own stream instead of a second one; a device prediction that reports a failed
    inner solve (test knob) is repeated on the host: the track of a filter that predicts on the host."""
    chained = _short_track(hm)
    three = _short_track(hm, attrs=[("chain", False)])
    _same_track(chained, three, upto=6)
    assert any(not np.array_equal(o[4], o[5]) for o in chained[0])        # projectmask moved something
    one_stream = _short_track(hm, tune=[("tail_split", 0)])
    _same_track(chained, one_stream, upto=6)
    failed = _short_track(hm, tune=[("newton_fail", 1)])
    host = _short_track(hm, attrs=[("newton_on_device", False)])
    _same_track(failed, host, upto=6)
