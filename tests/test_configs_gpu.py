"""The BASELINE.json configurations at their stated sizes, through the C-ABI (`pytest -m gpu`):

  config 2  Brox flow only, a single 512x512 synthetic warp pair -- all four analytic fields of the
            reference (synthetic/flowfields.py:3-7) against the C oracle, plus the analytic-field
            protocol of reference test_flow.py:120-138 with per-field bounds (measured + 10 %);
  config 3  full EKF mesh track on a 512x512 synthetic video with a ~40-vertex mesh: flow from the
            product's Brox, IteratedMSKalmanFilter.compute frame by frame, against the oracle's
            golden track (tools/make_golden.py config3: C Brox oracle + oracle tracker);
  config 4  lives in tests/test_fullsize_gpu.py (1024x1024, ~200 vertices);
  config 1  in tests/test_ekf_gpu.py (test_track_config1_matches_golden).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

# mean end-point error against the analytic field over the interior (border n/8 excluded), C oracle,
# seed 0, measured 0.0521 / 0.0517 / 0.1180 / 0.0455 px -- bounds = measured + 10 %: a change of the
# *algorithm* shows up here even though oracle and kernels would still agree with each other
ANALYTIC_BOUND_512 = {"translate_leftup": 0.0573, "translate_leftup_stretch": 0.0568, "rotate": 0.1298, "warp": 0.0501}


@pytest.mark.parametrize("name", ["translate_leftup", "translate_leftup_stretch", "rotate", "warp"])
def test_config2_single_512_pair_matches_oracle(hm, oracle_brox, name):
    from hydra_mi import brox, synth
    n = 512
    f0, f1, tu, tv = synth.warp_pair(n, name, 0)
    bf = brox.BroxOpticalFlow(n, n)                       # reference defaults, one pair per call
    assert bf.levels() == oracle_brox.levels(n, n)
    u, v = bf.calc(f0, f1)
    oracle_brox.set_threads(min(16, os.cpu_count() or 1))
    try:
        ru, rv = oracle_brox.calc(f0, f1)
    finally:
        oracle_brox.set_threads(1)
    epe = np.sqrt((u - ru) ** 2 + (v - rv) ** 2)
    assert epe.max() <= 1e-4, epe.max()                   # the contract (north_star)
    assert np.array_equal(u, ru) and np.array_equal(v, rv)   # what is achieved
    b = n // 8
    err = np.sqrt((u - tu) ** 2 + (v - tv) ** 2)[b:-b, b:-b]
    assert err.mean() <= ANALYTIC_BOUND_512[name], (name, err.mean())
    # launch tuning does not change a bit of it at this size either (several tiles per level)
    bf.tune("sor_threads", 512)
    u2, v2 = bf.calc(f0, f1)
    assert np.array_equal(u2, u) and np.array_equal(v2, v)


def test_track_config3_matches_golden(hm):
    path = os.path.join(GOLD, "config3_track.npz")
    from hydra_mi import brox, kalman, mesh, synth
    g = np.load(path)
    n, frames = int(g["n"]), int(g["frames"])
    video, masks, centre, radius = synth.disk_video(n, frames, "warp", 0)
    dm = mesh.disk_mesh(centre[0], centre[1], radius - 1.0, float(g["h0"]) * n)
    assert np.array_equal(dm.p, g["p"]) and np.array_equal(dm.t, g["t"])      # the golden's mesh
    assert 30 <= dm.size() <= 50
    bf = brox.BroxOpticalFlow(n, n)
    kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
    for k in range(1, frames):
        u, v = bf.calc(video[k - 1], video[k])
        e = kf.compute(video[k], np.dstack((u, v)), masks[k])
        X, Xg = kf.state.X.reshape(-1), g["X"][k - 1]
        rel = np.linalg.norm(X - Xg) / np.linalg.norm(Xg)
        assert rel <= 1e-5, (k, rel)                                          # the contract (north_star)
        assert kf.niter == int(g["iters"][k - 1]), k
        assert e[0] == int(g["err"][k - 1][0]) and e[3] == int(g["err"][k - 1][3])      # integer terms exact
        assert abs(e[1] - g["err"][k - 1][1]) <= 1e-6 * g["err"][k - 1][1]
        assert abs(e[2] - g["err"][k - 1][2]) <= 1e-6 * g["err"][k - 1][2]
    W, Wg = kf.state.W, g["W_last"]
    assert np.linalg.norm(W - Wg) <= 1e-5 * np.linalg.norm(Wg)


# RMS of (true - tracked) positions / velocities over the frames, measured on one MI355X (worst frame: 1.36 / 0.35 px
# for the translation, 1.75 / 0.43 for the rotation -- the accuracy of the reference's algorithm with its delta = 2 px
# differences on this texture, the states being those of the oracle to 1e-15) + 25 %
TRUE_STATE_BOUND = {"translate_leftup": (1.70, 0.44), "rotate": (2.19, 0.54)}


@pytest.mark.parametrize("name", ["translate_leftup", "rotate"])
def test_track_against_true_mesh_states(hm, name):
    """The ground-truth protocol of reference test_synthetic.py:28-105: track a synthetic video whose motion is an
    analytic field (synthetic/flowfields.py:3-7) and compare kf.state.X after every frame with the TRUE mesh states --
    the initial vertices advected through the field, velocities = the field at the vertices -- as RMS position and
    RMS velocity errors (:88-92).  Flow from the product's Brox, through the streaming pipeline."""
    from hydra_mi import kalman, mesh, synth
    from hydra_mi.pipeline import FlowEKFPipeline
    n, frames = 256, 8
    video, masks, c, r = synth.disk_video(n, frames, name, 5)
    dm = mesh.disk_mesh(c[0], c[1], r - 2.0, 0.1 * n)
    N = dm.size()
    field = synth.scaled_field(name, n)
    true_p = [np.asarray(dm.p, np.float64).copy()]
    for _ in range(frames - 1):
        vx, vy = field(true_p[-1][:, 0], true_p[-1][:, 1])
        true_p.append(true_p[-1] + np.column_stack((vx * np.ones(N), vy * np.ones(N))))
    kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
    pipe = FlowEKFPipeline(kf, video, masks, flow_batch=4)
    rms = []

    def check(k, e):
        X = kf.state.X.reshape(-1)
        pos, vel = X[:2 * N].reshape(-1, 2), X[2 * N:].reshape(-1, 2)
        vx, vy = field(true_p[k + 1][:, 0], true_p[k + 1][:, 1])
        tv = np.column_stack((vx * np.ones(N), vy * np.ones(N)))
        rms.append((float(np.sqrt(np.mean((pos - true_p[k + 1]) ** 2))), float(np.sqrt(np.mean((vel - tv) ** 2)))))
    pipe.run(on_frame=check)
    pipe.close()
    print("true-state RMS (%s): %s" % (name, ["%.3f / %.3f" % q for q in rms]))
    bp, bv = TRUE_STATE_BOUND[name]
    assert max(q[0] for q in rms) <= bp and max(q[1] for q in rms[1:]) <= bv, rms
