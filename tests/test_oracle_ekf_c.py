"""Pins oracle/ekf_ref_c.c (the C/OpenMP twin used for the large golden tracks, the full-size
oracle comparisons and bench.py's cpu_baseline) to oracle/ekf_ref.py and, through the same
known-answer family, to the reference's own test (reference test/test_cuda.py:198-266).  CPU only.
"""
import numpy as np
import pytest

from oracle import ekf_c, ekf_ref


def _random_case(n=72, seed=3):
    from hydra_mi import mesh, synth
    dm = mesh.disk_mesh(n / 2 - 0.5, n / 2 - 0.5, 0.3 * n, 0.15 * n)
    N = dm.size()
    tex = synth.noise_texture(n, seed).astype(np.uint8)
    rng = np.random.default_rng(seed)
    X = np.concatenate((dm.p.reshape(-1) + rng.normal(0, 0.8, 2 * N), rng.normal(0, 1.0, 2 * N)))
    Xobs = np.concatenate((dm.p.reshape(-1) + 1.25, np.full(2 * N, 0.5)))
    a = ekf_ref.Measurement(N, dm.t, dm.p, tex, 1e-3, 1.0, 1.0)
    b = ekf_c.Measurement(N, dm.t, dm.p, tex, 1e-3, 1.0, 1.0)
    y_im, yfx, yfy, ym = a.render(Xobs)
    y_m = (ym // 255).astype(np.uint8)
    flow = np.dstack((yfx, -yfy)).astype(np.float32) + rng.normal(0, 0.05, (n, n, 2)).astype(np.float32)
    return dm, N, X, a, b, y_im, flow, y_m


def test_render_bit_identical_to_numpy_oracle(hm):
    dm, N, X, a, b, y_im, flow, y_m = _random_case()
    for k in range(4):                                   # the state, and it moved off the frame's corner
        Xk = X.copy()
        Xk[:2 * N] -= 13.0 * k
        ra, rb = a.render(Xk), b.render(Xk)
        assert all(np.array_equal(p, q) for p, q in zip(ra, rb))


def test_render_folded_and_degenerate_mesh(hm):
    """overlapping triangles add and saturate, a zero-area triangle draws nothing, both windings draw"""
    tri = np.array([[0, 1, 2], [0, 2, 1], [0, 1, 1]])
    p = np.array([[4.0, 4.0], [28.0, 4.5], [4.25, 28.0]])
    tex = np.full((32, 32), 200, np.uint8)
    X = np.concatenate((p.reshape(-1), [1.5, 0, 1.5, 0.25, 1.5, 0]))
    ra = ekf_ref.render(X, 3, tri, p, tex, 32, 32)
    rb = ekf_c.Measurement(3, tri, p, tex, 1, 1, 1).render(X)
    assert all(np.array_equal(p_, q_) for p_, q_ in zip(ra, rb))
    assert ra[0].max() == 255


def test_jz_j_error_match_numpy_oracle(hm):
    dm, N, X, a, b, y_im, flow, y_m = _random_case()
    a.initjacobian(X, y_im, flow, y_m)
    b.initjacobian(X, y_im, flow, y_m)
    for k in (0, 1, 2 * N + 3, 4 * N - 1):
        Xp = X.copy()
        Xp[k] += 2.0
        ta, ca = a.jz(Xp)
        tb, cb = b.jz(Xp)
        assert np.allclose(ca, cb, rtol=1e-12, atol=1e-12) and abs(ta - tb) <= 1e-12 * max(1.0, abs(ta))
    for i, j in ((0, 0), (0, 1), (1, 2 * N + 1), (2 * N, 2 * N), (5, 5)):
        ja, jb = a.j(2.0, i, j), b.j(2.0, i, j)
        assert abs(ja - jb) <= 1e-12 * max(1.0, abs(ja))
    ea, eb = a.error(X, y_im, flow, y_m), b.error(X, y_im, flow, y_m)
    assert ea[0] == eb[0] and ea[3] == eb[3]                     # integer terms exact
    assert abs(ea[1] - eb[1]) <= 1e-12 * ea[1] and abs(ea[2] - eb[2]) <= 1e-12 * ea[2]
    assert np.array_equal(ea[4], eb[4]) and np.array_equal(ea[5], eb[5])


def test_whole_update_matches_numpy_oracle_and_is_thread_independent(hm):
    dm, N, X, a, b, y_im, flow, y_m = _random_case(48, 1)
    _, J = ekf_ref.adjacency(N, dm.t)
    Hz_a, Hzc_a = ekf_ref.jacobian(a, X, y_im, flow, y_m)
    H_a = ekf_ref.hessian_sparse(a, X, J)
    Hz_b, Hzc_b = ekf_ref.jacobian(b, X, y_im, flow, y_m)        # dispatches to the C loops
    H_b = ekf_ref.hessian_sparse(b, X, J)
    scale = np.abs(H_a).max()
    assert np.allclose(Hz_a, Hz_b, rtol=1e-11, atol=1e-11 * np.abs(Hz_a).max())
    assert np.allclose(Hzc_a, Hzc_b, rtol=1e-11, atol=1e-11 * np.abs(Hzc_a).max())
    assert np.allclose(H_a, H_b, rtol=1e-11, atol=1e-11 * scale)
    assert np.all(H_b[J == 0] == 0)
    b.set_threads(3)                                             # same sums whatever the team size
    Hz_c, Hzc_c = ekf_ref.jacobian(b, X, y_im, flow, y_m)
    H_c = ekf_ref.hessian_sparse(b, X, J)
    assert np.array_equal(Hz_b, Hz_c) and np.array_equal(Hzc_b, Hzc_c) and np.array_equal(H_b, H_c)


def test_ones_known_answers_with_c_twin(hm):
    """reference test/test_cuda.py:238-266 (eps = 1, the test predates the eps scaling)."""
    from hydra_mi import mesh
    nx = 680
    start, end = nx // 3, 2 * nx // 3
    dm = mesh.square4_mesh(start, end)
    frame = np.full((nx, nx), 128, np.uint8)
    flow = np.full((nx, nx, 2), 1.534, np.float32)
    X = np.concatenate((dm.p.reshape(-1), np.full(8, 1.534)))
    meas = ekf_c.Measurement(4, dm.t, dm.p, frame, 1.0, 1.0, 1.0)
    meas.initjacobian(X, frame, flow, np.ones((nx, nx), np.uint8))
    Xp = X.copy()
    Xp[:8] += 1
    total, c = meas.jz(Xp)
    expect_im = (226 + 227) * (128.0 / 255) * (128.0 / 255)
    expect_flow = 2 * (226 + 227) * (1.534 * 1.534)
    a = c[0] + c[1] + c[2]
    assert abs(a - expect_im - expect_flow) / a < 1e-5
    assert abs(c[0] - expect_im) / expect_im < 1e-12
    assert abs(c[3] - 453.0) < 1e-9
    Xv = X.copy()
    Xv[8:] += 1                                                  # velocities only: jz = 0 (:256-266)
    total, c = meas.jz(Xv)
    assert abs(total) < 1e-7 and abs(c[0]) < 1e-7 and abs(c[3]) < 1e-7


def test_tracker_with_c_twin_follows_numpy_tracker(hm):
    """two frames of BASELINE config 1 at 64x64: the oracle tracker gives the same state with either twin"""
    from hydra_mi import mesh, synth
    video, flow = synth.test_data(64, 64)
    dm = mesh.box_mesh(21.0, 22.0, 42.0, 43.0, 10.0)
    ta = ekf_ref.Tracker(dm.p, dm.t, dm.bars, dm.L, video[:, :, 0], nI=3)
    tb = ekf_ref.Tracker(dm.p, dm.t, dm.bars, dm.L, video[:, :, 0], nI=3, measurement=ekf_c.Measurement)
    for k in (1, 2):
        frame = video[:, :, k]
        mask = (frame > 0).astype(np.uint8)
        ea = ta.compute(frame, flow[:, :, :, k], mask)
        eb = tb.compute(frame, flow[:, :, :, k], mask)
        assert ta.niter == tb.niter and ea[0] == eb[0] and ea[3] == eb[3]
        rel = np.linalg.norm(ta.X - tb.X) / np.linalg.norm(ta.X)
        assert rel < 1e-9
