"""GPU parity of the EKF measurement path against oracle/ekf_ref.py, through the C-ABI.

Renders are integer / fixed-order binary32 work: the bar is bit-exact.  The reductions
are sums of those per-pixel values accumulated in binary64 in a different order than
numpy's: the bar is 1e-10 relative, far inside the contract's 1e-5 on the EKF state.
"""
import os

import numpy as np
import pytest

from oracle import ekf_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


class _Flow:
    pass


def _setup(hm, n=64, h0=11.0, seed=0, eps=(1e-3, 1.0, 1.0)):
    from hydra_mi import mesh, synth, renderer
    dm = mesh.disk_mesh((n - 1) / 2.0, (n - 1) / 2.0, 0.31 * n, h0)
    N = dm.size()
    tex = synth.noise_texture(n, seed + 2).astype(np.uint8)
    flow0 = np.zeros((n, n, 2), np.float32)
    R = renderer.Renderer(dm, np.zeros((N, 2)), flow0, n, tex, True, *eps)
    meas = ekf_ref.Measurement(N, dm.t, dm.p, tex, *eps)
    return dm, N, tex, R, meas


def _state(dm, rng, pos_sigma=0.7, vel_sigma=1.0):
    N = dm.size()
    return np.concatenate((dm.p.reshape(-1) + rng.normal(0, pos_sigma, 2 * N), rng.normal(0, vel_sigma, 2 * N)))


def _render(R, X):
    N = R.n
    R.update_vertex_buffer(X[:2 * N].reshape(-1, 2), X[2 * N:].reshape(-1, 2))
    return R.render()


@pytest.mark.parametrize("n,h0", [(64, 11.0), (96, 9.0), (130, 20.0)])
def test_render_bit_exact(hm, n, h0):
    dm, N, tex, R, meas = _setup(hm, n, h0)
    rng = np.random.default_rng(1)
    for trial in range(4):
        X = _state(dm, rng, pos_sigma=[0.0, 0.7, 2.5, 6.0][trial])
        got = _render(R, X)
        ref = meas.render(X)
        for name, g, r in zip(("im", "fx", "fy", "m"), got, ref):
            assert np.array_equal(g, r), (trial, name)
    # a mesh partly outside the frame
    X = _state(dm, rng)
    X[0:2 * N:2] -= 0.4 * n
    for g, r in zip(_render(R, X), meas.render(X)):
        assert np.array_equal(g, r)


def test_render_shared_edges_and_identity(hm):
    from hydra_mi import mesh, renderer
    n = 64
    dm = mesh.square4_mesh(10, 30)
    tex = (np.arange(n * n).reshape(n, n) % 251).astype(np.uint8)
    R = renderer.Renderer(dm, np.zeros((4, 2)), np.zeros((n, n, 2), np.float32), n, tex, True, 1, 1, 1)
    im, fx, fy, m = R.render()
    assert int((m == 255).sum()) == 400 and np.array_equal(im[10:30, 10:30], tex[10:30, 10:30])


def _observation(dm, meas, rng, n):
    N = dm.size()
    Xobs = np.concatenate((dm.p.reshape(-1) + 1.5, np.full(2 * N, 0.5)))
    y_im, yfx, yfy, ym = meas.render(Xobs)
    y_m = (ym // 255).astype(np.uint8)
    flow = (np.dstack((yfx, -yfy)) + rng.normal(0, 0.05, (n, n, 2))).astype(np.float32)
    return y_im, flow, y_m


def test_jz_j_error_match_cpu_twin(hm):
    n = 64
    dm, N, tex, R, meas = _setup(hm, n)
    rng = np.random.default_rng(3)
    X = _state(dm, rng)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    st = _Flow()
    st.X = X.reshape(-1, 1)
    R.update_vertex_buffer(X[:2 * N].reshape(-1, 2), X[2 * N:].reshape(-1, 2))
    R.initjacobian(y_im, flow, y_m)
    meas.initjacobian(X, y_im, flow, y_m)
    for k in (0, 1, 2 * N - 1, 2 * N, 4 * N - 1):
        Xp = X.copy()
        Xp[k] += 2.0
        st.X = Xp.reshape(-1, 1)
        got, gc = R.jz(st)
        ref, rc = meas.jz(Xp)
        assert np.allclose(gc, rc, rtol=1e-10, atol=1e-9), k
        assert abs(got - ref) <= 1e-10 * max(1.0, abs(ref))
    st.X = X.reshape(-1, 1)
    for (i, j) in [(0, 0), (0, 1), (1, 2 * N + 1), (2 * N, 2 * N), (2 * N, 2 * N + 1), (3, 3)]:
        got = R.j(st, 2.0, i, j)
        ref = meas.j(2.0, i, j)
        assert abs(got - ref) <= 1e-10 * max(1.0, abs(ref)), (i, j)
    e = R.error(st, y_im, flow, y_m)
    r = meas.error(X, y_im, flow, y_m)
    assert e[0] == r[0] and e[3] == r[3]                      # integer terms: exact
    assert abs(e[1] - r[1]) <= 1e-10 * r[1] and abs(e[2] - r[2]) <= 1e-10 * r[2]
    assert np.array_equal(e[4][:, :, 0], r[4]) and np.array_equal(e[5][:, :, 0], r[5])


def test_measure_matches_golden(hm):
    """hm_measure against the committed oracle vectors (tools/make_golden.py measure)."""
    from hydra_mi import mesh, renderer
    g = np.load(os.path.join(GOLD, "measure_64.npz"))
    dm = mesh.Mesh(g["p"], g["t"])
    N = dm.size()
    R = renderer.Renderer(dm, np.zeros((N, 2)), np.zeros((64, 64, 2), np.float32), 64, g["tex"], True, 1e-3, 1.0, 1.0)
    st = _Flow()
    st.X = g["X"].reshape(-1, 1)
    Hz, HTH, Hzc = R.measure(st, g["y_im"], g["flow"], g["y_m"])
    scale = np.abs(g["HTH"]).max()
    assert np.abs(HTH - g["HTH"]).max() <= 1e-9 * scale
    assert np.abs(Hz - g["Hz"]).max() <= 1e-9 * np.abs(g["Hz"]).max()
    assert np.abs(Hzc - g["Hzc"]).max() <= 1e-9 * np.abs(g["Hzc"]).max()
    assert np.array_equal(HTH, HTH.T)
    e = R.error(st, g["y_im"], g["flow"], g["y_m"])
    assert e[0] == int(g["err"][0]) and e[3] == int(g["err"][3])
    assert abs(e[1] - g["err"][1]) <= 1e-10 * g["err"][1]


def test_measure_equals_single_perturbation_operators(hm):
    """The fused kernel against the product's own one-render-per-perturbation operators and the oracle."""
    n = 48
    dm, N, tex, R, meas = _setup(hm, n, 10.0, seed=4)
    rng = np.random.default_rng(7)
    X = _state(dm, rng, pos_sigma=0.5)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    st = _Flow()
    st.X = X.reshape(-1, 1)
    Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
    rHz, rHzc = ekf_ref.jacobian(meas, X, y_im, flow, y_m)
    _, J = ekf_ref.adjacency(N, dm.t)
    rHTH = ekf_ref.hessian_sparse(meas, X, J)
    assert np.abs(Hz - rHz).max() <= 1e-9 * np.abs(rHz).max()
    assert np.abs(Hzc - rHzc).max() <= 1e-9 * np.abs(rHzc).max()
    assert np.abs(HTH - rHTH).max() <= 1e-9 * np.abs(rHTH).max()
    assert np.all(HTH[J == 0] == 0)
    # dense _hessian (kalman.py:521-536) has the same value: entries outside J are exactly 0
    R.update_vertex_buffer(X[:2 * N].reshape(-1, 2), X[2 * N:].reshape(-1, 2))
    R.initjacobian(y_im, flow, y_m)
    far = np.argwhere(J == 0)
    for i, j in far[:: max(1, len(far) // 6)][:6]:
        assert R.j(st, 2.0, int(i), int(j)) == 0.0


def test_dense_mesh_grows_the_difference_image_pool(hm):
    """The parked difference images of the measurement live in a pool of 8 frames' worth of pixels that grows on
    demand (the reference has no such limit: its renders are whole frames).  A fine mesh over most of a small frame --
    106 vertices on 48^2, star regions padded to whole 8x8 tiles: ~40 000 pixels against 18 432 -- makes hm_measure,
    hm_update_step and hm_update_run (each on a fresh context, so each meets the small pool) grow it and go on with the
    numbers of the oracle / of each other."""
    from oracle import ekf_c
    n = 48
    rng = np.random.default_rng(21)
    outs = []
    for path in ("measure", "step", "run"):
        dm, N, tex, R, meas = _setup(hm, n, 3.0, seed=6)
        assert N > 90
        if path == "measure":
            X = _state(dm, rng, pos_sigma=0.2)
            y_im, flow, y_m = _observation(dm, meas, rng, n)
            n4 = 4 * N
            M = np.random.default_rng(2).normal(size=(n4, n4))
            W = np.eye(n4) * 0.5 + 0.05 * (M @ M.T) / n4
        st = _Flow()
        st.X = X.reshape(-1, 1)
        R.update_frame(y_im, flow, y_m)
        if path == "measure":
            Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
            cm = ekf_c.Measurement(N, dm.t, dm.p, tex, 1e-3, 1.0, 1.0, threads=min(8, os.cpu_count() or 1))
            rHz, rHzc = cm.jacobian_all(X, y_im, flow, y_m)
            assert np.abs(Hz - rHz).max() <= 1e-9 * np.abs(rHz).max()
            _, J = ekf_ref.adjacency(N, dm.t)
            pi, pj = np.nonzero(np.triu(J == 1))
            sel = np.random.default_rng(3).choice(len(pi), 300, replace=False)
            vals = cm.hessian_pairs(pi[sel], pj[sel], 2.0)
            got = HTH[pi[sel], pj[sel]]
            assert np.abs(got - vals).max() <= 1e-9 * np.abs(HTH).max()
            assert np.all(HTH[J == 0] == 0) and np.array_equal(HTH, HTH.T)
        elif path == "step":
            R.update_begin(W, X)
            s1, _, e1 = R.update_step(st, y_im, flow, y_m)
            assert np.isfinite(s1).all()
            outs.append((s1, np.array(e1)))
        else:
            run = R.update_run(W, X, y_im, flow, y_m, 1, 1e-12)
            assert run[1]["niter"] == 1
            outs.append((None if run[1]["reverted"] else run[0].reshape(-1) - X, np.array(run[2][0])))
    assert np.array_equal(outs[0][1], outs[1][1])                      # the error sums of the first iterate
    if outs[1][0] is not None:
        assert np.allclose(outs[0][0].reshape(-1), outs[1][0], rtol=0, atol=1e-12 * np.abs(outs[0][0]).max())
    # the overflow inside a LOOP of iterations: by the time hm_update_run sees it, the render launch of that iteration has
    # put the star regions of its (meaningless) next iterate where the pool's new size is taken from -- the regions of the
    # measured iterate are computed again first.  Same iterations as on a context whose pool has grown already.
    dm, N, tex, R, meas = _setup(hm, n, 3.0, seed=6)
    R.update_frame(y_im, flow, y_m)
    first = R.update_run(W, X, y_im, flow, y_m, 3, 1e-12)              # grows the pool in its first iteration
    again = R.update_run(W, X, y_im, flow, y_m, 3, 1e-12)              # the pool is large enough now
    assert first[1]["niter"] == again[1]["niter"] >= 1
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[2], again[2])
    assert np.array_equal(first[2][0], outs[0][1])


def test_render_strip_height_changes_no_integer_term(hm):
    """hm_ctx_tune "render_rows": the strips of k_render_iter 8 instead of 16 rows high (twice as many partial sums: the
    buffer is sized for them) -- the same render, the same whole-number error terms, the flow terms to rounding."""
    n = 96
    outs = []
    for rows in (16, 8):
        dm, N, tex, R, meas = _setup(hm, n, 9.0, seed=3)
        R.tune("render_rows", rows)
        rng = np.random.default_rng(5)
        X = _state(dm, rng, pos_sigma=0.8)
        y_im, flow, y_m = _observation(dm, meas, rng, n)
        st = _Flow()
        st.X = X.reshape(-1, 1)
        R.update_frame(y_im, flow, y_m)
        outs.append(R.error(st, y_im, flow, y_m))
    a, b = outs
    assert a[0] == b[0] and a[3] == b[3]
    assert abs(a[1] - b[1]) <= 1e-12 * a[1] and abs(a[2] - b[2]) <= 1e-12 * a[2]
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


@pytest.mark.parametrize("case", ["outside", "folded", "blank", "saturated"])
def test_measure_edge_cases_match_oracle(hm, case):
    """The fused measurement where the rasteriser's corner cases matter: a mesh partly outside the
    frame (clipped star boxes), folded triangles (overlap: additive blending, 8-bit saturation),
    an empty observation (no mask, zero flow), a saturated texture."""
    n = 48
    dm, N, tex, R, meas = _setup(hm, n, 10.0, seed=14)
    rng = np.random.default_rng(17)
    X = _state(dm, rng, pos_sigma=0.5)
    if case == "outside":
        X[0:2 * N:2] -= 0.45 * n                  # x coordinates: a third of the mesh leaves the frame
        X[1:2 * N:2] += 0.30 * n
    elif case == "folded":
        X = _state(dm, rng, pos_sigma=4.0)        # vertices jump across their neighbours
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    if case == "blank":
        y_im = np.zeros_like(y_im); y_m = np.zeros_like(y_m); flow = np.zeros_like(flow)
    if case == "saturated":
        from hydra_mi import renderer
        tex = np.full((n, n), 255, np.uint8)
        R = renderer.Renderer(dm, np.zeros((N, 2)), np.zeros((n, n, 2), np.float32), n, tex, True, 1e-3, 1.0, 1.0)
        meas = ekf_ref.Measurement(N, dm.t, dm.p, tex, 1e-3, 1.0, 1.0)
        X = _state(dm, rng, pos_sigma=3.0)        # overlaps of a white texture: sums beyond 255
    st = _Flow()
    st.X = X.reshape(-1, 1)
    Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
    rHz, rHzc = ekf_ref.jacobian(meas, X, y_im, flow, y_m)
    _, J = ekf_ref.adjacency(N, dm.t)
    rHTH = ekf_ref.hessian_sparse(meas, X, J)
    tol = lambda a: 1e-9 * max(np.abs(a).max(), 1e-300)
    assert np.abs(Hz - rHz).max() <= tol(rHz)
    assert np.abs(Hzc - rHzc).max() <= tol(rHzc)
    assert np.abs(HTH - rHTH).max() <= tol(rHTH)
    e = R.error(st, y_im, flow, y_m)
    r = meas.error(X, y_im, flow, y_m)
    assert e[0] == r[0] and e[3] == r[3]


def test_masked_flow_path(hm):
    from hydra_mi.renderer import MaskedFlow
    n = 64
    dm, N, tex, R, meas = _setup(hm, n)
    rng = np.random.default_rng(9)
    X = _state(dm, rng)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    flow = flow + 0.3                                  # non-zero flow outside the mask
    st = _Flow()
    st.X = X.reshape(-1, 1)
    R.update_frame(y_im, flow, y_m)
    mf = MaskedFlow(flow, y_m)
    Hz, HTH, Hzc = R.measure(st, y_im, mf, y_m)
    rHz, _ = ekf_ref.jacobian(meas, X, y_im, ekf_ref.mask_flow(flow, y_m), y_m)
    assert np.abs(Hz - rHz).max() <= 1e-9 * np.abs(rHz).max()
    e = R.error(st, y_im, flow, y_m)                   # raw flow again, without a second upload
    r = meas.error(X, y_im, flow, y_m)
    assert abs(e[1] - r[1]) <= 1e-10 * r[1]


def test_track_config1_matches_golden(hm):
    """BASELINE config 1 end to end: IteratedMSKalmanFilter.compute over the translating square,
    state after every frame within 1e-5 relative of the oracle's (tools/make_golden.py config1)."""
    path = os.path.join(GOLD, "config1_track.npz")
    if not os.path.exists(path):
        pytest.skip("golden track not generated")
    from hydra_mi import mesh, synth, kalman
    g = np.load(path)
    video, flow = synth.test_data(128, 128)
    dm = mesh.Mesh(g["p"], g["t"], 15.0)
    kf = kalman.IteratedMSKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True)
    for k in range(g["X"].shape[0]):
        frame = video[:, :, k]
        mask = (frame > 0).astype(np.uint8)
        e = kf.compute(frame, flow[:, :, :, k], mask)
        X = kf.state.X.reshape(-1)
        rel = np.linalg.norm(X - g["X"][k]) / np.linalg.norm(g["X"][k])
        assert rel <= 1e-5, (k, rel)
        assert kf.niter == int(g["iters"][k]), k
        assert e[0] == int(g["err"][k][0]) and e[3] == int(g["err"][k][3])


def test_prediction_started_ahead_changes_nothing(hm):
    """IteratedMSKalmanFilter.predict_ahead (the next frame's hm_ms_newton on a worker thread from the end of the
    update on): the same states, bit for bit, with it and without; a state touched between frames is noticed and
    predicted again from what it is."""
    from hydra_mi import mesh, synth, kalman
    video, flow = synth.test_data(128, 128)
    g = np.load(os.path.join(GOLD, "config1_track.npz"))
    tracks = {}
    for ahead in (True, False):
        kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(g["p"], g["t"], 15.0), video[:, :, 0], flow[:, :, :, 0], True)
        kf.predict_ahead = ahead
        out = []
        for k in range(5):
            frame = video[:, :, k]
            if k == 3:
                kf.state.X[0, 0] += 0.25                 # the caller moves a vertex between frames
            kf.compute(frame, flow[:, :, :, k], (frame > 0).astype(np.uint8))
            out.append((kf.state.X.copy(), kf.niter, kf.newton_iterations))
        tracks[ahead] = out
        kf.close()
    for a, b in zip(tracks[True], tracks[False]):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]


def test_newton_on_device_matches_host_loop(hm):
    """The state prediction as one launch (hm_ms_worker_attach -> k_ms_newton4: four waves, a vertex per lane) against
    the host loop (hm_ms_newton) and the oracle (reference kalman.py:923-960): the per-vertex operations are the host's,
    the sums over the vector are added in another order -- 1e-12 relative, the same Newton iteration counts; a whole
    track with it stays within 1e-9 of the track with the host loop."""
    import ctypes
    from hydra_mi import mesh, synth, kalman, _lib
    video, flow = synth.test_data(128, 128)
    g = np.load(os.path.join(GOLD, "config1_track.npz"))
    kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(g["p"], g["t"], 15.0), video[:, :, 0], flow[:, :, :, 0], True)
    st = kf.state
    rng = np.random.default_rng(5)
    X0 = st.X.reshape(-1).copy()
    n2 = 2 * st.N
    X0[n2:] = rng.normal(0, 1.5, n2)                       # velocities: the springs get something to do
    X0[:n2] += rng.normal(0, 0.4, n2)
    bars = np.ascontiguousarray(kf._bars, np.int32)
    l0 = np.ascontiguousarray(st.l0[:, 0], np.float64)
    L = _lib.lib()
    Xh = X0.copy()
    ih = ctypes.c_int()
    _lib.check(L.hm_ms_newton(int(st.N), int(bars.shape[0]), _lib.ptr(bars), _lib.ptr(l0), float(kf.kappa), float(kf.M),
                              float(kf.deltat), int(kf.maxiter), float(kf.tol), _lib.ptr(Xh), ctypes.byref(ih)), "hm_ms_newton")
    Xd = X0.copy()
    idv = ctypes.c_int()
    rc = L.hm_newton_dev_start(st.renderer._h, int(st.N), int(bars.shape[0]), _lib.ptr(bars), _lib.ptr(l0), float(kf.kappa),
                               float(kf.M), float(kf.deltat), int(kf.maxiter), float(kf.tol), _lib.ptr(Xd))
    assert rc == 0, rc                                       # this mesh fits the kernel
    _lib.check(L.hm_newton_dev_finish(st.renderer._h, _lib.ptr(Xd), ctypes.byref(idv)), "hm_newton_dev_finish")
    assert idv.value == ih.value
    assert np.abs(Xd - Xh).max() <= 1e-12 * np.abs(Xh).max(), np.abs(Xd - Xh).max()
    n4 = 4 * st.N
    K = ekf_ref.incidence(st.N, bars)
    ref, _ = ekf_ref.ms_predict(X0, np.eye(n4), np.zeros((n4, n4)), K, l0, kf.kappa, kf.M, kf.deltat, kf.maxiter, kf.tol)
    assert np.abs(Xd - ref.reshape(-1)).max() <= 1e-9 * np.abs(ref).max()
    kf.close()
    tracks = {}
    for dev in (True, False):
        kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(g["p"], g["t"], 15.0), video[:, :, 0], flow[:, :, :, 0], True)
        kf.newton_on_device = dev
        out = []
        for k in range(5):
            frame = video[:, :, k]
            kf.compute(frame, flow[:, :, :, k], (frame > 0).astype(np.uint8))
            out.append((kf.state.X.copy(), kf.niter, kf.newton_iterations))
        tracks[dev] = out
        kf.close()
    for a, b in zip(tracks[True], tracks[False]):
        assert a[1:] == b[1:]
        assert np.abs(a[0] - b[0]).max() <= 1e-9 * np.abs(b[0]).max()


def test_covariance_prediction_queued_ahead_changes_nothing(hm):
    """IteratedMSKalmanFilter.cov_ahead (hm_update_arm_cov / hm_predict_take: the update queues F W F^T + Weps and the
    factorisation of the result for the next frame behind its own launches): the same states and covariances, bit
    for bit, with it and without; between frames the resident covariance is still the posterior; a state touched
    between frames is noticed and the prediction made again from what it is."""
    from hydra_mi import mesh, synth, kalman
    video, flow = synth.test_data(128, 128)
    g = np.load(os.path.join(GOLD, "config1_track.npz"))
    tracks, taken = {}, {}
    for ahead in (True, False):
        kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(g["p"], g["t"], 15.0), video[:, :, 0], flow[:, :, :, 0], True)
        kf.cov_ahead = ahead
        r = kf.state.renderer
        took = []
        take = r.predict_take
        def spy(*a, _take=take, _took=took, **k):
            out = _take(*a, **k)
            _took.append(out is not None)
            return out
        r.predict_take = spy
        out = []
        for k in range(6):
            frame = video[:, :, k]
            if k == 3:
                kf.state.X[0, 0] += 0.25                 # the caller moves a vertex between frames
            kf.compute(frame, flow[:, :, :, k], (frame > 0).astype(np.uint8))
            W = np.array(kf.state.W) if k in (1, 4) else None      # the posterior, fetched while the prediction is queued
            out.append((kf.state.X.copy(), kf.niter, W))
        out.append((np.array(kf.state.W), 0, None))
        tracks[ahead], taken[ahead] = out, took
        kf.close()
    assert taken[False] == []
    # asked from frame 1 on; frames 2 and 5: W had been fetched (a host array now), frame 3: the state was touched
    assert taken[True] == [True, False, False, True, False], taken[True]
    for a, b in zip(tracks[True], tracks[False]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]
        assert (a[2] is None) == (b[2] is None) and (a[2] is None or np.array_equal(a[2], b[2]))


def test_state_errors_are_loud(hm):
    from hydra_mi import mesh, renderer
    dm = mesh.square4_mesh(10, 30)
    tex = np.zeros((64, 64), np.uint8)
    R = renderer.Renderer(dm, np.zeros((4, 2)), np.zeros((64, 64, 2), np.float32), 64, tex, True, 1, 1, 1)
    st = _Flow()
    st.X = np.zeros((16, 1))
    with pytest.raises(RuntimeError):
        R.jz(st)                                        # no initjacobian yet (reference: assert, cuda_multi.py:611)
    with pytest.raises(NotImplementedError):
        renderer.Renderer(dm, np.zeros((4, 2)), np.zeros((64, 64, 2), np.float32), 64, tex, False, 1, 1, 1)
    with pytest.raises(RuntimeError):
        renderer.Renderer(dm, np.zeros((4, 2)), np.zeros((64, 64, 2), np.float32), 64, tex, True, 0.0, 1, 1)


# the last case has more than 248 vertices: 4N > 992, i.e. more than two column batches in k_back_row and a
# block count that is not a power of two for the recursive inverse
# 4N = 72, 200, 632, 1380 and 160, 108, 308: last block of 8, 8, 24, 4 and 32 (none), 12, 20 rows
@pytest.mark.parametrize("n,h0", [(64, 11.0), (96, 9.0), (160, 8.0), (256, 8.5), (64, 7.0), (96, 13.0), (128, 9.5)])
def test_device_dense_update_matches_host_algebra(hm, n, h0):
    """hm_update_begin/_step/_cov (blocked Cholesky on the device) against numpy on the host:
    step = (inv(W) + HTH)^-1 (Hz - HTH (X0 - X)), cov = (inv(W) + HTH)^-1."""
    dm, N, tex, R, meas = _setup(hm, n, h0, seed=6)
    rng = np.random.default_rng(11)
    X = _state(dm, rng, pos_sigma=0.5)
    X0 = X + rng.normal(0, 0.3, X.size)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    n4 = 4 * N
    M = rng.normal(size=(n4, n4))
    W = np.eye(n4) * 0.5 + 0.05 * (M @ M.T) / n4
    st = _Flow()
    st.X = X.reshape(-1, 1)
    Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
    R.update_begin(W, X0)
    step, Hzc2, err = R.update_step(st, y_im, flow, y_m)
    st2 = _Flow()
    st2.X = X0.reshape(-1, 1) + step
    e_ref = R.error(st2, y_im, flow, y_m)
    assert err[0] == e_ref[0] and err[3] == e_ref[3] and abs(err[1] - e_ref[1]) <= 1e-12 * e_ref[1]
    A = np.linalg.inv(W) + HTH
    ref = np.linalg.solve(A, Hz - HTH @ (X0 - X).reshape(-1, 1))
    assert np.array_equal(Hzc, Hzc2)
    assert np.linalg.norm(step - ref) <= 1e-9 * np.linalg.norm(ref)
    cov = R.update_cov(0)
    assert np.linalg.norm(cov - np.linalg.inv(A)) <= 1e-9 * np.linalg.norm(np.linalg.inv(A))
    # a second step keeps the first factor available as "the one before"
    st.X = (X + 0.1).reshape(-1, 1)
    Hz2, HTH2, _ = R.measure(st, y_im, flow, y_m)
    step2, _, _ = R.update_step(st, y_im, flow, y_m)
    A2 = np.linalg.inv(W) + HTH2
    ref2 = np.linalg.solve(A2, Hz2 - HTH2 @ (X0 - st.X.reshape(-1)).reshape(-1, 1))
    assert np.linalg.norm(step2 - ref2) <= 1e-9 * np.linalg.norm(ref2)
    assert np.linalg.norm(R.update_cov(1) - np.linalg.inv(A)) <= 1e-9 * np.linalg.norm(np.linalg.inv(A))
    assert np.linalg.norm(R.update_cov(0) - np.linalg.inv(A2)) <= 1e-9 * np.linalg.norm(np.linalg.inv(A2))


def test_device_dense_update_single_block(hm):
    """4N = 16: the whole system is one (partial) 32x32 block."""
    from hydra_mi import mesh, renderer
    n = 64
    dm = mesh.square4_mesh(14, 44)
    tex = (np.arange(n * n).reshape(n, n) * 7 % 251).astype(np.uint8)
    R = renderer.Renderer(dm, np.zeros((4, 2)), np.zeros((n, n, 2), np.float32), n, tex, True, 1e-3, 1.0, 1.0)
    meas = ekf_ref.Measurement(4, dm.t, dm.p, tex, 1e-3, 1.0, 1.0)
    rng = np.random.default_rng(2)
    X = _state(dm, rng, pos_sigma=0.4)
    X0 = X + rng.normal(0, 0.3, X.size)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    M = rng.normal(size=(16, 16))
    W = np.eye(16) * 0.5 + 0.05 * (M @ M.T) / 16
    st = _Flow()
    st.X = X.reshape(-1, 1)
    Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
    R.update_begin(W, X0)
    step, _, _ = R.update_step(st, y_im, flow, y_m)
    A = np.linalg.inv(W) + HTH
    ref = np.linalg.solve(A, Hz - HTH @ (X0 - X).reshape(-1, 1))
    assert np.linalg.norm(step - ref) <= 1e-9 * np.linalg.norm(ref)
    assert np.linalg.norm(R.update_cov(0) - np.linalg.inv(A)) <= 1e-9 * np.linalg.norm(np.linalg.inv(A))
    assert np.linalg.norm(R.update_cov(-1) - W) <= 1e-9 * np.linalg.norm(W)


def test_plain_kalman_filter_update_on_device(hm):
    """KalmanFilter.update (kalman.py:745-761) through the device path against the oracle."""
    from hydra_mi import kalman, mesh, synth
    video, flow = synth.test_data(64, 64)
    dm = mesh.box_mesh(21.0, 22.0, 42.0, 43.0, 10.0)
    kf = kalman.KalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True)
    tr_meas = ekf_ref.Measurement(dm.size(), kf.state.tri, dm.p, video[:, :, 0], 1e-3, 1e-3, 1e-3)
    frame, mask = video[:, :, 1], (video[:, :, 1] > 0).astype(np.uint8)
    kf.predict()
    Xp, Wp = kf.state.X.copy(), kf.state.W.copy()
    fm = ekf_ref.mask_flow(flow[:, :, :, 1], mask)
    kf.update(frame, fm, mask)
    Xr, Wr = ekf_ref.kf_update(tr_meas, Xp, Wp, kf.state.J, frame, fm, mask)
    assert np.linalg.norm(kf.state.X - Xr) <= 1e-8 * np.linalg.norm(Xr)
    assert np.linalg.norm(kf.state.W - Wr) <= 1e-7 * np.linalg.norm(Wr)


def test_device_covariance_prediction(hm):
    """hm_cov_predict against the host formula F W F^T + Weps, for both dynamics, with the covariance
    uploaded and with the copy that the previous call left on the device (DeviceCovariance)."""
    from hydra_mi import kalman
    n = 64
    dm, N, tex, R, meas = _setup(hm, n, 9.0, seed=3)
    rng = np.random.default_rng(21)
    n4 = 4 * N
    M = rng.normal(size=(n4, n4))
    W = np.eye(n4) * 0.3 + 0.02 * (M @ M.T) / n4
    F, Weps, _ = ekf_ref.initial_covariances(N, 0.1)
    got = R.cov_predict(W, None, None, 1.0, 0.0, 0.1)
    assert np.linalg.norm(got - (F @ W @ F.T + Weps)) <= 1e-13 * np.linalg.norm(got)
    # mass-spring F from the oracle's dense dfdy
    y = dm.p.reshape(-1) + rng.normal(0, 0.5, 2 * N)
    K = ekf_ref.incidence(N, dm.bars)
    l0 = ekf_ref.bar_lengths(K, dm.p.reshape(-1))
    dfdy = ekf_ref.ms_dfdy(K, l0, y, -1.0)
    e = np.eye(2 * N)
    Fm = np.block([[e, 0.05 * e], [0.05 * dfdy, e]])
    d = y.reshape(-1, 2)[dm.bars[:, 0]] - y.reshape(-1, 2)[dm.bars[:, 1]]
    l = np.sqrt((d * d).sum(1))
    k, c = -1.0 * (1 - l0 / l), -1.0 * l0 / l ** 3
    blocks = np.column_stack((k + c * d[:, 0] ** 2, c * d[:, 0] * d[:, 1], k + c * d[:, 1] ** 2))
    got = R.cov_predict(W, dm.bars, blocks, 0.05, 0.05, 0.1)
    ref = Fm @ W @ Fm.T + Weps
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)
    # prior straight from the device: predict -> begin(no upload) -> step -> cov -> predict(no upload)
    X = _state(dm, rng, pos_sigma=0.4)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    st = _Flow()
    st.X = X.reshape(-1, 1)
    tok = R.cov_predict(W, dm.bars, blocks, 0.05, 0.05, 0.1, fetch=False)
    assert np.array_equal(tok.fetch(), got)
    R.update_begin(tok, X)                       # the prior is used where it is: no upload
    step, _, _ = R.update_step(st, y_im, flow, y_m)
    Hz, HTH, _ = R.measure(st, y_im, flow, y_m)
    A = np.linalg.inv(ref) + HTH
    assert np.linalg.norm(step - np.linalg.solve(A, Hz)) <= 1e-8 * np.linalg.norm(step)
    R.update_begin(got.copy(), X)
    R.update_step(st, y_im, flow, y_m)
    assert not tok.valid()                       # that covariance has been replaced on the device
    with pytest.raises(RuntimeError):
        tok.fetch()
    Wtok = R.update_cov(0, fetch=False)
    Wp = Wtok.fetch()
    assert np.array_equal(R.update_cov(-1), got)                   # the prior is still there
    Wtok = R.update_cov(0, fetch=False)
    nxt = R.cov_predict(Wtok, dm.bars, blocks, 0.05, 0.05, 0.1)    # propagated where it is
    ref2 = Fm @ Wp @ Fm.T + Weps
    assert np.linalg.norm(nxt - ref2) <= 1e-12 * np.linalg.norm(ref2)


@pytest.mark.parametrize("stress", [False, True])
def test_fused_update_equals_stepwise_loop(hm, stress):
    """hm_update_run (the whole iterated update in one call) against the same loop written in
    Python over hm_update_begin / _step / _cov: same iterations, acceptance decisions, state and
    covariance, bit for bit.  stress: a vague prior and a fast flow, so that large steps (and,
    if they occur, mesh inversions and the rollback) are covered too."""
    from hydra_mi import mesh, synth, kalman
    n = 128
    video, flow = synth.test_data(n, n)
    if stress:
        flow = flow * 2.0
    kfs = []
    for fused in (True, False):
        dm = mesh.mask_mesh(video[:, :, 0] > 0, 12.0)
        kf = kalman.IteratedMSKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True)
        kf.fused_update = fused
        if stress:
            kf.state.W = kf.state.W * 400.0
        kfs.append(kf)
    for k in range(4):
        frame = video[:, :, k]
        mask = (frame > 0).astype(np.uint8)
        out = [kf.compute(frame, flow[:, :, :, k], mask) for kf in kfs]
        a, b = kfs
        assert (a.niter, a.reverted, a.converged) == (b.niter, b.reverted, b.converged), k
        assert np.array_equal(a.state.X, b.state.X), k
        assert out[0][:4] == out[1][:4], k
        Wa, Wb = a.state.W, b.state.W
        assert np.array_equal(Wa, Wb), k
        assert np.allclose(a.tv, b.tv, rtol=1e-9, atol=1e-12) and np.allclose(a.fv, b.fv, rtol=1e-9, atol=1e-12)
        assert np.allclose(a.mv, b.mv, rtol=1e-9, atol=1e-12)


def test_update_with_indefinite_covariance_fails_loudly_and_recovers(hm):
    """A prior that is not positive definite (or not finite) has no Cholesky factor: the update reports
    it (HM_ERR_NUMERIC -> FloatingPointError) instead of returning numbers, and the handle stays usable."""
    n = 64
    dm, N, tex, R, meas = _setup(hm, n)
    rng = np.random.default_rng(8)
    X = _state(dm, rng, pos_sigma=0.3)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    n4 = 4 * N
    R.update_frame(y_im, flow, y_m)
    good = np.eye(n4) * 0.5
    bad = good.copy()
    bad[5, 5] = -1.0
    with pytest.raises(FloatingPointError):
        R.update_run(bad, X, y_im, flow, y_m, 3, 1e-4)
    nan = good.copy()
    nan[n4 - 1, n4 - 1] = np.nan
    with pytest.raises(FloatingPointError):
        R.update_run(nan, X, y_im, flow, y_m, 3, 1e-4)
    Xk, info, errs, Hzc, gains, tok = R.update_run(good, X, y_im, flow, y_m, 3, 1e-4)
    assert info["niter"] >= 1 and np.all(np.isfinite(Xk)) and np.all(np.isfinite(tok.fetch()))


def test_update_run_without_iterations_keeps_the_prior(hm):
    dm, N, tex, R, meas = _setup(hm, 64, 9.0, seed=5)
    rng = np.random.default_rng(2)
    X = _state(dm, rng, pos_sigma=0.3)
    y_im, flow, y_m = _observation(dm, meas, rng, 64)
    n4 = 4 * N
    W = np.eye(n4) * 0.7
    R.update_frame(y_im, flow, y_m)
    Xk, info, errs, Hzc, gains, tok = R.update_run(W, X, y_im, flow, y_m, 0, 1e-4)
    assert info == dict(niter=0, accepted=0, reverted=False, converged=False) and errs.shape == (0, 4)
    assert np.array_equal(Xk.reshape(-1), X) and np.array_equal(tok.fetch(), W)


def test_dense_update_is_deterministic_under_contention(hm):
    """The update kernels exchange data between workgroups only across launches; their result must
    not depend on when workgroups start.  Run the same step on a quiet GPU and while a flow batch
    keeps every CU busy on another stream: bit-identical (this catches in-place races such as a
    panel workgroup overwriting a block its neighbours still have to read)."""
    torch = pytest.importorskip("torch")
    from hydra_mi import brox, synth
    n = 160
    dm, N, tex, R, meas = _setup(hm, n, 8.0, seed=8)
    rng = np.random.default_rng(31)
    X = _state(dm, rng, pos_sigma=0.5)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    n4 = 4 * N
    M = rng.normal(size=(n4, n4))
    W = np.eye(n4) * 0.5 + 0.05 * (M @ M.T) / n4
    st = _Flow()
    st.X = X.reshape(-1, 1)
    R.update_begin(W, X)
    quiet, _, e_quiet = R.update_step(st, y_im, flow, y_m)
    cov_quiet = R.update_cov(0)
    m = 512
    f0, f1, _, _ = synth.warp_pair(m, "warp", 0)
    B = 6
    F0 = torch.from_numpy(np.stack([f0] * B)).cuda()
    F1 = torch.from_numpy(np.stack([f1] * B)).cuda()
    U = torch.empty((B, m, m), dtype=torch.float32, device="cuda")
    V = torch.empty_like(U)
    torch.cuda.synchronize()
    bf = brox.BroxOpticalFlow(m, m, max_batch=B)
    for trial in range(3):
        bf.calc_dev(B, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())      # asynchronous
        R.update_begin(W, X)
        busy, _, e_busy = R.update_step(st, y_im, flow, y_m)
        cov_busy = R.update_cov(0)
        bf.sync()
        assert np.array_equal(busy, quiet) and e_busy == e_quiet and np.array_equal(cov_busy, cov_quiet), trial


def test_measure_split_changes_only_the_summation_order(hm):
    """hm_ctx_tune("measure_split"): more or fewer workgroups per vertex give the same sums up to
    the rounding of another summation order."""
    dm, N, tex, R, meas = _setup(hm, 96, 9.0, seed=4)
    rng = np.random.default_rng(12)
    X = _state(dm, rng, pos_sigma=0.5)
    y_im, flow, y_m = _observation(dm, meas, rng, 96)
    st = _Flow()
    st.X = X.reshape(-1, 1)
    R.update_frame(y_im, flow, y_m)
    Hz0, HTH0, Hzc0 = R.measure(st, y_im, flow, y_m)
    for split in (1, 2, 7, 16):
        R.tune("measure_split", split)
        Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
        assert np.abs(Hz - Hz0).max() <= 1e-12 * np.abs(Hz0).max(), split
        assert np.abs(HTH - HTH0).max() <= 1e-12 * np.abs(HTH0).max(), split
        assert np.abs(Hzc - Hzc0).max() <= 1e-12 * np.abs(Hzc0).max(), split
    R.tune("measure_split", 5)
    for split in (1, 3, 16):
        R.tune("edge_split", split)
        Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
        assert np.array_equal(Hz, Hz0) and np.array_equal(Hzc, Hzc0), split      # vertex jobs: untouched
        assert np.abs(HTH - HTH0).max() <= 1e-12 * np.abs(HTH0).max(), split
    with pytest.raises(RuntimeError):
        R.tune("measure_split", 17)
    with pytest.raises(RuntimeError):
        R.tune("edge_split", 0)
    with pytest.raises(RuntimeError):
        R.tune("no_such_knob", 1)


# ---- projectmask on the device (hm_project_mask) against oracle/ekf_ref.project_mask ---------------
# (the distance function is the reference's fd, the signed distance to the polygon through the object's border pixels;
# tests/test_host_cpu.py::test_mask_distance_is_the_polygon_distance_of_the_reference holds the oracle's restatement
# against imgproc.findObjectThreshold(mask)[2] and against known answers)
def _project_masks(n):
    yy, xx = np.mgrid[:n, :n]
    disk = ((xx - 0.5 * n) ** 2 + (yy - 0.45 * n) ** 2 < (0.3 * n) ** 2).astype(np.uint8)
    holed = disk.copy()
    holed[int(0.4 * n):int(0.5 * n), int(0.45 * n):int(0.55 * n)] = 0
    edge = np.zeros((n, n), np.uint8)
    edge[:n // 2, n // 3:] = 1                                      # touches two frame borders
    specks = disk.copy()
    specks[3, 5] = 1                                                # single far pixel
    specks[n - 2, n - 2] = 1
    # what the reference's contour pruning (imgproc.py:205-228) decides: a second, smaller object close to the mesh; a
    # 5 x 5 hole (contour area 34 < 40: counts as object) and a 5 x 6 one (40: stays a hole) with an island in it
    second = disk.copy()
    second[int(0.8 * n):int(0.8 * n) + 9, int(0.3 * n):int(0.3 * n) + 14] = 1
    pinholes = disk.copy()
    cy, cx = int(0.45 * n), int(0.5 * n)
    pinholes[cy - 12:cy - 7, cx - 10:cx - 5] = 0
    pinholes[cy + 4:cy + 9, cx + 3:cx + 9] = 0
    pinholes[cy + 6, cx + 5] = 1
    return {"disk": disk, "holed": holed, "edge": edge, "specks": specks, "second": second, "pinholes": pinholes}


def test_contour_pruning_matches_oracle(hm):
    """reference imgproc.py:198-228 on the device (k_ccl_*, csrc/project_kernels.h) against oracle/ekf_ref.pruned_object:
    the hand-made cases and random masks with specks, pinholes, islands and objects on the frame edge -- the same pixels."""
    from mask_cases import blobs as _blobs
    n = 96
    dm, N, tex, R, meas = _setup(hm, n, 9.0)
    for kind, mask in _project_masks(n).items():
        assert np.array_equal(R.prune_mask(mask).astype(bool), ekf_ref.pruned_object(mask)), kind
    rng = np.random.default_rng(23)
    for trial in range(60):
        mm = _blobs(rng, n, n, int(rng.integers(1, 6)))
        assert np.array_equal(R.prune_mask(mm).astype(bool), ekf_ref.pruned_object(mm)), trial
    ring = np.zeros((n, n), bool)
    ring[5:35, 5:35] = True
    ring[7:33, 7:33] = False                                        # contour area 841 beats the 400 solid pixels below
    ring[40:60, 50:70] = True
    assert np.array_equal(R.prune_mask(ring).astype(bool), ekf_ref.pruned_object(ring))
    for blank in (np.zeros((n, n), bool), np.ones((n, n), bool)):
        assert np.array_equal(R.prune_mask(blank).astype(bool), ekf_ref.pruned_object(blank))


@pytest.mark.parametrize("kind", ["disk", "holed", "edge", "specks", "second", "pinholes"])
def test_project_mask_matches_oracle(hm, kind):
    n = 96
    dm, N, tex, R, meas = _setup(hm, n, 9.0)
    mask = _project_masks(n)[kind]
    rng = np.random.default_rng(3)
    for trial, spread in enumerate((0.5, 4.0, 15.0, 60.0)):
        X = _state(dm, rng, pos_sigma=spread)
        if trial == 3:
            X[0:2] = (-7.3, 5.1)                                    # off the frame
            X[2:4] = (n + 11.0, n + 2.5)
            X[4:6] = (17.0, 33.0)                                   # on a pixel centre
        want = ekf_ref.project_mask(X, N, mask)[:, 0]
        got, moved = R.project_mask(X, mask)
        assert got.shape == X.shape
        inside_before = int((np.abs(want - X)[:2 * N].reshape(-1, 2).max(axis=1) > 0).sum())
        assert moved >= inside_before                               # moved counts d > 1, some of which may step by 0
        # same operations in the same order on exact integer distances: the same f64 numbers, also where
        # a vanishing gradient throws a vertex far away (the walk is the reference's, kalman.py:731-739)
        assert np.array_equal(got, want), (kind, trial, np.abs(got - want).max())


def test_project_mask_resident_observation_and_blank(hm):
    n = 64
    dm, N, tex, R, meas = _setup(hm, n)
    rng = np.random.default_rng(4)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    X = _state(dm, rng, pos_sigma=5.0)
    with pytest.raises(RuntimeError):
        R.project_mask(X)                                           # no observation yet
    R.update_frame(y_im, flow, y_m)
    got, moved = R.project_mask(X)                                  # the observation's mask, already on the device
    want = ekf_ref.project_mask(X, N, y_m)[:, 0]
    assert moved > 0 and np.array_equal(got, want)
    again, moved2 = R.project_mask(X, y_m)
    assert moved2 == moved and np.array_equal(again, got)
    for blank in (np.zeros((n, n), np.uint8), np.ones((n, n), np.uint8)):
        same, moved = R.project_mask(X, blank)
        assert moved == 0 and np.array_equal(same, X)
    # the resident path again, behind calls that left another mask's outline in the buffers, for other states (every
    # vertex writes its entries, moved or not), and with the outline queued ahead by a new observation
    for trial, spread in enumerate((0.3, 5.0, 20.0)):
        X2 = _state(dm, rng, pos_sigma=spread)
        got2, moved2 = R.project_mask(X2)
        want2 = ekf_ref.project_mask(X2, N, y_m)[:, 0]
        assert np.array_equal(got2, want2), trial
        assert (moved2 == 0) == np.array_equal(got2, X2) or moved2 > 0
    y_m3 = np.roll(y_m, 3, axis=1)
    R.update_frame(y_im, flow, y_m3)
    got3, moved3 = R.project_mask(X)
    assert np.array_equal(got3, ekf_ref.project_mask(X, N, y_m3)[:, 0])
    got4, moved4 = R.project_mask(X)                                # twice on the same outline: the counters are clean
    assert moved4 == moved3 and np.array_equal(got4, got3)
    with pytest.raises(ValueError):
        R.project_mask(X, np.zeros((n, n + 1), np.uint8))
    with pytest.raises(ValueError):
        R.project_mask(X[:-1], y_m)


def test_filter_projectmask_uses_the_device_and_matches_host_walk(hm):
    from hydra_mi import kalman
    n = 64
    dm, N, tex, R, meas = _setup(hm, n)
    kf = kalman.KalmanFilter(dm, tex, np.zeros((n, n, 2), np.float32), True)
    rng = np.random.default_rng(6)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    kf.state.X[:2 * N, 0] += rng.normal(0, 4.0, 2 * N)
    X0 = kf.state.X.copy()
    kf.projectmask(y_m)
    want = ekf_ref.project_mask(X0, N, y_m)
    assert kf.state.X.shape == X0.shape and np.array_equal(kf.state.X, want)
    assert not np.array_equal(kf.state.X, X0)


# ---- the reference's multi-perturbation operators (cuda_multi.py:81-248; kalman.py:452-489, 539-581) --------------
def _multi_case(hm, n=64, h0=11.0, seed=5):
    from oracle import partitions_ref
    dm, N, tex, R, meas = _setup(hm, n, h0, seed)
    rng = np.random.default_rng(seed)
    X = _state(dm, rng)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    E, labels = partitions_ref.jacobian_partitions(N, dm.t)
    Q, EH, EHi, lh = partitions_ref.hessian_partitions(N, dm.t)
    return dm, N, R, meas, X, y_im, flow, y_m, E, labels, Q, EH, EHi, lh


def test_jz_multi_and_j_multi_match_oracle(hm):
    """hm_jz_multi / hm_j_multi against the oracle's restatement of histogram_jz / histogram_j: every vertex of a
    partition perturbed in one render, sums separated by triangle label; also with a palette that leaves
    triangles unlabelled and with perturbations large enough to fold the mesh (ids add and saturate)."""
    dm, N, R, meas, X, y_im, flow, y_m, E, labels, Q, EH, EHi, lh = _multi_case(hm)
    st = _Flow()
    R.labels, R.labels_hess, R.Q = labels, lh, Q
    R.update_vertex_buffer(X[:2 * N].reshape(-1, 2), X[2 * N:].reshape(-1, 2))
    R.initjacobian(y_im, flow, y_m)
    meas.initjacobian(X, y_im, flow, y_m)
    for idx, e in enumerate(E):
        e = np.asarray(e)
        for comp, delta in ((0, 2.0), (1, -2.0), (2 * N, 2.0), (2 * N + 1, -2.0), (0, 9.0)):     # 9 px: folds triangles
            Xp = X.copy()
            Xp[comp + 2 * e] += delta
            st.X = Xp.reshape(-1, 1)
            R.update_vertex_buffer(Xp[:2 * N].reshape(-1, 2), Xp[2 * N:].reshape(-1, 2), idx)
            hz, hzc = R.jz_multi(st)
            rhz, rhzc = meas.jz_multi(Xp, labels[:, idx], N)
            scale = max(np.abs(rhzc).max(), 1e-30)
            assert np.abs(hzc - rhzc).max() <= 1e-10 * scale, (idx, comp)
            assert np.abs(hz[:, 0] - rhz).max() <= 1e-10 * scale
    st.X = X.reshape(-1, 1)
    for idx in (0, len(EH) // 2, len(EH) - 1):
        e = np.asarray(EH[idx]).reshape(-1, 2)
        for (o1, o2) in ((0, 0), (1, 2 * N), (2 * N, 2 * N + 1)):
            ee = np.column_stack((2 * e[:, 0] + o1, 2 * e[:, 1] + o2))
            h, hist, hc = R.j_multi(st, 2.0, ee, idx)
            rh, rnz, rhc = meas.j_multi(2.0, ee, lh[:, idx], len(Q))
            assert np.array_equal(hist, rnz > 0)
            assert np.abs(hc - rhc).max() <= 1e-10 * max(np.abs(rhc).max(), 1e-30)
            assert np.abs(h[0] - rh).max() <= 1e-10 * max(np.abs(rhc).max(), 1e-30)
    with pytest.raises(RuntimeError):
        R.update_vertex_buffer(X[:2 * N].reshape(-1, 2), X[2 * N:].reshape(-1, 2))       # palette -1: no labels
        R.jz_multi(st)


def test_multi_perturbation_assembly_equals_single_and_fused(hm):
    """The multi == single protocol of reference testbites/test_multipert_validation.py:98-189 on the device:
    KFState._jacobian_multi / _hessian_sparse_multi (partitions E / E_hessian, label-segmented sums) against the
    single-perturbation assembly, the fused measure() and the oracle's multi assembly.  The reference records
    agreement to round-off except a few mask-term entries (~2 %); with exact coverage rules and partitions whose
    stars do not interact the two agree to rounding."""
    from hydra_mi import kalman, renderer
    dm, N, R0, meas, X, y_im, flow, y_m, E, labels, Q, EH, EHi, lh = _multi_case(hm, 64, 12.0, 7)
    tex = meas.tex
    kf = kalman.IteratedMSKalmanFilter(dm, tex, np.zeros((64, 64, 2), np.float32), True)
    st = kf.state
    st.X = X.reshape(-1, 1).copy()
    assert [list(map(int, e)) for e in st.E] == E and np.array_equal(st.labels, labels)          # the palettes in use
    Hz_m, Hzc_m = st._jacobian_multi(y_im, flow, y_m)
    H_m = st._hessian_sparse_multi(y_im, flow, y_m)
    assert np.abs(st.X - X.reshape(-1, 1)).max() <= 1e-12                                       # state restored (x + d - 2d + d, as the reference does it)
    Hz_f, H_f, Hzc_f = st.update(y_im, flow, y_m)                                                # fused kernels
    sc, hs = np.abs(Hz_f).max(), np.abs(H_f).max()
    assert np.abs(Hz_m - Hz_f).max() <= 1e-9 * sc and np.abs(Hzc_m - Hzc_f).max() <= 1e-9 * np.abs(Hzc_f).max()
    assert np.abs(H_m - H_f).max() <= 1e-9 * hs
    assert not np.any((H_f != 0) & (H_m == 0))                                                   # no non-zero missing
    rHz, rHzc = ekf_ref.jacobian_multi(meas, X, E, labels, y_im, flow, y_m)
    assert np.abs(Hz_m - rHz).max() <= 1e-9 * sc
    rH = ekf_ref.hessian_sparse_multi(meas, X, Q, EH, EHi, lh, y_im, flow, y_m)
    assert np.abs(H_m - rH).max() <= 1e-9 * hs


def test_ms_predict_on_device_matches_oracle_and_host(hm):
    """hm_ms_predict (state prediction as one workgroup on the device + covariance prediction + the covariance
    half of the next update) against the oracle's IteratedMSKalmanFilter.predict (kalman.py:850-960: dense
    inverse per Newton iteration) and against the host Newton of the same library; then a whole frame with
    either predict gives the same state."""
    from hydra_mi import kalman
    n = 64
    dm, N, tex, R, meas = _setup(hm, n, 9.0, seed=4)
    rng = np.random.default_rng(33)
    n4 = 4 * N
    Mx = rng.normal(size=(n4, n4))
    W = np.eye(n4) * 0.3 + 0.02 * (Mx @ Mx.T) / n4
    X = np.concatenate((dm.p.reshape(-1) + rng.normal(0, 0.6, 2 * N), rng.normal(0, 1.5, 2 * N)))
    K = ekf_ref.incidence(N, dm.bars)
    l0 = ekf_ref.bar_lengths(K, dm.p.reshape(-1))
    _, Weps, _ = ekf_ref.initial_covariances(N, 0.1)
    rX, rW = ekf_ref.ms_predict(X, W, Weps, K, l0)
    tok = R.cov_predict(W, None, None, 0.0, 0.0, 0.0, fetch=False)           # W itself, resident on the device
    assert np.array_equal(tok.fetch(), W)
    Xp, its, tokp = R.ms_predict(tok, X, dm.bars, l0, -1.0, 1.0, 0.05, 1000, 1e-4, 0.1)
    assert its >= 20                                                        # 20 sub-steps, at least one iteration each
    assert np.linalg.norm(Xp - rX) <= 1e-9 * np.linalg.norm(rX)
    Wp = tokp.fetch()
    assert np.linalg.norm(Wp - rW) <= 1e-11 * np.linalg.norm(rW)
    # the host version of the same loop (hm_ms_newton): same iteration count, same numbers to rounding
    import ctypes
    from hydra_mi import _lib
    Xh = X.copy()
    b32 = np.ascontiguousarray(dm.bars, np.int32)
    itsh = ctypes.c_int()
    _lib.check(_lib.lib().hm_ms_newton(N, len(b32), _lib.ptr(b32), _lib.ptr(np.ascontiguousarray(l0)), -1.0, 1.0, 0.05,
                                       1000, 1e-4, _lib.ptr(Xh), ctypes.byref(itsh)))
    assert itsh.value == its and np.abs(Xh - Xp[:, 0]).max() <= 1e-12 * np.abs(Xh).max()
    # the prefactored covariance is picked up by the update: same step as with an explicit begin
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    st = _Flow()
    st.X = Xp
    R.update_begin(tokp, Xp)
    step_a, _, _ = R.update_step(st, y_im, flow, y_m)
    R.update_begin(Wp.copy(), Xp)
    step_b, _, _ = R.update_step(st, y_im, flow, y_m)
    assert np.array_equal(step_a, step_b)
    # a whole frame through the filter, device predict against host predict
    video, flowv = __import__("hydra_mi").synth.test_data(64, 64)
    from hydra_mi import mesh
    outs = []
    for dev_pred in (True, False):
        dmq = mesh.box_mesh(21.0, 22.0, 42.0, 43.0, 10.0)
        kf = kalman.IteratedMSKalmanFilter(dmq, video[:, :, 0], flowv[:, :, :, 0], True, nI=3)
        assert kf.device_predict is False                                   # the faster one is the default
        kf.device_predict = dev_pred
        for k in (1, 2):
            fr = video[:, :, k]
            kf.compute(fr, flowv[:, :, :, k], (fr > 0).astype(np.uint8))
        outs.append(kf.state.X.copy())
    assert np.linalg.norm(outs[0] - outs[1]) <= 1e-9 * np.linalg.norm(outs[1])


@pytest.mark.parametrize("n,h0", [(48, 30.0), (64, 20.0), (64, 11.0), (96, 9.0), (130, 9.0)])
def test_persistent_factorisation_equals_launch_per_step(hm, n, h0):
    """chol_flow = 1 (one persistent launch whose block tasks hand their results over through memory,
    csrc/chol_flow_kernels.h) gives the very bits of chol_flow = 0 (one launch per 32-column block step): steps,
    covariances of the last and the previous iterate, the whole iterated update -- for systems of one partial
    block up to several blocks with a partial last one, with and without right-hand-side rows (update steps /
    the inverse of the prior), and repeatedly on one handle (the pre-filled pattern is renewed every time)."""
    dm, N, tex, R, meas = _setup(hm, n, h0, seed=8)
    rng = np.random.default_rng(17)
    X = _state(dm, rng, pos_sigma=0.5)
    X0 = X + rng.normal(0, 0.3, X.size)
    y_im, flow, y_m = _observation(dm, meas, rng, n)
    n4 = 4 * N
    M = rng.normal(size=(n4, n4))
    W = np.eye(n4) * 0.5 + 0.05 * (M @ M.T) / n4
    st = _Flow()
    out = {}
    for mode in (1, 0, 1):
        R.tune("chol_flow", mode)
        st.X = X.reshape(-1, 1)
        R.update_frame(y_im, flow, y_m)
        R.update_begin(W, X0)
        s1, _, e1 = R.update_step(st, y_im, flow, y_m)
        st.X = (X + 0.1).reshape(-1, 1)
        s2, _, e2 = R.update_step(st, y_im, flow, y_m)
        c_prev, c_last = R.update_cov(1), R.update_cov(0)
        run = R.update_run(W, X0, y_im, flow, y_m, 4, 1e-12)
        got = (s1, s2, c_prev, c_last, run[0], run[2], run[5].fetch(), np.array(e1), np.array(e2))
        if mode in out:
            assert all(np.array_equal(a, b) for a, b in zip(out[mode], got))          # and it repeats
        out[mode] = got
    assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
    A = np.linalg.inv(W)
    assert np.isfinite(out[1][3]).all() and np.abs(out[1][3] - out[1][3].T).max() <= 1e-12 * np.abs(out[1][3]).max()
    with pytest.raises(RuntimeError):
        R.tune("chol_flow", 2)


def test_persistent_factorisation_survives_a_slow_chain(hm):
    """The tasks of the persistent factorisation launch that wait for the chain of diagonal blocks re-load hot for a
    few thousand rounds and then hand the watch to one lane that sleeps between polls (chol_flow_kernels.h,
    flow_fetch2).  With the chain stalled ~10 ms per block (test knob "chol_flow_stall") every such wait goes through
    the patient path: same bits, no time-out -- a slow chain (contention, a profiler) must not fail the frame.  And the
    launch needs a task workgroup besides the chain: chol_flow_wgs = 1 is refused."""
    dm, N, tex, R, meas = _setup(hm, 96, 9.0, seed=8)
    rng = np.random.default_rng(3)
    X = _state(dm, rng, pos_sigma=0.5)
    X0 = X + rng.normal(0, 0.3, X.size)
    y_im, flow, y_m = _observation(dm, meas, rng, 96)
    n4 = 4 * N
    assert n4 > 64                                                         # several block columns
    M = rng.normal(size=(n4, n4))
    W = np.eye(n4) * 0.5 + 0.05 * (M @ M.T) / n4
    st = _Flow()
    out = []
    for stall in (0, 2500, 0):
        R.tune("chol_flow_stall", stall)
        st.X = X.reshape(-1, 1)
        R.update_frame(y_im, flow, y_m)
        R.update_begin(W, X0)
        s1, _, e1 = R.update_step(st, y_im, flow, y_m)
        out.append((s1, R.update_cov(0), np.array(e1)))
    assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
    assert all(np.array_equal(a, b) for a, b in zip(out[0], out[2]))
    with pytest.raises(RuntimeError):
        R.tune("chol_flow_wgs", 1)
    R.tune("chol_flow_wgs", 2)                                             # the smallest launch that can make progress
    st.X = X.reshape(-1, 1)
    R.update_begin(W, X0)
    s2, _, _ = R.update_step(st, y_im, flow, y_m)
    assert np.array_equal(s2, out[0][0])
    R.tune("chol_flow_wgs", 256)
