"""Pins oracle/ekf_ref.py to the reference's own known answers for this path
(reference test/test_cuda.py:198-266, fixture recipe
test/createtestdata_kalmanfilter.py:37-54) and checks the raster rules the
oracle fixes where OpenGL leaves them to the driver.  CPU only.
"""
import numpy as np
import pytest

from oracle import ekf_ref


def _ones_case(hm):
    """680x680 frame of 128, flow 1.534, 4-vertex square [226,453], velocity 1.534."""
    from hydra_mi import mesh
    nx = 680
    start, end = nx // 3, 2 * nx // 3
    dm = mesh.square4_mesh(start, end)
    frame = np.full((nx, nx), 128, np.uint8)
    flow = np.full((nx, nx, 2), 1.534, np.float32)
    vel = np.full(dm.p.shape, 1.534)
    X = np.concatenate((dm.p.reshape(-1), vel.reshape(-1)))
    # the test predates the eps scaling and the mask term: eps = 1, mask term checked apart
    meas = ekf_ref.Measurement(4, dm.t, dm.p, frame, 1.0, 1.0, 1.0)
    return nx, start, end, dm, frame, flow, X, meas


def test_ones_initjacobian(hm):
    """test_cuda.py:216-236: residual images of the unperturbed render."""
    nx, start, end, dm, frame, flow, X, meas = _ones_case(hm)
    y_m = np.ones((nx, nx), np.uint8)
    meas.initjacobian(X, frame, flow, y_m)
    eps = 1e-6
    side = end - start
    assert abs(np.sum(meas.z) - 128 * (nx * nx - side * side) / 255.0) < 1e-6 * nx * nx
    assert abs(meas.zfx[start + 1, start + 1]) < eps
    assert abs(meas.zfy[start + 1, start + 1]) < eps
    assert abs(meas.zfx[start - 1, start - 1] - 1.534) < eps
    assert abs(meas.zfy[start - 1, start - 1] - 1.534) < eps
    assert np.sum(np.abs(meas.zfx) < eps) == side * side


def test_ones_jz(hm):
    """test_cuda.py:238-254: all vertices shifted by +1 in x and y."""
    nx, start, end, dm, frame, flow, X, meas = _ones_case(hm)
    y_m = np.ones((nx, nx), np.uint8)
    meas.initjacobian(X, frame, flow, y_m)
    Xp = X.copy()
    Xp[:8] += 1
    total, c = meas.jz(Xp)
    expect_im = (226 + 227) * (128.0 / 255) * (128.0 / 255)
    expect_flow = 2 * (226 + 227) * (1.534 * 1.534)
    a = c[0] + c[1] + c[2]
    assert abs(a - expect_im - expect_flow) / a < 1e-5
    assert abs(c[0] - expect_im) / expect_im < 1e-12
    assert abs(c[1] - expect_flow / 2) / expect_flow < 1e-6 and abs(c[2] - expect_flow / 2) / expect_flow < 1e-6
    # mask term (added to the reference after that test): the 453 uncovered pixels have z_m = 0,
    # the 453 newly covered ones have render difference 1 and z_m = (255 - 0)/255 = 1
    assert abs(c[3] - 453.0) < 1e-9
    assert abs(total - c.sum()) < 1e-9


def test_ones_jz_velocity_only(hm):
    """test_cuda.py:256-266: perturbing only the velocities leaves jz at 0."""
    nx, start, end, dm, frame, flow, X, meas = _ones_case(hm)
    meas.initjacobian(X, frame, flow, np.ones((nx, nx), np.uint8))
    Xp = X.copy()
    Xp[8:] += 1
    total, c = meas.jz(Xp)
    assert abs(c[0]) < 1e-7 and abs(c[3]) < 1e-7
    # flow residual is 0 inside the square, where the velocity render changed
    assert abs(total) < 1e-7


def test_ones_get_pixel_data(hm):
    """test_cuda.py:128-150 (get_pixel_data on the "ones" fixture, recipe test/createtestdata_kalmanfilter.py:52-54): the
    render itself, not only sums over it -- the texture target holds 128 on exactly (end - start)^2 pixels
    (np.sum(a) == (end - start)^2 * 128), the x-velocity target 1.534 inside the square to 1e-7, the y-velocity target
    -1.534 (renderer.py:513 draws -v_y: "Why negative?? Because I flipped it")."""
    nx, start, end, dm, frame, flow, X, meas = _ones_case(hm)
    a, b, c, d = meas.render(X)
    side = end - start
    assert int(np.sum(a.astype(np.int64))) == side * side * 128
    assert abs(float(b[start + 1, start + 1]) - 1.534) < 1e-7
    assert abs(-float(c[start + 1, start + 1]) - 1.534) < 1e-7
    assert abs(float(np.sum(c.astype(np.float64))) + side * side * 1.534) < side * side * 1e-7      # (:150, commented out there)
    assert int((d == 255).sum()) == side * side and int((d == 0).sum()) == nx * nx - side * side   # the mask target (:937)


def test_zeros_fixture(hm):
    """test_cuda.py:30-56 (the "zeros" fixture, recipe test/createtestdata_kalmanfilter.py:46-48: frame 0, flow 0, velocity
    0, the same 4-vertex square): every render target sums to exactly 0 (:53-56) and jz of any perturbation is exactly 0
    (:38-39) -- a black texture on a black frame with no flow carries no information."""
    from hydra_mi import mesh
    nx = 680
    start, end = nx // 3, 2 * nx // 3
    dm = mesh.square4_mesh(start, end)
    frame = np.zeros((nx, nx), np.uint8)
    flow = np.zeros((nx, nx, 2), np.float32)
    X = np.concatenate((dm.p.reshape(-1), np.zeros(8)))
    meas = ekf_ref.Measurement(4, dm.t, dm.p, frame, 1.0, 1.0, 1.0)
    a, b, c, d = meas.render(X)
    assert np.sum(a) == 0 and np.sum(b) == 0 and np.sum(c) == 0
    # the reference's observed mask term did not exist when that test was written: against the mask the render itself
    # covers (z_m = 0 everywhere) the whole of jz is 0, exactly, for position and velocity perturbations alike
    y_m = (d // 255).astype(np.uint8)
    meas.initjacobian(X, frame, flow, y_m)
    for k, delta in ((0, 3.0), (1, -3.0), (5, 3.0), (8, 3.0), (15, 3.0)):
        Xp = X.copy()
        Xp[k] += delta
        total, comp = meas.jz(Xp)
        assert comp[0] == 0.0 and comp[1] == 0.0 and comp[2] == 0.0
    Xp = X.copy()
    Xp[8:] += 1.0
    total, comp = meas.jz(Xp)
    assert total == 0.0


def test_square_coverage_and_shared_edge(hm):
    from hydra_mi import mesh
    dm = mesh.square4_mesh(10, 30)
    tex = np.arange(64 * 64, dtype=np.int64).reshape(64, 64) % 251
    tex = tex.astype(np.uint8)
    X = np.concatenate((dm.p.reshape(-1), np.zeros(8)))
    im, fx, fy, m = ekf_ref.render(X, 4, dm.t, dm.p, tex, 64, 64)
    assert m[10:30, 10:30].min() == 255 and int((m == 255).sum()) == 400     # [10,30) each way
    # identity render reproduces the texture: every pixel on the shared diagonal drawn once
    assert np.array_equal(im[10:30, 10:30], tex[10:30, 10:30])
    assert im[m == 0].max() == 0


def test_orientation_and_fixed_point_grid(hm):
    from hydra_mi import mesh
    dm = mesh.square4_mesh(8, 24)
    tex = np.full((32, 32), 77, np.uint8)
    X = np.concatenate((dm.p.reshape(-1), np.zeros(8)))
    a = ekf_ref.render(X, 4, dm.t, dm.p, tex, 32, 32)
    b = ekf_ref.render(X, 4, dm.t[:, ::-1], dm.p, tex, 32, 32)     # both windings are drawn
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    X2 = X.copy()
    X2[:8] += 1.0 / 1024                                            # below the 1/256 grid
    c = ekf_ref.render(X2, 4, dm.t, dm.p, tex, 32, 32)
    assert all(np.array_equal(p, q) for p, q in zip(a, c))


def test_additive_overlap_saturates(hm):
    tri = np.array([[0, 1, 2], [0, 1, 2]])
    p = np.array([[4.0, 4.0], [28.0, 4.0], [4.0, 28.0]])
    tex = np.full((32, 32), 200, np.uint8)
    X = np.concatenate((p.reshape(-1), [1.5, 0, 1.5, 0, 1.5, 0]))
    im, fx, fy, m = ekf_ref.render(X, 3, tri, p, tex, 32, 32)
    assert im.max() == 255 and m.max() == 255
    assert abs(fx.max() - 3.0) < 1e-5                               # r32f target adds without clamping


def test_velocity_channels_and_sign(hm):
    from hydra_mi import mesh
    dm = mesh.square4_mesh(8, 24)
    tex = np.full((32, 32), 50, np.uint8)
    vel = np.array([[1.0, 2.0]] * 4)
    X = np.concatenate((dm.p.reshape(-1), vel.reshape(-1)))
    im, fx, fy, m = ekf_ref.render(X, 4, dm.t, dm.p, tex, 32, 32)
    assert np.allclose(fx[m == 255], 1.0, atol=1e-6) and np.allclose(fy[m == 255], -2.0, atol=1e-6)
    assert np.all(fx[m == 0] == 0)


def test_error_wraps_like_uint8(hm):
    from hydra_mi import mesh
    dm = mesh.square4_mesh(8, 24)
    tex = np.full((32, 32), 250, np.uint8)
    X = np.concatenate((dm.p.reshape(-1), np.zeros(8)))
    meas = ekf_ref.Measurement(4, dm.t, dm.p, tex, 1, 1, 1)
    y_im = np.full((32, 32), 10, np.uint8)
    y_m = np.ones((32, 32), np.uint8)
    flow = np.zeros((32, 32, 2), np.float32)
    e_im, e_fx, e_fy, e_m, _, _ = meas.error(X, y_im, flow, y_m)
    pix = meas.render(X)[0]
    lit = np.sum(np.multiply(y_im - pix, y_im - pix))            # the reference's own expression
    assert e_im == int(lit)
    m = meas.render(X)[3]
    lit_m = np.sum(np.multiply(255 * y_m - m, 255 * y_m - m))
    assert e_m == int(lit_m)
    assert e_fx == 0.0 and e_fy == 0.0


def test_jacobian_sign_pulls_towards_observation(hm):
    """A square observed 2 px to the right of the state: Hz must push x positive."""
    from hydra_mi import mesh, synth
    n = 48
    dm = mesh.box_mesh(14, 14, 34, 34, 10)
    N = dm.size()
    rng = np.random.default_rng(0)
    tex = (synth.noise_texture(n, 1)).astype(np.uint8)
    X = np.concatenate((dm.p.reshape(-1), np.zeros(2 * N)))
    meas = ekf_ref.Measurement(N, dm.t, dm.p, tex, 1e-3, 1.0, 1.0)
    Xobs = X.copy()
    Xobs[0:2 * N:2] += 2
    y_im, _, _, y_m = meas.render(Xobs)
    y_m = (y_m // 255).astype(np.uint8)
    flow = np.zeros((n, n, 2), np.float32)
    Hz, Hzc = ekf_ref.jacobian(meas, X, y_im, flow, y_m)
    assert Hz[0:2 * N:2].sum() > 0
    Jv, J = ekf_ref.adjacency(N, dm.t)
    HTH = ekf_ref.hessian_sparse(meas, X, J)
    assert np.allclose(HTH, HTH.T) and np.all(np.diag(HTH) >= 0)
    assert np.all(HTH[J == 0] == 0)


def test_multi_perturbation_oracle_equals_single(hm):
    """The oracle's restatement of the label-segmented kernels (cuda_multi.py:81-248) assembled as
    kalman.py:452-489 / 539-581 gives the single-perturbation Jacobian and Hessian (the protocol of
    testbites/test_multipert_validation.py:98-189), and the id image follows the palette rules."""
    from hydra_mi import mesh, synth
    from oracle import partitions_ref
    n = 48
    dm = mesh.disk_mesh(23.5, 23.5, 15.0, 11.0)
    N = dm.size()
    tex = synth.noise_texture(n, 1).astype(np.uint8)
    rng = np.random.default_rng(2)
    X = np.concatenate((dm.p.reshape(-1) + rng.normal(0, 0.5, 2 * N), rng.normal(0, 1.0, 2 * N)))
    meas = ekf_ref.Measurement(N, dm.t, dm.p, tex, 1e-3, 1.0, 1.0)
    y_im, yfx, yfy, ym = meas.render(np.concatenate((dm.p.reshape(-1) + 1.0, np.full(2 * N, 0.3))))
    y_m = (ym // 255).astype(np.uint8)
    flow = np.dstack((yfx, -yfy)).astype(np.float32)
    E, labels = partitions_ref.jacobian_partitions(N, dm.t)
    Q, EH, EHi, lh = partitions_ref.hessian_partitions(N, dm.t)
    ids = ekf_ref.render(X, N, dm.t, dm.p, tex, n, n, labels[:, 0])[4]
    m = meas.render(X)[3]
    assert set(np.unique(ids[m == 255])) <= set(int(v) for v in E[0]) | {65535}      # labelled or white
    assert np.all(ids[m == 0] == 0)
    Hz1, Hzc1 = ekf_ref.jacobian(meas, X, y_im, flow, y_m)
    Hzm, Hzcm = ekf_ref.jacobian_multi(meas, X, E, labels, y_im, flow, y_m)
    assert np.abs(Hz1 - Hzm).max() <= 1e-12 * np.abs(Hz1).max() and np.abs(Hzc1 - Hzcm).max() <= 1e-12 * np.abs(Hzc1).max()
    _, J = ekf_ref.adjacency(N, dm.t)
    H1 = ekf_ref.hessian_sparse(meas, X, J)
    Hm = ekf_ref.hessian_sparse_multi(meas, X, Q, EH, EHi, lh, y_im, flow, y_m)
    assert np.abs(H1 - Hm).max() <= 1e-12 * np.abs(H1).max() and not np.any((H1 != 0) & (Hm == 0))
