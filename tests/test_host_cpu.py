"""Host logic of the product on the CPU: file format, input generators, meshes, filter
algebra (with the measurement supplied by the oracle), sharding over two gloo ranks."""
import os
import struct
import sys

import numpy as np
import pytest

from oracle import ekf_ref, partitions_ref
from mask_cases import blobs as _blobs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- .mat wire format (reference src/optical_flow_ext.cpp:47-170) -----------------------
def test_mat_format_bytes(hm, tmp_path):
    from hydra_mi import matio
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    fn = str(tmp_path / "f_000_x.mat")
    matio.write_mat(fn, a)
    raw = open(fn, "rb").read()
    assert struct.unpack("<iii", raw[:12]) == (5, 4, 3)          # CV_32FC1, width, height
    assert raw[12:] == a.tobytes()
    assert np.array_equal(matio.read_mat(fn), a)
    d = np.arange(6, dtype=np.float64).reshape(2, 3)
    fn2 = str(tmp_path / "d.mat")
    matio.write_mat(fn2, d)
    assert struct.unpack("<iii", open(fn2, "rb").read(12)) == (6, 3, 2)
    assert np.array_equal(matio.read_mat(fn2), d)
    with pytest.raises(ValueError):
        matio.write_mat(fn2, np.zeros((2, 2), np.int32))
    with pytest.raises(IOError):
        matio.read_mat(str(tmp_path / "missing.mat"))


def test_flowstream_reads_tool_output(hm, tmp_path):
    from hydra_mi import matio
    from hydra_mi.renderer import FlowStream
    prefix = str(tmp_path / "flow")
    for k in range(2):
        matio.write_flow(prefix, k, np.full((4, 5), k + 0.5, np.float32), np.full((4, 5), -k, np.float32))
    fs = FlowStream(prefix)
    assert fs.isOpened()
    ok, peeked = fs.peek()
    ok, f0 = fs.read()
    ok1, f1 = fs.read()
    ok2, f2 = fs.read()
    assert ok and ok1 and not ok2 and f2 is None
    assert f0.shape == (4, 5, 2) and f0.dtype == np.float32 and np.array_equal(peeked, f0)
    assert np.all(f1[:, :, 0] == 1.5) and np.all(f1[:, :, 1] == -1)


# ---- input generators ---------------------------------------------------------------------
def test_synth_test_data_matches_reference_layout(hm):
    """reference synth.py:10-42 including its quirk: box rows [43,86) (flipped), flow rows [42,85)."""
    from hydra_mi import synth
    video, flow = synth.test_data(128, 128)
    assert video.shape == (128, 128, 10) and flow.shape == (128, 128, 2, 10)
    rows = np.nonzero(video[:, :, 0].any(axis=1))[0]
    cols = np.nonzero(video[:, :, 0].any(axis=0))[0]
    assert (rows.min(), rows.max(), cols.min(), cols.max()) == (43, 85, 42, 84)
    assert set(np.unique(video[:, :, 0])) == {0, 128, 255}
    f = np.nonzero(flow[:, :, 0, 0])
    assert (f[0].min(), f[0].max(), f[1].min(), f[1].max()) == (42, 84, 42, 84)
    assert np.all(flow[42:85, 42:85, :, 0] == -3)
    # frame k is frame 0 moved up-left by 3k
    assert np.array_equal(video[:-9, :-9, 3], video[9:, 9:, 0])
    assert np.all(flow[33:76, 33:76, :, 3] == -3) and flow[:, :, :, 3].sum() == -3 * 2 * 43 * 43


def test_flowfields_values(hm):
    from hydra_mi import synth
    assert synth.flowfields["translate_leftup"]((10.0, 20.0), 0) == (-1.5, -1.5)
    assert synth.flowfields["rotate"]((300.0, 325.0), 0) == (-1.0, 0.0)
    assert synth.flowfields["translate_leftup_stretch"]((600.0, 0.0), 0) == (1.0, -1.0)
    vx, vy = synth.flowfields["warp"]((150.0, 450.0), 0)
    assert abs(vx - (-(0.5 - 1) * 150 * (150 / 800. - 1) / 250.)) < 1e-12
    f0, f1, tu, tv = synth.warp_pair(48, "translate_leftup", 0)
    assert f0.dtype == np.uint8 and np.allclose(tu, -1.5 * 48 / 600)


# ---- filter host logic with the oracle as measurement backend ---------------------------------
class OracleRenderer:
    """Duck-type of renderer.Renderer on top of oracle/ekf_ref (tests only)."""

    def __init__(self, dm, tex, eps):
        self.meas = ekf_ref.Measurement(dm.size(), dm.t, dm.p, tex, *eps)
        self.force = None
        self.J = ekf_ref.adjacency(dm.size(), dm.t)[1]

    def setforce(self, f):
        self.force = f

    def update_frame(self, *a):
        pass

    def measure(self, state, y_im, y_flow, y_m, deltaX=2.0):
        X = state.X.reshape(-1)
        Hz, Hzc = ekf_ref.jacobian(self.meas, X, y_im, np.asarray(y_flow), y_m, deltaX)
        return Hz, ekf_ref.hessian_sparse(self.meas, X, self.J, deltaX), Hzc

    def error(self, state, y_im, y_flow, y_m, want_flow=True):
        return self.meas.error(state.X.reshape(-1), y_im, np.asarray(y_flow), y_m)


def _tiny_case(hm):
    from hydra_mi import mesh, synth
    video, flow = synth.test_data(48, 48)
    dm = mesh.box_mesh(16.0, 17.0, 31.0, 32.0, 8.0)
    return video, flow, dm


def test_filter_host_logic_matches_oracle_tracker(hm):
    from hydra_mi import kalman, mesh
    video, flow, dm = _tiny_case(hm)
    eps = (1e-3, 1.0, 1.0)
    dm2 = mesh.Mesh(dm.p, dm.t, dm.h0)
    kf = kalman.IteratedMSKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True, nI=2,
                                       renderer=OracleRenderer(dm, video[:, :, 0], eps))
    tr = ekf_ref.Tracker(dm2.p, dm2.t, dm2.bars, dm2.L, video[:, :, 0], nI=2)
    assert kf.size() == 4 * dm.size() and kf.state.X.shape == (kf.size(), 1)
    for k in (1, 2):
        frame = video[:, :, k]
        mask = (frame > 0).astype(np.uint8)
        e = kf.compute(frame, flow[:, :, :, k], mask)
        r = tr.compute(frame, flow[:, :, :, k], mask)
        X, Xr = kf.state.X.reshape(-1), tr.X.reshape(-1)
        assert np.linalg.norm(X - Xr) / np.linalg.norm(Xr) < 1e-9
        assert np.linalg.norm(kf.state.W - tr.W) / np.linalg.norm(tr.W) < 1e-7
        assert e[0] == r[0] and e[3] == r[3] and abs(e[1] - r[1]) < 1e-6 * max(1.0, r[1])
        assert kf.niter == tr.niter


def test_plain_and_iterated_filters(hm):
    """KalmanFilter (:626-765) and IteratedKalmanFilter (:767-831) defaults and one step each."""
    from hydra_mi import kalman
    video, flow, dm = _tiny_case(hm)
    frame, mask = video[:, :, 1], (video[:, :, 1] > 0).astype(np.uint8)
    kf = kalman.KalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True,
                             renderer=OracleRenderer(dm, video[:, :, 0], (1e-3, 1e-3, 1e-3)))
    assert (kf.state.eps_F, kf.state.eps_Z, kf.state.eps_J, kf.state.eps_M) == (1, 1e-3, 1e-3, 1e-3)
    N = kf.N
    X0, W0 = kf.state.X.copy(), kf.state.W.copy()
    kf.predict()
    F, Weps, _ = ekf_ref.initial_covariances(N, 1)
    assert np.allclose(kf.state.X, F @ X0) and np.allclose(kf.state.W, F @ W0 @ F.T + Weps)
    Xp, Wp = kf.state.X.copy(), kf.state.W.copy()
    kf.update(frame, ekf_ref.mask_flow(flow[:, :, :, 1], mask), mask)
    Xr, Wr = ekf_ref.kf_update(kf.state.renderer.meas, Xp, Wp, kf.state.J, frame,
                               ekf_ref.mask_flow(flow[:, :, :, 1], mask), mask)
    assert np.allclose(kf.state.X, Xr, rtol=1e-9, atol=1e-9) and np.allclose(kf.state.W, Wr, rtol=1e-7, atol=1e-12)
    ikf = kalman.IteratedKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True,
                                      renderer=OracleRenderer(dm, video[:, :, 0], (1e-3, 1e-3, 1e10)))
    assert (ikf.nI, ikf.reltol, ikf.state.eps_F, ikf.state.eps_M) == (10, 1e-4, 1e-3, 1e10)
    ms = kalman.IteratedMSKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True,
                                       renderer=OracleRenderer(dm, video[:, :, 0], (1e-3, 1, 1)))
    assert (ms.state.eps_F, ms.state.eps_J, ms.state.eps_M, ms.kappa, ms.deltat, ms.tol) == (1e-1, 1, 1, -1, 0.05, 1e-4)


def test_mesh_inversion_rolls_back(hm):
    """kalman.py:806-811: a step that flips a triangle is discarded, the last good state kept."""
    from hydra_mi import kalman
    video, flow, dm = _tiny_case(hm)

    class Exploding(OracleRenderer):
        def measure(self, state, y_im, y_flow, y_m, deltaX=2.0):
            Hz, HTH, Hzc = OracleRenderer.measure(self, state, y_im, y_flow, y_m, deltaX)
            Hz = Hz.copy()
            Hz[0] = 1e9                                   # throws vertex 0 far away
            return Hz, HTH, Hzc

    kf = kalman.IteratedKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True,
                                     renderer=Exploding(dm, video[:, :, 0], (1e-3, 1e-3, 1e10)))
    X0, W0 = kf.state.X.copy(), kf.state.W.copy()
    mask = (video[:, :, 0] > 0).astype(np.uint8)
    kf.update(video[:, :, 0], flow[:, :, :, 0], mask)
    assert kf.reverted and np.array_equal(kf.state.X, X0) and np.array_equal(kf.state.W, W0)


def test_state_structure_and_partitions(hm):
    from hydra_mi import kalman, mesh
    dm = mesh.disk_mesh(31.5, 31.5, 22.0, 9.0)
    tex = np.zeros((64, 64), np.uint8)
    kf = kalman.KalmanFilter(dm, tex, np.zeros((64, 64, 2), np.float32), True,
                             renderer=OracleRenderer(dm, tex, (1, 1, 1)))
    st = kf.state
    N = st.N
    assert np.array_equal(st.vertices(), dm.p.astype(np.float32)) and np.all(st.velocities() == 0)
    Jv, J = ekf_ref.adjacency(N, st.tri)
    assert np.array_equal(st.Jv, Jv) and np.array_equal(st.J, J)
    assert np.array_equal(st.K, ekf_ref.incidence(N, dm.bars))
    assert np.all(st.ori != 0) and kalman.stats.meshpts == N
    # every vertex in exactly one class, no two adjacent vertices share a class
    allv = sorted(v for e in st.E for v in e)
    assert allv == list(range(N))
    for e in st.E:
        for a in e:
            for b in e:
                assert a == b or Jv[a, b] == 0
    assert st.labels.shape == (len(st.tri), len(st.E))
    # pairs: Q lists every adjacent-or-equal pair once; each lands in a class -- an off-diagonal pair in
    # exactly one, a diagonal pair (i,i) in one or two (the reference schedules it again, see
    # oracle/partitions_ref.py and testbites/test_multipert_validation.py:330-332)
    assert len(st.Q) == int(np.triu(Jv).sum())
    idx = [int(i) for e in st.E_hessian_idx for i in e]
    assert sorted(set(idx)) == list(range(len(st.Q)))
    for i in set(idx):
        assert idx.count(i) == 1 or (idx.count(i) == 2 and st.Q[i][0] == st.Q[i][1])
    for pairs in st.E_hessian:
        verts = [set(p) for p in pairs.tolist()]
        for i in range(len(verts)):
            for j in range(i + 1, len(verts)):
                assert not any(Jv[a, b] for a in verts[i] for b in verts[j] if a != b) or verts[i] & verts[j] == set()


def _bare_state(kalman, dm):
    st = kalman.KFState.__new__(kalman.KFState)       # the partition builders need N, tri, Jv only
    st.N = dm.size()
    st.tri = np.asarray(dm.t)
    st.Jv = partitions_ref.adjacency(st.N, dm.t)
    return st


def test_partitions_hand_traced_square(hm):
    """The 4-vertex, 2-triangle square of the reference's unit-test fixture, traced by hand through
    reference kalman.py:223-272 and :305-389 (adjacency 0-1, 0-2, 1-2, 1-3, 2-3; 0-3 not adjacent)."""
    from hydra_mi import kalman, mesh
    st = _bare_state(kalman, mesh.square4_mesh(10, 30))
    E, labels = st._vertex_partitions()
    assert [list(map(int, e)) for e in E] == [[0, 3], [1], [2]]
    assert labels.tolist() == [[0, 1, 2], [3, 1, 2]]
    Q, EH, EHi, lh = st._pair_partitions()
    assert Q.tolist() == [[0, 0], [0, 1], [0, 2], [1, 1], [1, 2], [1, 3], [2, 2], [2, 3], [3, 3]]
    # (0,0) and (3,3) share no triangle and go first; every other pair clashes with all that remain;
    # the diagonal pairs (1,1) and (2,2) are scheduled twice (P is intersected with A before the current
    # pair leaves A, and a diagonal pair is its own p_self match)
    assert [e.tolist() for e in EH] == [[[0, 0], [3, 3]], [[0, 1]], [[0, 2]], [[1, 1]], [[1, 1]], [[1, 2]], [[1, 3]],
                                        [[2, 2]], [[2, 2]], [[2, 3]]]
    assert [list(map(int, e)) for e in EHi] == [[0, 8], [1], [2], [3], [3], [4], [5], [6], [6], [7]]
    assert lh.tolist() == [[0, 1, 2, 3, 3, 4, 5, 6, 6, 7], [8, 1, 2, 3, 3, 4, 5, 6, 6, 7]]


@pytest.mark.parametrize("which", ["config1", "disk", "box_fine"])
def test_partitions_equal_reference_restatement(hm, which):
    """E, labels, Q, E_hessian, E_hessian_idx, labels_hess of KFState are integer work: they equal the
    statement-by-statement restatement of reference kalman.py:223-272, 305-389 (oracle/partitions_ref.py)
    and, for the BASELINE config-1 mesh, the committed fixture."""
    from hydra_mi import kalman, mesh
    dm = {"config1": lambda: mesh.box_mesh(42.0, 43.0, 85.0, 86.0, 15.0),
          "disk": lambda: mesh.disk_mesh(31.5, 31.5, 22.0, 9.0),
          "box_fine": lambda: mesh.box_mesh(10.0, 12.0, 70.0, 50.0, 7.0)}[which]()
    st = _bare_state(kalman, dm)
    E, labels = st._vertex_partitions()
    Q, EH, EHi, lh = st._pair_partitions()
    rE, rlabels = partitions_ref.jacobian_partitions(st.N, dm.t)
    rQ, rEH, rEHi, rlh = partitions_ref.hessian_partitions(st.N, dm.t)
    assert [list(map(int, e)) for e in E] == rE and np.array_equal(labels, rlabels)
    assert np.array_equal(Q, rQ) and np.array_equal(lh, rlh)
    assert len(EH) == len(rEH) and all(np.array_equal(a, b) for a, b in zip(EH, rEH))
    assert len(EHi) == len(rEHi) and all(np.array_equal(a, b) for a, b in zip(EHi, rEHi))
    if which == "config1":
        g = np.load(os.path.join(ROOT, "tests", "golden", "partitions_config1.npz"))
        assert np.array_equal(dm.t, g["t"])
        assert np.array_equal(np.concatenate(E), g["E_flat"]) and np.array_equal([len(e) for e in E], g["E_len"])
        assert np.array_equal(labels, g["labels"]) and np.array_equal(Q, g["Q"]) and np.array_equal(lh, g["labels_hess"])
        assert np.array_equal(np.concatenate(EHi), g["EH_idx_flat"]) and np.array_equal([len(e) for e in EHi], g["EH_len"])


def test_projectmask_pulls_outliers_back(hm):
    from hydra_mi import kalman, mesh
    dm = mesh.square4_mesh(20, 40)
    tex = np.zeros((64, 64), np.uint8)
    kf = kalman.KalmanFilter(dm, tex, np.zeros((64, 64, 2), np.float32), True,
                             renderer=OracleRenderer(dm, tex, (1, 1, 1)))
    mask = np.zeros((64, 64), np.uint8)
    mask[20:41, 20:41] = 1
    X0 = kf.state.X.copy()
    kf.projectmask(mask)                                   # all vertices within 1 px: untouched
    assert np.array_equal(kf.state.X, X0)
    kf.state.X[0, 0] -= 6.0                                 # vertex 0 six pixels outside
    Xo = ekf_ref.project_mask(kf.state.X, 4, mask)
    kf.projectmask(mask)
    assert np.allclose(kf.state.X, Xo)
    assert kf.state.X[0, 0] > X0[0, 0] - 1.5 and kf.state.X[8, 0] > 0     # pulled back, velocity follows


def test_mask_distance_is_the_polygon_distance_of_the_reference(hm):
    """N3: the distance function of projectmask is the reference's fd (imgproc.py:195-235): the signed distance to the
    polygon through the centres of the object's border pixels (-cv2.pointPolygonTest).  The product's host form
    (imgproc.outline_distance, exact over all sides) equals the oracle's restatement bit for bit; for a mask the
    reference's contour pruning leaves alone (one object, holes >= 40 px) it equals findObjectThreshold(mask).fd, the
    restated reference function, for points outside the object; known answers on a square."""
    from hydra_mi import imgproc, kalman
    rng = np.random.default_rng(5)
    H, W = 90, 120
    yy, xx = np.mgrid[:H, :W]
    m = (((xx - 50) ** 2 + (yy - 40) ** 2 < 20 ** 2) | ((xx > 95) & (yy < 30) & (xx - 95 + 30 - yy > 8))).astype(np.uint8)
    m[36:44, 46:54] = 0                                                                        # a hole of 64 px
    p = np.column_stack((rng.uniform(-8, W + 8, 400), rng.uniform(-8, H + 8, 400)))
    p[:20] = np.round(p[:20])                                                                  # on pixel centres
    got = kalman._mask_distance(m)(p)
    want = ekf_ref.outline_distance(m)(p)
    assert np.array_equal(got, want)
    one = (xx - 50) ** 2 + (yy - 40) ** 2 < 20 ** 2                                            # a single object with its hole
    one = one & ~((yy >= 36) & (yy < 44) & (xx >= 46) & (xx < 54))
    ref_fd = imgproc.findObjectThreshold(np.where(one, 200, 0).astype(np.uint8), 7)[2]
    mine = kalman._mask_distance(one.astype(np.uint8))
    outside = mine(p) > 0
    assert outside.sum() > 100
    assert np.allclose(mine(p)[outside], ref_fd(p)[outside], rtol=0, atol=1e-12)
    assert np.array_equal(np.sign(mine(p)), np.sign(ref_fd(p)))
    # known answers: the square [20, 40]^2 of pixel centres
    sq = np.zeros((64, 64), np.uint8)
    sq[20:41, 20:41] = 1
    fd = kalman._mask_distance(sq)
    q = np.array([[10.0, 30.0], [30.0, 45.5], [17.0, 16.0], [30.0, 30.0], [20.0, 25.0], [42.0, 43.0], [-3.0, 30.0]])
    assert np.allclose(fd(q), [10.0, 5.5, 5.0, -10.0, 0.0, np.sqrt(13.0), 23.0], rtol=0, atol=1e-12)
    for blank in (np.zeros_like(m),):
        assert np.array_equal(kalman._mask_distance(blank)(p), np.zeros(len(p)))


def test_contour_pruning_follows_the_reference(hm):
    """N3, reference imgproc.py:198-228 (called per frame from kalman.py:725): of the contours cv2.findContours(RETR_TREE)
    finds, the largest level-0 one stays with its level-1 holes of cv2.contourArea >= 40; smaller objects, smaller holes
    and everything deeper go.  Known answers for oracle/ekf_ref.pruned_object (areas of polygons through pixel centres,
    worked out by hand), and the product's host form (imgproc._object_and_holes, written independently) equal to it on
    random masks with specks, pinholes, holes with islands and objects on the frame edge."""
    from hydra_mi import imgproc
    m = np.zeros((60, 80), bool)
    m[5:45, 5:55] = True                       # the object: 40 x 50 pixels
    m[10:15, 10:15] = False                    # 5 x 5 hole: contour through the 20 pixels around it, area 6 x 6 - 4 / 2 = 34 -> filled
    m[12, 12] = True                           # ... with an island in it: filled with the hole
    m[20:25, 10:16] = False                    # 5 x 6 hole: area 6 x 7 - 2 = 40 -> kept
    m[30:38, 30:40] = False                    # 8 x 10 hole, kept, with an object inside: the object goes (level 2)
    m[33:35, 33:36] = True
    m[50:58, 60:78] = True                     # a second object, 8 x 18 = 144 px (contour area 7 x 17 = 119): goes
    m[2, 70] = True                            # a speck
    want = np.zeros_like(m)
    want[5:45, 5:55] = True
    want[20:25, 10:16] = False
    want[30:38, 30:40] = False
    assert np.array_equal(ekf_ref.pruned_object(m), want)
    assert np.array_equal(imgproc._object_and_holes(m), want)
    # cv2.contourArea of the OUTER contour decides, not the pixel count: a ring of 232 pixels around 784 pixels of
    # background (area 29 x 29 = 841) beats a solid square of 400 pixels (area 19 x 19 = 361) -- and its inside is a kept hole
    r = np.zeros((70, 90), bool)
    r[5:35, 5:35] = True
    r[7:33, 7:33] = False
    r[40:60, 50:70] = True
    want = np.zeros_like(r)
    want[5:35, 5:35] = True
    want[7:33, 7:33] = False
    assert np.array_equal(ekf_ref.pruned_object(r), want) and np.array_equal(imgproc._object_and_holes(r), want)
    # background that reaches the frame edge is outside, not a hole (findContours: the frame is surrounded by background)
    u = np.zeros((40, 40), bool)
    u[0:30, 5:35] = True
    u[0:20, 12:28] = False                     # a pocket open towards the top edge
    assert np.array_equal(ekf_ref.pruned_object(u), u) and np.array_equal(imgproc._object_and_holes(u), u)
    # nothing of area >= 40: nothing is kept (the reference fails with an IndexError there)
    tiny = np.zeros((20, 20), bool)
    tiny[3:9, 3:9] = True                      # 36 px, area 25
    assert not ekf_ref.pruned_object(tiny).any() and not imgproc._object_and_holes(tiny).any()
    assert not ekf_ref.pruned_object(np.zeros((9, 9), bool)).any()
    rng = np.random.default_rng(11)
    for trial in range(40):
        H, W = int(rng.integers(24, 90)), int(rng.integers(24, 130))
        mm = _blobs(rng, H, W, int(rng.integers(1, 5)))
        assert np.array_equal(imgproc._object_and_holes(mm), ekf_ref.pruned_object(mm)), trial
    # and the distance functions built on them agree, pruning included
    mm = _blobs(np.random.default_rng(5), 70, 100, 3)
    p = np.column_stack((rng.uniform(-5, 105, 200), rng.uniform(-5, 75, 200)))
    assert np.array_equal(imgproc.outline_distance(mm)(p), ekf_ref.outline_distance(mm)(p))
    assert not np.array_equal(ekf_ref.outline_distance(mm)(p), ekf_ref.outline_distance(mm, prune=False)(p))      # (it matters here)


def test_signed_distance_and_distmesh(hm, tmp_path):
    """N3: imgproc.findObjectThreshold (mask, kept contours, signed distance to the outline -- reference
    imgproc.py:175-248) and DistMesh.createMesh (distmesh_dyn.py:42-139) on shapes whose distance is known."""
    from hydra_mi import distmesh_dyn, imgproc
    n = 160
    yy, xx = np.mgrid[:n, :n]
    img = np.zeros((n, n), np.uint8)
    img[30:121, 40:131] = 200                                       # a square object, rows 30..120, columns 40..130
    img[70:81, 80:91] = 0                                           # with an 11 x 11 hole (kept: >= 40 px)
    img[50:53, 60:63] = 0                                           # a 3 x 3 pinhole (dropped: < 40 px)
    img[5:12, 5:12] = 90                                            # and a second, smaller object (dropped)
    mask, ctrs, fd = imgproc.findObjectThreshold(img, 9)
    assert np.array_equal(mask, (img > 9).astype(np.uint8))
    assert ctrs.nC == 2 and [lv for _, lv in ctrs.traverse()] == [0, 1]      # outer outline + the one real hole
    # exact outside the object: the polygon runs through the centres of the boundary pixels
    q = np.array([[20.0, 75.0], [140.5, 75.0], [85.0, 10.0], [30.0, 20.0], [85.0, 75.0], [8.0, 8.0], [61.0, 51.0], [84.5, 60.0]])
    want = np.array([20.0, 10.5, 20.0, np.hypot(10.0, 10.0), 6.0, np.hypot(32.0, 22.0), None, None], dtype=object)
    got = fd(q)
    for g, w in zip(got[:4], want[:4]):
        assert abs(g - w) < 1e-12
    assert abs(got[4] - 6.0) < 1e-12          # centre of the hole: outside the object, 6 px from the hole's border pixels (79 / 91)
    assert abs(got[5] - want[5]) < 1e-12      # the small object does not count
    assert got[6] < 0 and got[7] < 0          # the pinhole is filled; a point of the body is inside
    # signs against the filled region on a dense sample, magnitudes against the exact distance outside
    rng = np.random.default_rng(1)
    p = rng.uniform(0, n - 1, (4000, 2))
    d = fd(p)
    sq = lambda x, y: np.maximum(np.maximum(40 - x, x - 130), np.maximum(30 - y, y - 120))
    hole = lambda x, y: np.maximum(np.maximum(79 - x, x - 91), np.maximum(69 - y, y - 81))       # < 0 inside the hole's polygon
    outside_sq = (sq(p[:, 0], p[:, 1]) > 0)
    ex = np.hypot(np.maximum(0, np.maximum(40 - p[:, 0], p[:, 0] - 130)), np.maximum(0, np.maximum(30 - p[:, 1], p[:, 1] - 120)))
    assert np.abs(d[outside_sq] - ex[outside_sq]).max() < 1e-9
    in_hole = hole(p[:, 0], p[:, 1]) < 0
    assert np.all(d[in_hole] > 0) and np.all(d[~outside_sq & ~in_hole & (sq(p[:, 0], p[:, 1]) < -1e-9) & (hole(p[:, 0], p[:, 1]) > 1e-9)] < 0)
    # DistMesh on a disk: deterministic, inside the object, near-uniform bars, well-shaped triangles
    disk = ((xx - 79.5) ** 2 + (yy - 79.5) ** 2 <= 50.0 ** 2).astype(np.uint8) * 180
    mask, ctrs, fd = imgproc.findObjectThreshold(disk, 9)
    dm = distmesh_dyn.DistMesh(disk, h0=16)
    dm.createMesh(ctrs, fd, disk)
    dm2 = distmesh_dyn.DistMesh(disk, h0=16)
    dm2.createMesh(ctrs, fd, disk)
    assert np.array_equal(dm.p, dm2.p) and np.array_equal(dm.t, dm2.t)
    assert 30 <= dm.size() <= 60 and dm.iterations < dm.maxiter
    assert fd(dm.p).max() < 1.0                                      # vertices on or inside the outline (1 px)
    assert np.array_equal(np.unique(dm.t), np.arange(dm.size()))     # every vertex is used
    a, b = dm.p[dm.t[:, 1]] - dm.p[dm.t[:, 0]], dm.p[dm.t[:, 2]] - dm.p[dm.t[:, 0]]
    cr = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    assert np.abs(cr / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)).min() > 0.4
    assert np.abs(cr).sum() / 2 > 0.93 * np.pi * 50.0 ** 2           # the mesh covers the disk
    assert dm.L.min() > 0.6 * 16 and dm.L.max() < 2.2 * 16
    bars = np.unique(np.sort(np.vstack((dm.t[:, [0, 1]], dm.t[:, [1, 2]], dm.t[:, [2, 0]])), axis=1), axis=0)
    # (L are the lengths before the last move of the points, as in the reference, :104 / :119: the filter measures its own)
    assert np.array_equal(bars, dm.bars) and np.allclose(dm.L, np.linalg.norm(dm.p[bars[:, 0]] - dm.p[bars[:, 1]], axis=1), rtol=2e-2)
    # the 19 pickled fields round-trip (distmesh_dyn.py:205-222)
    fn = str(tmp_path / "mesh.pkl")
    dm.save(fn)
    dm3 = distmesh_dyn.DistMesh(disk, h0=16)
    dm3.load(fn)
    assert np.array_equal(dm3.p, dm.p) and np.array_equal(dm3.t, dm.t) and dm3.size() == dm.size()
    # updateMesh: the outline moves by 3 px, the mesh follows without re-triangulating
    disk2 = ((xx - 82.5) ** 2 + (yy - 79.5) ** 2 <= 50.0 ** 2).astype(np.uint8) * 180
    _, ctrs2, fd2 = imgproc.findObjectThreshold(disk2, 9)
    nt = len(dm.t)
    dm.updateMesh(ctrs2, fd2, disk2)
    assert fd2(dm.p).max() < 1.0 and len(dm.t) <= nt


def test_videostream_and_shared_gray_conversion(hm, tmp_path):
    """VideoStream mirrors reference renderer.py:739-805 on an array source; a colour video is converted the way
    cvtColor(BGR2GRAY) does (renderer.py:752, src/optical_flow_ext.cpp:366-368) by ONE function both command
    line tools use, so the tracker sees the frames its flow files were computed on."""
    from hydra_mi import pipeline
    rng = np.random.default_rng(0)
    col = rng.integers(0, 256, (4, 12, 10, 3), dtype=np.uint8)
    fn = str(tmp_path / "colour.npy")
    np.save(fn, col)
    gray = pipeline.load_video(fn)
    want = np.rint(0.114 * col[..., 0] + 0.587 * col[..., 1] + 0.299 * col[..., 2]).astype(np.uint8)
    assert gray.dtype == np.uint8 and np.array_equal(gray, want)
    assert np.array_equal(pipeline.to_gray(gray), gray)                   # gray stays gray
    sys.path.insert(0, ROOT)
    import inspect
    import optical_flow_ext
    import run_kalmanfilter
    assert "pipeline.load_video" in inspect.getsource(optical_flow_ext.main)
    assert "VideoStream(args.fn_in" in inspect.getsource(run_kalmanfilter.main)
    vs = pipeline.VideoStream(fn, 100)
    assert (vs.nx, vs.ny) == (12, 10) and vs.isOpened()
    mask, ctrs, fd = vs.backsub()
    assert np.array_equal(mask, (want[0] > 100).astype(np.uint8)) and callable(fd)
    assert np.array_equal(vs.current_frame(), want[0] * mask) and np.array_equal(vs.current_frame(False), want[0])
    assert np.array_equal(vs.backsub(col[0]), mask[:, :, None] * col[0])
    for k in (1, 2, 3):
        ret, frame, grayframe, m = vs.read()
        assert ret and np.array_equal(m, (want[k] > 100).astype(np.uint8)) and np.array_equal(grayframe, want[k] * m)
    ret, frame, grayframe, m = vs.read()
    assert ret is False and frame is None and not vs.isOpened()
    with pytest.raises(ValueError):
        np.save(fn, col.astype(np.float32))
        pipeline.load_video(fn)


def test_video_sources_and_native_flow_tool_cpu_side(hm, tmp_path):
    """N2: load_video reads .npy, .npz and multi-page TIFF stacks (gray and RGB) to the same gray frames; the native flow
    tool (csrc/optical_flow_ext.cpp, the reference's C++ CLI over the C-ABI) is built, prints its usage with exit code
    1 without arguments and refuses a missing / malformed video with the reference's kind of message."""
    import subprocess
    from PIL import Image
    from hydra_mi import pipeline
    rng = np.random.default_rng(4)
    gray = rng.integers(0, 256, (3, 10, 12), dtype=np.uint8)
    bgr = rng.integers(0, 256, (3, 10, 12, 3), dtype=np.uint8)
    np.save(str(tmp_path / "g.npy"), gray)
    np.savez(str(tmp_path / "g.npz"), gray)
    pages = [Image.fromarray(f) for f in gray]
    pages[0].save(str(tmp_path / "g.tif"), save_all=True, append_images=pages[1:])
    pages = [Image.fromarray(np.ascontiguousarray(f[..., ::-1])) for f in bgr]              # TIFF pages are RGB
    pages[0].save(str(tmp_path / "c.tiff"), save_all=True, append_images=pages[1:])
    for fn in ("g.npy", "g.npz", "g.tif"):
        assert np.array_equal(pipeline.load_video(str(tmp_path / fn)), gray), fn
    assert np.array_equal(pipeline.load_video(str(tmp_path / "c.tiff")), pipeline.to_gray(bgr))
    exe = os.path.join(ROOT, "kalman-hydra_amd", "optical_flow_ext")
    assert os.access(exe, os.X_OK), "run `python __graft_entry__.py build`"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stdout and "[alpha] [gamma] [scale_factor] [inner_it] [outer_it] [solver_it]" in r.stdout
    r = subprocess.run([exe, str(tmp_path / "missing.npy"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to open the video" in r.stderr and "alpha = 0.197" in r.stdout
    np.save(str(tmp_path / "f.npy"), gray.astype(np.float32))
    r = subprocess.run([exe, str(tmp_path / "f.npy"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 1 and "uint8" in r.stderr


# ---- sharding over ranks (gloo, world_size 2) ---------------------------------------------------
def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import hydra_mi  # noqa: F401
    from hydra_mi import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 7
    f0 = np.arange(n * 4 * 5, dtype=np.uint8).reshape(n, 4, 5)
    f1 = f0 + 1

    def fake_flow(a, b):
        return torch.from_numpy(a.astype(np.float32) * 2 + rank * 0), torch.from_numpy(b.astype(np.float32) - 3)

    u, v = batch.flow_batch_sharded(f0, f1, fake_flow)
    ok = bool(torch.equal(u, torch.from_numpy(f0.astype(np.float32) * 2))
              and torch.equal(v, torch.from_numpy(f1.astype(np.float32) - 3)))
    # the gather the bench ends with: unequal blocks to rank 0 only
    ub, _, mine_r = batch.flow_batch_sharded(f0, f1, fake_flow, gather=False)
    g = batch.gather_to_root(ub, n, dst=0)
    ok = ok and ((g is None) if rank != 0 else bool(torch.equal(g, u)))
    mine = batch.shard(n, rank, world)
    out.put((rank, ok, list(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_two_ranks_gloo(hm):
    import torch.multiprocessing as mp
    from hydra_mi import batch
    assert [list(batch.shard(7, r, 2)) for r in range(2)] == [[0, 1, 2, 3], [4, 5, 6]]
    assert sum(len(batch.shard(256, r, 8)) for r in range(8)) == 256 and len(batch.shard(256, 3, 8)) == 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res[0][1] and res[1][1]
    assert res[0][2] == [0, 1, 2, 3] and res[1][2] == [4, 5, 6]


def test_bench_flowbatch_two_ranks_gloo_stub_flow(hm):
    """bench.py's BASELINE config 5 path -- batch.shard of the pair indices (pair i = seed i), one block per rank, no
    exchange while computing, batch.gather_to_root of all flow planes, one JSON line from rank 0 -- launched the way the
    driver launches it (torch.distributed.run, 2 ranks), on gloo with a CPU stand-in for the flow kernel."""
    import json
    import subprocess
    port = 29700 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "flowbatch", "--backend", "gloo", "--stub-flow", "--size", "48", "--pairs-per-gpu", "3"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["unit"] == "pairs/sec" and out["stub_flow"] is True and out["roofline"] is None
    d = out["distributed"]
    assert d["world"] == 2 and d["backend"] == "gloo" and [r["rank"] for r in d["ranks"]] == [0, 1]
    assert d["gathered_bytes"] == 6 * 2 * 48 * 48 * 4                       # every pair's two planes reached rank 0
    assert "seeds 0..5" in out["config"]["workload"] and out["value"] > 0


def test_bench_starts_its_own_ranks_or_refuses(hm):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE in the environment) must not print a
    one-rank line labelled as N were asked for: it starts torch.distributed.run --nproc-per-node N itself, as a child
    process, before it has imported anything that touches the GPU, and leaves with the child's exit code; under a
    launcher whose WORLD_SIZE is not N it refuses to run."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "flowbatch",
           "--backend", "gloo", "--stub-flow", "--size", "32", "--pairs-per-gpu", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "launching" in res.stderr and "torch.distributed.run" in res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["distributed"]["world"] == 2
    # a launcher with another world size: refused, with the command to run in the message
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "does not match WORLD_SIZE 3" in (bad.stderr + bad.stdout)
    # one rank asked for, one rank run: no launcher needed
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--workload",
                          "flowbatch", "--backend", "gloo", "--stub-flow", "--size", "32", "--pairs-per-gpu", "2"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert one.returncode == 0 and json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_newton_worker_equals_synchronous_call(hm):
    """hm_ms_newton_start / _finish (the next frame's state prediction on a host thread, include/hydra_mi.h) give
    the numbers and the iteration count of hm_ms_newton; a worker takes one job at a time and can be reused."""
    import ctypes
    from hydra_mi import _lib, mesh
    dm = mesh.disk_mesh(64, 64, 50, 12.0)
    N = dm.size()
    bars = np.ascontiguousarray(dm.bars, np.int32)
    l0 = np.linalg.norm(dm.p[bars[:, 0]] - dm.p[bars[:, 1]], axis=1)
    rng = np.random.default_rng(0)
    X0 = np.concatenate((dm.p.reshape(-1) + rng.normal(0, .3, 2 * N), rng.normal(0, .3, 2 * N)))
    L = _lib.lib()
    args = (N, len(bars), _lib.ptr(bars), _lib.ptr(l0), -1.0, 1.0, 0.05, 1000, 1e-4)
    X, its = X0.copy(), ctypes.c_int()
    _lib.check(L.hm_ms_newton(*args, _lib.ptr(X), ctypes.byref(its)), "hm_ms_newton")
    assert its.value >= 20 and not np.array_equal(X, X0)
    w = ctypes.c_void_p()
    _lib.check(L.hm_ms_worker_create(ctypes.byref(w)), "create")
    try:
        Y, its2 = np.empty_like(X0), ctypes.c_int()
        assert L.hm_ms_newton_finish(w, _lib.ptr(Y), ctypes.byref(its2)) != 0          # nothing started yet
        for rep in range(3):
            _lib.check(L.hm_ms_newton_start(w, *args, _lib.ptr(X0)), "start")
            if rep == 0:                                                              # a second start supersedes the first
                Z0 = X0 + 1e-3
                _lib.check(L.hm_ms_newton_start(w, *args, _lib.ptr(Z0)), "start")
                _lib.check(L.hm_ms_newton_finish(w, _lib.ptr(Y), ctypes.byref(its2)), "finish")
                assert not np.array_equal(X, Y)
                _lib.check(L.hm_ms_newton_start(w, *args, _lib.ptr(X0)), "start")
            _lib.check(L.hm_ms_newton_finish(w, _lib.ptr(Y), ctypes.byref(its2)), "finish")
            assert np.array_equal(X, Y) and its.value == its2.value
    finally:
        L.hm_ms_worker_destroy(w)


def test_flow_series_are_sized_from_the_measured_model(hm):
    """FlowEKFPipeline._first_series / _next_series (host logic, no GPU): with the calibrated times of a series of 2 and of B
    pairs and the filter's frame time, a phase starts with the smallest series the ramp can follow and every next series
    is the largest one that is done (beside the filter: 1.3 x) when the filter is through with the last, at least one
    pair more, never more than B; without measurements the fixed ramp 2, 3, 5, 8."""
    from hydra_mi.pipeline import FlowEKFPipeline
    p = object.__new__(FlowEKFPipeline)
    p.B, p.concurrent_series, p.first_series, p.adaptive_first, p.model_ramp = 8, False, 0, True, True
    p._series_s, p._frame_s = {}, None
    assert p._first_series() == 2 and [p._next_series(n) for n in (2, 3, 5, 8)] == [3, 5, 8, 8]
    # a series of n pairs alone: 3.9 + 1.46 n ms (1024^2 on an MI355X), a frame of the filter 3.2 ms
    p._series_s = {2: 6.82e-3, 8: 15.58e-3}
    p._frame_s = 3.2e-3
    n1 = p._first_series()
    assert n1 == 4                                           # 1.3 x T(4) = 12.7 ms <= 4 x 3.2 ms, 1.3 x T(3) = 10.8 > 9.6
    sizes = [n1]
    while sizes[-1] < p.B:
        sizes.append(p._next_series(sizes[-1]))
    assert sizes == [4, 5, 6, 7, 8]
    assert p._next_series(8) == 8
    p._frame_s = 10e-3                                       # a slow filter: the flow is never waited for, series double
    assert p._first_series() == 2 and p._next_series(2) == 4 and p._next_series(4) == 8
    p.B = 16
    p._frame_s = 3.2e-3
    assert [p._next_series(n) for n in (8, 10, 14)] == [10, 14, 16]
    p.model_ramp = False                                     # the fixed rule again
    assert p._next_series(4) == 6


def test_concurrent_series_are_sized_from_what_is_queued_in_front_of_them(hm):
    """FlowEKFPipeline._next_concurrent (host logic, no GPU): with two handles the next series is the largest one that is done
    (1.6 x its time alone and what the series in flight still have to do: they share the chip) when the filter has used up
    the pairs that are ready and the series in flight; never smaller
    than the series in front of it, larger when that one is overdue, None without measurements."""
    import time
    from hydra_mi.pipeline import FlowEKFPipeline
    p = object.__new__(FlowEKFPipeline)
    p.B, p.concurrent_series, p.first_series, p.adaptive_first, p.model_ramp, p.trace = 8, True, 0, True, True, None
    p._series_s, p._frame_s, p._flying, p._ready, p._cursor, p._flow_late = {}, None, [], (0, 0), 0, False
    assert p._next_concurrent() is None and p._first_series() == 1
    p._series_s, p._frame_s = {2: 6.82e-3, 8: 15.58e-3}, 3.7e-3          # a series of n pairs alone: 6.82 + 1.46 (n - 2) ms
    now = time.perf_counter()
    # the start of a phase: one pair in flight since just now -- needed in 1.6 x 5.36 + 3.7 = 12.3 ms: 1.6 x T(2) = 10.9 fits
    p._flying = [{"lo": 0, "hi": 1, "t0": now}]
    assert p._next_concurrent() == 2
    # two pairs ready and unused, a series of two launched 7 ms ago (2.4 ms of its work left): needed in 14.8 ms, and the
    # chip is shared with what that series still has to do: 1.6 x (2.4 + T(2)) = 14.8 does not quite fit -- as large as the last
    p._ready, p._cursor = (1, 3), 1
    p._flying = [{"lo": 3, "hi": 5, "t0": now - 7e-3}]
    assert p._next_concurrent() == 2
    p._flying = [{"lo": 3, "hi": 5, "t0": now - 11e-3}]      # ... launched 11 ms ago it is done: 1.6 x T(3) = 13.2 fits
    assert p._next_concurrent() == 3
    # full series: 8 ready, 8 in flight: a series of 8 fits many times over
    p._ready, p._cursor = (10, 18), 10
    p._flying = [{"lo": 18, "hi": 26, "t0": now - 5e-3}]
    assert p._next_concurrent() == 8
    # nothing ready, and the series in flight (3 pairs) should have been done long ago: the flow is late -- a larger series
    p._ready, p._cursor = (5, 5), 5
    p._flying = [{"lo": 5, "hi": 8, "t0": now - 1.0}]
    assert p._next_concurrent() == 5
    p._flying = [{"lo": 5, "hi": 6, "t0": now - 1.0}]
    assert p._next_concurrent() == 2
    # never a series of one pair behind the opening one, whatever fits (a slow calibration, a fast filter)
    p._ready, p._cursor, p._flying, p._frame_s = (5, 6), 5, [{"lo": 6, "hi": 7, "t0": now}], 0.5e-3
    assert p._next_concurrent() == 2
    p._frame_s = 3.7e-3
    # a filter that converges in an iteration per frame (0.9 ms): nothing "fits", the series stay as large as the last one,
    # also when both handles are free (nothing in flight: the series the filter works through counts) ...
    p._frame_s = 0.9e-3
    p._ready, p._cursor, p._flying = (40, 48), 47, []
    assert p._next_concurrent() == 8
    p._ready, p._cursor, p._flying = (40, 43), 42, []
    assert p._next_concurrent() == 3
    # ... and grow when the filter has just waited for the flow
    p._flow_late = True
    assert p._next_concurrent() == 5
