"""Parity at BASELINE.json's full sizes (1024^2 frames, ~200-vertex mesh, config 4): the flow against
the C oracle (it finishes a 1024^2 pair in seconds); the fused measurement against the oracle's
jz / j (oracle/ekf_ref_c.c, the C twin of ekf_ref.Measurement: one full-frame render per
perturbation as the reference's CPU path does, cuda.py:972-1010) on a sample of components and on a
whole vertex, and against the product's own fine-grained operators; symmetry and sparsity of HTH;
the information-form identity of the update."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_FULL = 1024


def test_brox_full_size_pair_matches_oracle(hm, oracle_brox):
    from hydra_mi import brox, synth
    f0, f1, tu, tv = synth.warp_pair(N_FULL, "translate_leftup_stretch", 3)          # a BASELINE config 5 pair
    bf = brox.BroxOpticalFlow(N_FULL, N_FULL, max_batch=2)
    assert bf.levels() == oracle_brox.levels(N_FULL, N_FULL)
    u, v = bf.calc(f0, f1)
    ru, rv = oracle_brox.calc(f0, f1)
    assert np.sqrt((u - ru) ** 2 + (v - rv) ** 2).max() <= 1e-4                        # the contract
    assert np.array_equal(u, ru) and np.array_equal(v, rv)                            # what is achieved
    b = N_FULL // 8
    # and it is the flow: mean end-point error against the analytic field over the interior, C oracle measured
    # 0.04721 px for this pair -- bound = measured + 10 % (a change of the algorithm would show here)
    assert np.sqrt((u - tu) ** 2 + (v - tv) ** 2)[b:-b, b:-b].mean() <= 0.0519
    # a series of pairs gives each pair the numbers it gets alone
    g0, g1, _, _ = synth.warp_pair(N_FULL, "rotate", 4)
    ub, vb = bf.calc_batch(np.stack((f0, g0)), np.stack((f1, g1)))
    assert np.array_equal(ub[0], u) and np.array_equal(vb[0], v)
    u2, v2 = bf.calc(g0, g1)
    assert np.array_equal(ub[1], u2) and np.array_equal(vb[1], v2)


def _scene():
    from hydra_mi import mesh, renderer, synth
    n = N_FULL
    video, masks, c, r = synth.disk_video(n, 2, "translate_leftup", 0)
    dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)                                # the bench's mesh: 201 vertices
    N = dm.size()
    R = renderer.Renderer(dm, np.zeros((N, 2)), np.zeros((n, n, 2), np.float32), n, video[0], True, 1e-3, 1.0, 1.0)
    rng = np.random.default_rng(1)
    flow = np.zeros((n, n, 2), np.float32) - 2.5 + rng.normal(0, 0.05, (n, n, 2)).astype(np.float32)
    X = np.concatenate((dm.p.reshape(-1) + rng.normal(0, 0.4, 2 * N), rng.normal(-2.5, 0.3, 2 * N)))
    return dm, N, R, video[1], flow, masks[1], X


class _S:
    pass


def test_measure_full_size_against_oracle(hm):
    """Hz and HTH of the fused kernels at 1024^2 / 201 vertices against the oracle's central / forward
    differences of full-frame renders (kalman.py:499-515, 595-603): 24 random components of Hz, all four
    of one vertex with their channel split, the 4x4 diagonal block of that vertex, the 4x4 block of an
    adjacent pair, and a non-adjacent pair (exactly zero)."""
    from oracle import ekf_c
    dm, N, R, y_im, flow, y_m, X = _scene()
    st = _S()
    st.X = X.reshape(-1, 1)
    R.update_frame(y_im, flow, y_m)
    Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
    meas = ekf_c.Measurement(N, dm.t, dm.p, R.current_frame, 1e-3, 1.0, 1.0, threads=min(16, os.cpu_count() or 1))
    rng = np.random.default_rng(11)
    v, w = int(dm.t[11][0]), int(dm.t[11][1])                                         # an adjacent pair
    own = [2 * v, 2 * v + 1, 2 * N + 2 * v, 2 * N + 2 * v + 1]
    oth = [2 * w, 2 * w + 1, 2 * N + 2 * w, 2 * N + 2 * w + 1]
    idx = np.array(sorted(set(rng.choice(4 * N, 24, replace=False).tolist() + own)), np.int32)
    rHz, rHzc = meas.jacobian_all(X, y_im, flow, y_m, 2.0, idx)
    scale = np.abs(Hz).max()
    assert np.abs(Hz[idx, 0] - rHz[:, 0]).max() <= 1e-9 * scale
    assert np.abs(Hzc[idx] - rHzc).max() <= 1e-9 * np.abs(Hzc).max()
    adj = np.eye(N, dtype=bool)
    for a, b, c in dm.t:
        adj[a, b] = adj[b, a] = adj[a, c] = adj[c, a] = adj[b, c] = adj[c, b] = True
    far = int(np.flatnonzero(~adj[v])[0])
    pairs = [(i, j) for i in own for j in own if j >= i] + [(i, j) for i in own for j in oth] + [(2 * v, 2 * far)]
    vals = meas.hessian_pairs([p[0] for p in pairs], [p[1] for p in pairs], 2.0)
    hs = np.abs(HTH).max()
    for (i, j), ref in zip(pairs, vals):
        assert abs(HTH[i, j] - ref) <= 1e-9 * max(hs * 1e-3, abs(ref)), (i, j, HTH[i, j], ref)
        assert HTH[j, i] == HTH[i, j]
    assert vals[-1] == 0.0 and HTH[2 * v, 2 * far] == 0.0
    e = R.error(st, y_im, flow, y_m)
    r = meas.error(X, y_im, flow, y_m)
    assert e[0] == r[0] and e[3] == r[3] and abs(e[1] - r[1]) <= 1e-10 * r[1] and abs(e[2] - r[2]) <= 1e-10 * r[2]
    assert np.array_equal(e[4][:, :, 0], r[4]) and np.array_equal(e[5][:, :, 0], r[5])


def test_measure_full_size_against_fine_grained_operators(hm):
    dm, N, R, y_im, flow, y_m, X = _scene()
    assert 180 <= N <= 220
    st = _S()
    st.X = X.reshape(-1, 1)
    R.update_frame(y_im, flow, y_m)
    Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
    n4 = 4 * N
    assert np.array_equal(HTH, HTH.T) and np.all(np.diag(HTH) >= 0)
    # sparsity: non-adjacent vertices have disjoint supports (kalman.py:202-205)
    adj = np.eye(N, dtype=bool)
    for a, b, c in dm.t:
        adj[a, b] = adj[b, a] = adj[a, c] = adj[c, a] = adj[b, c] = adj[c, b] = True
    J = np.kron(np.ones((2, 2), bool), np.kron(adj, np.ones((2, 2), bool)))
    assert not HTH[~J].any() and np.abs(HTH[J]).max() > 0
    assert np.allclose(Hzc.sum(axis=1), Hz[:, 0], rtol=1e-12, atol=1e-9 * np.abs(Hz).max())
    # the fused star kernel against one full-frame render per perturbation (hm_jz / hm_j)
    R.update_vertex_buffer(X[:2 * N].reshape(-1, 2), X[2 * N:].reshape(-1, 2))
    R.initjacobian(y_im, flow, y_m)
    rng = np.random.default_rng(5)
    for p in rng.choice(n4, 6, replace=False):
        s = _S()
        Xp, Xm = X.copy(), X.copy()
        Xp[p] += 2.0
        Xm[p] -= 2.0
        s.X = Xp.reshape(-1, 1)
        jp, _ = R.jz(s)
        s.X = Xm.reshape(-1, 1)
        jm, _ = R.jz(s)
        ref = (jp - jm) / 4.0
        assert abs(Hz[p, 0] - ref) <= 1e-9 * max(1.0, abs(ref), np.abs(Hz).max() * 1e-3), p
    st.X = X.reshape(-1, 1)
    v, w = int(dm.t[7][0]), int(dm.t[7][1])                                           # an adjacent pair
    far = int(np.flatnonzero(~adj[v])[0])
    for (i, j) in [(2 * v, 2 * v), (2 * v, 2 * v + 1), (2 * v, 2 * w + 1), (2 * v + 1, 2 * N + 2 * w),
                   (2 * N + 2 * v, 2 * N + 2 * v), (2 * v, 2 * far)]:
        ref = R.j(st, 2.0, i, j) / 2.0 / 2.0                                          # kalman.py:598
        assert abs(HTH[i, j] - ref) <= 1e-9 * max(1.0, abs(ref)), (i, j)


def test_update_full_size_information_identity(hm):
    """One iteration of the update from X0: W1 = inv(inv(W0) + HTH(X0)), X1 = X0 + W1 Hz(X0)
    (kalman.py:785-799 with X = X0)."""
    dm, N, R, y_im, flow, y_m, X0 = _scene()
    n4 = 4 * N
    rng = np.random.default_rng(9)
    M = rng.normal(size=(n4, n4))
    W0 = np.eye(n4) * 0.5 + 0.05 * (M @ M.T) / n4
    st = _S()
    st.X = X0.reshape(-1, 1)
    R.update_frame(y_im, flow, y_m)
    Hz, HTH, Hzc = R.measure(st, y_im, flow, y_m)
    X1, info, errs, Hzc1, gains, tok = R.update_run(W0, X0, y_im, flow, y_m, 1, 1e-4)
    assert info["niter"] == 1 and errs.shape == (1, 4) and np.array_equal(Hzc1, Hzc)
    W1 = tok.fetch()
    assert np.abs(W1 - W1.T).max() <= 1e-12 * np.abs(W1).max()
    A = np.linalg.inv(W0) + HTH
    assert np.linalg.norm(W1 @ A - np.eye(n4)) <= 1e-8
    if not info["reverted"]:
        step = np.linalg.solve(A, Hz)
        assert np.linalg.norm(X1.reshape(-1, 1) - X0.reshape(-1, 1) - step) <= 1e-9 * np.linalg.norm(step)
        e = R.error(_state_of(X1), y_im, flow, y_m)
        assert e[0] == int(errs[0][0]) and e[3] == int(errs[0][3]) and abs(e[1] - errs[0][1]) <= 1e-12 * e[1]
    # gains (kalman.py:828-830)
    assert np.allclose(gains[0], W1 @ Hzc[:, 0], rtol=1e-9, atol=1e-12)
    assert np.allclose(gains[1], W1 @ (Hzc[:, 1] + Hzc[:, 2]), rtol=1e-9, atol=1e-12)


def _state_of(X):
    s = _S()
    s.X = np.asarray(X).reshape(-1, 1)
    return s


def test_pipeline_equals_sequential_at_full_size(hm):
    """BASELINE config 4's frame loop (1024^2, 201 vertices): the streaming pipeline -- flow series of 1, 2, 4 pairs
    computed on the flow handle's stream while the filter works -- gives the bits of one `bf.calc` + `kf.compute`
    after the other, twice over (a race between the two would not show every time; tools/determinism_check.py is
    the longer form of this)."""
    from hydra_mi import kalman, mesh, synth, brox
    from hydra_mi.pipeline import FlowEKFPipeline
    n, frames = 1024, 7
    video, masks, c, r = synth.disk_video(n, frames, "translate_leftup", 0)
    dm0 = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)

    def make():
        return kalman.IteratedMSKalmanFilter(mesh.Mesh(dm0.p, dm0.t, dm0.h0), video[0], np.zeros((n, n, 2), np.float32), True)

    kf, bf, ref = make(), brox.BroxOpticalFlow(n, n), []
    for k in range(frames - 1):
        u, v = bf.calc(video[k], video[k + 1])
        kf.compute(video[k + 1], np.dstack((u, v)), masks[k + 1])
        ref.append((kf.state.X.copy(), kf.niter))
    for rep in range(2):
        kf, got = make(), []
        pipe = FlowEKFPipeline(kf, video, masks, flow_batch=8)
        pipe.run(0, frames - 1, on_frame=lambda k, e: got.append((kf.state.X.copy(), kf.niter)))
        pipe.close()
        for k in range(frames - 1):
            assert got[k][1] == ref[k][1] and np.array_equal(got[k][0], ref[k][0]), (rep, k)


GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _config4_inputs():
    from hydra_mi import mesh, synth
    g = np.load(os.path.join(GOLD, "config4_track.npz"))
    n, frames = int(g["n"]), int(g["frames"])
    video, masks, centre, radius = synth.disk_video(n, frames, "translate_leftup", 0)
    dm = mesh.disk_mesh(centre[0], centre[1], radius - 1.0, float(g["h0"]) * n)
    assert np.array_equal(dm.p, g["p"]) and np.array_equal(dm.t, g["t"])          # the golden's mesh: the bench's 201 vertices
    assert dm.size() == 201 and n == N_FULL
    return g, n, frames, video, masks, dm


def _check_config4_frame(g, k, kf, e):
    X, Xg = kf.state.X.reshape(-1), g["X"][k - 1]
    rel = np.linalg.norm(X - Xg) / np.linalg.norm(Xg)
    assert rel <= 1e-5, (k, rel)                                                    # the contract (north_star)
    assert kf.niter == int(g["iters"][k - 1]), (k, kf.niter)
    ge = g["err"][k - 1]
    assert e[0] == int(ge[0]) and e[3] == int(ge[3]), k                             # integer error terms exact
    assert abs(e[1] - ge[1]) <= 1e-6 * ge[1] and abs(e[2] - ge[2]) <= 1e-6 * ge[2], k
    Wd = np.diag(kf.state.W)
    assert np.linalg.norm(Wd - g["W_diag"][k - 1]) <= 1e-5 * np.linalg.norm(g["W_diag"][k - 1]), k
    return rel


def test_track_config4_matches_golden(hm):
    """BASELINE config 4 at state level (EKF state <= 1e-5 rel): 1024^2 video, the bench's
    201-vertex mesh, flow from the product's Brox, IteratedMSKalmanFilter.compute frame by frame, against the golden
    track of tools/make_golden.py config4 -- the C Brox oracle + the oracle's tracker with the reference's iterated
    update (kalman.py:774-831) over full-frame renders per perturbation (cuda.py:972-1010), 10 + 10 + 10 iterations of
    1 608 jz + 10 906 j evaluations each.  Same iteration counts, integer error terms exact, covariance diagonal and
    sampled covariance rows <= 1e-5."""
    from hydra_mi import brox, kalman
    g, n, frames, video, masks, dm = _config4_inputs()
    bf = brox.BroxOpticalFlow(n, n)
    kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
    worst = 0.0
    for k in range(1, frames):
        u, v = bf.calc(video[k - 1], video[k])
        e = kf.compute(video[k], np.dstack((u, v)), masks[k])
        worst = max(worst, _check_config4_frame(g, k, kf, e))
    W = kf.state.W
    rows = g["W_last_rows"]
    assert np.linalg.norm(W[::67] - rows) <= 1e-5 * np.linalg.norm(rows)
    print("config 4 golden track: worst state difference %.2e rel" % worst)


def test_track_config4_through_the_pipeline_matches_golden(hm):
    """The same golden track through the streaming component the bench times (FlowEKFPipeline: frames uploaded one by
    one into the frame ring, flow series beside the filter, flow handed over in device memory)."""
    from hydra_mi import kalman
    from hydra_mi.pipeline import FlowEKFPipeline
    g, n, frames, video, masks, dm = _config4_inputs()
    kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
    pipe = FlowEKFPipeline(kf, video, masks, flow_batch=8)
    pipe.run(on_frame=lambda k, e: _check_config4_frame(g, k + 1, kf, e))
    assert len(pipe.frame_done) == frames - 1
    pipe.close()


# mean end-point error (px) of the C oracle against the analytic field, interior of the 1024^2 frame, for the pairs
# of BASELINE config 5 with seeds 0..31 (tools: oracle/brox_oracle.calc on synth.warp_pair(1024,
# "translate_leftup_stretch", seed)); the test allows measured + 10 %
CONFIG5_EPE = [0.0433, 0.0424, 0.0484, 0.0472, 0.0466, 0.0454, 0.0409, 0.0463, 0.0482, 0.0461, 0.0483, 0.0452, 0.0478,
               0.0464, 0.0440, 0.0417, 0.0482, 0.0475, 0.0476, 0.0475, 0.0439, 0.0385, 0.0471, 0.0473, 0.0416, 0.0447,
               0.0456, 0.0469, 0.0467, 0.0431, 0.0467, 0.0411]


def test_config5_rank_share_matches_oracle(hm, oracle_brox):
    """BASELINE config 5, one rank's share of the 256-pair batch (hydra_mi.batch.shard(256, rank 0 of 8) = pairs
    0..31, pair i = seed i of the `translate_leftup_stretch` warp) computed the way `bench.py --workload
    flowbatch` computes it -- frames resident in HBM, hm_brox_calc_dev in launch series of 8 pairs: five pairs with
    distinct seeds are bit-equal to the C oracle, and every one of the 32 is the flow of its analytic field (mean
    end-point error within 10 % of what the oracle measures for that seed)."""
    from hydra_mi import batch, brox, synth
    from hydra_mi.pipeline import DeviceBuffer
    n, B = N_FULL, 8
    mine = batch.shard(256, 0, 8)
    assert list(mine) == list(range(32))
    P = len(mine)
    f0 = np.empty((P, n, n), np.uint8)
    f1 = np.empty((P, n, n), np.uint8)
    truth = []
    for j, i in enumerate(mine):
        f0[j], f1[j], tu, tv = synth.warp_pair(n, "translate_leftup_stretch", batch.pair_seed(i))
        truth.append((tu, tv))
    d0, d1 = DeviceBuffer(f0.nbytes), DeviceBuffer(f1.nbytes)
    du, dv = DeviceBuffer(P * n * n * 4), DeviceBuffer(P * n * n * 4)
    d0.upload(f0)
    d1.upload(f1)
    bf = brox.BroxOpticalFlow(n, n, max_batch=B)
    for s in range(0, P, B):
        bf.calc_dev(B, d0.ptr + s * n * n, d1.ptr + s * n * n, du.ptr + s * n * n * 4, dv.ptr + s * n * n * 4)
    bf.sync()
    U = du.download(np.empty((P, n, n), np.float32))
    V = dv.download(np.empty((P, n, n), np.float32))
    b = n // 8
    for j in range(P):
        tu, tv = truth[j]
        epe = np.sqrt((U[j] - tu) ** 2 + (V[j] - tv) ** 2)[b:-b, b:-b].mean()
        assert epe <= 1.1 * CONFIG5_EPE[j], (j, epe)
    oracle_brox.set_threads(min(16, os.cpu_count() or 1))
    try:
        for j in (0, 9, 14, 22, 31):
            ru, rv = oracle_brox.calc(f0[j], f1[j])
            assert np.array_equal(U[j], ru) and np.array_equal(V[j], rv), j
    finally:
        oracle_brox.set_threads(1)
    for d in (d0, d1, du, dv):
        d.close()
