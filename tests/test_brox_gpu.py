"""GPU parity of the Brox path: every HIP kernel and the whole pipeline against
oracle/brox_ref.c on the same seeded inputs, through the C-ABI.

The oracle and the kernels use the same binary32 operation order with FMA
contraction off, so the bar is BIT-EXACT equality (np.array_equal; it treats
+0 and -0 as equal), well inside the 1e-4 end-point-error tolerance the
build contract allows.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(16, 16), (37, 29), (64, 64), (65, 64), (100, 77), (200, 150), (131, 259)]


def _rand(shape, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return rng.uniform(lo, hi, shape).astype(np.float32)


def _smooth(shape, seed):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    return ndimage.gaussian_filter(rng.random(shape), 2.0).astype(np.float32)


@pytest.mark.parametrize("w,h", SIZES)
def test_blur_resample_deriv(hm, oracle_brox, w, h):
    from hydra_mi import brox
    img = _smooth((h, w), 1)
    assert np.array_equal(brox.op_blur(img, 0.8), oracle_brox.blur(img, 0.8))
    assert np.array_equal(brox.op_blur(img, 0.5), oracle_brox.blur(img, 0.5))
    wd, hd = int(np.ceil(w * 0.8)), int(np.ceil(h * 0.8))
    assert np.array_equal(brox.op_resample(img, wd, hd, 1.0), oracle_brox.resample(img, wd, hd, 1.0))
    up = brox.op_resample(img, w + 13, h + 7, 1.25)
    assert np.array_equal(up, oracle_brox.resample(img, w + 13, h + 7, 1.25))
    dx, dy = brox.op_deriv(img)
    rx, ry = oracle_brox.deriv(img)
    assert np.array_equal(dx, rx) and np.array_equal(dy, ry)


@pytest.mark.parametrize("w,h", SIZES)
def test_fused_launches_equal_the_separate_operators(hm, oracle_brox, w, h):
    """calc builds a pyramid level, the derivative images of a level and the prolongation of u + du in one
    launch each; every one gives the bits of the oracle's separate operators applied in turn."""
    from hydra_mi import brox
    img, img1 = _smooth((h, w), 1), _smooth((h, w), 2)
    for scale in (0.8, 0.5):
        wd, hd = int(np.ceil(w * scale)), int(np.ceil(h * scale))
        want = oracle_brox.resample(oracle_brox.blur(img, scale), wd, hd, 1.0)
        assert np.array_equal(brox.op_pyr_down(img, wd, hd, scale), want), scale
    Ix0, Iy0 = oracle_brox.deriv(img)
    I1x, I1y = oracle_brox.deriv(img1)
    I1xx, I1xy = oracle_brox.deriv(I1x)
    _, I1yy = oracle_brox.deriv(I1y)
    for name, g, r in zip("Ix0 Iy0 I1x I1y I1xx I1xy I1yy".split(), brox.op_deriv_all(img, img1),
                          (Ix0, Iy0, I1x, I1y, I1xx, I1xy, I1yy)):
        assert np.array_equal(g, r), name
    u, v = _rand((h, w), 3, -3, 3), _rand((h, w), 4, -3, 3)
    du, dv = _rand((h, w), 5, -0.5, 0.5), _rand((h, w), 6, -0.5, 0.5)
    wf, hf = int(np.ceil(w / 0.8)), int(np.ceil(h / 0.8)) + 1
    gu, gv = brox.op_add_prolong(u, v, du, dv, wf, hf)
    assert np.array_equal(gu, oracle_brox.resample(u + du, wf, hf, np.float32(wf) / np.float32(w)))
    assert np.array_equal(gv, oracle_brox.resample(v + dv, wf, hf, np.float32(hf) / np.float32(h)))
    su, sv = brox.op_add_prolong(u, v, du, dv, w, h)                  # the level-0 form
    assert np.array_equal(su, u + du) and np.array_equal(sv, v + dv)


def _level_fields(oracle_brox, w, h, seed):
    I0 = _smooth((h, w), seed)
    I1 = _smooth((h, w), seed + 1)
    Ix0, Iy0 = oracle_brox.deriv(I0)
    I1x, I1y = oracle_brox.deriv(I1)
    I1xx, I1xy = oracle_brox.deriv(I1x)
    _, I1yy = oracle_brox.deriv(I1y)
    u = _rand((h, w), seed + 2, -3, 3)
    v = _rand((h, w), seed + 3, -3, 3)
    return (I0, Ix0, Iy0, I1, I1x, I1y, I1xx, I1xy, I1yy, u, v)


@pytest.mark.parametrize("w,h", SIZES)
def test_warp_prepare(hm, oracle_brox, w, h):
    from hydra_mi import brox
    f = _level_fields(oracle_brox, w, h, 10)
    got = brox.op_warp(*f)
    ref = oracle_brox.warp(*f)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    u, v = f[9], f[10]
    du = _rand((h, w), 20, -0.5, 0.5)
    dv = _rand((h, w), 21, -0.5, 0.5)
    got = brox.op_prepare(u, v, du, dv, ref, 0.197, 50.0)
    exp = oracle_brox.prepare(u, v, du, dv, ref, 0.197, 50.0)
    for name, g, r in zip("nu nv a12 idu idv sx sy".split(), got, exp):
        assert np.array_equal(g, r), name


@pytest.mark.parametrize("kind", ["smooth", "rough", "jump", "outside"])
def test_warp_window_and_direct_sampling_agree_with_oracle(hm, oracle_brox, kind):
    """k_warp in both variants -- direct reads, and the window of a block's taps staged in LDS when it
    fits (smooth flow) with the fall-back when it does not (rough flow, a discontinuity) -- is
    bit-identical to the oracle."""
    from hydra_mi import brox
    w, h = 200, 150
    f = list(_level_fields(oracle_brox, w, h, 50))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    if kind == "smooth":
        f[9] = (6.0 + 0.02 * xx).astype(np.float32); f[10] = (-4.0 + 0.03 * yy).astype(np.float32)
    elif kind == "rough":
        f[9] = _rand((h, w), 60, -40, 40); f[10] = _rand((h, w), 61, -40, 40)
    elif kind == "jump":
        f[9] = np.where(xx < 97, -7.5, 31.25).astype(np.float32); f[10] = np.where(yy < 70, 12.0, -9.0).astype(np.float32)
    else:
        f[9] = np.full((h, w), 500.0, np.float32); f[10] = np.zeros((h, w), np.float32)
    ref = oracle_brox.warp(*f)
    for window in (False, True):
        got = brox.op_warp(*f, window=window)
        for name, g_, r_ in zip("Iz Ix Iy Ixz Iyz Ixx Ixy Iyy".split(), got, ref):
            assert np.array_equal(g_, r_), (kind, window, name)


@pytest.mark.parametrize("w,h", SIZES + [(300, 300)])
@pytest.mark.parametrize("fuse", [0, 1, 2, 5, 105, 205, 10, 210])
def test_sor(hm, oracle_brox, w, h, fuse):
    from hydra_mi import brox
    f = _level_fields(oracle_brox, w, h, 30)
    warped = oracle_brox.warp(*f)
    du = _rand((h, w), 40, -0.5, 0.5)
    dv = _rand((h, w), 41, -0.5, 0.5)
    coef = oracle_brox.prepare(f[9], f[10], du, dv, warped, 0.197, 50.0)
    gdu, gdv = brox.op_sor(du, dv, coef, 10, fuse=fuse)
    rdu, rdv = oracle_brox.sor(du, dv, coef, 10)
    assert np.array_equal(gdu, rdu) and np.array_equal(gdv, rdv)


def _pair(hm, n, name, seed=0):
    from hydra_mi import synth
    f0, f1, tu, tv = synth.warp_pair(n, name, seed)
    return f0, f1, tu, tv


@pytest.mark.parametrize("n,name", [(64, "warp"), (128, "translate_leftup"), (200, "rotate"),
                                    (256, "translate_leftup_stretch")])
def test_calc_matches_oracle(hm, oracle_brox, n, name):
    from hydra_mi import brox
    f0, f1, tu, tv = _pair(hm, n, name)
    bf = brox.BroxOpticalFlow(n, n)
    assert bf.levels() == oracle_brox.levels(n, n)
    u, v = bf.calc(f0, f1)
    ru, rv = oracle_brox.calc(f0, f1)
    epe = np.sqrt((u - ru) ** 2 + (v - rv) ** 2)
    assert epe.max() <= 1e-4, epe.max()           # the contract's tolerance
    assert np.array_equal(u, ru) and np.array_equal(v, rv)   # what is actually achieved
    # and the flow is a sensible estimate of the analytic field (informational bound)
    b = n // 8
    err = np.sqrt((u - tu) ** 2 + (v - tv) ** 2)[b:-b, b:-b]
    assert err.mean() < 0.25


def test_calc_nonsquare_and_params(hm, oracle_brox):
    from hydra_mi import brox, synth
    f0, f1, _, _ = synth.warp_pair(160, "warp", 3)
    f0, f1 = np.ascontiguousarray(f0[:100, :150]), np.ascontiguousarray(f1[:100, :150])
    for kw, okw in [(dict(alpha=0.4, gamma=25.0, inner_iterations=5, solver_iterations=5),
                     dict(alpha=0.4, gamma=25.0, inner=5, solver=5)),
                    (dict(scale_factor=0.5, outer_iterations=3), dict(scale=0.5, outer=3)),
                    (dict(outer_iterations=1, solver_iterations=7), dict(outer=1, solver=7))]:
        bf = brox.BroxOpticalFlow(150, 100, **kw)
        u, v = bf.calc(f0, f1)
        ru, rv = oracle_brox.calc(f0, f1, **okw)
        assert np.array_equal(u, ru) and np.array_equal(v, rv), kw


def test_calc_batch_and_tuning_do_not_change_results(hm, oracle_brox):
    from hydra_mi import brox, synth
    n = 96
    pairs = [synth.warp_pair(n, name, seed) for seed, name in enumerate(["warp", "rotate", "translate_leftup"])]
    F0 = np.stack([p[0] for p in pairs])
    F1 = np.stack([p[1] for p in pairs])
    bf = brox.BroxOpticalFlow(n, n, max_batch=3)
    U, V = bf.calc_batch(F0, F1)
    for i in range(3):
        ru, rv = oracle_brox.calc(F0[i], F1[i])
        assert np.array_equal(U[i], ru) and np.array_equal(V[i], rv)
    for key, val in [("sor_fuse", 1), ("sor_fuse", 2), ("sor_threads", 512), ("sor_fuse", 0), ("sor_threads", 1024),
                     ("warp_window", 1), ("warp_window", 0), ("coarse_max", 0), ("coarse_max", 32), ("coarse_max", 64), ("sor_deep", 0), ("sor_fuse", 10),
                     # the handle's two streams: compute-unit mask, the whole chip for one call, back, no mask
                     ("whole_chip", 1), ("cu_reserve", 32), ("whole_chip", 1), ("whole_chip", 0), ("cu_reserve", 0), ("whole_chip", 0)]:
        bf.tune(key, val)
        U2, V2 = bf.calc_batch(F0, F1)
        assert np.array_equal(U, U2) and np.array_equal(V, V2), (key, val)


def test_two_lanes_give_the_bits_of_one(hm, oracle_brox):
    """hm_brox_tune "lanes" = 2: a series of four or more pairs on the CU-masked stream runs as two halves side by side,
    in disjoint parts of the handle's planes -- every pair the oracle's flow, whatever half it was in."""
    from hydra_mi import brox, synth
    n = 112
    names = ["warp", "rotate", "translate_leftup", "translate_leftup_stretch", "warp", "rotate", "translate_leftup"]
    pairs = [synth.warp_pair(n, name, seed) for seed, name in enumerate(names)]
    F0 = np.stack([p[0] for p in pairs])
    F1 = np.stack([p[1] for p in pairs])
    bf = brox.BroxOpticalFlow(n, n, max_batch=7)
    U, V = bf.calc_batch(F0, F1)
    for i in (0, 3, 4, 6):                        # the first and last pair of either half (4 + 3)
        ru, rv = oracle_brox.calc(F0[i], F1[i])
        assert np.array_equal(U[i], ru) and np.array_equal(V[i], rv), i
    bf.tune("lanes", 2)
    bf.tune("cu_reserve", 32)
    for count in (7, 4, 5, 3):                    # 3: below four pairs one lane
        U2, V2 = bf.calc_batch(F0[:count], F1[:count])
        assert np.array_equal(U[:count], U2) and np.array_equal(V[:count], V2), count
    bf.tune("whole_chip", 1)                      # the plain stream has no twin: one lane
    U2, V2 = bf.calc_batch(F0, F1)
    assert np.array_equal(U, U2) and np.array_equal(V, V2)
    with pytest.raises(RuntimeError):
        bf.tune("lanes", 2)                       # after cu_reserve: refused (the twin stream takes its mask there)


@pytest.mark.parametrize("w,h,kw,okw", [
    (64, 64, {}, {}),                                                        # the whole pyramid inside k_coarse
    (57, 33, dict(inner_iterations=3, solver_iterations=4), dict(inner=3, solver=4)),
    (17, 64, {}, {}),
    (100, 90, dict(scale_factor=0.97, outer_iterations=70), dict(scale=0.97, outer=70)),   # > 32 coarse levels: two launches
    (150, 100, dict(scale_factor=0.5), dict(scale=0.5)),                     # the exit level is 4x the last fused one
])
def test_coarse_levels_in_one_launch(hm, oracle_brox, w, h, kw, okw):
    """k_coarse (the levels of at most 64 x 64 px in one launch per pair and tile size) against the oracle and
    against the launch-per-operator path."""
    from hydra_mi import brox, synth
    n = max(w, h)
    f0, f1, _, _ = synth.warp_pair(n + (n & 1), "warp", 5)
    F0 = np.stack([np.ascontiguousarray(f0[:h, :w]), np.ascontiguousarray(f1[:h, :w])])
    F1 = np.stack([np.ascontiguousarray(f1[:h, :w]), np.ascontiguousarray(f0[:h, :w])])
    bf = brox.BroxOpticalFlow(w, h, max_batch=2, **kw)
    assert sum(1 for lw, lh in bf.levels() if lw <= 64 and lh <= 64) >= 1
    ref = [oracle_brox.calc(F0[i], F1[i], **okw) for i in range(2)]
    for cmax in (64, 32, 0):
        bf.tune("coarse_max", cmax)
        U, V = bf.calc_batch(F0, F1)
        for i in range(2):
            assert np.array_equal(U[i], ref[i][0]) and np.array_equal(V[i], ref[i][1]), (cmax, i)


def test_coarse_launch_with_pairs_levels_apart(hm, oracle_brox):
    """The workgroups of a k_coarse launch (one per pair) go through the levels on their own: a pair must not
    touch what another pair uses at a different level.  With the test knob pairs 8.. wait ~0.3 ms at every level
    between writing and reading their derivative planes (and drop their L1) while pairs 0..7 run whole levels
    ahead; workgroups z and z + 8 usually share an XCD, i.e. an L2, so what one writes the other reads.  (Found as
    an intermittent difference between the pipelined and the sequential frame loop at 1024^2: with the per-level
    offset z * plane of the launch-per-operator path, the planes of pair 2 at one level overlapped those of pair 10
    three levels down.)"""
    from hydra_mi import brox, synth
    n, B = 64, 12
    names = ["warp", "rotate", "translate_leftup", "translate_leftup_stretch"]
    pairs = [synth.warp_pair(n, names[s % 4], s)[:2] for s in range(B)]
    F0 = np.stack([p[0] for p in pairs]); F1 = np.stack([p[1] for p in pairs])
    ref = [oracle_brox.calc(F0[i], F1[i]) for i in range(B)]
    bf = brox.BroxOpticalFlow(n, n, max_batch=B)
    for cmax in (32, 64):
        bf.tune("coarse_max", cmax)
        for stagger in (1, 0):
            bf.tune("coarse_stagger", stagger)
            U, V = bf.calc_batch(F0, F1)
            for i in range(B):
                assert np.array_equal(U[i], ref[i][0]) and np.array_equal(V[i], ref[i][1]), (cmax, stagger, i)


def test_identical_frames_give_zero_flow(hm):
    from hydra_mi import brox, synth
    f0 = synth.warp_pair(64, "warp")[0]
    u, v = brox.BroxOpticalFlow(64, 64).calc(f0, f0)
    assert np.abs(u).max() == 0.0 and np.abs(v).max() == 0.0


def test_errors_are_loud(hm):
    from hydra_mi import brox
    with pytest.raises(RuntimeError):
        brox.BroxOpticalFlow(64, 64, scale_factor=1.5)
    bf = brox.BroxOpticalFlow(64, 64)
    with pytest.raises(ValueError):
        bf.calc(np.zeros((32, 32), np.uint8), np.zeros((32, 32), np.uint8))
    with pytest.raises(TypeError):
        bf.calc(np.zeros((64, 64), np.float32), np.zeros((64, 64), np.float32))
    with pytest.raises(RuntimeError):
        bf.tune("sor_fuse", 3)        # 3 does not divide solver_iterations = 10


def test_profile_totals(hm, oracle_brox):
    """Profiling does not change the result; switching it off keeps the recorded totals for profile_read;
    the totals are those of the launch plan (pixels x fused iterations, pixels per pass)."""
    from hydra_mi import brox, synth
    n = 80
    f0, f1, _, _ = synth.warp_pair(n, "rotate", 1)
    ru, rv = oracle_brox.calc(f0, f1)
    bf = brox.BroxOpticalFlow(n, n, max_batch=2)
    with pytest.raises(RuntimeError):
        bf.tune("graph", 1)           # the round-1 hipGraph knob is gone
    bf.tune("coarse_max", 32)
    bf.profile(True)
    u, v = bf.calc(f0, f1)
    bf.profile(False)
    u2, v2 = bf.calc(f0, f1)          # not recorded
    ms, launches, pxit, px = bf.profile_read()
    assert launches > 0 and ms > 0 and pxit > 0 and 0 < px <= pxit
    assert np.array_equal(u, ru) and np.array_equal(v, rv) and np.array_equal(u2, ru) and np.array_equal(v2, rv)
    # every level that is launched per operator (the small levels run inside k_coarse and have no k_sor launches):
    # inner x solver red-black iterations over its pixels
    want = sum(w * h for w, h in bf.levels() if w > 32 or h > 32) * 10 * 10
    assert pxit == want
    per_call = launches
    bf.profile(True)
    bf.calc(f0, f1)
    ms2, launches2, pxit2, _ = bf.profile_read()
    assert launches2 == per_call and pxit2 == want


def test_repeated_series_with_caller_reallocating_buffers(hm):
    """A caller that allocates fresh device buffers for every series (a new video each time): the
    flows must not depend on how often the handle has been used or where the buffers live.  (The
    round-1 hipGraph replay failed exactly this from the fifth launch on and was removed.)"""
    torch = pytest.importorskip("torch")
    from hydra_mi import brox, synth
    n, B = 256, 4
    pairs = [synth.warp_pair(n, name, s) for s, name in enumerate(["warp", "rotate", "translate_leftup", "warp", "rotate"])]
    F0 = np.stack([p[0] for p in pairs[:B]]); F1 = np.stack([p[1] for p in pairs[:B]])
    bf = brox.BroxOpticalFlow(n, n, max_batch=B)
    first = None
    for it in range(4):
        d0 = torch.from_numpy(F0).cuda(); d1 = torch.from_numpy(F1).cuda()
        U = torch.empty((2, B, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
        torch.cuda.synchronize()
        for rep in range(2):
            bf.calc_dev(B, d0.data_ptr(), d1.data_ptr(), U[rep].data_ptr(), V[rep].data_ptr())
        bf.sync()
        got = (U.cpu().numpy(), V.cpu().numpy())
        assert np.isfinite(got[0]).all() and np.isfinite(got[1]).all(), it
        assert np.array_equal(got[0][0], got[0][1]) and np.array_equal(got[1][0], got[1][1]), it
        if first is None:
            first = got
        assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1]), it
        keep = (d0, d1, U, V) if it % 2 == 0 else None        # vary what is freed in between
