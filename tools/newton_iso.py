"""Development aid: hm_newton_dev_start / _finish on the same input many times (alone, and with a flow series running
beside it) -- every result must be the first one's, bit for bit."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth, brox, _lib
n = 1024
video, masks, c, r = synth.disk_video(n, 10, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
st = kf.state
rng = np.random.default_rng(1)
X0 = st.X.reshape(-1).copy()
n2 = 2 * st.N
X0[n2:] = rng.normal(0, 1.5, n2)
bars = np.ascontiguousarray(kf._bars, np.int32)
l0 = np.ascontiguousarray(st.l0[:, 0], np.float64)
L = _lib.lib()
h = st.renderer._h

def once(x):
    X = x.copy()
    its = ctypes.c_int()
    rc = L.hm_newton_dev_start(h, int(st.N), int(bars.shape[0]), _lib.ptr(bars), _lib.ptr(l0), float(kf.kappa), float(kf.M),
                               float(kf.deltat), int(kf.maxiter), float(kf.tol), _lib.ptr(X))
    assert rc == 0
    _lib.check(L.hm_newton_dev_finish(h, _lib.ptr(X), ctypes.byref(its)), "finish")
    return X, its.value

ref, its = once(X0)
bad = 0
for i in range(3000):
    got, it2 = once(X0)
    if not np.array_equal(got, ref) or it2 != its:
        bad += 1
        if bad <= 3:
            print("launch %d differs: max |dX| %.3g, its %d vs %d" % (i, np.abs(got - ref).max(), it2, its), flush=True)
print("alone: %d of 3000 launches differ" % bad, flush=True)
# with a flow series beside it
import torch
bf = brox.BroxOpticalFlow(n, n, max_batch=8)
F0 = torch.from_numpy(np.stack([video[0]] * 8)).cuda(); F1 = torch.from_numpy(np.stack([video[1]] * 8)).cuda()
U = torch.empty((8, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
bad = 0
for rep in range(30):
    bf.calc_dev(8, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
    for i in range(40):
        got, it2 = once(X0)
        if not np.array_equal(got, ref) or it2 != its:
            bad += 1
            if bad <= 3:
                print("beside flow, launch %d.%d differs: max |dX| %.3g" % (rep, i, np.abs(got - ref).max()), flush=True)
    bf.sync()
print("beside a flow series: %d of 1200 launches differ" % bad, flush=True)
# alternating two different inputs (stale input would show)
X1 = X0.copy(); X1[:n2] += rng.normal(0, 0.5, n2)
r1, _ = once(X1)
bad = 0
for i in range(1000):
    a, _ = once(X0); b, _ = once(X1)
    if not np.array_equal(a, ref) or not np.array_equal(b, r1):
        bad += 1
        if bad <= 3:
            print("alternating %d differs: %.3g %.3g" % (i, np.abs(a - ref).max(), np.abs(b - r1).max()), flush=True)
print("alternating inputs: %d of 1000 pairs differ" % bad, flush=True)
