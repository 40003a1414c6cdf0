"""Development aid: time of IteratedMSKalmanFilter.predict at the bench's size, device Newton against host Newton."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth
n = 1024
video, masks, c, r = synth.disk_video(n, 4, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
for dev in (True, False):
    kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(dm.p, dm.t, dm.h0), video[0], np.zeros((n, n, 2), np.float32), True)
    kf.device_predict = dev
    flow = np.zeros((n, n, 2), np.float32) - 1.28
    kf.compute(video[1], flow, masks[1])
    ts = []
    for rep in range(6):
        kf.state.X[2 * kf.N:] += 0.3 * np.random.default_rng(rep).normal(size=(2 * kf.N, 1))
        t0 = time.perf_counter(); kf.predict(); kf.state.renderer.update_begin(kf.state._W, kf.state.X); _ = kf.state.renderer.cov_fetch()[0, 0]; ts.append(time.perf_counter() - t0)
    print("device_predict", dev, "predict + covariance half, ms:", ["%.3f" % (1e3 * t) for t in ts], "newton iterations", kf.newton_iterations, flush=True)
