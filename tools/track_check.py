"""Does the tracker follow the advected disk?  (sanity check of flow + EKF together)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import brox, kalman, mesh, synth
n, F = 256, 12
for name in ("translate_leftup", "warp"):
    video, masks, c, r = synth.disk_video(n, F, name, 0)
    dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.1 * n)
    p0 = dm.p.copy()
    bf = brox.BroxOpticalFlow(n, n)
    kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
    field = synth.scaled_field(name, n)
    truth = p0.copy()
    for k in range(F - 1):
        u, v = bf.calc(video[k], video[k + 1])
        e = kf.compute(video[k + 1], np.dstack((u, v)), masks[k + 1])
        vx, vy = field(truth[:, 0], truth[:, 1])
        truth = truth + np.column_stack((np.broadcast_to(vx, truth[:, 0].shape), np.broadcast_to(vy, truth[:, 0].shape)))
    est = kf.state.vertices()
    print("%-18s mean displacement est (%.2f, %.2f) true (%.2f, %.2f)  rms position error %.3f px  last iters %d"
          % (name, *(est - p0).mean(0), *(truth - p0).mean(0), np.sqrt(((est - truth) ** 2).sum(1).mean()), kf.niter))
