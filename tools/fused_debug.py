"""Debug aid: fused hm_update_run against the step-wise loop on a stressed track."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import mesh, synth, kalman

n = 128
video, flow = synth.test_data(n, n)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
wmul = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
flow = flow * scale
for fused in (False, True):
    dm = mesh.mask_mesh(video[:, :, 0] > 0, 12.0)
    kf = kalman.IteratedMSKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True)
    kf.fused_update = fused
    kf.state.W = kf.state.W * wmul
    for k in range(4):
        frame = video[:, :, k]
        mask = (frame > 0).astype(np.uint8)
        try:
            e = kf.compute(frame, flow[:, :, :, k], mask)
        except FloatingPointError as ex:
            print("fused", fused, "frame", k, "FPE", ex)
            break
        print("fused", fused, "frame", k, "niter", kf.niter, "rev", kf.reverted, "conv", kf.converged, "err", e[:4],
              "X", float(np.abs(kf.state.X).max()), "Wmax", float(np.abs(kf.state.W).max()))
