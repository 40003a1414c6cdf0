"""Development aid: a few frames of the 1024^2 bench scenario with a given measure_split
(run under rocprofv3 --kernel-trace --stats to read k_measure_vertex's duration)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, kalman, mesh, synth
from hydra_mi.renderer import DeviceObservation
n = 1024
split = int(sys.argv[1]) if len(sys.argv) > 1 else 3
video, masks, c, r = synth.disk_video(n, 5, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
dv = torch.from_numpy(video).cuda(); dmk = torch.from_numpy(masks).cuda()
du = torch.empty((n, n), dtype=torch.float32, device="cuda"); dvv = torch.empty_like(du)
bf = brox.BroxOpticalFlow(n, n)
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
kf.state.renderer.tune("measure_split", split)
if len(sys.argv) > 2:
    kf.state.renderer.tune("edge_split", int(sys.argv[2]))
for k in range(4):
    bf.calc_dev(1, dv[k].data_ptr(), dv[k + 1].data_ptr(), du.data_ptr(), dvv.data_ptr()); bf.sync()
    obs = DeviceObservation(dv[k + 1].data_ptr(), du.data_ptr(), dvv.data_ptr(), dmk[k + 1].data_ptr(), y_m_host=masks[k + 1])
    kf.compute(obs, None, None)
    print(split, k, kf.niter, float(np.abs(kf.state.X).sum()))
