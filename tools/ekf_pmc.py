"""Development aid: a few frames of the filter at 1024^2 / 201 vertices for rocprofv3 --pmc runs (flow computed once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, kalman, mesh, synth
from hydra_mi.renderer import DeviceObservation
n = 1024
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
video, masks, c, r = synth.disk_video(n, nf + 1, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
dv = torch.from_numpy(video).cuda(); dmk = torch.from_numpy(masks).cuda()
U = torch.empty((nf, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
bf = brox.BroxOpticalFlow(n, n, max_batch=4)
for k in range(nf):
    bf.calc_dev(1, dv[k].data_ptr(), dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr())
bf.sync()
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
for key in ("render_rows", "measure_split", "edge_split"):            # A/B knobs: HYDRA_TUNE_render_rows=8 ...
    if os.environ.get("HYDRA_TUNE_" + key):
        kf.state.renderer.tune(key, int(os.environ["HYDRA_TUNE_" + key]))
for k in range(nf):
    obs = DeviceObservation(dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr(), dmk[k + 1].data_ptr())
    kf.compute(obs, None, None)
    print("frame", k, "iterations", kf.niter, flush=True)
