"""Development aid: the filter alone at 1024^2 / 201 vertices (flow precomputed, nothing beside it): ms per frame and per
IEKF iteration over frames 2..11.  HYDRA_MI_SO selects another build of the library (A/B experiments)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, kalman, mesh, synth
from hydra_mi.renderer import DeviceObservation
n, nf = 1024, 12
video, masks, c, r = synth.disk_video(n, nf + 1, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
dv = torch.from_numpy(video).cuda(); dmk = torch.from_numpy(masks).cuda()
U = torch.empty((nf, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
bf = brox.BroxOpticalFlow(n, n, max_batch=4)
for k in range(0, nf, 4):
    bf.calc_dev(4, dv[k].data_ptr(), dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr())
bf.sync()
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
kf.return_flow = False
its = 0
for k in range(nf):
    if k == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter(); its = 0
    nxt = dmk[k + 2].data_ptr() if k + 2 <= nf else None
    obs = DeviceObservation(dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr(), dmk[k + 1].data_ptr(), next_mask=nxt)
    kf.compute(obs, None, None)
    its += kf.niter
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s: %.3f ms/frame, %.2f iterations/frame, %.1f us per iteration incl. frame overhead, state checksum %.12e"
      % (os.environ.get("HYDRA_MI_SO", "libhydra_mi.so").split("/")[-1], 1e3 * dt / (nf - 2), its / (nf - 2), 1e6 * dt / its, float(np.abs(kf.state.X).sum())))
kf.close(); bf.close()
