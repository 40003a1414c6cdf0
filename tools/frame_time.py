"""Development aid: wall time of the phases of IteratedMSKalmanFilter.compute at 1024^2 (flow precomputed)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, kalman, mesh, synth
from hydra_mi.renderer import DeviceObservation
n = 1024
nf = 12
video, masks, c, r = synth.disk_video(n, nf + 1, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
dv = torch.from_numpy(video).cuda(); dmk = torch.from_numpy(masks).cuda()
U = torch.empty((nf, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
bf = brox.BroxOpticalFlow(n, n, max_batch=4)
for k in range(0, nf, 4):
    bf.calc_dev(4, dv[k].data_ptr(), dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr())
bf.sync()
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
T = {}
def tick(name, t0):
    t1 = time.perf_counter(); T[name] = T.get(name, 0.0) + (t1 - t0); return t1
R = kf.state.renderer
iters = 0
nits = 0
for k in range(nf):
    obs = DeviceObservation(dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr(), dmk[k + 1].data_ptr(), y_m_host=masks[k + 1])
    if k == 2:
        T.clear(); iters = 0; tstart = time.perf_counter()
    t = time.perf_counter()
    R.set_observation_dev(obs); t = tick("set_obs", t)
    blocks = kf._spring_blocks(); t = tick("spring_blocks", t)
    kf.orig_x = kf.state.X.copy(); kf._newton(); t = tick("newton", t)
    kf.state.W = R.cov_predict(kf.state._W, kf._bars, blocks, kf.deltat, kf.deltat / kf.M, kf.state.eps_F, fetch=False); t = tick("cov_predict", t)
    kf.pred_x = kf.state.X.copy()
    kf.projectmask(obs); t = tick("projectmask", t)
    kf.update(obs, obs.masked, obs); t = tick("update", t)
    e = kf.error(obs, obs.raw, obs); t = tick("error", t)
    iters += kf.niter
    nits = nits + kf.newton_iterations if k >= 2 else 0
tot = time.perf_counter() - tstart
m = nf - 2
print("newton iterations/frame", nits / m); print("frames", m, "ms/frame", 1e3 * tot / m, "iters/frame", iters / m)
for k_, v in T.items():
    print("%-14s %8.3f ms/frame" % (k_, 1e3 * v / m))
