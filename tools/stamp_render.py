"""Development aid: where the workgroups of k_render_iter spend their time.  Needs the instrumented build
(hipcc ... -DHM_STAMP -shared brox.hip ekf.hip predict.cpp -o build_exp/libhydra_mi_stamp.so): every workgroup writes
wall_clock64() stamps (100 MHz) at its start, after the triangle tests, after the pixel loop, before the reduction
and at its end."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hydra_mi
from hydra_mi import _lib
_lib.SO_PATH = os.path.join(ROOT, "build_exp", "libhydra_mi_stamp.so")
from hydra_mi import brox, kalman, mesh, synth
from hydra_mi.renderer import DeviceObservation
L = _lib.lib()
L.hm_debug_stamps.restype = ctypes.c_int
L.hm_debug_stamps.argtypes = [ctypes.c_void_p]
n, nf = 1024, 2
video, masks, c, r = synth.disk_video(n, nf + 1, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
dv = torch.from_numpy(video).cuda(); dmk = torch.from_numpy(masks).cuda()
U = torch.empty((nf, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
bf = brox.BroxOpticalFlow(n, n, max_batch=4)
for k in range(nf):
    bf.calc_dev(1, dv[k].data_ptr(), dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr())
bf.sync()
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
nwg = kf.N + 1024 + 8
stamps = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
_lib.check(L.hm_debug_stamps(stamps.data_ptr()), "hm_debug_stamps")
for k in range(nf):
    obs = DeviceObservation(dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr(), dmk[k + 1].data_ptr())
    kf.compute(obs, None, None)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64) * 0.01          # microseconds; the stamps of the LAST launch
N = kf.N
t0 = s[:N + 1024, 0].min()
reg, st = s[:N], s[N:N + 1024]
print("kernel span %.1f us (first start to last end)" % (s[:N + 1024, 4].max() - t0))
print("region workgroups: start %.1f..%.1f us after the first, duration mean %.1f max %.1f us" %
      ((reg[:, 0] - t0).min(), (reg[:, 0] - t0).max(), (reg[:, 4] - reg[:, 0]).mean(), (reg[:, 4] - reg[:, 0]).max()))
print("strip workgroups: start %.1f..%.1f us, duration mean %.1f max %.1f us" %
      ((st[:, 0] - t0).min(), (st[:, 0] - t0).max(), (st[:, 4] - st[:, 0]).mean(), (st[:, 4] - st[:, 0]).max()))
for a, b, name in ((0, 1, "triangle tests"), (1, 2, "setups + pixel loop"), (2, 3, "texels, stores, error terms"), (3, 4, "reduction")):
    d = st[:, b] - st[:, a]
    print("  %-28s mean %.2f  p90 %.2f  max %.2f us" % (name, d.mean(), np.percentile(d, 90), d.max()))
busy = (st[:, 2] - st[:, 1]) > 1.0
print("  strips with candidates: %d of %d; their total mean %.1f us, the others %.1f us" %
      (busy.sum(), len(st), (st[busy, 4] - st[busy, 0]).mean(), (st[~busy, 4] - st[~busy, 0]).mean()))
print("  last strip ends %.1f us after the first start; last region %.1f" % ((st[:, 4] - t0).max(), (reg[:, 4] - t0).max()))
