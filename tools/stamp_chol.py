"""Development aid: where a column of the chain of k_chol_flow spends its time.  Needs the instrumented build
(hipcc ... -DHM_STAMP -shared brox.hip ekf.hip predict.cpp -o build_exp/libhydra_mi_stamp.so): lane 0 of the chain's
waves writes clock64() stamps per column -- 0 top of the column, 1 products done, 2 / 3 start / end of the B wave's
eight strips, 5 T wave done with its stores, 7 prefetch of the next column's operands done, 6 after the closing barrier."""
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
stamps = torch.zeros((4096 + 64, 8), dtype=torch.int64, device="cuda")
_lib.check(L.hm_debug_stamps(stamps.data_ptr()), "hm_debug_stamps")
for k in range(nf):
    obs = DeviceObservation(dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr(), dmk[k + 1].data_ptr())
    kf.compute(obs, None, None)
torch.cuda.synchronize()
s = stamps.cpu().numpy()[4096:4096 + 26].astype(np.float64)      # the stamps of the LAST factorisation
nb = 26
span = s[nb - 1, 6] - s[0, 0]
print("chain: %d columns, %.0f clocks from the top of column 0 to the end of the last (%.0f per column)" % (nb, span, span / nb))
names = (("0 -> 1", 0, 1, "products (X = P T^T, X X^T), block to LDS"), ("1 -> 2", 1, 2, "barrier"),
         ("2 -> 3", 2, 3, "B wave: eight strips"), ("3 -> 5", 3, 5, "T wave's tail (last strip, its stores)"),
         ("5 -> 6", 5, 6, "closing barrier"), ("2 -> 7", 2, 7, "(prefetch of the next column's operands, beside the strips)"),
         ("0 -> 6", 0, 6, "whole column"))
mid = s[2:nb - 1]
for tag, a, b, what in names:
    d = mid[:, b] - mid[:, a]
    print("  %-7s mean %7.0f  min %7.0f  max %7.0f clocks   %s" % (tag, d.mean(), d.min(), d.max(), what))
gap = s[3:nb - 1, 0] - s[2:nb - 2, 6]
print("  6 -> next 0: mean %.0f clocks" % gap.mean())
