"""Development aid: the 1024^2 disk video of several seeds tracked one after the other (single thread)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, kalman, mesh, synth
from hydra_mi.renderer import DeviceObservation
n = 1024
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 14
B = int(sys.argv[3]) if len(sys.argv) > 3 else 7
bf = brox.BroxOpticalFlow(n, n, max_batch=B)
if os.environ.get('NOGRAPH'): bf.tune('graph', 0)
for v in [int(x) for x in sys.argv[1].split(",")]:
    video, masks, c, r = synth.disk_video(n, nf + 1, "translate_leftup", v)
    dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
    dv = torch.from_numpy(video).cuda(); dmk = torch.from_numpy(masks).cuda()
    U = torch.empty((nf, n, n), dtype=torch.float32, device="cuda"); Vv = torch.empty_like(U)
    if os.environ.get('TSYNC'): torch.cuda.synchronize()
    for k in range(0, nf, B):
        nbp = min(B, nf - k)
        bf.calc_dev(nbp, dv[k].data_ptr(), dv[k + 1].data_ptr(), U[k].data_ptr(), Vv[k].data_ptr())
    bf.sync()
    print("seed", v, "flow finite", bool(torch.isfinite(U).all()), bool(torch.isfinite(Vv).all()), "abs max", float(U.abs().max()), float(Vv.abs().max()), flush=True)
    if os.environ.get('FLOWONLY'): continue
    kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
    its = []
    try:
        for k in range(nf):
            obs = DeviceObservation(dv[k + 1].data_ptr(), U[k].data_ptr(), Vv[k].data_ptr(), dmk[k + 1].data_ptr(), y_m_host=masks[k + 1])
            kf.compute(obs, None, None)
            its.append(kf.niter)
        print("seed", v, "ok iterations", its, flush=True)
    except FloatingPointError as e:
        print("seed", v, "FAILED at frame", len(its), its, e, flush=True)
