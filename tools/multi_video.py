"""Development aid: V independent videos tracked concurrently on one GPU (one thread, filter and stream each);
flow precomputed.  Prints aggregate frames/s."""
import sys, os, time, threading, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, kalman, mesh, synth
from hydra_mi.renderer import DeviceObservation
faulthandler.dump_traceback_later(90, exit=True)
n, nf = 1024, 14
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pipes = []
bf = brox.BroxOpticalFlow(n, n, max_batch=7)
for v in range(V):
    video, masks, c, r = synth.disk_video(n, nf + 1, "translate_leftup", v)
    dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
    dv = torch.from_numpy(video).cuda(); dmk = torch.from_numpy(masks).cuda()
    U = torch.empty((nf, n, n), dtype=torch.float32, device="cuda"); Vv = torch.empty_like(U)
    torch.cuda.synchronize()
    for k in range(0, nf, 7):
        bf.calc_dev(7, dv[k].data_ptr(), dv[k + 1].data_ptr(), U[k].data_ptr(), Vv[k].data_ptr())
    bf.sync()
    kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
    pipes.append((kf, dv, dmk, U, Vv, masks))
barrier = threading.Barrier(V + 1)
iters = [0] * V
failed = []
def run(v):
    kf, dv, dmk, U, Vv, masks = pipes[v]
    try:
        for k in range(nf):
            if k == 2:
                barrier.wait()
            obs = DeviceObservation(dv[k + 1].data_ptr(), U[k].data_ptr(), Vv[k].data_ptr(), dmk[k + 1].data_ptr(), y_m_host=masks[k + 1])
            kf.compute(obs, None, None)
            if k >= 2:
                iters[v] += kf.niter
    except Exception as e:
        failed.append((v, repr(e)))
        barrier.abort()
        return
    barrier.wait()
ths = [threading.Thread(target=run, args=(v,)) for v in range(V)]
for t in ths: t.start()
barrier.wait(); torch.cuda.synchronize(); t0 = time.perf_counter()
barrier.wait(); torch.cuda.synchronize(); el = time.perf_counter() - t0
for t in ths: t.join()
print("videos %d: %.1f frames/s aggregate, %.2f ms per frame per video, iterations/frame %.2f %s" % (V, V * (nf - 2) / el, 1e3 * el / (nf - 2), sum(iters) / (V * (nf - 2)), failed))
