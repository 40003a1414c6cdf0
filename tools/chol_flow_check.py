"""Development aid: the persistent factorisation launch (chol_flow = 1) against the launch-per-step form (0):
bit-identical step / covariance, and the time of an update iteration with either."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import mesh, renderer, synth


class S:
    pass


def scene(n, h0):
    video, masks, c, r = synth.disk_video(n, 2, "translate_leftup", 0)
    dm = mesh.disk_mesh(c[0], c[1], r - 1.0, h0 * n)
    N = dm.size()
    R = renderer.Renderer(dm, np.zeros((N, 2)), np.zeros((n, n, 2), np.float32), n, video[0], True, 1e-3, 1.0, 1.0)
    rng = np.random.default_rng(1)
    flow = np.zeros((n, n, 2), np.float32) - 2.5 + rng.normal(0, 0.05, (n, n, 2)).astype(np.float32)
    X = np.concatenate((dm.p.reshape(-1) + rng.normal(0, 0.4, 2 * N), rng.normal(-2.5, 0.3, 2 * N)))
    n4 = 4 * N
    M = rng.normal(size=(n4, n4))
    W0 = np.eye(n4) * 0.5 + 0.05 * (M @ M.T) / n4
    return dm, N, R, video[1], flow, masks[1], X, W0


for n, h0 in ((256, 0.2), (256, 0.08), (512, 0.06), (1024, 0.047)):
    dm, N, R, y_im, flow, y_m, X, W0 = scene(n, h0)
    out = {}
    for mode in (0, 1):
        R.tune("chol_flow", mode)
        if os.environ.get("FLOW_WGS"):
            R.tune("chol_flow_wgs", int(os.environ["FLOW_WGS"]))
        R.update_frame(y_im, flow, y_m)
        res = R.update_run(W0, X, y_im, flow, y_m, 3, 1e-12)
        Wd = res[5].fetch()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter(); R.update_run(W0, X, y_im, flow, y_m, 3, 1e-12); ts.append(time.perf_counter() - t0)
        out[mode] = (res[0], Wd, res[2], min(ts))
    same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    print("n %d N %d (4N = %d): identical %s; 3 iterations + prior inverse: step launches %.3f ms, one launch %.3f ms"
          % (n, N, 4 * N, same, 1e3 * out[0][3], 1e3 * out[1][3]), flush=True)
