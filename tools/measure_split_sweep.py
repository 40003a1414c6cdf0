"""Development aid: time of an update iteration against the workgroups per vertex / edge job of the measurement kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "chol_flow_check.py")).read().split("for n, h0 in")[0])
dm, N, R, y_im, flow, y_m, X, W0 = scene(1024, 0.047)
R.update_frame(y_im, flow, y_m)


def run():
    R.update_run(W0, X, y_im, flow, y_m, 4, 1e-12)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter(); R.update_run(W0, X, y_im, flow, y_m, 4, 1e-12); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)


for key, vals in (("measure_split", (3, 4, 5, 6, 8, 10, 12, 16)), ("edge_split", (1, 2, 3, 4, 6))):
    for v in vals:
        R.tune(key, v)
        print("%s %2d: 4 iterations + prior inverse %.3f ms" % (key, v, run()), flush=True)
    R.tune(key, 5 if key == "measure_split" else 2)
