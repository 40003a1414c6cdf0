"""Development aid: where the host time of IteratedMSKalmanFilter.predict goes at the bench's size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth
n = 1024
video, masks, c, r = synth.disk_video(n, 12, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(dm.p, dm.t, dm.h0), video[0], np.zeros((n, n, 2), np.float32), True)
flow = np.zeros((n, n, 2), np.float32) - 1.28
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T.setdefault(name, []).append(time.perf_counter() - t0)
    setattr(obj, name, g)
wrap(kf, "_spring_blocks"); wrap(kf, "_newton"); wrap(kf.state.renderer, "cov_predict"); wrap(kf.state.renderer, "update_prefactor")
wrap(kf, "predict"); wrap(kf, "projectmask"); wrap(kf, "update"); wrap(kf, "error"); wrap(kf, "_after_update"); wrap(kf, "compute")
wrap(kf.state.renderer, "set_observation_dev"); wrap(kf.state.renderer, "update_run"); wrap(kf.state.renderer, "project_mask")
from hydra_mi.pipeline import FlowEKFPipeline
pipe = FlowEKFPipeline(kf, video, masks)
pipe.run(0, 11)
for k, v in T.items():
    print("%-18s ms per frame (last 6): %s" % (k, " ".join("%.3f" % (1e3 * x) for x in v[-6:])))
print("newton iterations of the last frame", kf.newton_iterations)
