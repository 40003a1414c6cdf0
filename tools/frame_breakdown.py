"""Development aid: where the host spends a frame of the streaming pipeline outside the update's iterations --
wall-clock around the pieces of IteratedMSKalmanFilter.compute (predict: spring blocks, queueing the covariance half,
queueing the prior's factorisation, waiting for the Newton worker; projectmask; update; error) at 1024^2 / 201 vertices."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth
from hydra_mi.pipeline import FlowEKFPipeline

n, warm, frames = 1024, 5, 20
video, masks, c, r = synth.disk_video(n, warm + frames + 1, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
pipe = FlowEKFPipeline(kf, video, masks, flow_batch=8)
T = collections.defaultdict(float)

def timed(obj, name, tag=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[tag or name] += time.perf_counter() - t0
    setattr(obj, name, g)

rdr = kf.state.renderer
for o, nm in ((kf, "predict"), (kf, "_spring_blocks"), (rdr, "cov_predict"), (rdr, "update_prefactor"), (kf, "_newton"),
              (kf, "projectmask"), (kf, "update"), (kf, "_after_update"), (kf, "error"), (rdr, "set_observation_dev"),
              (pipe, "flow_ready"), (rdr, "update_run")):
    if hasattr(o, nm):
        timed(o, nm)
pipe.run(0, warm)
T.clear()
t0 = time.perf_counter()
pipe.run(warm, warm + frames)
wall = time.perf_counter() - t0
print("%d frames, %.3f ms per frame" % (frames, 1e3 * wall / frames))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-22s %.3f ms per frame" % (k, 1e3 * v / frames))
pipe.close()
