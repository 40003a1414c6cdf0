"""Development aid: the flow-series schedule of the streaming pipeline (sizes, per-frame waits) at the bench's size.
argv: flow_batch [warm frames] [timed frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth
from hydra_mi.pipeline import FlowEKFPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 1024
video, masks, c, r = synth.disk_video(n, warm + frames + 1, "translate_leftup", 0)
dm = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
kf = kalman.IteratedMSKalmanFilter(dm, video[0], np.zeros((n, n, 2), np.float32), True)
pipe = FlowEKFPipeline(kf, video, masks, flow_batch=B)
if os.environ.get("SPECULATE") is not None:
    kf.state.renderer.tune("speculate", int(os.environ["SPECULATE"]))
pipe.run(0, warm)
print("after warm-up: series times", {k: round(1e3 * v, 2) for k, v in pipe._series_s.items()}, "frame %.2f ms" % (1e3 * pipe._frame_s))
lines = []
T = {}
def timed(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)
for o, nm in ((kf, "predict"), (kf, "_newton"), (kf, "projectmask"), (kf, "update"), (kf, "error"), (kf.state.renderer, "set_observation_dev"),
              (kf.state.renderer, "update_run"), (kf.state.renderer, "predict_take")):
    timed(o, nm)
def tr(msg):
    lines.append(msg + "  | " + " ".join("%s %.2f" % (k, 1e3 * v) for k, v in T.items()) + " newton its %s reverted %s converged %s" % (getattr(kf, "newton_iterations", None), getattr(kf, "reverted", None), getattr(kf, "converged", None)))
    T.clear()
pipe.trace = tr
import gc
_g = {}
def _gccb(phase, info):
    if phase == "start":
        _g["t"] = time.perf_counter()
    else:
        lines.append("    gc generation %d: %.2f ms, %d collected" % (info["generation"], 1e3 * (time.perf_counter() - _g["t"]), info["collected"]))
gc.callbacks.append(_gccb)
if os.environ.get("GC_FREEZE"):
    gc.collect(); gc.freeze()
t0 = time.perf_counter()
pipe.run(warm, warm + frames)
wall = time.perf_counter() - t0
print("B=%d: %d frames, %.3f ms per frame, %.1f frames/s" % (B, frames, 1e3 * wall / frames, frames / wall))
for l in lines:
    print(" ", l)
pipe.close()
