"""Development aid: where the interpreter spends a frame of the streaming pipeline (cProfile over 20 frames at 1024^2)."""
import os, sys, cProfile, pstats
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
pipe.run(0, warm)
pr = cProfile.Profile()
pr.enable()
pipe.run(warm, warm + frames)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
pipe.close()
