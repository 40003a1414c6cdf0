"""How close the device track of BASELINE config 1 is to the committed oracle track (the test asks for 1e-5)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import mesh, synth, kalman
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config1_track.npz"))
video, flow = synth.test_data(128, 128)
dm = mesh.Mesh(g["p"], g["t"], 15.0)
kf = kalman.IteratedMSKalmanFilter(dm, video[:, :, 0], flow[:, :, :, 0], True)
for k in range(g["X"].shape[0]):
    frame = video[:, :, k]
    kf.compute(frame, flow[:, :, :, k], (frame > 0).astype(np.uint8))
    X = kf.state.X.reshape(-1)
    print("frame %d: rel %.3e  iterations %d (oracle %d)" % (k, np.linalg.norm(X - g["X"][k]) / np.linalg.norm(g["X"][k]), kf.niter, int(g["iters"][k])))
W = kf.state.W
print("W rel", np.linalg.norm(W - g["W_last"]) / np.linalg.norm(g["W_last"]))
