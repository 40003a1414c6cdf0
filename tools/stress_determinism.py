"""Development aid: the bench's protocol (a warm-up phase, then 20 frames through the streaming pipeline) repeated in one
process; every repetition must give the first one's states and iteration counts bit for bit.  argv: repetitions"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth
from hydra_mi.pipeline import FlowEKFPipeline

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, warm, frames = 1024, 5, 20
video, masks, c, r = synth.disk_video(n, warm + frames + 1, "translate_leftup", 0)
dm0 = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)


def once():
    kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(dm0.p, dm0.t, dm0.h0), video[0], np.zeros((n, n, 2), np.float32), True)
    pipe = FlowEKFPipeline(kf, video, masks, flow_batch=8)
    out = []
    cb = lambda k, e: out.append((kf.state.X.copy(), kf.niter, kf.newton_iterations, e[0], e[3]))
    pipe.run(0, warm, on_frame=cb)
    pipe.run(warm, warm + frames, on_frame=cb)
    W = np.array(kf.state.W)
    pipe.close()
    kf.close()
    return out, W


ref, Wref = once()
print("iterations", [o[1] for o in ref], flush=True)
bad = 0
for i in range(reps):
    got, W = once()
    diff = [k for k in range(len(ref)) if not np.array_equal(ref[k][0], got[k][0]) or ref[k][1:] != got[k][1:]]
    if diff or not np.array_equal(W, Wref):
        bad += 1
        k = diff[0] if diff else -1
        print("repetition %d differs: first at frame %d (iterations %s vs %s, newton %s vs %s, max |dX| %.3g); covariance equal %s"
              % (i, k, ref[k][1], got[k][1], ref[k][2], got[k][2], np.abs(ref[k][0] - got[k][0]).max(), np.array_equal(W, Wref)), flush=True)
print("%d of %d repetitions differ" % (bad, reps))
