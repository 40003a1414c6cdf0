"""Development aid: iteration counts / states of the bench's 1024^2 track under the switches of round 3 (device Newton,
covariance queued ahead, speculative measurement) -- frame by frame without the pipeline, flows computed once."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth, brox

n, frames = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 67
video, masks, c, r = synth.disk_video(n, frames, "translate_leftup", 0)
dm0 = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)
bf = brox.BroxOpticalFlow(n, n)
flows = []
for k in range(frames - 1):
    u, v = bf.calc(video[k], video[k + 1])
    flows.append(np.dstack((u, v)))


def track(dev, cov, spec):
    kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(dm0.p, dm0.t, dm0.h0), video[0], np.zeros((n, n, 2), np.float32), True)
    kf.newton_on_device, kf.cov_ahead = dev, cov
    kf.state.renderer.tune("speculate", spec)
    out = []
    for k in range(frames - 1):
        kf.compute(video[k + 1], flows[k], masks[k + 1])
        out.append((kf.state.X.copy(), kf.niter, kf.newton_iterations))
    kf.close()
    return out


ref = track(False, False, 0)
print("host Newton, nothing ahead: iterations", sum(o[1] for o in ref), [o[1] for o in ref], flush=True)
for name, args in (("the same again", (False, False, 0)), ("device Newton", (True, False, 0)), ("covariance ahead", (False, True, 0)),
                   ("speculative measurement", (False, False, 1)), ("all three", (True, True, 1)), ("all three again", (True, True, 1))):
    got = track(*args)
    d = [float(np.abs(a[0] - b[0]).max()) for a, b in zip(ref, got)]
    first = next((k for k in range(len(ref)) if ref[k][1] != got[k][1]), None)
    print("%-26s iterations %d, Newton its equal %s, max |dX| %.3g (frame %d), first frame with another iteration count: %s"
          % (name, sum(o[1] for o in got), [o[2] for o in ref] == [o[2] for o in got], max(d), int(np.argmax(d)), first), flush=True)
