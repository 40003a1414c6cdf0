"""Development aid: the bench's workload (1024^2, 201 vertices) run several times -- pipelined with different
schedules, and frame by frame without any overlap -- must give bit-identical states and iteration counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth, brox
from hydra_mi.pipeline import FlowEKFPipeline

n, frames = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 25
video, masks, c, r = synth.disk_video(n, frames, "translate_leftup", 0)
dm0 = mesh.disk_mesh(c[0], c[1], r - 1.0, 0.047 * n)


def pipelined(reserve, batch, two=False):
    kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(dm0.p, dm0.t, dm0.h0), video[0], np.zeros((n, n, 2), np.float32), True)
    if os.environ.get("CHOL_FLOW"):
        kf.state.renderer.tune("chol_flow", int(os.environ["CHOL_FLOW"]))
    pipe = FlowEKFPipeline(kf, video, masks, flow_batch=batch, cu_reserve=reserve, concurrent_series=two)
    out = []
    pipe.run(0, frames - 1, on_frame=lambda k, e: out.append((kf.state.X.copy(), kf.niter)))
    pipe.close()
    return out


def sequential():
    kf = kalman.IteratedMSKalmanFilter(mesh.Mesh(dm0.p, dm0.t, dm0.h0), video[0], np.zeros((n, n, 2), np.float32), True)
    if os.environ.get("CHOL_FLOW"):
        kf.state.renderer.tune("chol_flow", int(os.environ["CHOL_FLOW"]))
    bf = brox.BroxOpticalFlow(n, n)
    out = []
    for k in range(frames - 1):
        u, v = bf.calc(video[k], video[k + 1])
        kf.compute(video[k + 1], np.dstack((u, v)), masks[k + 1])
        out.append((kf.state.X.copy(), kf.niter))
    return out


ref = sequential()
print("sequential: iterations", [o[1] for o in ref], flush=True)
for name, fn in [("pipelined batch 8", lambda: pipelined(32, 8)), ("pipelined batch 8 again", lambda: pipelined(32, 8)),
                 ("pipelined batch 8, no CUs reserved", lambda: pipelined(0, 8)), ("pipelined batch 3", lambda: pipelined(32, 3)),
                 ("pipelined batch 8, two series in flight", lambda: pipelined(32, 8, True))]:
    got = fn()
    bad = [k for k in range(len(ref)) if not np.array_equal(ref[k][0], got[k][0]) or ref[k][1] != got[k][1]]
    print("%-40s identical to sequential: %s%s" % (name, not bad, "" if not bad else "; first difference at frame %d (iterations %d vs %d, max |dX| %.3g)"
          % (bad[0] + 1, ref[bad[0]][1], got[bad[0]][1], np.abs(ref[bad[0]][0] - got[bad[0]][0]).max())), flush=True)
