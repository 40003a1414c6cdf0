"""Development aid: the streaming ring test with switches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import kalman, mesh, synth
from hydra_mi.pipeline import FlowEKFPipeline, VideoStream, threshold_mask
n, F = 96, 16
video, _, c, r = synth.disk_video(n, F, "warp", 2)
masks = np.stack([threshold_mask(f, 9) for f in video])
zero = np.zeros((n, n, 2), np.float32)

def track(make, **kw):
    kf = kalman.IteratedMSKalmanFilter(mesh.disk_mesh(c[0], c[1], r - 1.0, 14.0), video[0] * masks[0], zero, True, nI=3)
    for k, v in kw.items():
        setattr(kf, k, v)
    pipe = make(kf)
    got = []
    lines = []
    pipe.trace = lines.append
    pipe.run(on_frame=lambda k, e: got.append((kf.state.X.copy(), e[:4], kf.niter)))
    pipe.close(); kf.close()
    return got, lines

def mk(resident=False, ramp=True, src="array"):
    def f(kf):
        if src == "array":
            p = FlowEKFPipeline(kf, video, masks, observed=video * masks, flow_batch=2, resident=resident)
        else:
            p = FlowEKFPipeline(kf, VideoStream(video, 9), flow_batch=2)
        p.model_ramp = ramp
        return p
    return f

ref, _ = track(mk(resident=True, ramp=False))
for name, make, kw in (("resident, model ramp", mk(True, True), {}), ("ring, fixed ramp", mk(False, False), {}), ("ring, model ramp", mk(False, True), {}),
                       ("stream source, model ramp", mk(False, True, "stream"), {}), ("stream source, fixed ramp", mk(False, False, "stream"), {}),
                       ("ring, model ramp, host newton", mk(False, True), dict(newton_on_device=False)),
                       ("ring, model ramp again", mk(False, True), {})):
    got, lines = track(make, **kw)
    bad = [k for k in range(len(ref)) if not np.array_equal(got[k][0], ref[k][0]) or got[k][1] != ref[k][1] or got[k][2] != ref[k][2]]
    print("%-32s differing frames: %s" % (name, bad), flush=True)
    if bad:
        for l in lines[max(0, bad[0] - 2):bad[0] + 2]:
            print("    ", l[:150])

print("--- the test's structure: nothing closed in between ---")
def track2(make):
    kf = kalman.IteratedMSKalmanFilter(mesh.disk_mesh(c[0], c[1], r - 1.0, 14.0), video[0] * masks[0], zero, True, nI=3)
    pipe = make(kf)
    got = []
    lines = []
    pipe.trace = lines.append
    pipe.run(on_frame=lambda k, e: got.append((kf.state.X.copy(), e[:4], kf.niter)))
    return pipe, kf, got, lines
for rep in range(3):
    pr, kfr, ref2, _ = track2(lambda kf: FlowEKFPipeline(kf, video, masks, observed=video * masks, flow_batch=2, resident=True))
    ps, kfs, got, l1 = track2(lambda kf: FlowEKFPipeline(kf, video, masks, observed=video * masks, flow_batch=2))
    pv, kfv, gotv, l2 = track2(lambda kf: FlowEKFPipeline(kf, VideoStream(video, 9), flow_batch=2))
    for name, g, lines, p in (("ring", got, l1, ps), ("stream source", gotv, l2, pv)):
        bad = [k for k in range(len(ref2)) if not np.array_equal(g[k][0], ref2[k][0]) or g[k][1] != ref2[k][1] or g[k][2] != ref2[k][2]]
        print("rep %d %-16s pinned %s differing frames: %s; vs first reference: %s" % (rep, name, getattr(p.source, "pinned", None), bad,
              [k for k in range(len(ref)) if not np.array_equal(g[k][0], ref[k][0])]), flush=True)
        if bad:
            for l in lines[max(0, bad[0] - 2):bad[0] + 2]:
                print("    ", l[:150])
    print("   resident vs first reference:", [k for k in range(len(ref)) if not np.array_equal(ref2[k][0], ref[k][0])])
