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
    if os.environ.get("SPECULATE") is not None:
        kf.state.renderer.tune("speculate", int(os.environ["SPECULATE"]))
    if os.environ.get("COV_AHEAD") is not None:
        kf.cov_ahead = os.environ["COV_AHEAD"] != "0"
    if os.environ.get("NEWTON_DEV") is not None:
        kf.newton_on_device = os.environ["NEWTON_DEV"] != "0"
    if os.environ.get("PREDICT_AHEAD") is not None:
        kf.predict_ahead = os.environ["PREDICT_AHEAD"] != "0"
    pipe = FlowEKFPipeline(kf, video, masks, flow_batch=8)
    if os.environ.get("MODEL_RAMP") is not None:
        pipe.model_ramp = os.environ["MODEL_RAMP"] != "0"
    if os.environ.get("GC_FREEZE") is not None:
        pipe.gc_freeze = os.environ["GC_FREEZE"] != "0"

    out = []
    extra = []
    cb = lambda k, e: (out.append((kf.state.X.copy(), kf.niter, kf.newton_iterations, e[0], e[3])),
                       extra.append((kf.orig_x.copy(), kf.pred_x.copy(), getattr(kf, "proj_x", kf.pred_x).copy())))
    once.extra = extra
    try:
        pipe.run(0, warm, on_frame=cb)
        pipe.run(warm, warm + frames, on_frame=cb)
    except Exception as exc:
        print("FAILED after %d frames: %s" % (len(out), str(exc)[:150]), flush=True)
        print("   iterations so far", [o[1] for o in out], flush=True)
        raise
    W = np.array(kf.state.W)
    pipe.close()
    kf.close()
    kf.state.renderer.close()
    return out, W


ref, Wref = once()
ref_extra = once.extra
print("iterations", [o[1] for o in ref], flush=True)
bad = 0
for i in range(reps):
    got, W = once()
    diff = [k for k in range(len(ref)) if not np.array_equal(ref[k][0], got[k][0]) or ref[k][1:] != got[k][1:]]
    if diff or not np.array_equal(W, Wref):
        bad += 1
        k = diff[0] if diff else -1
        ex = once.extra
        if diff and k > 0:
            bad_pred = ex[k][1].reshape(-1)
            print("   bad prediction vs: previous frame's prediction %.3g, previous frame's state-before-predict %.3g, this frame's state-before-predict %.3g, the right prediction %.3g"
                  % (np.abs(bad_pred - ref_extra[k - 1][1].reshape(-1)).max(), np.abs(bad_pred - ref_extra[k - 1][0].reshape(-1)).max(),
                     np.abs(bad_pred - ref_extra[k][0].reshape(-1)).max(), np.abs(bad_pred - ref_extra[k][1].reshape(-1)).max()), flush=True)
            nn = bad_pred.size // 2
            print("   positions differ by %.3g, velocities by %.3g" % (np.abs(bad_pred[:nn] - ref_extra[k][1].reshape(-1)[:nn]).max(),
                                                                          np.abs(bad_pred[nn:] - ref_extra[k][1].reshape(-1)[nn:]).max()), flush=True)
        if diff:
            print("   frame %d: state before predict equal %s, predicted state equal %s (max diff %.3g)"
                  % (k, np.array_equal(ex[k][0], ref_extra[k][0]), np.array_equal(ex[k][1], ref_extra[k][1]),
                     np.abs(ex[k][1] - ref_extra[k][1]).max()), flush=True)
        print("repetition %d differs: first at frame %d (iterations %s vs %s, newton %s vs %s, max |dX| %.3g); covariance equal %s"
              % (i, k, ref[k][1], got[k][1], ref[k][2], got[k][2], np.abs(ref[k][0] - got[k][0]).max(), np.array_equal(W, Wref)), flush=True)
print("%d of %d repetitions differ" % (bad, reps))
