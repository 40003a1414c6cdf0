"""Generates tests/golden/*.npz from the oracle (run here, on the CPU; the GPU box only reads them).

The reference itself cannot be imported or run in this image (Python 2 sources, no OpenGL /
PyCUDA / OpenCV), so the vectors come from oracle/, which is pinned to the
reference's own known answers by tests/test_oracle_ekf.py.

    python tools/make_golden.py [config1|config3|config4|partitions|brox|measure|all]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hydra_mi                                     # noqa: E402  (input generators only)
from hydra_mi import mesh, synth                    # noqa: E402
from oracle import brox_oracle, ekf_c, ekf_ref      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def config1(frames=10):
    """BASELINE config 1: 128x128 translating square (synth.py:10-42), gridsize 15, exact flow."""
    video, flow = synth.test_data(128, 128)
    # the box occupies rows [43,86) (flipped) and columns [42,85): mesh over its extent
    dm = mesh.box_mesh(42.0, 43.0, 85.0, 86.0, 15.0)
    tr = ekf_ref.Tracker(dm.p, dm.t, dm.bars, dm.L, video[:, :, 0])
    Xs, errs, iters = [], [], []
    t0 = time.time()
    for k in range(frames):
        frame = video[:, :, k]
        mask = (frame > 0).astype(np.uint8)
        e = tr.compute(frame, flow[:, :, :, k], mask)
        Xs.append(tr.X.reshape(-1).copy())
        errs.append([float(e[0]), e[1], e[2], float(e[3])])
        iters.append(tr.niter)
        print("frame %d: iters %d, err %s, %.0fs" % (k, tr.niter, errs[-1], time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "config1_track.npz"), p=dm.p, t=dm.t, bars=dm.bars, L=dm.L,
                        X=np.array(Xs), err=np.array(errs), iters=np.array(iters), W_last=tr.W)


def config3(frames=5, n=512, h0=0.12):
    """BASELINE config 3: 512x512 video of a textured disk advected by the reference's `warp` field
    (synthetic/flowfields.py:7), ~40-vertex mesh (h0 = 0.12 W), flow from the C Brox oracle with the
    reference defaults, IteratedMSKalmanFilter defaults.  The measurement model is the C twin of
    ekf_ref.Measurement (oracle/ekf_ref_c.c; tests/test_oracle_ekf_c.py holds the two together): the NumPy
    one needs ~4 300 full-frame renders per IEKF iteration here."""
    video, masks, centre, radius = synth.disk_video(n, frames, "warp", 0)
    dm = mesh.disk_mesh(centre[0], centre[1], radius - 1.0, h0 * n)
    threads = max(1, min(16, os.cpu_count() or 1))
    meas = lambda *a: ekf_c.Measurement(*a, threads=threads)
    tr = ekf_ref.Tracker(dm.p, dm.t, dm.bars, dm.L, video[0], measurement=meas)
    brox_oracle.set_threads(threads)
    Xs, errs, iters = [], [], []
    t0 = time.time()
    for k in range(1, frames):
        u, v = brox_oracle.calc(video[k - 1], video[k])
        e = tr.compute(video[k], np.dstack((u, v)).astype(np.float32), masks[k])
        Xs.append(tr.X.reshape(-1).copy())
        errs.append([float(e[0]), e[1], e[2], float(e[3])])
        iters.append(tr.niter)
        print("frame %d: iters %d, err %s, %.0fs" % (k, tr.niter, errs[-1], time.time() - t0), flush=True)
    brox_oracle.set_threads(1)
    np.savez_compressed(os.path.join(OUT, "config3_track.npz"), n=n, frames=frames, h0=h0, p=dm.p, t=dm.t, bars=dm.bars,
                        L=dm.L, X=np.array(Xs), err=np.array(errs), iters=np.array(iters), W_last=tr.W)


def config4(frames=4, n=1024, h0=0.047):
    """BASELINE config 4: 1024x1024 video of the textured disk advected by `translate_leftup`
    (synthetic/flowfields.py:3), the bench's 201-vertex mesh (h0 = 0.047 W), flow from the C Brox oracle with
    the reference defaults, IteratedMSKalmanFilter defaults (reference kalman.py:774-831, 835-960), the
    measurement model the C/OpenMP twin of ekf_ref.Measurement: every jz / j evaluation a full-frame render
    (cuda.py:972-1010).  ~3 min per frame on 8 cores."""
    video, masks, centre, radius = synth.disk_video(n, frames, "translate_leftup", 0)
    dm = mesh.disk_mesh(centre[0], centre[1], radius - 1.0, h0 * n)
    threads = max(1, min(16, os.cpu_count() or 1))
    meas = lambda *a: ekf_c.Measurement(*a, threads=threads)
    tr = ekf_ref.Tracker(dm.p, dm.t, dm.bars, dm.L, video[0], measurement=meas)
    brox_oracle.set_threads(threads)
    Xs, errs, iters, Wd = [], [], [], []
    t0 = time.time()
    for k in range(1, frames):
        u, v = brox_oracle.calc(video[k - 1], video[k])
        e = tr.compute(video[k], np.dstack((u, v)).astype(np.float32), masks[k])
        Xs.append(tr.X.reshape(-1).copy())
        Wd.append(np.diag(tr.W).copy())
        errs.append([float(e[0]), e[1], e[2], float(e[3])])
        iters.append(tr.niter)
        print("frame %d: iters %d, err %s, %.0fs" % (k, tr.niter, errs[-1], time.time() - t0), flush=True)
    brox_oracle.set_threads(1)
    np.savez_compressed(os.path.join(OUT, "config4_track.npz"), n=n, frames=frames, h0=h0, p=dm.p, t=dm.t, bars=dm.bars,
                        L=dm.L, X=np.array(Xs), err=np.array(errs), iters=np.array(iters), W_diag=np.array(Wd),
                        W_last_rows=tr.W[::67].copy())


def partitions():
    """The perturbation partitions of the BASELINE config-1 mesh from the statement-by-statement restatement
    of reference kalman.py:223-272, 305-389 (oracle/partitions_ref.py)."""
    from oracle import partitions_ref
    dm = mesh.box_mesh(42.0, 43.0, 85.0, 86.0, 15.0)
    N = dm.size()
    E, labels = partitions_ref.jacobian_partitions(N, dm.t)
    Q, EH, EHi, lh = partitions_ref.hessian_partitions(N, dm.t)
    np.savez_compressed(os.path.join(OUT, "partitions_config1.npz"), p=dm.p, t=dm.t,
                        E_flat=np.concatenate(E).astype(np.int64), E_len=np.array([len(e) for e in E]), labels=labels,
                        Q=Q, EH_idx_flat=np.concatenate(EHi).astype(np.int64), EH_len=np.array([len(e) for e in EHi]),
                        labels_hess=lh)


def brox():
    """Brox flow of small synthetic warps (reference defaults)."""
    out = {}
    for n, name in [(64, "warp"), (96, "rotate"), (128, "translate_leftup_stretch")]:
        f0, f1, tu, tv = synth.warp_pair(n, name, 0)
        u, v = brox_oracle.calc(f0, f1)
        out["f0_%d" % n], out["f1_%d" % n], out["u_%d" % n], out["v_%d" % n] = f0, f1, u, v
    np.savez_compressed(os.path.join(OUT, "brox_small.npz"), **out)


def measure():
    """Hz / HTH / error of one state against one observation on a 64x64 frame."""
    n = 64
    dm = mesh.disk_mesh(31.5, 31.5, 20.0, 11.0)
    N = dm.size()
    tex = synth.noise_texture(n, 2).astype(np.uint8)
    rng = np.random.default_rng(5)
    X = np.concatenate((dm.p.reshape(-1) + rng.normal(0, 0.7, 2 * N), rng.normal(0, 1.0, 2 * N)))
    Xobs = np.concatenate((dm.p.reshape(-1) + 1.5, np.full(2 * N, 0.5)))
    meas = ekf_ref.Measurement(N, dm.t, dm.p, tex, 1e-3, 1.0, 1.0)
    y_im, yfx, yfy, ym = meas.render(Xobs)
    y_m = (ym // 255).astype(np.uint8)
    flow = np.dstack((yfx, -yfy)).astype(np.float32) + rng.normal(0, 0.05, (n, n, 2)).astype(np.float32)
    Hz, Hzc = ekf_ref.jacobian(meas, X, y_im, flow, y_m)
    _, J = ekf_ref.adjacency(N, dm.t)
    HTH = ekf_ref.hessian_sparse(meas, X, J)
    err = meas.error(X, y_im, flow, y_m)
    np.savez_compressed(os.path.join(OUT, "measure_64.npz"), p=dm.p, t=dm.t, tex=tex, X=X, y_im=y_im, flow=flow,
                        y_m=y_m, Hz=Hz, Hzc=Hzc, HTH=HTH, err=np.array([float(err[0]), err[1], err[2], float(err[3])]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("brox", "all"):
        brox()
    if what in ("measure", "all"):
        measure()
    if what in ("config1", "all"):
        config1()
    if what in ("config3", "all"):
        config3()
    if what == "config4":                                # not in "all": half an hour of CPU
        config4()
    if what in ("partitions", "all"):
        partitions()
