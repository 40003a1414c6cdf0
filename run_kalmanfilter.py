#!/usr/bin/env python
"""run_kalmanfilter.py [input_video] [optic_flow_path] [output_file] -- HydraGL tracker on MI355X.

Same arguments as the reference CLI (reference run_kalmanfilter.py:38-53):

    fn_in            input video.  No OpenCV here: a .npy / .npz array of shape (frames, H, W)
                     or (frames, H, W, 3), 8-bit
    flow_in          optic flow path prefix: <flow_in>_%03d_x.mat / _y.mat as written by the
                     flow tool (optical_flow_ext.py); if no flow files exist the flow is
                     computed in-process with the same Brox defaults
    fn_out           output file: the tracked states of all frames (np.savez)
    -n/--name        name for saving run images (accepted, unused: no screenshots on this path)
    -t/--threshold   threshold intensity below which is background (default 9)
    -s/--gridsize    edge length for mesh (default 22)
    -c/--cuda        whether to do the analysis on the GPU (default True; there is no CPU path)
"""
import argparse
import sys

import numpy as np

import hydra_mi  # noqa: F401
from hydra_mi import brox, kalman, mesh
from hydra_mi.renderer import FlowStream


def load_video(fn):
    a = np.load(fn)
    if hasattr(a, "files"):
        a = a[a.files[0]]
    a = np.asarray(a)
    if a.ndim == 4:
        a = a[..., 0]
    if a.ndim != 3 or a.dtype != np.uint8:
        raise SystemExit("%s: expected an 8-bit array of shape (frames, H, W[, 3])" % fn)
    return a


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("fn_in", default="./video/johntest_brightcontrast_short.npy", nargs="?",
                        help="input video file (.npy/.npz, frames x H x W, uint8)")
    parser.add_argument("flow_in", default="./video/johntest_brightcontrast_short/flow", nargs="?",
                        help="input optic flow path")
    parser.add_argument("fn_out", default="./video/johntest_brightcontrast_short_output.npz", nargs="?",
                        help="output file (tracked states)")
    parser.add_argument("-n", "--name", default="johntest_brightcontrast_short", nargs="?",
                        help="name for saving run images")
    parser.add_argument("-t", "--threshold", default=9, type=int,
                        help="threshold intensity below which is background")
    parser.add_argument("-s", "--gridsize", default=22, type=int,
                        help="edge length for mesh (smaller is finer; unstable much further below 18)")
    parser.add_argument("-c", "--cuda", default=True, type=bool, help="whether or not to do analysis on the GPU")
    args = parser.parse_args(argv)
    if len(sys.argv) == 1 and argv is None:
        print("No command line arguments provided, using defaults")

    video = load_video(args.fn_in)
    frame = video[0]
    mask = (frame > args.threshold).astype(np.uint8)
    distmesh = mesh.mask_mesh(mask, float(args.gridsize))
    frame = frame * mask

    flowstream = FlowStream(args.flow_in)
    ret_flow, flowframe = flowstream.peek()
    bf = None
    if not ret_flow:
        print("Cannot read flow stream at %s*: computing Brox flow in-process" % args.flow_in)
        bf = brox.BroxOpticalFlow(frame.shape[1], frame.shape[0])
        flowframe = np.dstack(bf.calc(video[0], video[1])) if len(video) > 1 else np.zeros(frame.shape + (2,), np.float32)

    kf = kalman.IteratedMSKalmanFilter(distmesh, frame, flowframe, cuda=args.cuda, sparse=True, multi=True)
    states, errors = [], []
    for count in range(1, len(video)):
        print("Frame %d" % count)
        gray = video[count]
        m = (gray > args.threshold).astype(np.uint8)
        if bf is None:
            ret_flow, flowframe = flowstream.read()
            if not ret_flow:
                break
        else:
            flowframe = np.dstack(bf.calc(video[count - 1], video[count]))
        e = kf.compute(gray * m, flowframe, m)
        states.append(kf.state.X.reshape(-1).copy())
        errors.append([float(e[0]), float(e[1]), float(e[2]), float(e[3])])
    np.savez(args.fn_out, X=np.array(states), err=np.array(errors), p=distmesh.p, t=kf.state.tri)
    print("Finished: %d frames, states in %s" % (len(states), args.fn_out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
