#!/usr/bin/env python
"""run_kalmanfilter.py [input_video] [optic_flow_path] [output_file] -- HydraGL tracker on MI355X.

Same arguments as the reference CLI (reference run_kalmanfilter.py:38-53):

    fn_in            input video.  No OpenCV here: a .npy / .npz array of shape (frames, H, W)
                     or (frames, H, W, 3) (BGR, converted as cvtColor BGR2GRAY does), 8-bit
    flow_in          optic flow path prefix: <flow_in>_%03d_x.mat / _y.mat as written by the
                     flow tool (optical_flow_ext.py); if no flow files exist the flow is
                     computed in-process with the same Brox defaults, streamed to the filter on
                     the GPU (hydra_mi.pipeline.FlowEKFPipeline)
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
from hydra_mi import kalman
from hydra_mi.distmesh_dyn import DistMesh
from hydra_mi.pipeline import FlowEKFPipeline, VideoStream
from hydra_mi.renderer import FlowStream


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("fn_in", default="./video/johntest_brightcontrast_short.npy", nargs="?",
                        help="input video file (.npy/.npz, frames x H x W[ x 3], uint8)")
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

    capture = VideoStream(args.fn_in, args.threshold)          # reference run_kalmanfilter.py:58
    frame = capture.current_frame()
    mask, ctrs, fd = capture.backsub()
    distmesh = DistMesh(frame, h0=args.gridsize)               # reference run_kalmanfilter.py:62-63
    distmesh.createMesh(ctrs, fd, frame, plot=False)

    flowstream = FlowStream(args.flow_in)
    ret_flow, flowframe = flowstream.peek()
    states, errors = [], []

    def keep(kf, e):
        states.append(kf.state.X.reshape(-1).copy())
        errors.append([float(e[0]), float(e[1]), float(e[2]), float(e[3])])

    if ret_flow:
        # the reference's loop (:78-89): one flow file per frame
        kf = kalman.IteratedMSKalmanFilter(distmesh, frame, flowframe, cuda=args.cuda, sparse=True, multi=True)
        count = 0
        while capture.isOpened():
            count += 1
            ret, _, grayframe, m = capture.read()
            ret_flow, flowframe = flowstream.read()
            if ret is False or ret_flow is False:
                break
            print("Frame %d" % count)
            keep(kf, kf.compute(grayframe, flowframe, m))
    else:
        # no flow files: flow and filter in one process, the flow of the coming frames computed on the GPU
        # beside the filter (hydra_mi.pipeline; replaces the file hand-off of reference README.md:26-31)
        print("Cannot read flow stream at %s*: computing Brox flow in-process" % args.flow_in)
        kf = kalman.IteratedMSKalmanFilter(distmesh, frame, np.zeros(frame.shape + (2,), np.float32), cuda=args.cuda,
                                           sparse=True, multi=True)
        pipe = FlowEKFPipeline(kf, capture)          # frames, masks and background-subtracted frames read from the stream

        def on_frame(k, e):
            print("Frame %d" % (k + 1))
            keep(kf, e)
        pipe.run(on_frame=on_frame)
        pipe.close()
    np.savez(args.fn_out, X=np.array(states), err=np.array(errors), p=distmesh.p, t=kf.state.tri)
    print("Finished: %d frames, states in %s" % (len(states), args.fn_out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
