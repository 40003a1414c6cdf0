#!/usr/bin/env python
"""optical_flow_ext.py <video> <output prefix> [alpha] [gamma] [scale_factor] [inner_it] [outer_it] [solver_it]

The flow tool of the reference (reference src/optical_flow_ext.cpp:441-507) on MI355X:
for every consecutive frame pair computes the Brox flow and writes
<prefix>_%03d_x.mat and <prefix>_%03d_y.mat (format: src/optical_flow_ext.cpp:47-108).
Parameters and defaults as there (:453-488): alpha 0.197, gamma 50, scale_factor 0.8,
inner 10, outer 77, solver 10.  The video is a .npy / .npz array (frames, H, W[, 3]) uint8
(no OpenCV on this path).  All pairs of the video go to the GPU as one batch per
`HYDRA_MI_FLOW_BATCH` (default 16) pairs -- they are independent.
"""
import os
import sys

import numpy as np

import hydra_mi  # noqa: F401
from hydra_mi import brox, matio, pipeline


def main(av):
    if len(av) < 3:
        print(__doc__)
        return 1
    try:                                             # BGR -> gray as cvtColor(BGR2GRAY) (:366-368)
        a = pipeline.load_video(av[1])
    except (OSError, ValueError) as exc:
        sys.stderr.write("Failed to open the video: %s\n" % exc)
        return 1
    prefix = av[2]
    vals = [0.197, 50.0, 0.8, 10, 77, 10]
    for i, s in enumerate(av[3:9]):
        vals[i] = float(s) if i < 3 else int(s)
    alpha, gamma, scale, inner, outer, solver = vals
    print("Using Brox optic flow parameters:\n   alpha = %g\n   gamma = %g\n   scale_factor = %g\n"
          "   inner_iterations = %d\n   outer_iterations = %d\n   solver_iterations = %d"
          % (alpha, gamma, scale, inner, outer, solver))
    B = max(1, int(os.environ.get("HYDRA_MI_FLOW_BATCH", "16")))
    n = a.shape[0] - 1
    bf = brox.BroxOpticalFlow(a.shape[2], a.shape[1], alpha, gamma, scale, inner, outer, solver, max_batch=min(B, max(n, 1)))
    for s in range(0, n, B):
        e = min(n, s + B)
        fx, fy = bf.calc_batch(np.ascontiguousarray(a[s:e]), np.ascontiguousarray(a[s + 1:e + 1]))
        for k in range(s, e):
            matio.write_flow(prefix, k, fx[k - s], fy[k - s])
    print("Finished.")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
