"""The two command-line tools end to end on a small synthetic video (reference CLIs:
src/optical_flow_ext.cpp:441-507 and run_kalmanfilter.py:9-93)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flow_tool_then_tracker(hm, tmp_path, oracle_brox):
    from hydra_mi import matio, synth
    sys.path.insert(0, ROOT)
    import optical_flow_ext
    import run_kalmanfilter
    n, F = 96, 5
    video, masks, c, r = synth.disk_video(n, F, "translate_leftup", 0)
    vid = str(tmp_path / "video.npy")
    np.save(vid, video)
    prefix = str(tmp_path / "flow")
    # positional parameters as in the reference: alpha gamma scale inner outer solver
    assert optical_flow_ext.main(["optical_flow_ext", vid, prefix, "0.197", "50", "0.8", "5", "77", "5"]) == 0
    for k in range(F - 1):
        fx = matio.read_mat(prefix + "_%03d_x.mat" % k)
        fy = matio.read_mat(prefix + "_%03d_y.mat" % k)
        ru, rv = oracle_brox.calc(video[k], video[k + 1], inner=5, solver=5)
        assert fx.dtype == np.float32 and np.array_equal(fx, ru) and np.array_equal(fy, rv)
    assert not os.path.exists(prefix + "_%03d_x.mat" % (F - 1))
    out = str(tmp_path / "states.npz")
    assert run_kalmanfilter.main([vid, prefix, out, "-s", "14", "-t", "9"]) == 0
    res = np.load(out)
    assert res["X"].shape[0] == F - 1 and np.all(np.isfinite(res["X"]))
    N = res["p"].shape[0]
    disp = res["X"][-1][:2 * N].reshape(-1, 2).mean(0) - res["p"].mean(0)
    true = np.array(synth.scaled_field("translate_leftup", n)(0.0, 0.0)) * (F - 1)
    assert np.linalg.norm(disp - true) < 1.0           # the mesh follows the object
    # without flow files the tracker computes the flow itself
    out2 = str(tmp_path / "states2.npz")
    assert run_kalmanfilter.main([vid, str(tmp_path / "nothing"), out2, "-s", "14"]) == 0
    res2 = np.load(out2)
    assert res2["X"].shape == res["X"].shape and np.all(np.isfinite(res2["X"]))
    assert np.abs(res2["X"][-1][:2 * N] - res["X"][-1][:2 * N]).max() < 1.5   # flows with 10/10 vs 5/5 iterations
