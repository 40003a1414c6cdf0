"""Development aid: SOR time of the finest level (1024^2) per pair for series of 1..8 pairs, 512 threads throughout -- does a
smaller working set (k_prepare's seven planes of fewer pairs: 28 B/px) come back from the L2 / Infinity Cache faster?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, synth
n = 1024
f0, f1, _, _ = synth.warp_pair(n, "translate_leftup_stretch", 0)
for b in (8, 4, 2, 1, 8):
    F0 = torch.from_numpy(np.stack([f0] * b)).cuda(); F1 = torch.from_numpy(np.stack([f1] * b)).cuda()
    U = torch.empty((b, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
    bf = brox.BroxOpticalFlow(n, n, max_batch=b)
    bf.tune("sor_threads", 512)
    for _ in range(2):
        bf.calc_dev(b, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
    bf.sync()
    bf.profile(True)
    reps = 4
    for _ in range(reps):
        bf.calc_dev(b, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
    lv = bf.profile_levels()
    bf.profile(False)
    for k in (0, 1, 2):
        l = lv[k]
        print("batch %d level %d (%dx%d): %d SOR launches, %.1f us per launch, %.1f us per launch and pair" %
              (b, k, l["w"], l["h"], l["launches"] // reps, 1e3 * l["ms"] / l["launches"], 1e3 * l["ms"] / l["launches"] / b), flush=True)
