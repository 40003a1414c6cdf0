"""One batched Brox call at 1024^2 for PMC collection (rocprofv3 --pmc ...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, synth
n, B = 1024, 8
f0, f1, _, _ = synth.warp_pair(n, "translate_leftup_stretch", 0)
F0 = torch.from_numpy(np.stack([f0] * B)).cuda(); F1 = torch.from_numpy(np.stack([f1] * B)).cuda()
U = torch.empty((B, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
torch.cuda.synchronize()
bf = brox.BroxOpticalFlow(n, n, max_batch=B); bf.tune("sor_threads", 512)
for _ in range(2):
    bf.calc_dev(B, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
bf.sync()
print("done")
