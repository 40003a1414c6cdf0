"""One flow series of B pairs of n x n (argv: n B [reps] [key=value tunes]) for kernel traces of a single series."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, synth
n, B = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
f0, f1, _, _ = synth.warp_pair(n, "translate_leftup_stretch", 0)
F0 = torch.from_numpy(np.stack([f0] * B)).cuda(); F1 = torch.from_numpy(np.stack([f1] * B)).cuda()
U = torch.empty((B, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
torch.cuda.synchronize()
bf = brox.BroxOpticalFlow(n, n, max_batch=B)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    bf.tune(k, int(v))
for _ in range(2):
    bf.calc_dev(B, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
bf.sync()
t = time.perf_counter()
for _ in range(reps):
    bf.calc_dev(B, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
bf.sync()
print("n=%d B=%d: %.3f ms per series" % (n, B, 1e3 * (time.perf_counter() - t) / reps))
