"""Development aid: 8 pairs of 1024^2 as one series of 8 against two concurrent series of 4 (two handles, two streams)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hydra_mi
from hydra_mi import brox, synth

n = 1024
f0, f1, _, _ = synth.warp_pair(n, "translate_leftup_stretch", 0)
def bufs(b):
    F0 = torch.from_numpy(np.stack([f0] * b)).cuda(); F1 = torch.from_numpy(np.stack([f1] * b)).cuda()
    U = torch.empty((b, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
    return F0, F1, U, V
def timeit(handles, reps=5):
    for _ in range(2):
        for bf, b, (F0, F1, U, V) in handles: bf.calc_dev(b, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
    for bf, _, _ in handles: bf.sync()
    t = time.perf_counter()
    for _ in range(reps):
        for bf, b, (F0, F1, U, V) in handles: bf.calc_dev(b, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
    for bf, _, _ in handles: bf.sync()
    return (time.perf_counter() - t) / reps * 1e3
one = [(brox.BroxOpticalFlow(n, n, max_batch=8), 8, bufs(8))]
print("one series of 8: %.2f ms" % timeit(one), flush=True)
for split in ((4, 4), (2, 2, 2, 2), (6, 2), (5, 3)):
    hs = [(brox.BroxOpticalFlow(n, n, max_batch=b), b, bufs(b)) for b in split]
    print("concurrent series of %s: %.2f ms per 8 pairs" % (split, timeit(hs)), flush=True)
    for bf, _, _ in hs: bf.close() if hasattr(bf, "close") else None
print("one series of 8: %.2f ms" % timeit(one), flush=True)
