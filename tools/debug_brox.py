import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hydra_mi
from hydra_mi import brox, synth
from oracle import brox_oracle as bo
t0 = time.time()
f0, f1, _, _ = synth.warp_pair(64, "warp", 0)
def cmp(tag, kw, okw):
    t = time.time()
    bf = brox.BroxOpticalFlow(64, 64, **kw)
    u, v = bf.calc(f0, f1)
    ru, rv = bo.calc(f0, f1, **okw)
    d = max(np.abs(u - ru).max(), np.abs(v - rv).max())
    print("%-40s maxdiff %.3e  |u|max %.3f  (%.2fs)" % (tag, d, np.abs(ru).max(), time.time() - t), flush=True)
for o in (3, 4, 5, 6, 7, 8):
    cmp("outer%d" % o, dict(outer_iterations=o), dict(outer=o))
for o in (4, 8):
    cmp("outer%d inner1 solver1" % o, dict(outer_iterations=o, inner_iterations=1, solver_iterations=1), dict(outer=o, inner=1, solver=1))
    cmp("outer%d gamma0" % o, dict(outer_iterations=o, gamma=0.0), dict(outer=o, gamma=0.0))
print("total", time.time() - t0)
