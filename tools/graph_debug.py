"""Development aid: the hipGraph replay problem seen with ROCm 7.2 (why hm_brox_tune "graph" is off by default).
A handle's launch series replayed as a graph gives garbage from about the fifth launch on when the caller
allocates device memory between calls; the same calls with direct launches are bit-identical every time.

  python tools/graph_debug.py fixed|realloc [graph]     (graph: switch the replay on)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_mi
from hydra_mi import brox, synth
n, nf, B = 1024, 14, 7
mode = sys.argv[1] if len(sys.argv) > 1 else "realloc"
bf = brox.BroxOpticalFlow(n, n, max_batch=B)
if "graph" in sys.argv[2:]:
    bf.tune("graph", 1)
video, masks, c, r = synth.disk_video(n, nf + 1, "translate_leftup", 0)


def alloc():
    dv = torch.from_numpy(video).cuda()
    U = torch.empty((nf, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
    torch.cuda.synchronize()
    return dv, U, V


dv, U, V = alloc()
ref = None
for it in range(6):
    if mode == "realloc":
        dv, U, V = alloc()
    for k in range(0, nf, B):
        bf.calc_dev(B, dv[k].data_ptr(), dv[k + 1].data_ptr(), U[k].data_ptr(), V[k].data_ptr())
    bf.sync()
    same = ref is None or bool(torch.equal(U, ref))
    if ref is None:
        ref = U.clone()
    print(mode, it, "finite", bool(torch.isfinite(U).all()), "same as first", same, "abs max", float(U.abs().max()), flush=True)
