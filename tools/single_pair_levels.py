"""Development aid: where one 1024^2 pair (and a series of 8) spends its time, by pyramid level -- a kernel trace of one
call, every launch attributed to its level by its grid: python tools/single_pair_levels.py <pairs> under rocprofv3, then
python tools/single_pair_levels.py --trace <csv> <pairs>."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(pairs):
    import numpy as np, torch
    import hydra_mi
    from hydra_mi import brox, synth
    n = 1024
    f0, f1, _, _ = synth.warp_pair(n, "translate_leftup_stretch", 0)
    F0 = torch.from_numpy(np.stack([f0] * pairs)).cuda(); F1 = torch.from_numpy(np.stack([f1] * pairs)).cuda()
    U = torch.empty((pairs, n, n), dtype=torch.float32, device="cuda"); V = torch.empty_like(U)
    bf = brox.BroxOpticalFlow(n, n, max_batch=pairs)
    for _ in range(3):
        bf.calc_dev(pairs, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
    bf.sync()
    bf.close() if hasattr(bf, "close") else None

def trace(path, pairs):
    import csv, collections
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
                  for r in csv.DictReader(open(path)))
    # the last call: from the last k_u8_to_f32 on
    start = max(i for i, r in enumerate(rows) if r[2].startswith("k_u8_to_f32"))
    rows = rows[start:]
    t0, t1 = rows[0][0], rows[-1][1]
    print("one call of %d pairs: %.3f ms, %d launches" % (pairs, (t1 - t0) / 1e6, len(rows)))
    # levels in launch order: a new level of the solve starts with k_deriv / k_deriv_all / k_coarse
    phase, cur = [], None
    busy = collections.OrderedDict()
    for k, (s, e, n) in enumerate(rows):
        if n.startswith(("k_deriv_all", "k_coarse")) or (n.startswith("k_deriv") and not rows[k - 1][2].startswith("k_deriv")):
            cur = "solve %d" % (len([p for p in busy if p.startswith("solve")]))
        elif cur is None:
            cur = "pyramid"
        d = busy.setdefault(cur, [0.0, 0, s, e, 0.0])
        d[0] += (e - s) / 1e3; d[1] += 1; d[3] = e
        if n.startswith("k_sor"): d[4] += (e - s) / 1e3
    for name, (kern, cnt, s, e, sor) in busy.items():
        print("  %-10s %4d launches  wall %8.1f us  kernels %8.1f us (k_sor %7.1f)  idle between launches %7.1f us"
              % (name, cnt, (e - s) / 1e3, kern, sor, (e - s) / 1e3 - kern))

if __name__ == "__main__":
    if sys.argv[1] == "--trace":
        trace(sys.argv[2], int(sys.argv[3]))
    else:
        run(int(sys.argv[1]))
