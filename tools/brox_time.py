"""Ad-hoc timing of the Brox pipeline variants on one GPU (development aid, not the bench)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hydra_mi
from hydra_mi import brox, synth

def run(n, batch, reps, tunes):
    f0, f1, _, _ = synth.warp_pair(n, "translate_leftup_stretch", 0)
    F0 = torch.from_numpy(np.stack([f0] * batch)).cuda()
    F1 = torch.from_numpy(np.stack([f1] * batch)).cuda()
    U = torch.empty((batch, n, n), dtype=torch.float32, device="cuda")
    V = torch.empty_like(U)
    torch.cuda.synchronize()
    bf = brox.BroxOpticalFlow(n, n, max_batch=batch)
    for tune in tunes:
        for k, v in tune.items():
            bf.tune(k, v)
        for _ in range(2):
            bf.calc_dev(batch, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
        bf.sync()
        t = time.perf_counter()
        for _ in range(reps):
            bf.calc_dev(batch, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
        t_enq = (time.perf_counter() - t) / reps          # host time to enqueue a series
        bf.sync()
        dt = (time.perf_counter() - t) / reps
        bf.profile(True)
        for _ in range(reps):
            bf.calc_dev(batch, F0.data_ptr(), F1.data_ptr(), U.data_ptr(), V.data_ptr())
        ms, launches, pxit, _ = bf.profile_read()
        bf.profile(False)
        print(json.dumps(dict(n=n, batch=batch, tune=tune, ms_per_call=dt * 1e3, enqueue_ms=t_enq * 1e3, pairs_per_s=batch / dt,
                              sor_ms_per_call=ms / reps, sor_launches=launches // reps,
                              sor_GBps_alg=52.0 * pxit / (ms * 1e-3) / 1e9 if ms else None)), flush=True)

if __name__ == "__main__":
    tunes = [dict(sor_fuse=0, sor_threads=256), dict(sor_fuse=0, sor_threads=512), dict(sor_fuse=0, sor_threads=1024),
             dict(sor_fuse=2, sor_threads=512), dict(sor_fuse=2, sor_threads=1024)]
    if len(sys.argv) > 1 and sys.argv[1] == "dry":        # where a SOR launch spends its time: loads + stores only
        run(1024, 8, 3, [dict(sor_fuse=0, sor_threads=512, sor_dry=0), dict(sor_fuse=0, sor_threads=512, sor_dry=1)])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "series":     # the product's configuration: a series of 8 pairs, 2 pairs, one pair
        run(1024, 8, 5, [dict()])
        run(1024, 2, 5, [dict()])
        run(1024, 1, 5, [dict()])
        run(512, 1, 5, [dict()])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coarse64":   # levels up to 64 px inside k_coarse<64> against launch per operator
        ct = [dict(coarse_max=32), dict(coarse_max=64), dict(coarse_max=32), dict(coarse_max=64)]
        for b in (1, 2, 4, 8):
            run(1024, b, 5, ct)
        run(512, 1, 5, ct)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "series8":    # series of 8, 6, 5, 4 pairs as the product runs them
        for b in (8, 6, 5, 4):
            run(1024, b, 5, [dict()])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wide":       # the 128 x 64 SOR tile from which level size on
        wt = [dict(sor_wide=0), dict(sor_wide=1024), dict(sor_wide=640), dict(sor_wide=400), dict(sor_wide=256), dict(sor_wide=128), dict(sor_wide=0)]
        run(1024, 8, 5, wt)
        run(1024, 4, 5, [dict(sor_wide=0), dict(sor_wide=400)])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coarse":     # the coarse end of the pyramid: one launch against launch per operator
        ct = [dict(coarse_max=32, sor_deep=0), dict(sor_deep=1), dict(sor_deep=2), dict(sor_deep=3), dict(sor_deep=4), dict(sor_deep=8)]
        run(64, 1, 20, ct)
        run(512, 1, 10, ct)
        run(1024, 1, 10, ct)
        run(1024, 8, 5, ct)
        sys.exit(0)
    run(512, 1, 5, tunes[:3])
    run(1024, 1, 5, tunes)
    run(1024, 8, 3, tunes)
