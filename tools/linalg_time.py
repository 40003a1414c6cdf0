"""Dense f64 algebra of the filter update at n = 4N: host LAPACK vs rocSOLVER through torch."""
import time, os, sys
import numpy as np, scipy.linalg as sla
import torch
from threadpoolctl import threadpool_limits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 804
rng = np.random.default_rng(0)
M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n); b = rng.normal(size=(n, 4))
def tm(f, reps=5):
    f(); t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps * 1e3
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 4, 8, 16):
    with threadpool_limits(limits=th):
        print("host threads %2d: lu_factor+solve %.2f ms  cho %.2f ms  inv %.2f ms  gemm %.2f ms" % (
            th, tm(lambda: sla.lu_solve(sla.lu_factor(A, check_finite=False), b, check_finite=False)),
            tm(lambda: sla.cho_solve(sla.cho_factor(A, check_finite=False), b, check_finite=False)),
            tm(lambda: np.linalg.inv(A)), tm(lambda: A @ A)))
Ad = torch.from_numpy(A).cuda(); bd = torch.from_numpy(b).cuda()
def sync(f):
    def g():
        f(); torch.cuda.synchronize()
    return g
print("gpu: solve %.2f ms  lu_factor+lu_solve %.2f ms  cholesky+solve %.2f ms  inv %.2f ms  cholesky_inverse %.2f ms gemm %.2f ms" % (
    tm(sync(lambda: torch.linalg.solve(Ad, bd))),
    tm(sync(lambda: torch.linalg.lu_solve(*torch.linalg.lu_factor(Ad), bd))),
    tm(sync(lambda: torch.cholesky_solve(bd, torch.linalg.cholesky(Ad)))),
    tm(sync(lambda: torch.linalg.inv(Ad))),
    tm(sync(lambda: torch.cholesky_inverse(torch.linalg.cholesky(Ad)))),
    tm(sync(lambda: Ad @ Ad))))
x = torch.linalg.solve(Ad, bd).cpu().numpy(); xr = np.linalg.solve(A, b)
print("rel diff gpu/host solve", np.linalg.norm(x - xr) / np.linalg.norm(xr))
