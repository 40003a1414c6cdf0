"""Experiment driver for tools/gbar_bench.hip (build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/gbar_bench.hip
-o build_exp/libgbar.so): microseconds per grid-wide barrier inside one persistent launch against microseconds per
dependent launch, for a few grid sizes and amounts of data handed over at each step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hydra_mi  # noqa: F401  (one HIP runtime in the process)
from hydra_mi import _lib
_lib.lib()
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_exp", "libgbar.so"))
for f in (L.gbar_run, L.launch_run):
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_int] * 4
for wgs in (8, 32, 64, 128, 256, 512):
    for kb in (0, 4, 64):
        if wgs * kb > 256 * 1024:
            continue
        b = L.gbar_run(wgs, 256, 2000, kb)
        l = L.launch_run(wgs, 256, 500, kb)
        print("wgs %4d  touch %3d KB/wg: barrier %.2f us, dependent launch %.2f us" % (wgs, kb, b, l), flush=True)
