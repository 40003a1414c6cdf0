"""Development aid: the bench with K idle extra HIP streams in the process (HM_EXTRA_STREAMS=K), created before anything
else of the product -- how many hardware queues a process may hold before the filter's launches start to wait."""
import ctypes, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hydra_mi
from hydra_mi import _lib
held = []
for _ in range(int(os.environ.get("HM_EXTRA_STREAMS", "0"))):
    s = ctypes.c_void_p()
    _lib.check(_lib.lib().hm_copy_stream_create(0, ctypes.byref(s)), "hm_copy_stream_create")
    held.append(s)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
