"""ctypes binding of libhydra_mi.so (declared in include/hydra_mi.h).

There is no CPU fallback: if the library has not been built (``python
__graft_entry__.py build``) importing a compute entry point raises, and on a
machine without a GPU every compute call returns HM_ERR_HIP, which surfaces
here as RuntimeError.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HYDRA_MI_SO: another build of the same library, for A/B experiments -- tools/ only)
SO_PATH = os.environ.get("HYDRA_MI_SO") or os.path.join(_HERE, "libhydra_mi.so")

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_vp = ctypes.c_void_p

#: every symbol include/hydra_mi.h declares -> (restype, argtypes)
SIGNATURES = {
    "hm_last_error": (ctypes.c_char_p, []),
    "hm_version": (ctypes.c_char_p, []),
    "hm_device_count": (ctypes.c_int, []),
    "hm_dev_alloc": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(c_vp)]),
    "hm_dev_free": (ctypes.c_int, [ctypes.c_int, c_vp]),
    "hm_dev_upload": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, ctypes.c_uint64]),
    "hm_dev_download": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, ctypes.c_uint64]),
    "hm_copy_stream_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    "hm_copy_stream_destroy": (ctypes.c_int, [ctypes.c_int, c_vp]),
    "hm_copy_stream_sync": (ctypes.c_int, [ctypes.c_int, c_vp]),
    "hm_host_alloc": (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(c_vp)]),
    "hm_host_free": (ctypes.c_int, [c_vp]),
    "hm_dev_upload_async": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, ctypes.c_uint64, c_vp]),
    "hm_brox_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(c_vp)]),
    "hm_brox_destroy": (ctypes.c_int, [c_vp]),
    "hm_brox_calc": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "hm_brox_calc_batch": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp]),
    "hm_brox_calc_dev": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp]),
    "hm_brox_sync": (ctypes.c_int, [c_vp]),
    "hm_brox_stream": (c_vp, [c_vp]),
    "hm_brox_levels": (ctypes.c_int, [c_vp, c_i32p, c_i32p, ctypes.c_int]),
    "hm_brox_set_omega": (ctypes.c_int, [c_vp, ctypes.c_float]),
    "hm_brox_tune": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    "hm_brox_profile": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "hm_brox_profile_read": (ctypes.c_int, [c_vp, c_f64p, ctypes.POINTER(ctypes.c_longlong), c_f64p, c_f64p]),
    "hm_brox_profile_levels": (ctypes.c_int, [c_vp, ctypes.c_int, c_f64p, ctypes.POINTER(ctypes.c_longlong), c_f64p, c_f64p]),
    "hm_op_blur": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_vp]),
    "hm_op_resample": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_float]),
    "hm_op_deriv": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    "hm_op_pyr_down": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_vp, ctypes.c_int, ctypes.c_int]),
    "hm_op_deriv_all": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "hm_op_add_prolong": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_int,
                                         ctypes.c_int]),
    "hm_op_warp": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp), ctypes.c_int]),
    "hm_op_prepare": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                     ctypes.c_float, ctypes.POINTER(c_vp)]),
    "hm_op_sor": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_float]),
    "hm_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                     c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(c_vp)]),
    "hm_ctx_destroy": (ctypes.c_int, [c_vp]),
    "hm_set_texture": (ctypes.c_int, [c_vp, c_vp]),
    "hm_set_observation": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "hm_set_observation_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "hm_render": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "hm_initjacobian": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int]),
    "hm_jz": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_f64p, c_f64p]),
    "hm_j": (ctypes.c_int, [c_vp, c_vp, ctypes.c_double, ctypes.c_int, ctypes.c_int, c_f64p]),
    "hm_jz_multi": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int, c_vp, c_vp]),
    "hm_j_multi": (ctypes.c_int, [c_vp, c_vp, ctypes.c_double, ctypes.c_int, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, c_vp]),
    "hm_error": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_f64p, c_vp, c_vp]),
    "hm_measure": (ctypes.c_int, [c_vp, c_vp, ctypes.c_double, ctypes.c_int, c_vp, c_vp, c_vp]),
    "hm_update_begin": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "hm_update_step": (ctypes.c_int, [c_vp, c_vp, ctypes.c_double, ctypes.c_int, c_vp, c_vp, c_f64p]),
    "hm_update_cov": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp]),
    "hm_update_prefactor": (ctypes.c_int, [c_vp]),
    "hm_update_run": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                     ctypes.POINTER(ctypes.c_int), c_vp, c_vp, c_vp, c_vp]),
    "hm_update_last_error": (ctypes.c_int, [c_vp, c_vp, c_f64p]),
    "hm_cov_fetch": (ctypes.c_int, [c_vp, c_vp]),
    "hm_project_mask": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "hm_cov_predict": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, c_vp]),
    "hm_ms_newton": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_int, ctypes.c_double, c_vp, ctypes.POINTER(ctypes.c_int)]),
    "hm_ms_worker_create": (ctypes.c_int, [ctypes.POINTER(c_vp)]),
    "hm_ms_worker_destroy": (ctypes.c_int, [c_vp]),
    "hm_ms_newton_start": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_int, ctypes.c_double, c_vp]),
    "hm_ms_newton_finish": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(ctypes.c_int)]),
    "hm_ms_worker_attach": (ctypes.c_int, [c_vp, c_vp]),
    "hm_newton_dev_start": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_int, ctypes.c_double, c_vp]),
    "hm_newton_dev_finish": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(ctypes.c_int)]),
    "hm_prune_mask": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "hm_chain_project": (ctypes.c_int, [c_vp]),
    "hm_prepare_mask": (ctypes.c_int, [c_vp, c_vp]),
    "hm_update_arm_mask": (ctypes.c_int, [c_vp, c_vp]),
    "hm_update_tail": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "hm_chain_states": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "hm_update_arm_newton": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_int, ctypes.c_double]),
    "hm_update_arm_cov": (ctypes.c_int, [c_vp, ctypes.c_double]),
    "hm_predict_take": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_double]),
    "hm_ms_predict": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_int, ctypes.c_double, ctypes.c_double, c_vp, ctypes.POINTER(ctypes.c_int),
                                     ctypes.c_int]),
    "hm_ctx_tune": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    "hm_ctx_sync": (ctypes.c_int, [c_vp]),
    "hm_ctx_stream": (c_vp, [c_vp]),
}

_lib = None


def _one_hip_runtime():
    """Keep ONE HIP runtime in the process.  PyTorch wheels carry their own libamdhip64; if
    libhydra_mi.so pulled in the system copy first and torch were imported afterwards, torch would
    find the runtime already initialised by a different build and report "No HIP GPUs are available".
    So when torch is installed its copy is loaded (globally) before ours, whatever the import order;
    HYDRA_MI_HIP_RUNTIME=<path> overrides, HYDRA_MI_HIP_RUNTIME=system skips this."""
    choice = os.environ.get("HYDRA_MI_HIP_RUNTIME", "")
    if choice == "system":
        return
    path = choice
    if not path:
        import importlib.util
        import sys
        if "torch" in sys.modules:
            return                         # already in the process
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def lib():
    """The loaded library; raises if it was never built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "%s is missing: the HIP extension has not been built "
                "(run `python __graft_entry__.py build`); there is no CPU fallback" % SO_PATH)
        _one_hip_runtime()
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None:                 # calling it later raises AttributeError -- loudly
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def missing_symbols():
    """Declared in include/hydra_mi.h but absent from the built library (must be empty)."""
    L = lib()
    return [name for name in SIGNATURES if getattr(L, name, None) is None]


HM_ERR_NUMERIC = -4


def check(rc, what=""):
    if rc != 0:
        msg = lib().hm_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what or "libhydra_mi call", rc,
                                                         msg.decode() if msg else "?"))


# ---- lifetime -----------------------------------------------------------------------------------
# Everything that owns native state (pipelines with their helper threads and copy streams, filters with their worker
# thread, renderer / flow handles with their streams and page-locked blocks, device buffers) registers here and is
# closed -- threads joined, streams drained and destroyed, page-locked memory released -- before the interpreter and
# the HIP runtime finalise: a process that exits with helper threads alive and streams registered crashed inside
# __cxa_finalize when a profiler's tool library was finalising beside it (rocprofv3 -- python bench.py, round 3).
_live = {}          # id -> (order, serial, weakref): lower order closes first; within an order, newest first
_serial = [0]


def register(obj, order):
    """order: 0 pipelines, 1 filters, 2 renderer handles, 3 flow handles, 4 device / page-locked buffers."""
    import weakref
    _serial[0] += 1
    key = id(obj)
    _live[key] = (order, _serial[0], weakref.ref(obj, lambda _r, k=key: _live.pop(k, None)))


def close_all():
    """Close every live native object (idempotent; runs at interpreter exit)."""
    for order, serial, ref in sorted(_live.values(), key=lambda e: (e[0], -e[1])):
        obj = ref()
        if obj is None:
            continue
        try:
            obj.close()
        except Exception:          # noqa: BLE001 -- teardown goes on with the others
            pass
    _live.clear()


import atexit  # noqa: E402

atexit.register(close_all)


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_vp)


def ptr_array(arrs):
    return (c_vp * len(arrs))(*[a.ctypes.data for a in arrs])


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
