"""ctypes front-end of oracle/brox_ref.c  (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.  Parity against the reference's
Brox (OpenCV cudaoptflow, reference src/optical_flow_ext.cpp:310,317) is
UNPINNED -- see the header of brox_ref.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbrox_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int)

#: reference defaults, src/optical_flow_ext.cpp:453-488
DEFAULTS = dict(alpha=0.197, gamma=50.0, scale=0.8, inner=10, outer=77, solver=10)


def build(force=False):
    src = os.path.join(_HERE, "brox_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbrox_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.brox_ref_calc_f32.restype = ctypes.c_int
        _lib.brox_ref_calc_u8.restype = ctypes.c_int
        _lib.brox_ref_levels.restype = ctypes.c_int
        _lib.brox_ref_gauss.restype = ctypes.c_int
        # one thread unless a caller asks for more: the OpenMP loops are there for the
        # timed CPU baseline, and a default-sized team on a many-core host with a small
        # CPU quota (the GPU box) is orders of magnitude slower than one thread
        _lib.brox_ref_set_threads(ctypes.c_int(1))
    return _lib


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def set_threads(n):
    lib().brox_ref_set_threads(ctypes.c_int(int(n)))


def levels(W, H, scale=0.8, outer=77):
    ws = np.zeros(128, np.int32)
    hs = np.zeros(128, np.int32)
    n = lib().brox_ref_levels(int(W), int(H), ctypes.c_float(scale), int(outer), _p(ws, _i32p), _p(hs, _i32p))
    return [(int(ws[i]), int(hs[i])) for i in range(n)]


def gauss(scale=0.8):
    g = np.zeros(33, np.float32)
    R = lib().brox_ref_gauss(ctypes.c_float(scale), _p(g))
    return g[: 2 * R + 1].copy()


def calc(f0, f1, alpha=0.197, gamma=50.0, scale=0.8, inner=10, outer=77, solver=10):
    """Brox flow of two gray frames (u8, or f32 already in [0,1]) -> (u, v) f32 HxW."""
    f0 = np.ascontiguousarray(f0)
    f1 = np.ascontiguousarray(f1)
    assert f0.shape == f1.shape and f0.ndim == 2 and f0.dtype == f1.dtype
    H, W = f0.shape
    u = np.empty((H, W), np.float32)
    v = np.empty((H, W), np.float32)
    args = (int(W), int(H), ctypes.c_float(alpha), ctypes.c_float(gamma), ctypes.c_float(scale),
            int(inner), int(outer), int(solver), _p(u), _p(v))
    if f0.dtype == np.uint8:
        rc = lib().brox_ref_calc_u8(_p(f0, _u8p), _p(f1, _u8p), *args)
    else:
        f0 = f0.astype(np.float32, copy=False)
        f1 = f1.astype(np.float32, copy=False)
        rc = lib().brox_ref_calc_f32(_p(f0), _p(f1), *args)
    if rc != 0:
        raise ValueError("brox_ref_calc: bad arguments (rc=%d)" % rc)
    return u, v


# ---- single operators, for kernel-level parity tests ------------------------
def blur(img, scale=0.8):
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty_like(img)
    lib().brox_ref_blur(_p(img), _p(out), img.shape[1], img.shape[0], ctypes.c_float(scale))
    return out


def resample(img, wd, hd, mul=1.0):
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty((hd, wd), np.float32)
    lib().brox_ref_resample(_p(img), img.shape[1], img.shape[0], _p(out), int(wd), int(hd), ctypes.c_float(mul))
    return out


def deriv(img):
    img = np.ascontiguousarray(img, np.float32)
    dx = np.empty_like(img)
    dy = np.empty_like(img)
    lib().brox_ref_deriv(_p(img), _p(dx), _p(dy), img.shape[1], img.shape[0])
    return dx, dy


def warp(I0, Ix0, Iy0, I1, I1x, I1y, I1xx, I1xy, I1yy, u, v):
    arrs = [np.ascontiguousarray(a, np.float32) for a in (I0, Ix0, Iy0, I1, I1x, I1y, I1xx, I1xy, I1yy, u, v)]
    H, W = arrs[0].shape
    outs = [np.empty((H, W), np.float32) for _ in range(8)]
    lib().brox_ref_warp(*[_p(a) for a in arrs], int(W), int(H), *[_p(o) for o in outs])
    return outs  # Iz, Ix, Iy, Ixz, Iyz, Ixx, Ixy, Iyy


def prepare(u, v, du, dv, warped, alpha=0.197, gamma=50.0):
    arrs = [np.ascontiguousarray(a, np.float32) for a in (u, v, du, dv) + tuple(warped)]
    H, W = arrs[0].shape
    outs = [np.empty((H, W), np.float32) for _ in range(7)]
    lib().brox_ref_prepare(*[_p(a) for a in arrs], *[_p(o) for o in outs], int(W), int(H),
                           ctypes.c_float(alpha), ctypes.c_float(gamma))
    return outs  # nu, nv, a12, idu, idv, sx, sy


def sor(du, dv, coef, iters):
    du = np.array(du, np.float32, order="C", copy=True)
    dv = np.array(dv, np.float32, order="C", copy=True)
    coef = [np.ascontiguousarray(a, np.float32) for a in coef]
    H, W = du.shape
    lib().brox_ref_sor(_p(du), _p(dv), *[_p(c) for c in coef], int(W), int(H), int(iters))
    return du, dv
