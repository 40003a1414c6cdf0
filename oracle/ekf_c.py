"""ctypes front-end of oracle/ekf_ref_c.c  (TEST INFRASTRUCTURE ONLY).

The C/OpenMP twin of ``ekf_ref.Measurement``: same constructor, same methods, same
numbers (renders bit-identical, sums to rounding -- tests/test_oracle_ekf_c.py), fast
enough that a whole ``KFState.update`` of the reference's CPU path (kalman.py:491-518,
583-606: 2*4N jz + nzj j evaluations, every one a full-frame render) can be run and
timed at 512^2 and 1024^2.  Used for the golden tracks of the larger BASELINE configs
(tools/make_golden.py), for the oracle comparisons at full size and for bench.py's
``cpu_baseline``.  Only tests/, tools/make_golden.py, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libekf_oracle.so")
_lib = None

_vp = ctypes.c_void_p


def build(force=False):
    src = os.path.join(_HERE, "ekf_ref_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libekf_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.ekf_c_create.restype = _vp
        L.ekf_c_create.argtypes = [ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.ekf_c_destroy.argtypes = [_vp]
        L.ekf_c_destroy.restype = None
        L.ekf_c_set_threads.argtypes = [_vp, ctypes.c_int]
        L.ekf_c_set_threads.restype = None
        L.ekf_c_max_threads.restype = ctypes.c_int
        L.ekf_c_render.argtypes = [_vp] * 6
        L.ekf_c_initjacobian.argtypes = [_vp] * 6
        L.ekf_c_jz.argtypes = [_vp] * 4
        L.ekf_c_j.argtypes = [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_int, _vp]
        L.ekf_c_error.argtypes = [_vp] * 9
        L.ekf_c_jacobian.argtypes = [_vp, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp]
        L.ekf_c_hessian_pairs.argtypes = [_vp, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def max_threads():
    return int(lib().ekf_c_max_threads())


class Measurement:
    """Same interface as ekf_ref.Measurement (the CPU twin of CUDAGL, cuda.py:929-1010)."""

    def __init__(self, N, tri, uv, tex, eps_Z, eps_J, eps_M, threads=1):
        self.N = int(N)
        self.tri = np.ascontiguousarray(tri, np.int64)
        self.uv = np.ascontiguousarray(uv, np.float32)
        tex = np.asarray(tex)
        self.tex = np.ascontiguousarray(tex if tex.ndim == 2 else tex[:, :, 0], np.uint8)
        self.H, self.W = self.tex.shape
        self.eps_Z, self.eps_J, self.eps_M = float(eps_Z), float(eps_J), float(eps_M)
        self._h = lib().ekf_c_create(self.N, int(self.tri.shape[0]), _p(self.tri), _p(self.uv), _p(self.tex), self.W,
                                     self.H, self.eps_Z, self.eps_J, self.eps_M)
        if not self._h:
            raise MemoryError("ekf_c_create")
        self.set_threads(threads)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ekf_c_destroy(self._h)
            self._h = None

    def set_threads(self, n):
        """OpenMP threads of jacobian_all / hessian_all (one evaluation per thread at a time)."""
        self.threads = max(1, int(n))
        lib().ekf_c_set_threads(self._h, self.threads)

    def _X(self, X):
        X = np.ascontiguousarray(np.asarray(X, np.float64).reshape(-1))
        assert X.size == 4 * self.N
        return X

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d)%s" % (what, rc, ": initjacobian has not been called" if rc == -2 else ""))

    def render(self, X):
        im = np.empty((self.H, self.W), np.uint8)
        m = np.empty_like(im)
        fx = np.empty((self.H, self.W), np.float32)
        fy = np.empty_like(fx)
        self._check(lib().ekf_c_render(self._h, _p(self._X(X)), _p(im), _p(fx), _p(fy), _p(m)), "render")
        return im, fx, fy, m

    def _obs(self, y_im, y_flow, y_m):
        y_im = np.asarray(y_im)
        y_m = np.asarray(y_m)
        if y_im.dtype != np.uint8 or y_m.dtype != np.uint8:
            raise TypeError("the C twin takes the uint8 frame / mask the reference works with")
        return (np.ascontiguousarray(y_im), np.ascontiguousarray(y_flow[:, :, 0], np.float32),
                np.ascontiguousarray(y_flow[:, :, 1], np.float32), np.ascontiguousarray(y_m))

    def initjacobian(self, X, y_im, y_flow, y_m):
        a, fx, fy, m = self._obs(y_im, y_flow, y_m)
        self.X0 = self._X(X).copy()
        self._check(lib().ekf_c_initjacobian(self._h, _p(self.X0), _p(a), _p(fx), _p(fy), _p(m)), "initjacobian")

    def jz(self, Xp):
        tot = ctypes.c_double()
        c = np.empty(4)
        self._check(lib().ekf_c_jz(self._h, _p(self._X(Xp)), ctypes.addressof(tot), _p(c)), "jz")
        return tot.value, c

    def j(self, deltaX, i, jdx):
        out = ctypes.c_double()
        self._check(lib().ekf_c_j(self._h, float(deltaX), int(i), int(jdx), ctypes.addressof(out)), "j")
        return out.value

    def error(self, X, y_im, y_flow, y_m):
        a, fx, fy, m = self._obs(y_im, y_flow, y_m)
        err = np.empty(4)
        pfx = np.empty((self.H, self.W), np.float32)
        pfy = np.empty_like(pfx)
        self._check(lib().ekf_c_error(self._h, _p(self._X(X)), _p(a), _p(fx), _p(fy), _p(m), _p(err), _p(pfx), _p(pfy)),
                    "error")
        return int(err[0]), float(err[1]), float(err[2]), int(err[3]), pfx, pfy

    # -- whole passes of KFState.update (ekf_ref.jacobian / hessian_sparse call these when present) ----
    def jacobian_all(self, X, y_im, y_flow, y_m, deltaX=2.0, idx=None):
        """_jacobian (kalman.py:491-518) -> (Hz [n,1], Hzc [n,4]) for the state indices idx (default all)."""
        self.initjacobian(X, y_im, y_flow, y_m)
        idx = np.arange(4 * self.N, dtype=np.int32) if idx is None else np.ascontiguousarray(idx, np.int32)
        Hz = np.zeros(idx.size)
        Hzc = np.zeros((idx.size, 4))
        self._check(lib().ekf_c_jacobian(self._h, float(deltaX), int(idx.size), _p(idx), _p(Hz), _p(Hzc)), "jacobian")
        return Hz.reshape(-1, 1), Hzc

    def hessian_pairs(self, pi, pj, deltaX=2.0):
        pi = np.ascontiguousarray(pi, np.int32)
        pj = np.ascontiguousarray(pj, np.int32)
        out = np.zeros(pi.size)
        self._check(lib().ekf_c_hessian_pairs(self._h, float(deltaX), int(pi.size), _p(pi), _p(pj), _p(out)), "hessian")
        return out

    def hessian_all(self, X, J, deltaX=2.0):
        """_hessian_sparse (kalman.py:583-606): initjacobian must have been called at X."""
        n = 4 * self.N
        pi, pj = np.nonzero(np.triu(np.asarray(J) == 1))
        vals = self.hessian_pairs(pi, pj, deltaX)
        HTH = np.zeros((n, n))
        HTH[pi, pj] = vals
        HTH[pj, pi] = vals
        return HTH
