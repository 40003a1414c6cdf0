"""Brox optical flow on MI355X -- host mirror of the interface the reference uses.

The reference computes its flow with ``cv::cuda::BroxOpticalFlow::create(alpha,
gamma, scale_factor, inner, outer, solver)->calc(frame0f, frame1f, flow)``
(reference src/optical_flow_ext.cpp:310,317) after converting the 8-bit frames
with 1/255 (:314-315), and stores flowx / flowy as separate f32 planes
(:322-328).  ``BroxOpticalFlow`` below keeps that shape: ``create`` with the same
parameters and defaults (:453-488), ``calc`` on two gray frames returning
(flowx, flowy).  All arithmetic runs in libhydra_mi.so (csrc/brox.hip).
"""
import ctypes
import os

import numpy as np

from . import _lib

#: reference defaults, src/optical_flow_ext.cpp:453-488
DEFAULTS = dict(alpha=0.197, gamma=50.0, scale_factor=0.8, inner_iterations=10,
                outer_iterations=77, solver_iterations=10)


class BroxOpticalFlow:
    def __init__(self, width, height, alpha=0.197, gamma=50.0, scale_factor=0.8, inner_iterations=10,
                 outer_iterations=77, solver_iterations=10, max_batch=1, device=0):
        self._h = None
        L = _lib.lib()
        h = _lib.c_vp()
        _lib.check(L.hm_brox_create(int(device), int(width), int(height), int(max_batch), alpha, gamma,
                                    scale_factor, int(inner_iterations), int(outer_iterations),
                                    int(solver_iterations), ctypes.byref(h)), "hm_brox_create")
        self._h = h
        _lib.register(self, 3)
        for kv in os.environ.get("HYDRA_MI_BROX_TUNE", "").split(","):     # experiments: "key=value,..." for hm_brox_tune
            if "=" in kv:
                k, v = kv.split("=")
                _lib.check(L.hm_brox_tune(h, k.strip().encode(), int(v)), "hm_brox_tune")
        self.width, self.height, self.max_batch, self.device = int(width), int(height), int(max_batch), int(device)
        self.params = dict(alpha=alpha, gamma=gamma, scale_factor=scale_factor, inner_iterations=inner_iterations,
                           outer_iterations=outer_iterations, solver_iterations=solver_iterations)

    @classmethod
    def create(cls, width, height, alpha=0.197, gamma=50.0, scale_factor=0.8, inner_iterations=10,
               outer_iterations=77, solver_iterations=10, **kw):
        """Same argument order as cuda::BroxOpticalFlow::create, preceded by the frame size."""
        return cls(width, height, alpha, gamma, scale_factor, inner_iterations, outer_iterations,
                   solver_iterations, **kw)

    def close(self):
        if self._h is not None:
            _lib.lib().hm_brox_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host arrays -----------------------------------------------------------------
    def _frames(self, f0, f1, n):
        f0 = np.ascontiguousarray(f0)
        f1 = np.ascontiguousarray(f1)
        if f0.dtype != np.uint8 or f1.dtype != np.uint8:
            raise TypeError("frames must be 8-bit gray (the reference converts u8 * 1/255 itself)")
        want = (n, self.height, self.width) if n is not None else (self.height, self.width)
        if f0.shape != want or f1.shape != want:
            raise ValueError("frames must have shape %s, got %s and %s" % (want, f0.shape, f1.shape))
        return f0, f1

    def calc(self, frame0, frame1):
        """Flow of one pair of u8 gray frames -> (flowx, flowy) f32 HxW."""
        f0, f1 = self._frames(frame0, frame1, None)
        fx = np.empty((self.height, self.width), np.float32)
        fy = np.empty_like(fx)
        _lib.check(_lib.lib().hm_brox_calc(self._h, _lib.ptr(f0), _lib.ptr(f1), _lib.ptr(fx), _lib.ptr(fy)),
                   "hm_brox_calc")
        return fx, fy

    def calc_batch(self, frames0, frames1):
        """n independent pairs, arrays of shape (n, H, W) -> (flowx, flowy) of shape (n, H, W)."""
        n = int(np.shape(frames0)[0])
        f0, f1 = self._frames(frames0, frames1, n)
        fx = np.empty((n, self.height, self.width), np.float32)
        fy = np.empty_like(fx)
        _lib.check(_lib.lib().hm_brox_calc_batch(self._h, n, _lib.ptr(f0), _lib.ptr(f1), _lib.ptr(fx),
                                                 _lib.ptr(fy)), "hm_brox_calc_batch")
        return fx, fy

    # -- device pointers (torch tensors or raw addresses) -------------------------------
    def calc_dev(self, n, d_frame0, d_frame1, d_flowx, d_flowy):
        """Asynchronous on the handle's stream; arguments are device addresses (int)."""
        _lib.check(_lib.lib().hm_brox_calc_dev(self._h, int(n), int(d_frame0), int(d_frame1), int(d_flowx),
                                               int(d_flowy)), "hm_brox_calc_dev")

    def sync(self):
        _lib.check(_lib.lib().hm_brox_sync(self._h), "hm_brox_sync")

    @property
    def stream(self):
        return _lib.lib().hm_brox_stream(self._h)

    def levels(self):
        ws = np.zeros(128, np.int32)
        hs = np.zeros(128, np.int32)
        n = _lib.lib().hm_brox_levels(self._h, ws.ctypes.data_as(_lib.c_i32p), hs.ctypes.data_as(_lib.c_i32p), 128)
        return [(int(ws[i]), int(hs[i])) for i in range(n)]

    def set_omega(self, omega):
        _lib.check(_lib.lib().hm_brox_set_omega(self._h, omega), "hm_brox_set_omega")

    def tune(self, key, value):
        _lib.check(_lib.lib().hm_brox_tune(self._h, key.encode(), int(value)), "hm_brox_tune")

    def profile(self, enable=True):
        _lib.check(_lib.lib().hm_brox_profile(self._h, 1 if enable else 0), "hm_brox_profile")

    def profile_read(self):
        """(sor_ms, sor_launches, sor_pixel_iterations, sor_pixels) since the last read."""
        ms = ctypes.c_double()
        n = ctypes.c_longlong()
        pxit = ctypes.c_double()
        px = ctypes.c_double()
        _lib.check(_lib.lib().hm_brox_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(pxit),
                                                   ctypes.byref(px)), "hm_brox_profile_read")
        return ms.value, n.value, pxit.value, px.value

    def profile_levels(self):
        """Per pyramid level (0 = the full frame) since profile(True): list of dicts ms, launches,
        pixel_iterations, pixels, w, h."""
        cap = 128
        ms, pxit, px = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        ln = np.zeros(cap, np.int64)
        L = _lib.lib().hm_brox_profile_levels(self._h, cap, ms.ctypes.data_as(_lib.c_f64p),
                                              ln.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                              pxit.ctypes.data_as(_lib.c_f64p), px.ctypes.data_as(_lib.c_f64p))
        if L < 0:
            _lib.check(L, "hm_brox_profile_levels")
        geo = self.levels()
        return [{"level": k, "w": geo[k][0], "h": geo[k][1], "ms": float(ms[k]), "launches": int(ln[k]),
                 "pixel_iterations": float(pxit[k]), "pixels": float(px[k])} for k in range(min(L, cap))]


# ---- single operators (the kernels calc() is made of), host arrays -----------------------
def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def op_blur(img, scale_factor=0.8):
    img = _f32(img)
    out = np.empty_like(img)
    _lib.check(_lib.lib().hm_op_blur(_lib.ptr(img), img.shape[1], img.shape[0], scale_factor, _lib.ptr(out)))
    return out


def op_resample(img, wd, hd, mul=1.0):
    img = _f32(img)
    out = np.empty((hd, wd), np.float32)
    _lib.check(_lib.lib().hm_op_resample(_lib.ptr(img), img.shape[1], img.shape[0], _lib.ptr(out), wd, hd, mul))
    return out


def op_deriv(img):
    img = _f32(img)
    dx, dy = np.empty_like(img), np.empty_like(img)
    _lib.check(_lib.lib().hm_op_deriv(_lib.ptr(img), img.shape[1], img.shape[0], _lib.ptr(dx), _lib.ptr(dy)))
    return dx, dy


def op_pyr_down(img, wd, hd, scale_factor=0.8):
    """one pyramid level as calc builds it: Gaussian + resampling in one launch"""
    img = _f32(img)
    out = np.empty((hd, wd), np.float32)
    _lib.check(_lib.lib().hm_op_pyr_down(_lib.ptr(img), img.shape[1], img.shape[0], scale_factor, _lib.ptr(out), wd, hd))
    return out


def op_deriv_all(I0, I1):
    """-> Ix0, Iy0, I1x, I1y, I1xx, I1xy, I1yy (one launch)"""
    I0, I1 = _f32(I0), _f32(I1)
    outs = [np.empty_like(I0) for _ in range(7)]
    _lib.check(_lib.lib().hm_op_deriv_all(_lib.ptr(I0), _lib.ptr(I1), I0.shape[1], I0.shape[0], _lib.ptr_array(outs)))
    return outs


def op_add_prolong(u, v, du, dv, wd, hd):
    """(u + du, v + dv) prolonged to wd x hd with the flow rescaled (same size: the sums themselves)"""
    u, v, du, dv = _f32(u), _f32(v), _f32(du), _f32(dv)
    u2, v2 = np.empty((hd, wd), np.float32), np.empty((hd, wd), np.float32)
    _lib.check(_lib.lib().hm_op_add_prolong(_lib.ptr(u), _lib.ptr(v), _lib.ptr(du), _lib.ptr(dv), u.shape[1], u.shape[0],
                                            _lib.ptr(u2), _lib.ptr(v2), wd, hd))
    return u2, v2


def op_warp(*fields, window=False):
    """fields = I0,Ix0,Iy0,I1,I1x,I1y,I1xx,I1xy,I1yy,u,v -> Iz,Ix,Iy,Ixz,Iyz,Ixx,Ixy,Iyy
    (window: the LDS-window variant of the kernel instead of direct reads)"""
    ins = [_f32(a) for a in fields]
    assert len(ins) == 11
    H, W = ins[0].shape
    outs = [np.empty((H, W), np.float32) for _ in range(8)]
    _lib.check(_lib.lib().hm_op_warp(_lib.ptr_array(ins), W, H, _lib.ptr_array(outs), 1 if window else 0))
    return outs


def op_prepare(u, v, du, dv, warped, alpha=0.197, gamma=50.0):
    ins = [_f32(a) for a in (u, v, du, dv) + tuple(warped)]
    assert len(ins) == 12
    H, W = ins[0].shape
    outs = [np.empty((H, W), np.float32) for _ in range(7)]
    _lib.check(_lib.lib().hm_op_prepare(_lib.ptr_array(ins), W, H, alpha, gamma, _lib.ptr_array(outs)))
    return outs


def op_sor(du, dv, coef, iterations, fuse=0, omega=1.99):
    du = np.array(du, np.float32, order="C", copy=True)
    dv = np.array(dv, np.float32, order="C", copy=True)
    coef = [_f32(c) for c in coef]
    assert len(coef) == 7
    H, W = du.shape
    _lib.check(_lib.lib().hm_op_sor(_lib.ptr(du), _lib.ptr(dv), _lib.ptr_array(coef), W, H, int(iterations),
                                    int(fuse), omega))
    return du, dv
