"""Measurement model of the mesh tracker on MI355X.

Host mirror of the reference's ``renderer.Renderer`` (reference
renderer.py:197-737) for the part that is on the hot path: the four off-screen
renders of the textured mesh and the reductions against the observed frame
that the reference splits between OpenGL (renderer.py:310-325) and the CUDA
kernels of ``CUDAGL`` / ``CUDAGL_multi`` (cuda.py, cuda_multi.py).  Here both
live in libhydra_mi.so (csrc/ekf.hip): a software rasteriser and fused
perturb-and-reduce kernels; nothing is rendered on a CPU and there is no
fallback.  The on-screen views, screenshots and key bindings of the reference
canvas are visualisation and are not part of this path.

Same method names and argument meaning as the reference:
``update_vertex_buffer``, ``render``, ``initjacobian``, ``jz``, ``jz_multi``,
``j``, ``j_multi``, ``error``, ``update_frame``, ``get_flow``; plus ``measure``,
the fused form of ``KFState.update`` (kalman.py:437-449).

``FlowStream`` reads the ``<path>_%03d_x.mat`` / ``_y.mat`` files the flow tool
writes (reference renderer.py:807-874).
"""
import ctypes
import os

import numpy as np

from . import _lib
from . import matio


class ChainedPredictionFailed(RuntimeError):
    """hm_update_run, chained to a state prediction on the device (hm_chain_project), found that the prediction's inner
    solve had given up: nothing was updated; predict on the host and update again."""


class Renderer:
    def __init__(self, distmesh, vel, flow, nx, im1, cuda, eps_Z, eps_J, eps_M, labels=None, labels_hess=None,
                 Q=None, showtracking=False, force=None, multi=True, device=0):
        if not cuda:
            raise NotImplementedError(
                "cuda=False: this build has no CPU measurement path (the reference's NumPy twin, "
                "cuda.py:929-1010, is restated under oracle/ as test infrastructure only)")
        self._h = None
        self.Q = Q
        self.cuda = cuda
        self.force = force
        self.labels, self.labels_hess = labels, labels_hess
        self.tri = np.ascontiguousarray(distmesh.t, np.int32)
        self.n = int(distmesh.p.shape[0])
        im1 = np.asarray(im1)
        tex = im1 if im1.ndim == 2 else im1[:, :, 0]      # the r8 target keeps channel 0 (cuda.py:931)
        self.ny, self.nx = int(tex.shape[0]), int(tex.shape[1])
        if int(nx) != self.ny:
            raise ValueError("nx=%d does not match the frame (%d rows)" % (nx, self.ny))
        uv = np.ascontiguousarray(distmesh.p, np.float32)  # texture coordinates = initial vertices (renderer.py:579)
        L = _lib.lib()
        h = _lib.c_vp()
        _lib.check(L.hm_ctx_create(int(device), self.nx, self.ny, self.n, int(self.tri.shape[0]), _lib.ptr(self.tri),
                                   _lib.ptr(uv), eps_Z, eps_J, eps_M, ctypes.byref(h)), "hm_ctx_create")
        self._h = h
        self._worker = None                     # a state-prediction worker attached to this handle (attach_worker)
        _lib.register(self, 2)
        _lib.check(L.hm_set_texture(self._h, _lib.ptr(np.ascontiguousarray(tex, np.uint8))), "hm_set_texture")
        self.vertices = np.array(distmesh.p, np.float64)
        self.velocities = np.array(vel, np.float64).reshape(self.vertices.shape)
        self._obs = None
        self._palette = None
        for kv in os.environ.get("HYDRA_MI_TUNE", "").split(","):          # experiments: "key=value,..." for hm_ctx_tune
            if "=" in kv:
                k, v = kv.split("=")
                _lib.check(L.hm_ctx_tune(h, k.strip().encode(), int(v)), "hm_ctx_tune")
        self.frame_in_place = False             # set by KalmanFilter.compute() while the frame it uploaded is current
        self._cov_serial = 0                    # names the covariance resident on the device (DeviceCovariance)
        self.current_frame = tex
        self.current_flowx, self.current_flowy = flow[:, :, 0], flow[:, :, 1]

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if self._h is not None:
            if self._worker is not None:        # the worker keeps a pointer to this handle: back to its host thread first
                try:
                    _lib.lib().hm_ms_worker_attach(self._worker, None)
                finally:
                    self._worker = None
            _lib.lib().hm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ----------------------------------------------------------------------------
    def update_vertex_buffer(self, vertices, velocities, multi_idx=-1, hess=False):
        """reference renderer.py:503-556: the vertex buffer and the label palette of the mask render
        (multi_idx: the partition whose labels colour the triangles, -1: none; hess: the pair partitions).
        The palette only matters to jz_multi / j_multi."""
        self.vertices = np.asarray(vertices, np.float64).reshape(-1, 2)
        self.velocities = np.asarray(velocities, np.float64).reshape(-1, 2)
        table = self.labels_hess if hess else self.labels
        if multi_idx is None or multi_idx < 0 or table is None:
            self._palette = None                                      # every triangle (255, 255, 255): no label
        else:
            self._palette = np.ascontiguousarray(np.asarray(table)[:, multi_idx], np.int32)

    def _X(self, state=None):
        if state is not None:
            return np.ascontiguousarray(np.asarray(state.X, np.float64).reshape(-1))
        return np.ascontiguousarray(np.concatenate((self.vertices.reshape(-1), self.velocities.reshape(-1))))

    def render(self):
        """The four renders of the current vertex buffer -> (im u8, fx f32, fy f32, m u8)."""
        n = self.nx * self.ny
        im = np.empty((self.ny, self.nx), np.uint8)
        m = np.empty_like(im)
        fx = np.empty((self.ny, self.nx), np.float32)
        fy = np.empty_like(fx)
        _lib.check(_lib.lib().hm_render(self._h, _lib.ptr(self._X()), _lib.ptr(im), _lib.ptr(fx), _lib.ptr(fy),
                                        _lib.ptr(m)), "hm_render")
        self.last_render = (im, fx, fy, m)
        return self.last_render

    def get_flow(self):
        """reference renderer.py:666-672."""
        _, fx, fy, _ = self.render()
        return fx, fy

    # -- observation ------------------------------------------------------------------------
    def update_frame(self, y_im, y_flow, y_m):
        """reference renderer.py:656-664: the frame compute() works on.  Uploaded once."""
        self.set_observation(y_im, y_flow, y_m)

    def set_observation(self, y_im, y_flow, y_m):
        y_im = np.asarray(y_im)
        if y_im.ndim == 3:
            y_im = y_im[:, :, 0]
        a = np.ascontiguousarray(y_im, np.uint8)
        fx = np.ascontiguousarray(y_flow[:, :, 0], np.float32)
        fy = np.ascontiguousarray(y_flow[:, :, 1], np.float32)
        m = np.ascontiguousarray(y_m, np.uint8)
        if a.shape != (self.ny, self.nx) or fx.shape != a.shape or m.shape != a.shape:
            raise ValueError("observation arrays must be %dx%d" % (self.ny, self.nx))
        _lib.check(_lib.lib().hm_set_observation(self._h, _lib.ptr(a), _lib.ptr(fx), _lib.ptr(fy), _lib.ptr(m)),
                   "hm_set_observation")
        self._obs = (y_im, y_flow, y_m)

    def set_observation_dev(self, obs):
        """A DeviceObservation: frame, flow planes (e.g. what hm_brox_calc_dev just wrote) and mask
        already in device memory.  The caller keeps them alive and ordered before this call."""
        _lib.check(_lib.lib().hm_set_observation_dev(self._h, int(obs.d_y_im), int(obs.d_flowx), int(obs.d_flowy),
                                                     int(obs.d_y_m)), "hm_set_observation_dev")
        self._obs = obs

    def _masked_flag(self, y_im, y_flow, y_m):
        """Which of the two device copies of the observed flow an operator uses (0 raw, 1 mask-multiplied),
        uploading the arrays first unless they are known to be the observation in place.

        The reference's operators always work on the arrays they are given (renderer.py:485-501, 674-679).
        Host arrays are therefore uploaded on every call -- a caller may have refilled its buffers in
        place -- except inside KalmanFilter.compute(), which has just uploaded them itself
        (``frame_in_place``): there the arrays are recognised by identity and the copies are skipped."""
        o = self._obs
        if isinstance(y_im, DeviceObservation):
            if o is not y_im:
                self.set_observation_dev(y_im)
            return 1 if y_flow is y_im.masked else 0
        if self.frame_in_place and o is not None and len(o) == 3 and y_im is o[0] and y_m is o[2]:
            if y_flow is o[1]:
                return 0
            if getattr(y_flow, "_hm_masked_from", None) is o[1]:
                return 1
        raw = getattr(y_flow, "_hm_masked_from", None)
        if raw is not None:                       # a MaskedFlow: upload the flow it was made from, use the masked copy
            self.set_observation(y_im, raw, y_m)
            return 1
        self.set_observation(y_im, y_flow, y_m)
        return 0

    # -- the reference's operator interface ---------------------------------------------------
    def initjacobian(self, y_im, y_flow, y_m):
        """reference renderer.py:674-679 / cuda.py:940-950: current render becomes the reference."""
        self._masked = self._masked_flag(y_im, y_flow, y_m)
        _lib.check(_lib.lib().hm_initjacobian(self._h, _lib.ptr(self._X()), self._masked), "hm_initjacobian")

    def jz(self, state=None):
        """reference renderer.py:681-694 / cuda.py:972-980 -> (jz, [im, fx, fy, m] components)."""
        out = ctypes.c_double()
        comp = (ctypes.c_double * 4)()
        _lib.check(_lib.lib().hm_jz(self._h, _lib.ptr(self._X(state)), getattr(self, "_masked", 0),
                                    ctypes.byref(out), comp), "hm_jz")
        return out.value, np.array(comp[:])

    def _labels_now(self, who):
        if getattr(self, "_palette", None) is None:
            raise RuntimeError("%s: no label palette selected (update_vertex_buffer / KFState.refresh with the "
                               "partition index first, reference kalman.py:459, 545)" % who)
        return self._palette

    def jz_multi(self, state):
        """reference renderer.py:696-709 / cuda_multi.py:721-845 -> (hz [N,1], hzc [N,4]): the terms of jz summed
        per vertex label of the palette in place (hm_jz_multi)."""
        lab = self._labels_now("jz_multi")
        hz = np.zeros(self.n)
        hzc = np.zeros((self.n, 4))
        _lib.check(_lib.lib().hm_jz_multi(self._h, _lib.ptr(self._X(state)), getattr(self, "_masked", 0), _lib.ptr(lab),
                                          self.n, _lib.ptr(hz), _lib.ptr(hzc)), "hm_jz_multi")
        return hz.reshape(-1, 1), hzc

    def j(self, state, deltaX, i, j):
        """reference renderer.py:711-721 / cuda.py:982-1010."""
        out = ctypes.c_double()
        _lib.check(_lib.lib().hm_j(self._h, _lib.ptr(self._X(state)), float(deltaX), int(i), int(j),
                                   ctypes.byref(out)), "hm_j")
        return out.value

    def j_multi(self, state, deltaX, ee, labelidx, ee_idx=None):
        """reference renderer.py:723-737 / cuda_multi.py:979-1129 -> (h [1,|Q|], nz [|Q|] bool, hcomp [|Q|,4]): the
        terms of j for the pairs ee (|e| x 2 state indices, all perturbed at once) summed per pair label of the
        hessian palette `labelidx` (hm_j_multi)."""
        lab = np.ascontiguousarray(np.asarray(self.labels_hess)[:, labelidx], np.int32)
        ee = np.ascontiguousarray(np.asarray(ee).reshape(-1, 2), np.int32)
        nq = len(self.Q)
        h = np.zeros(nq)
        nz = np.zeros(nq)
        hc = np.zeros((nq, 4))
        _lib.check(_lib.lib().hm_j_multi(self._h, _lib.ptr(self._X(state)), float(deltaX), int(ee.shape[0]), _lib.ptr(ee),
                                         _lib.ptr(lab), nq, _lib.ptr(h), _lib.ptr(nz), _lib.ptr(hc)), "hm_j_multi")
        return h.reshape(1, -1), nz > 0, hc

    def measure(self, state, y_im, y_flow, y_m, deltaX=2.0):
        """KFState.update (kalman.py:437-449) fused: -> (Hz [4N,1], HTH [4N,4N], Hz_components [4N,4])."""
        masked = self._masked_flag(y_im, y_flow, y_m)
        n4 = 4 * self.n
        Hz = np.empty(n4)
        Hzc = np.empty((n4, 4))
        HTH = np.empty((n4, n4))
        _lib.check(_lib.lib().hm_measure(self._h, _lib.ptr(self._X(state)), float(deltaX), masked, _lib.ptr(Hz),
                                         _lib.ptr(Hzc), _lib.ptr(HTH)), "hm_measure")
        return Hz.reshape(-1, 1), HTH, Hzc

    # -- the dense update on the device (information form) --------------------------------------
    def _cov_arg(self, W, who):
        """NULL for the covariance resident on the device, else a contiguous host array."""
        if isinstance(W, DeviceCovariance):
            if not W.valid(self):
                raise RuntimeError("%s: this DeviceCovariance is no longer resident on the device" % who)
            return None
        return np.ascontiguousarray(W, np.float64)

    def _cov_result(self, fetch):
        self._cov_serial += 1
        tok = DeviceCovariance(self, self._cov_serial)
        return tok.fetch() if fetch else tok

    def tune(self, key, value):
        _lib.check(_lib.lib().hm_ctx_tune(self._h, key.encode(), int(value)), "hm_ctx_tune")

    def project_mask(self, X, y_m=None):
        """KalmanFilter.projectmask (kalman.py:724-742) on the device -> (X projected, number of
        vertices that were outside).  y_m: host mask, or None for the mask of the observation in place."""
        Xp = np.array(X, np.float64).reshape(-1)
        if Xp.shape[0] != 4 * self.n:
            raise ValueError("state of %d entries for a mesh of %d vertices" % (Xp.shape[0], self.n))
        mask = None
        if y_m is not None:
            mask = np.ascontiguousarray(np.asarray(y_m) > 0.5).view(np.uint8)
            if mask.shape != (self.ny, self.nx):
                raise ValueError("mask of shape %r for frames of %r" % (mask.shape, (self.ny, self.nx)))
        moved = ctypes.c_int(0)
        _lib.check(_lib.lib().hm_project_mask(self._h, _lib.ptr(mask), _lib.ptr(Xp), ctypes.byref(moved)),
                   "hm_project_mask")
        return Xp.reshape(np.shape(X)), moved.value

    def prune_mask(self, y_m):
        """hm_prune_mask: the reference's contour pruning of a mask (imgproc.py:198-228) as project_mask applies it."""
        mask = np.ascontiguousarray(np.asarray(y_m) > 0.5).view(np.uint8)
        if mask.shape != (self.ny, self.nx):
            raise ValueError("mask of shape %r for frames of %r" % (mask.shape, (self.ny, self.nx)))
        out = np.empty_like(mask)
        _lib.check(_lib.lib().hm_prune_mask(self._h, _lib.ptr(mask), _lib.ptr(out)), "hm_prune_mask")
        return out

    def cov_fetch(self):
        n4 = 4 * self.n
        W = np.empty((n4, n4))
        _lib.check(_lib.lib().hm_cov_fetch(self._h, _lib.ptr(W)), "hm_cov_fetch")
        return W

    def cov_predict(self, W, bars, blocks, a, s, eps_F, fetch=True):
        """hm_cov_predict: F W F^T + Weps on the device.  W is a host array or the DeviceCovariance
        the last update left on the device; fetch=False returns a DeviceCovariance instead of
        copying 4N x 4N doubles back."""
        Win = self._cov_arg(W, "cov_predict")
        nb = 0 if bars is None else int(len(bars))
        b = None if nb == 0 else np.ascontiguousarray(bars, np.int32)
        blk = None if nb == 0 else np.ascontiguousarray(blocks, np.float64)
        _lib.check(_lib.lib().hm_cov_predict(self._h, _lib.ptr(Win), nb, _lib.ptr(b), _lib.ptr(blk), float(a), float(s),
                                             float(eps_F), None), "hm_cov_predict")
        return self._cov_result(fetch)

    def ms_predict(self, W, X, bars, l0, kappa, M, dt, maxiter, tol, eps_F, prefactor=True):
        """hm_ms_predict: IteratedMSKalmanFilter.predict (kalman.py:850-863) for the DeviceCovariance W in one native
        call -> (X predicted [4N,1], Newton iterations, DeviceCovariance of the predicted covariance)."""
        if self._cov_arg(W, "ms_predict") is not None:
            raise TypeError("ms_predict takes the DeviceCovariance resident on the device")
        Xp = np.ascontiguousarray(np.asarray(X, np.float64).reshape(-1)).copy()
        b = np.ascontiguousarray(bars, np.int32)
        l0 = np.ascontiguousarray(np.asarray(l0, np.float64).reshape(-1))
        its = ctypes.c_int()
        _lib.check(_lib.lib().hm_ms_predict(self._h, int(b.shape[0]), _lib.ptr(b), _lib.ptr(l0), float(kappa), float(M),
                                            float(dt), int(maxiter), float(tol), float(eps_F), _lib.ptr(Xp),
                                            ctypes.byref(its), 1 if prefactor else 0), "hm_ms_predict")
        return Xp.reshape(-1, 1), its.value, self._cov_result(False)

    def update_prefactor(self, W):
        """hm_update_prefactor: queue the factorisation / inversion of the DeviceCovariance W now (it
        does not need the predicted state); the next update_begin / update_run with W picks it up."""
        if self._cov_arg(W, "update_prefactor") is not None:
            raise TypeError("update_prefactor takes the DeviceCovariance resident on the device")
        _lib.check(_lib.lib().hm_update_prefactor(self._h), "hm_update_prefactor")

    def update_begin(self, W_prior, X0):
        """Factor the prior covariance on the device and keep inv(W), X0 there (hm_update_begin).
        A DeviceCovariance is used where it is."""
        W = self._cov_arg(W_prior, "update_begin")
        x0 = np.ascontiguousarray(np.asarray(X0, np.float64).reshape(-1))
        _lib.check(_lib.lib().hm_update_begin(self._h, _lib.ptr(W), _lib.ptr(x0)), "hm_update_begin")
        if W is not None:
            self._cov_serial += 1

    def update_run(self, W_prior, X0, y_im, y_flow, y_m, max_iter, reltol, deltaX=2.0, fetch=False, tail=True):
        """hm_update_run: the whole iterated update (kalman.py:774-831) ->
        (X kept [4N,1], info dict, errs [niter,4], Hz_components [4N,4], gains [3,4N], covariance).
        tail=False: the call does not wait for the kernels that form the Hz components and the gains (None in their place);
        update_tail() fetches them, until the next update_run."""
        masked = self._masked_flag(y_im, y_flow, y_m)
        W = self._cov_arg(W_prior, "update_run")
        n4 = 4 * self.n
        X = np.ascontiguousarray(np.asarray(X0, np.float64).reshape(-1)).copy()
        info = (ctypes.c_int * 4)()
        errs = np.zeros((max(int(max_iter), 1), 4))
        Hzc = np.empty((n4, 4)) if tail else None
        gains = np.empty((3, n4)) if tail else None
        rc = _lib.lib().hm_update_run(self._h, _lib.ptr(W), _lib.ptr(X), float(deltaX), masked, int(max_iter),
                                      float(reltol), info, _lib.ptr(errs), _lib.ptr(Hzc), _lib.ptr(gains), None)
        if rc == 2:                                 # a chained run whose state prediction gave up: the caller predicts on the host
            raise ChainedPredictionFailed()         # (nothing was updated: the predicted covariance is still the resident one)
        self._cov_serial += 1
        if rc == _lib.HM_ERR_NUMERIC:
            raise FloatingPointError(_lib.lib().hm_last_error().decode())
        _lib.check(rc, "hm_update_run")
        out = dict(niter=info[0], accepted=info[1], reverted=bool(info[2]), converged=bool(info[3]))
        return X.reshape(-1, 1), out, errs[:info[0]], Hzc, gains, self._cov_result(fetch)

    def arm_newton(self, worker, bars, l0, kappa, M, dt, maxiter, tol):
        """hm_update_arm_newton: the next update_run starts hm_ms_newton on `worker` with the state it ends with, as
        soon as that state is known."""
        bars = np.ascontiguousarray(bars, np.int32)
        l0 = np.ascontiguousarray(l0, np.float64)
        _lib.check(_lib.lib().hm_update_arm_newton(self._h, worker, int(bars.shape[0]), _lib.ptr(bars), _lib.ptr(l0),
                                                   float(kappa), float(M), float(dt), int(maxiter), float(tol)),
                   "hm_update_arm_newton")

    def attach_worker(self, worker, on=True):
        """hm_ms_worker_attach: the state predictions started on `worker` run as a launch on this renderer's device."""
        _lib.check(_lib.lib().hm_ms_worker_attach(worker, self._h if on else None), "hm_ms_worker_attach")
        self._worker = worker if on else None

    def detach_worker(self, worker):
        """(the worker is about to be destroyed)"""
        if self._worker is not None and self._h is not None:
            _lib.lib().hm_ms_worker_attach(worker, None)
        self._worker = None

    def chain_project(self):
        """hm_chain_project: projectmask of the state prediction in flight queued behind it, its result the prior mean
        of the next update_run (whose X0 is then ignored) -- True when queued, False when there is nothing to chain."""
        rc = _lib.lib().hm_chain_project(self._h)
        if rc == 1:
            return False
        _lib.check(rc, "hm_chain_project")
        return True

    def update_tail(self):
        """hm_update_tail -> (Hz components [4N,4], gains [3,4N]) of the last update_run(tail=False)."""
        n4 = 4 * self.n
        Hzc, gains = np.empty((n4, 4)), np.empty((3, n4))
        _lib.check(_lib.lib().hm_update_tail(self._h, _lib.ptr(Hzc), _lib.ptr(gains)), "hm_update_tail")
        return Hzc, gains

    def arm_mask(self, d_mask):
        """hm_update_arm_mask: the next update_run queues the outline of this mask (device address: the NEXT frame's) when
        its state is final."""
        _lib.check(_lib.lib().hm_update_arm_mask(self._h, ctypes.c_void_p(int(d_mask))), "hm_update_arm_mask")

    def prepare_mask(self, d_mask):
        """hm_prepare_mask: pruning + outline of the NEXT observation's mask (device address) queued a frame ahead."""
        _lib.check(_lib.lib().hm_prepare_mask(self._h, ctypes.c_void_p(int(d_mask))), "hm_prepare_mask")

    def chain_states(self):
        """hm_chain_states -> (predicted state, projected state, Newton iterations, vertices moved) of the last chained
        update_run."""
        n4 = 4 * self.n
        pred, proj = np.empty(n4), np.empty(n4)
        its, moved = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib().hm_chain_states(self._h, _lib.ptr(pred), _lib.ptr(proj), ctypes.byref(its), ctypes.byref(moved)),
                   "hm_chain_states")
        return pred, proj, its.value, moved.value

    def arm_cov(self, eps_F):
        """hm_update_arm_cov: the next update_run (armed with arm_newton as well) also queues the covariance half of the
        next frame's prediction -- cov_predict at the state it ends with, then update_prefactor -- behind its own last
        launches; predict_take makes them current."""
        _lib.check(_lib.lib().hm_update_arm_cov(self._h, float(eps_F)), "hm_update_arm_cov")

    def predict_take(self, W, X, bars, l0, kappa, a, s, eps_F):
        """hm_predict_take: the DeviceCovariance of the prediction update_run queued ahead from the resident covariance W,
        if it was made from exactly these inputs (its factorisation for the next update is queued as well, as after
        update_prefactor) -- else None, and the caller calls cov_predict / update_prefactor itself."""
        if not (isinstance(W, DeviceCovariance) and W.valid(self)):
            return None
        x = np.ascontiguousarray(np.asarray(X, np.float64).reshape(-1))
        b = np.ascontiguousarray(bars, np.int32)
        l0 = np.ascontiguousarray(np.asarray(l0, np.float64).reshape(-1))
        rc = _lib.lib().hm_predict_take(self._h, _lib.ptr(x), int(b.shape[0]), _lib.ptr(b), _lib.ptr(l0), float(kappa),
                                        float(a), float(s), float(eps_F))
        if rc == 1:
            return None
        _lib.check(rc, "hm_predict_take")
        return self._cov_result(False)

    def update_step(self, state, y_im, y_flow, y_m, deltaX=2.0, want_error=True):
        """hm_update_step: measurement at state.X, the solve and (want_error) Renderer.error of the new
        iterate X0 + step -> (step [4N,1], Hz_components [4N,4], (e_im, e_fx, e_fy, e_m) or None)."""
        masked = self._masked_flag(y_im, y_flow, y_m)
        n4 = 4 * self.n
        step = np.empty(n4)
        Hzc = np.empty((n4, 4))
        err = (ctypes.c_double * 4)()
        _lib.check(_lib.lib().hm_update_step(self._h, _lib.ptr(self._X(state)), float(deltaX), masked,
                                             _lib.ptr(step), _lib.ptr(Hzc), err if want_error else None),
                   "hm_update_step")
        e = (int(err[0]), err[1], err[2], int(err[3])) if want_error else None
        return step.reshape(-1, 1), Hzc, e

    def update_cov(self, which=0, fetch=True):
        """hm_update_cov: covariance of the last step (0), of the one before (1) or the prior (-1)."""
        _lib.check(_lib.lib().hm_update_cov(self._h, int(which), None), "hm_update_cov")
        return self._cov_result(fetch)

    def error(self, state, y_im, y_flow, y_m, want_flow=True):
        """reference renderer.py:485-501 -> (e_im, e_fx, e_fy, e_m, fx, fy).

        want_flow=False skips the device-to-host copy of the two rendered flow planes (the IEKF
        loop only looks at the four sums); fx, fy are then None."""
        masked = self._masked_flag(y_im, y_flow, y_m)
        err = (ctypes.c_double * 4)()
        fx = fy = None
        if not want_flow and not masked:
            # the state an update_run has just kept: the sums came out of the render of its last iterate
            rc = _lib.lib().hm_update_last_error(self._h, _lib.ptr(self._X(state)), err)
            if rc == 0:
                return int(err[0]), err[1], err[2], int(err[3]), None, None
            if rc < 0:
                _lib.check(rc, "hm_update_last_error")
        if want_flow:
            fx = np.empty((self.ny, self.nx), np.float32)
            fy = np.empty_like(fx)
        _lib.check(_lib.lib().hm_error(self._h, _lib.ptr(self._X(state)), masked, err, _lib.ptr(fx), _lib.ptr(fy)),
                   "hm_error")
        if want_flow:      # the reference returns read_pixels arrays of shape (ny, nx, 1)
            fx, fy = fx[:, :, None], fy[:, :, None]
        return int(err[0]), err[1], err[2], int(err[3]), fx, fy


class DeviceCovariance:
    """A 4N x 4N covariance that lives on the device: what the last cov_predict / update_cov /
    update_run of one Renderer left there.  fetch() copies it to the host; it stops being valid
    when that renderer produces the next one."""

    def __init__(self, renderer, serial):
        self._renderer, self._serial = renderer, serial

    def valid(self, renderer=None):
        r = self._renderer
        return (renderer is None or renderer is r) and r._cov_serial == self._serial

    def fetch(self):
        if not self.valid():
            raise RuntimeError("this DeviceCovariance is no longer resident on the device")
        return self._renderer.cov_fetch()


class DeviceObservation:
    """One observed frame held in device memory (addresses as integers): u8 frame, f32 flow x / y,
    u8 mask in {0,1}, all W*H row-major.  ``raw`` and ``masked`` are the tokens to pass as y_flow."""

    class _Token:
        def __init__(self, name):
            self.name = name

    def __init__(self, d_y_im, d_flowx, d_flowy, d_y_m, y_m_host=None, next_mask=None):
        self.d_y_im, self.d_flowx, self.d_flowy, self.d_y_m = d_y_im, d_flowx, d_flowy, d_y_m
        self.y_m_host = y_m_host
        self.next_mask = next_mask      # device address of the NEXT frame's mask when it is resident already (hm_prepare_mask)
        self.raw = DeviceObservation._Token("raw")
        self.masked = DeviceObservation._Token("masked")


class MaskedFlow(np.ndarray):
    """y_m * y_flow (kalman.py:679-682) that remembers the raw flow it was made from, so the
    renderer can use the device-side product instead of uploading the frame a second time."""

    def __new__(cls, y_flow, y_m):
        out = np.dstack((y_m * y_flow[:, :, 0], y_m * y_flow[:, :, 1])).astype(np.float32).view(cls)
        out._hm_masked_from = y_flow
        return out

    def __array_finalize__(self, obj):
        self._hm_masked_from = getattr(obj, "_hm_masked_from", None)


class FlowStream:
    """reference renderer.py:807-874: flow frames <path>_%03d_x.mat / _y.mat."""

    def __init__(self, path):
        self.path = path
        self.frame = 0

    def _names(self):
        return (self.path + "_%03d_x.mat" % self.frame, self.path + "_%03d_y.mat" % self.frame)

    def peek(self):
        fn_x, fn_y = self._names()
        try:
            self.flowx = matio.read_mat(fn_x)
            self.flowy = matio.read_mat(fn_y)
        except IOError:
            return False, None
        return True, np.dstack((self.flowx, self.flowy)).astype(np.float32)

    def read(self):
        ret, flow = self.peek()
        self.frame += 1
        return ret, flow

    def isOpened(self):
        fn_x, fn_y = self._names()
        return os.path.isfile(fn_x) and os.path.isfile(fn_y)
