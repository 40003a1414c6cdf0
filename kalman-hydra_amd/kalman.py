"""Extended / iterated Kalman filter of the mesh tracker -- the reference's filter API.

Drop-in for the classes of reference kalman.py on the hot path: ``KFState``
(:106-624), ``KalmanFilter`` (:626-765), ``IteratedKalmanFilter`` (:767-831),
``IteratedMSKalmanFilter`` (:834-960) and the module-level ``stats`` (:25-104).
Constructor arguments, defaults, method names, the state layout
``X = [x0,y0,...,vx0,vy0,...]`` (:178) and the return value of ``compute`` are the
reference's.  What changes is where the work is done:

* ``KFState.update`` (:437-449) is one call into libhydra_mi.so
  (``Renderer.measure`` -> ``hm_measure``): the mesh is rasterised and every
  finite-difference perturbation of kalman.py:491-518 / :583-606 is evaluated and
  reduced on the GPU, inside the bounding box of the triangles it moves.  The
  result has the single-perturbation semantics of the reference
  (``multi=False``); ``multi=True`` -- the reference's way of batching several
  perturbations into one render -- is accepted and gives the same numbers.
* the dense algebra of the update runs on the device (``hm_update_run`` /
  ``hm_update_begin`` / ``_step`` / ``_cov``): one blocked Cholesky factorisation per IEKF
  iteration on the f64 matrix cores instead of an explicit inverse per iteration; the
  covariance is formed once, for the state that is kept, and stays on the device between
  predict and update.  Same mathematics, rounding-level differences.
* the mass-spring predict is native code (``hm_ms_newton``: block-eliminated Newton, the spring
  operator applied bar by bar), the covariance prediction a device kernel (``hm_cov_predict``).

There is no CPU measurement path: ``cuda=False`` raises.
"""
import ctypes
import os
import sys
import time

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import _lib
from .renderer import DeviceCovariance, Renderer, MaskedFlow, DeviceObservation


class Statistics:
    """reference kalman.py:25-62: counters the timing study reads (timing_synthetic.py:60-90)."""

    _scalars = ("niter", "meshpts", "gridsize", "jacobianpartitions", "hessianpartitions", "nzj")
    _lists = dict(jacobianrenderstc=2, hessianrenderstc=2, renders=1, statepredtime=1, stateupdatetc=2,
                  hessinc=1, jacinc=1, hessincsparse=1)

    def __init__(self):
        for k in self._scalars:
            setattr(self, k, 0)
        for k, n in self._lists.items():
            setattr(self, k, [0] * n)

    def reset(self):
        for k in self._scalars:
            setattr(self, k, 0)
        for k, n in self._lists.items():
            getattr(self, k)[:] = [0] * n


stats = Statistics()

# Host BLAS is only used when a caller supplies its own measurement object (the `renderer=` seam of
# the constructors, which the CPU tests of the host logic use); the product path does its dense
# algebra on the device.  For that seam: a default-sized BLAS thread team on a many-core host with a
# small CPU quota (a container, the GPU box) is slower than a handful of threads by orders of
# magnitude, so the team is capped (HYDRA_MI_BLAS_THREADS overrides the cap).
try:
    from threadpoolctl import ThreadpoolController as _ThreadpoolController
except ImportError:                                   # pragma: no cover
    _ThreadpoolController = None
_BLAS_THREADS = int(os.environ.get("HYDRA_MI_BLAS_THREADS", "8"))


def _spd_factor(A):
    """Cholesky factor of the information matrix, or an LU factor if it is not positive definite
    (the finite-difference HTH is symmetric but only approximately positive semi-definite)."""
    try:
        return ("cho", sla.cho_factor(A, lower=True, check_finite=False))
    except np.linalg.LinAlgError:
        return ("lu", sla.lu_factor(A, check_finite=False))


def _factor_solve(fac, b):
    kind, f = fac
    if kind == "cho":
        return sla.cho_solve(f, b, check_finite=False)
    return sla.lu_solve(f, b, check_finite=False)


def _spd_inverse(A):
    """inv(A) for a symmetric matrix: through the Cholesky factor when A is positive definite."""
    try:
        c, info = sla.lapack.dpotrf(A, lower=1, clean=0, overwrite_a=0)
        if info == 0:
            inv, info = sla.lapack.dpotri(c, lower=1, overwrite_c=1)
            if info == 0:
                low = np.tril(inv)
                return low + np.tril(inv, -1).T
    except Exception:
        pass
    return np.linalg.inv(A)


class _blas_cap:
    """Cap the BLAS / OpenMP pools for the host algebra of one filter step.  The pools are looked up
    once (threadpoolctl scans the loaded libraries, ~0.5 ms); filters whose algebra runs on the
    device (enabled=False) skip it altogether."""
    _controller = None

    def __init__(self, enabled=True):
        self._ctx = None
        self._enabled = enabled and _ThreadpoolController is not None

    def __enter__(self):
        if not self._enabled:
            return
        if _blas_cap._controller is None:
            _blas_cap._controller = _ThreadpoolController()
        self._ctx = _blas_cap._controller.limit(limits=_BLAS_THREADS)
        self._ctx.__enter__()

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


def _log(msg):
    sys.stdout.write(msg + "\n")


def _rows_unique(a):
    return np.unique(np.asarray(a), axis=0)


class KFState:
    _ori_stale = False

    def __init__(self, distmesh, im, flow, cuda, eps_F=1, eps_Z=1e-3, eps_J=1e-3, eps_M=1e-3, vel=None,
                 sparse=True, multi=True, verbose=False, renderer=None, device=0):
        self.multi = multi
        self.sparse = sparse
        self.verbose = verbose
        self._ver = np.array(distmesh.p, np.float32)
        self._vel = np.zeros(self._ver.shape, np.float32) if vel is None else np.asarray(vel).reshape(self._ver.shape)
        self.tex = im
        self.nx, self.ny = im.shape[0], im.shape[1]
        self.M = self.nx * self.ny
        self.NZ = self.M
        self.eps_F, self.eps_Z, self.eps_J, self.eps_M = eps_F, eps_Z, eps_J, eps_M
        self.N = distmesh.size()
        self.u = self._ver

        # orientation of the faces and removal of slivers, |sin(angle at vertex 0)| <= 0.06 (:148-161)
        tri = np.asarray(distmesh.t)
        a = self._ver[tri[:, 1]] - self._ver[tri[:, 0]]
        b = self._ver[tri[:, 2]] - self._ver[tri[:, 0]]
        cr = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
        sine = cr / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
        keep = np.abs(sine) > 0.06
        if verbose:
            _log("Removing %d faces for being too flat" % int(np.sum(~keep)))
        self.sineface = sine[keep]
        self.ori = np.sign(cr)[keep]
        self.tri = tri[keep]
        distmesh.t = self.tri
        self.NT = self.tri.shape[0]

        # bars that no remaining face uses are dropped (:163-175)
        t = self.tri
        allbars = np.vstack((t[:, [0, 1]], t[:, [1, 2]], t[:, [0, 2]]))
        allbars = _rows_unique(np.sort(allbars, axis=1))
        have = set(map(tuple, allbars))
        sel = np.array([tuple(bar) in have for bar in np.asarray(distmesh.bars)], bool)
        distmesh.bars = np.asarray(distmesh.bars)[sel]
        distmesh.L = np.asarray(distmesh.L)[sel]

        # state, transition and covariances (:178-186)
        N = self.N
        self.X = np.vstack((self._ver.reshape(-1, 1), self._vel.reshape(-1, 1))).astype(np.float64)
        e = np.eye(2 * N)
        z = np.zeros((2 * N, 2 * N))
        self.F = np.block([[e, e], [z, e]])
        self.Weps = eps_F * np.block([[e / 4, e / 2], [e / 2, e]])
        self.W = np.block([[1e-2 * e, z], [z, e]])

        # vertex adjacency and the sparsity pattern of HTH (:189-205)
        Jv = np.eye(N)
        for k in range(3):
            for l in range(3):
                Jv[t[:, k], t[:, l]] = 1
        self.Jv = Jv
        self.J = np.kron(np.ones((2, 2)), np.kron(Jv, np.ones((2, 2))))

        # spring incidence and rest lengths (:206-218)
        bars = distmesh.bars
        self.I = bars.shape[0]
        Kp = np.zeros((N, self.I))
        Kp[bars[:, 0], np.arange(self.I)] = 1
        Kp[bars[:, 1], np.arange(self.I)] = -1
        self.Kp = Kp
        self.K = np.kron(Kp, np.eye(2))
        self.Ks = sp.csr_matrix(self.K)
        self.l0 = self.lengths()
        self.L = self.lengths()

        # perturbation partitions of the reference's multi-render scheme (:223-389); kept as
        # attributes for API compatibility, the native path does not need them
        self.E, self.labels = self._vertex_partitions()
        self.Q, self.E_hessian, self.E_hessian_idx, self.labels_hess = self._pair_partitions()

        # `renderer` lets a caller supply the measurement object (tests of the host logic do);
        # by default it is the HIP one -- there is no CPU implementation in this package
        self.renderer = renderer if renderer is not None else Renderer(
            distmesh, self._vel, flow, self.nx, im, cuda, eps_Z, eps_J, eps_M, self.labels, self.labels_hess,
            self.Q, showtracking=False, device=device)

        stats.meshpts = self.N
        stats.gridsize = getattr(distmesh, "h0", 0)
        stats.nzj = np.sum(self.J) / 2 + (self.N * 4) / 2
        stats.hessinc[0] = 2 + (stats.meshpts * 4) * (stats.meshpts * 4)
        stats.jacinc[0] = 2 + stats.meshpts * 4 * 2
        stats.hessincsparse[0] = 2 + stats.nzj * 2
        stats.jacobianpartitions = len(self.E)
        stats.hessianpartitions = len(self.E_hessian)

    # -- partitions ---------------------------------------------------------------------------
    def _vertex_partitions(self):
        """Greedy classes of mutually non-adjacent vertices (:223-255) and, per class, the vertex
        each triangle is attributed to, -1 if none (:263-272)."""
        nbr = [set(np.nonzero(self.Jv[q])[0]) - {q} for q in range(self.N)]
        free = set(range(self.N))
        queue = list(range(self.N))
        classes = []
        while queue:
            later = set()
            members = []
            while queue:
                q = queue[0]
                members.append(q)
                later = (later | nbr[q]) & free
                free.discard(q)
                queue = [x for x in queue if x != q and x not in nbr[q]]
            queue = sorted(later)
            classes.append(members)
        labels = -np.ones((len(self.tri), len(classes)))
        for k, members in enumerate(classes):
            for node in members:                       # later members overwrite earlier ones, as in the reference
                labels[np.any(self.tri == node, axis=1), k] = node
        return classes, labels

    def _pair_partitions(self):
        """Same idea for the vertex pairs Q = {(i,j): i<=j adjacent} of the sparse Hessian (:305-389)."""
        N = self.N
        Q = np.array([[i, j] for i in range(N) for j in range(i, N) if self.Jv[i, j]])
        nbr = [set(np.nonzero(self.Jv[q])[0]) for q in range(N)]
        diag = {int(q[0]): k for k, q in enumerate(Q) if q[0] == q[1]}
        free = set(range(len(Q)))
        queue = list(range(len(Q)))
        classes, classes_idx = [], []
        while queue:
            later = set()
            members = []
            while queue:
                qi = queue[0]
                a, b = int(Q[qi][0]), int(Q[qi][1])
                near = (nbr[a] | nbr[b]) - {a, b}
                clash = {k for k in queue if int(Q[k][0]) in near or int(Q[k][1]) in near}
                clash |= {diag[a], diag[b]} & set(queue)
                members.append(qi)
                # the reference intersects its "do later" set with the pairs not yet placed BEFORE it
                # marks the current one as placed (:341-344), and a diagonal pair clashes with itself
                # (p_self1 / p_self2, :331-332): (i,i) is scheduled once more and lands in a second
                # class -- kept, the lists must equal the reference's
                later = (later | clash) & free
                free.discard(qi)
                queue = [k for k in queue if k != qi and k not in clash]
            queue = sorted(later)
            idx = np.array(sorted(members), float)
            classes_idx.append(idx)
            classes.append(Q[idx.astype(int)].reshape(-1, 2))
        labels = -np.ones((len(self.tri), len(classes)))
        for k, pairs in enumerate(classes):
            for i, (n1, n2) in enumerate(pairs):
                hit = np.any(self.tri == n1, axis=1) | np.any(self.tri == n2, axis=1)
                labels[hit, k] = classes_idx[k][i]
        return Q, classes, classes_idx, labels

    # -- geometry -------------------------------------------------------------------------------
    def vertices(self):
        return self.X[0:2 * self.N].reshape((-1, 2))

    def velocities(self):
        return self.X[2 * self.N:].reshape((-1, 2))

    def lengths(self):
        d = self.Ks.T.dot(self.vertices().reshape(-1, 1)).reshape(-1, 2)
        return np.sqrt((d * d).sum(axis=1)).reshape(-1, 1)

    # The covariance may live on the device between the calls of a filter step (renderer.DeviceCovariance);
    # reading state.W brings it to the host once, assigning replaces it.
    @property
    def W(self):
        if isinstance(self._W, DeviceCovariance):
            self._W = self._W.fetch()
        return self._W

    @W.setter
    def W(self, value):
        self._W = value

    def update_orientation(self):
        ver = self.vertices()
        a = ver[self.tri[:, 1]] - ver[self.tri[:, 0]]
        b = ver[self.tri[:, 2]] - ver[self.tri[:, 0]]
        cr = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
        self.ori = np.sign(cr)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.sineface = cr / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))

    # ori / sineface (:410-414): the fused update tests the orientation of the faces natively (hm_update_run) and only
    # marks these stale; they are formed from the state when somebody looks (eight NumPy passes over the faces, ~50 us of
    # every frame otherwise)
    @property
    def ori(self):
        if self._ori_stale:
            self._ori_stale = False
            self.update_orientation()
        return self._ori

    @ori.setter
    def ori(self, value):
        self._ori, self._ori_stale = value, False

    @property
    def sineface(self):
        if self._ori_stale:
            self._ori_stale = False
            self.update_orientation()
        return self._sineface

    @sineface.setter
    def sineface(self, value):
        self._sineface = value

    def size(self):
        return self.X.shape[0]

    def setforce(self, f):
        self.renderer.force = f

    def get_flow(self):
        return self.renderer.get_flow()

    def refresh(self, multi_idx=-1, hess=False):
        self.renderer.update_vertex_buffer(self.vertices(), self.velocities(), multi_idx, hess)

    def render(self):
        stats.renders[0] += 1
        return self.renderer.render()

    # -- linearisation of the measurement ------------------------------------------------------------
    def update(self, y_im, y_flow, y_m):
        """(Hz, HTH, Hz_components) at the current state -- reference kalman.py:437-449."""
        t0 = time.time()
        Hz, HTH, Hzc = self.renderer.measure(self, y_im, y_flow, y_m)
        dt = time.time() - t0
        stats.stateupdatetc[0] += dt
        stats.stateupdatetc[1] += 1
        # the reference splits this time between its Jacobian and Hessian passes; here it is one launch
        stats.jacobianrenderstc[0] += dt / 2
        stats.jacobianrenderstc[1] += stats.jacinc[0]
        stats.hessianrenderstc[0] += dt / 2
        stats.hessianrenderstc[1] += stats.hessincsparse[0] if self.sparse else stats.hessinc[0]
        return Hz, HTH, Hzc

    # one perturbation per render, through the fine-grained operators (slow; used by the parity tests)
    def _jacobian(self, y_im, y_flow, y_m, deltaX=2):
        n = self.size()
        Hz = np.zeros((n, 1))
        Hzc = np.zeros((n, 4))
        self.refresh()
        self.renderer.initjacobian(y_im, y_flow, y_m)
        for idx in range(n):
            for sign in (+1, -1):
                self.X[idx, 0] += sign * deltaX
                hz, hzc = self.renderer.jz(self)
                self.X[idx, 0] -= sign * deltaX
                Hz[idx, 0] += sign * hz / deltaX
                Hzc[idx, :] += sign * hzc / deltaX
            Hz[idx, 0] /= 2
            Hzc[idx, :] /= 2
        return Hz, Hzc

    def _jacobian_multi(self, y_im, y_flow, y_m, deltaX=2):
        """reference kalman.py:452-489: the vertices of one partition (mutually non-adjacent, so their stars do
        not overlap) are perturbed in ONE render and the sums are separated by triangle label
        (Renderer.jz_multi -> hm_jz_multi).  8 renders per partition instead of 8 per vertex.  The product's
        update() does not use it (measure() evaluates single perturbations inside their stars); it is the
        reference's operator sequence, kept runnable for the multi == single protocol
        (testbites/test_multipert_validation.py:98-189)."""
        n = self.size()
        Hz = np.zeros((n, 1))
        Hzc = np.zeros((n, 4))
        for idx, e in enumerate(self.E):
            e = np.asarray(e, np.int64)
            self.refresh(idx)
            self.renderer.initjacobian(y_im, y_flow, y_m)
            for i in range(2):
                for j in range(2):
                    ee = i + 2 * self.N * j + 2 * e
                    self.X[ee, 0] += deltaX
                    self.refresh(idx)
                    hz, hzc = self.renderer.jz_multi(self)
                    Hz[ee, 0] = hz[e, 0] / deltaX
                    Hzc[ee, :] = hzc[e, :] / deltaX
                    self.X[ee, 0] -= 2 * deltaX
                    self.refresh(idx)
                    hz, hzc = self.renderer.jz_multi(self)
                    Hz[ee, 0] = Hz[ee, 0] - hz[e, 0] / deltaX
                    Hzc[ee, :] -= hzc[e, :] / deltaX
                    self.X[ee, 0] += deltaX
                    Hz[ee, 0] = Hz[ee, 0] / 2
                    Hzc[ee, :] = Hzc[ee, :] / 2
        self.refresh()
        return Hz, Hzc

    def _hessian_sparse(self, y_im, y_flow, y_m, deltaX=2, pattern=None):
        n = self.size()
        HTH = np.zeros((n, n))
        self.refresh()
        self.renderer.initjacobian(y_im, y_flow, y_m)
        for i in range(n):
            for j in range(i, n):
                if pattern is None or pattern[i, j] == 1:
                    HTH[i, j] = self.renderer.j(self, deltaX, i, j) / deltaX / deltaX
                    HTH[j, i] = HTH[i, j]
        return HTH

    def _hessian_sparse_multi(self, y_im, y_flow, y_m, deltaX=2):
        """reference kalman.py:539-581: per partition of vertex pairs 16 x 2 renders, sums separated by pair label
        (Renderer.j_multi -> hm_j_multi)."""
        n = self.size()
        HTH = np.zeros((n, n))
        for idx, e in enumerate(self.E_hessian):
            e = np.asarray(e, np.int64).reshape(-1, 2)
            self.refresh(idx, hess=True)
            self.renderer.initjacobian(y_im, y_flow, y_m)
            for i1 in range(2):
                for j1 in range(2):
                    for i2 in range(2):
                        for j2 in range(2):
                            off1 = i1 + 2 * self.N * j1
                            off2 = i2 + 2 * self.N * j2
                            ee = np.column_stack((2 * e[:, 0] + off1, 2 * e[:, 1] + off2))
                            h, hist, _ = self.renderer.j_multi(self, deltaX, ee, idx, self.E_hessian_idx[idx])
                            for qi in np.flatnonzero(hist):
                                q = self.Q[qi]
                                q1, q2 = 2 * q[0] + off1, 2 * q[1] + off2
                                HTH[q1, q2] = h[0, qi] / deltaX / deltaX
                                HTH[q2, q1] = HTH[q1, q2]
        self.refresh()
        return HTH

    def _hessian(self, y_im, y_flow, y_m, deltaX=2):
        return self._hessian_sparse(y_im, y_flow, y_m, deltaX, None)


class KalmanFilter:
    def __init__(self, distmesh, im, flow, cuda, vel=None, sparse=True, multi=True, eps_F=1, eps_Z=1e-3,
                 eps_J=1e-3, eps_M=1e-3, verbose=False, renderer=None, device=0):
        self.distmesh = distmesh
        self.N = distmesh.size()
        self.verbose = verbose
        self.state = KFState(distmesh, im, flow, cuda, vel=vel, sparse=sparse, multi=multi, eps_F=eps_F,
                             eps_Z=eps_Z, eps_J=eps_J, eps_M=eps_M, verbose=verbose, renderer=renderer,
                             device=device)
        _lib.register(self, 1)
        self.predtime = 0
        self.updatetime = 0
        self.projecttime = 0
        # compute() returns the rendered flow planes as host arrays, as the reference does (renderer.py:485-501).
        # A streaming caller that only wants the four error sums sets this to False: the planes stay on the
        # device (8 MB less over PCIe per 1024^2 frame) and compute() returns None in their place.
        self.return_flow = True

    def size(self):
        return self.N * 4

    def close(self):
        """Releases the native state of the filter: its renderer handle (streams, page-locked blocks, device memory)."""
        r = getattr(getattr(self, "state", None), "renderer", None)
        if r is not None and hasattr(r, "close"):
            r.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _say(self, msg):
        if self.verbose:
            _log(msg)

    def compute(self, y_im, y_flow, y_m, maskflow=True, imageoutput=None):
        """One frame: predict, project onto the mask, update; returns the error tuple (:676-700).

        ``y_im`` may be a renderer.DeviceObservation (frame, flow and mask already in device
        memory, e.g. the flow straight from hm_brox_calc_dev); y_flow / y_m are then ignored."""
        if isinstance(y_im, DeviceObservation):
            obs = y_im
            self.state.renderer.set_observation_dev(obs)
            y_flow, y_m = obs.raw, obs
            y_flow_mask = obs.masked if maskflow is True else obs.raw
        else:
            self.state.renderer.update_frame(y_im, y_flow, y_m)
            y_flow_mask = MaskedFlow(y_flow, y_m) if maskflow is True else y_flow
        r = self.state.renderer
        r.frame_in_place = True          # the operators below work on the frame just uploaded: no second copy
        try:
            with _blas_cap(enabled=not hasattr(r, "update_run")):
                t0 = time.time()
                if self._compute_chained(y_im, y_flow_mask, y_m):
                    t1 = t2 = t0 + (self._chain_predict_s or 0.0)
                else:
                    self.predict()
                    t1 = time.time()
                    self.projectmask(y_m)
                    t2 = time.time()
                    self.update(y_im, y_flow_mask, y_m)
                self._after_update()
                t3 = time.time()
            self.predtime += t1 - t0
            self.projecttime += t2 - t1
            self.updatetime += t3 - t2
            self._say("Prediction time: %g\nProjection time: %g\nUpdate time: %g" % (t1 - t0, t2 - t1, t3 - t2))
            if imageoutput is not None:
                self._say("imageoutput=%r ignored: screenshots are not part of this path" % (imageoutput,))
            return self.error(y_im, y_flow, y_m, want_flow=self.return_flow)
        finally:
            r.frame_in_place = False

    _chain_predict_s = None

    def _compute_chained(self, y_im, y_flow, y_m):
        """Hook: predict -> projectmask -> update as one chain on the device, where a filter can (the mass-spring filter
        with its prediction started ahead on the device); False: the three steps are taken one by one."""
        return False

    def _after_update(self):
        """Hook for work that only needs the state the update ended with (the mass-spring filter starts the next
        frame's state prediction here)."""

    def predict(self):
        """Constant-velocity prediction (:703-718): X <- F X, W <- F W F^T + Weps, F = [[I, I], [0, I]]."""
        t0 = time.time()
        st = self.state
        self.orig_x = st.X.copy()
        st.X = st.F.dot(st.X)
        self.pred_x = st.X.copy()
        if hasattr(st.renderer, "cov_predict"):
            st.W = st.renderer.cov_predict(st._W, None, None, 1.0, 0.0, st.eps_F, fetch=False)
        else:
            st.W = st.F.dot(st.W.dot(st.F.T)) + st.Weps
        stats.statepredtime[0] += time.time() - t0

    def projectmask(self, y_m):
        """Vertices more than 1 px outside the object are pulled back onto its outline (:724-742).

        The signed distance is the reference's (imgproc.py:195-235): the mask's contours pruned as there (:205-228: the
        largest object and its holes of area >= 40), then the distance to the polygon through the centres of the border
        pixels of what is left, negative inside (imgproc.outline_distance; on the device csrc/project_kernels.h, k_ccl_*
        + k_outline + k_project_mask).  Steps, step size and the stale d / index set follow the reference."""
        if y_m is None:
            return
        r = self.state.renderer
        if hasattr(r, "project_mask"):    # the same walk on the device (hm_project_mask)
            X, moved = r.project_mask(self.state.X, None if isinstance(y_m, DeviceObservation) else y_m)
            if moved:
                self.state.X = X
            self.proj_x, self.moved = np.array(self.state.X), moved
            return
        if isinstance(y_m, DeviceObservation):
            y_m = y_m.y_m_host
        p = self.state.vertices()
        near = ~_inside(y_m, p)           # vertices whose four surrounding pixels are not all object
        if not near.any():
            return                        # d <= 0 everywhere: nothing to project
        fd = _mask_distance(y_m)
        ddeps = 1e-1
        p_orig = p.copy()
        d = np.zeros(len(p))              # surely-inside vertices have d <= 0: never selected below
        d[near] = fd(p[near])
        ix = d > 1
        for _ in range(10):
            if ix.any():
                gx = (fd(p[ix] + [ddeps, 0]) - d[ix]) / ddeps
                gy = (fd(p[ix] + [0, ddeps]) - d[ix]) / ddeps
                g2 = gx ** 2 + gy ** 2
                with np.errstate(divide="ignore", invalid="ignore"):
                    step = np.where(g2 > 0, d[ix] / g2, 0.0)
                p[ix] -= (step * np.vstack((gx, gy))).T
        self.state.X[0:2 * self.N] = p.reshape(-1, 1)
        self.state.X[2 * self.N:] += (p - p_orig).reshape(-1, 1)

    def update(self, y_im, y_flow, y_m):
        """Single EKF step in information form (:745-761): W = inv(inv(W) + HTH), X += W Hz."""
        st = self.state
        X, W = st.X, st.W
        if hasattr(st.renderer, "update_step"):
            st.renderer.update_begin(W, X)
            step, Hzc, _ = st.renderer.update_step(st, y_im, y_flow, y_m, want_error=False)   # X0 - X = 0: step = W_new Hz
            if not np.all(np.isfinite(step)):
                raise FloatingPointError("information matrix of the update is not positive definite")
            Wn = st.renderer.update_cov(0)
            st.X = X + step
        else:
            Hz, HTH, Hzc = st.update(y_im, y_flow, y_m)
            Wn = _spd_inverse(_spd_inverse(W) + HTH)
            st.X = X + Wn.dot(Hz)
        st.W = Wn
        self.tv = Wn.dot(Hzc[:, 0])
        self.fv = Wn.dot(Hzc[:, 1] + Hzc[:, 2])
        self.mv = Wn.dot(Hzc[:, 3])

    def error(self, y_im, y_flow, y_m, want_flow=True):
        return self.state.renderer.error(self.state, y_im, y_flow, y_m, want_flow=want_flow)


class IteratedKalmanFilter(KalmanFilter):
    def __init__(self, distmesh, im, flow, cuda, sparse=True, multi=True, nI=10, eps_F=1e-3, eps_Z=1e-3,
                 eps_J=1e-3, eps_M=1e10, **kw):
        KalmanFilter.__init__(self, distmesh, im, flow, cuda, sparse=sparse, multi=multi, eps_F=eps_F, eps_Z=eps_Z,
                              eps_J=eps_J, eps_M=eps_M, **kw)
        self.nI = nI
        self.reltol = 1e-4
        self.fused_update = True         # hm_update_run; False: the same loop in Python over hm_update_step
        # the diagnostics of reference kalman.py:826-828 (tv, fv, mv = W Hzc[:, 0], W (Hzc[:, 1] + Hzc[:, 2]), W Hzc[:, 3])
        # are formed on the device by the fused update; True: they cross to the host when they are read, not every frame
        self.lazy_gains = True
        self._gains = (None, None, None)

    def _gain(self, k):
        if self._gains is None:
            _, g = self.state.renderer.update_tail()
            self._gains = (g[0], g[1], g[2])
        return self._gains[k]

    def _set_gain(self, k, v):
        g = list(self._gains) if self._gains is not None else [None, None, None]
        g[k] = v
        self._gains = tuple(g)

    tv = property(lambda self: self._gain(0), lambda self, v: self._set_gain(0, v))
    fv = property(lambda self: self._gain(1), lambda self, v: self._set_gain(1, v))
    mv = property(lambda self: self._gain(2), lambda self, v: self._set_gain(2, v))


    def _update_fused(self, y_im, y_flow, y_m):
        """The whole loop below in one native call (hm_update_run): same numbers, the iterate and
        the covariance stay on the device."""
        st = self.state
        t0 = time.time()
        # (the gains -- tv, fv, mv below -- are fetched when somebody reads them: the call does not wait for their kernels)
        lazy = bool(self.lazy_gains) and hasattr(st.renderer, "update_tail")
        X, info, errs, Hzc, gains, W = st.renderer.update_run(st._W, st.X, y_im, y_flow, y_m, self.nI, self.reltol, tail=not lazy)
        stats.stateupdatetc[0] += time.time() - t0
        stats.stateupdatetc[1] += info["niter"]
        for i, e in enumerate(errs):
            self._say("   IEKF K = %d" % i)
            if info["reverted"] and i == len(errs) - 1:
                self._say("** Mesh inconsistent ** Reverting to last good state and continuing")
            else:
                self._say("-- e_im: %d, e_fx: %d, e_fy: %d, e_m: %d" % tuple(e))
        st.X = X
        st.W = W
        st._ori_stale = True             # (update_orientation when ori / sineface are next read)
        self.niter = info["niter"]
        stats.niter += self.niter
        self.reverted, self.converged = info["reverted"], info["converged"]
        if lazy:
            self._gains = None                # tv / fv / mv: renderer.update_tail() on first access
        else:
            self._gains = (gains[0], gains[1], gains[2])

    def update(self, y_im, y_flow, y_m):
        """Iterated EKF in information form (:774-831).

        Per iteration the reference forms W = inv(invW0 + HTH) and
        X = X0 + W Hz - W HTH (X0 - X).  Only W (Hz - HTH (X0 - X)) is needed to move the state,
        so each iteration factorises invW0 + HTH once (Cholesky) and solves; the covariance is
        formed for the state that is kept (the last accepted one).  With the HIP renderer the
        factorisations run on the device (hm_update_begin / _step / _cov) and only the 4N-vector
        of the step crosses PCIe per iteration."""
        st = self.state
        if self.fused_update and hasattr(st.renderer, "update_run"):
            return self._update_fused(y_im, y_flow, y_m)
        on_device = hasattr(st.renderer, "update_step")
        X = st.X
        X_orig, X_old = X.copy(), X.copy()
        W_old = st._W
        if on_device:
            st.renderer.update_begin(W_old, X_orig)
        else:
            W_old = st.W
            invW_orig = _spd_inverse(W_old)
        A = A_old = None                 # host path: information matrices of X and X_old (None = the prior)
        accepted = 0                     # iterations whose state was kept
        eold = 0.0
        conv = False
        reverted = False
        Hzc = np.zeros((st.size(), 4))
        self.niter = 0
        for i in range(self.nI):
            self._say("   IEKF K = %d" % i)
            t0 = time.time()
            step = None
            if on_device:
                step, Hzc, err = st.renderer.update_step(st, y_im, y_flow, y_m)
                if not np.all(np.isfinite(step)):
                    # inv(W) is positive definite and HTH is a Gram matrix of the difference images
                    # (positive semi-definite), so this only happens with non-finite inputs
                    raise FloatingPointError("the update system inv(W) + HTH is not positive definite "
                                             "(non-finite state, covariance or observation?)")
                dt = time.time() - t0
                stats.stateupdatetc[0] += dt
                stats.stateupdatetc[1] += 1
            else:
                Hz, HTH, Hzc = st.update(y_im, y_flow, y_m)
                A = invW_orig + HTH
                step = _factor_solve(_spd_factor(A), Hz - HTH.dot(X_orig - X))
            X = X_orig + step
            st.X = X
            self.niter += 1
            st.update_orientation()
            if np.any(st.ori < 0):
                st.X = X_old
                A = A_old
                reverted = True
                self._say("** Mesh inconsistent ** Reverting to last good state and continuing")
                break
            if on_device:                # error of the new iterate came back with the step
                e_im, e_fx, e_fy, e_m = err
            else:
                e_im, e_fx, e_fy, e_m, _, _ = st.renderer.error(st, y_im, y_flow, y_m, want_flow=False)
            enew = float(np.sqrt(float(e_im) ** 2 + e_fx ** 2 + e_fy ** 2 + float(e_m) ** 2))
            self._say("-- e_im: %d, e_fx: %d, e_fy: %d, e_m: %d" % (e_im, e_fx, e_fy, e_m))
            accepted += 1
            if abs(enew - eold) / enew < self.reltol:
                conv = True
                break
            eold = enew
            X_old = X.copy()
            A_old = A
        stats.niter += self.niter
        if on_device:
            if reverted:
                st.W = st.renderer.update_cov(-1 if accepted == 0 else 1, fetch=False)
            else:
                st.W = st.renderer.update_cov(-1 if self.niter == 0 else 0, fetch=False)
        else:
            st.W = W_old if A is None else _spd_inverse(A)
        self.reverted = reverted
        Wd = st.W
        self.tv = Wd.dot(Hzc[:, 0])
        self.fv = Wd.dot(Hzc[:, 1] + Hzc[:, 2])
        self.mv = Wd.dot(Hzc[:, 3])
        self.converged = conv


class IteratedMSKalmanFilter(IteratedKalmanFilter):
    """Iterated filter with mass-spring dynamics (:834-960)."""

    def __init__(self, distmesh, im, flow, cuda, sparse=True, multi=True, nI=10, eps_F=1e-1, eps_Z=1e-3,
                 eps_J=1, eps_M=1, **kw):
        IteratedKalmanFilter.__init__(self, distmesh, im, flow, cuda, sparse=sparse, multi=multi, eps_F=eps_F,
                                      eps_Z=eps_Z, eps_J=eps_J, eps_M=eps_M, nI=nI, **kw)
        self.M = 1
        self.kappa = -1
        self.deltat = 0.05
        self.maxiter = 1000
        self.tol = 1e-4
        self.force = lambda l1, l2: -self.kappa * (l1 - l2)
        self.state.setforce(self.force)
        self._bar_pattern()
        # True: hm_ms_predict, the whole prediction in one native call with the Newton iterations of the state as a
        # single workgroup on the device.  Measured at 201 vertices (tools/predict_time.py): 1.21 ms against 0.68 ms
        # for the default below -- ~500 dependent conjugate-gradient steps of a 402-unknown system are ~2 us each
        # on one CU and ~0.8 us on a host core.  False: hm_cov_predict, hm_update_prefactor (the covariance half
        # of the update queued from a helper thread) and hm_ms_newton on the host, side by side.
        self.device_predict = False
        # True: the next frame's state prediction (hm_ms_newton, 0.42 ms of host time at 201 vertices) starts on a
        # worker thread as soon as the update has its final state, instead of at the top of the next compute()
        self.predict_ahead = True
        # True (with predict_ahead): the update also queues the covariance half of the next frame's prediction -- F W F^T +
        # Weps at the state it ends with, and the factorisation of the result -- behind its own last launches
        # (hm_update_arm_cov); predict() takes it if the state, the springs and the parameters are still the same.
        self.cov_ahead = True
        self._cov_armed = False
        # True: the prediction started ahead runs as one launch on the device (hm_ms_worker_attach: k_ms_newton4, four
        # waves, a vertex per lane) instead of on the worker's host thread, when the mesh fits the kernel (<= 256
        # vertices); the two agree to rounding
        self.newton_on_device = True
        # True: compute() chains predict -> projectmask -> update on the device when it can (_compute_chained)
        self.chain = True
        self._worker_dev = None
        self._worker, self._ahead, self._armed = None, None, None

    def _jacobian(self):
        """d f / d y of the spring force at the current vertices (:865-902), sparse.

        The reference forms  -K [diag(kappa (1 - l0/l)) x I2] K^T - K diag(d) [dk x 1_2]  with
        dk_i = kappa l0_i / l_i^3 * d_i^T (K^T)_i and d = K^T y, as dense products.  Bar i between
        vertices a and b contributes the 2x2 block  B_i = k_i I + c_i d_i d_i^T  (k_i = kappa (1 - l0_i/l_i),
        c_i = kappa l0_i / l_i^3) with sign - on (a,a), (b,b) and + on (a,b), (b,a); assembled here
        directly from those blocks."""
        st = self.state
        bars = self._bars
        y = st.vertices()
        d = y[bars[:, 0]] - y[bars[:, 1]]
        l = np.sqrt((d * d).sum(axis=1))
        l0 = st.l0[:, 0]
        k = self.kappa * (1 - l0 / l)
        c = self.kappa * l0 / l ** 3
        Bm = c[:, None, None] * (d[:, :, None] * d[:, None, :])
        Bm[:, 0, 0] += k
        Bm[:, 1, 1] += k
        vals = np.concatenate((-Bm, -Bm, Bm, Bm)).reshape(-1)
        n2 = 2 * st.N
        return sp.csr_matrix((vals, (self._jrows, self._jcols)), shape=(n2, n2))

    def _spring_blocks(self):
        """Per spring the symmetric block (Bxx, Bxy, Byy) of dfdy at the current vertices (see _jacobian)."""
        st = self.state
        y = st.vertices()
        d = y[self._bars[:, 0]] - y[self._bars[:, 1]]
        l = np.sqrt((d * d).sum(axis=1))
        l0 = st.l0[:, 0]
        k = self.kappa * (1 - l0 / l)
        c = self.kappa * l0 / (l * l * l)           # (the operations of the native spring_blocks, csrc/ekf.hip: same bits)
        return np.column_stack((k + c * d[:, 0] * d[:, 0], c * d[:, 0] * d[:, 1], k + c * d[:, 1] * d[:, 1]))

    def _bar_pattern(self):
        """Row / column indices of the four 2x2 blocks every bar contributes to dfdy."""
        bars = np.asarray(self.distmesh.bars, np.int64)
        self._bars = bars
        a, b = bars[:, 0], bars[:, 1]
        al = np.arange(2)
        rows, cols = [], []
        for (p, q) in ((a, a), (b, b), (a, b), (b, a)):
            rows.append((2 * p[:, None, None] + al[None, :, None]) + 0 * al[None, None, :])
            cols.append((2 * q[:, None, None] + al[None, None, :]) + 0 * al[None, :, None])
        self._jrows = np.concatenate(rows).reshape(-1)
        self._jcols = np.concatenate(cols).reshape(-1)

    def _dfdx(self):
        n2 = 2 * self.state.N
        e = sp.eye(n2, format="csr")
        A = self._jacobian() * (self.deltat / self.M)
        return sp.bmat([[e, self.deltat * e], [A, e]], format="csr")

    def _dgdx(self):
        n2 = 2 * self.state.N
        e = sp.eye(n2, format="csr")
        A = self._jacobian() * (self.deltat / self.M)
        return sp.bmat([[e, -self.deltat * e], [-A, e]], format="csc")

    def predict(self):
        t0 = time.time()
        st = self.state
        self.orig_x = st.X.copy()
        # F = [[I, dt I], [dt/M dfdy, I]] at the state before the step (:856)
        if self.device_predict and hasattr(st.renderer, "ms_predict") and isinstance(st._W, DeviceCovariance):
            self._take_ahead()               # a prediction started ahead on the host is not what this path uses
            # the whole prediction in one native call (hm_ms_predict): the Newton iterations of the state as one
            # workgroup on a second stream while the covariance prediction and, for the fused update, the
            # factorisation / inversion it starts with are queued on the filter's stream
            st.X, self.newton_iterations, st.W = st.renderer.ms_predict(
                st._W, st.X, self._bars, st.l0[:, 0], self.kappa, self.M, self.deltat, self.maxiter, self.tol, st.eps_F,
                prefactor=self.fused_update)
        elif hasattr(st.renderer, "cov_predict"):
            # the covariance half first: it is queued on the device (prediction, then the
            # factorisation and inversion the update starts with) and runs while the host works
            # through the Newton iterations of the state
            taken = None
            if self._cov_armed and hasattr(st.renderer, "predict_take"):
                # the update of the last frame queued exactly this behind its own launches (hm_update_arm_cov)
                taken = st.renderer.predict_take(st._W, st.X, self._bars, st.l0[:, 0], self.kappa, self.deltat,
                                                 self.deltat / self.M, st.eps_F)
            self._cov_armed = False
            if taken is not None:
                st.W = taken
            else:
                blocks = self._spring_blocks()
                st.W = st.renderer.cov_predict(st._W, self._bars, blocks, self.deltat, self.deltat / self.M, st.eps_F,
                                               fetch=False)
                if self.fused_update and hasattr(st.renderer, "update_prefactor"):
                    st.renderer.update_prefactor(st._W)
            self._newton()
        else:
            A = self._jacobian() * (self.deltat / self.M)
            self._newton()
            st.W = _fwft(A, self.deltat, st.W) + st.Weps
        self.pred_x = st.X.copy()
        stats.statepredtime[0] += time.time() - t0

    def _compute_chained(self, y_im, y_flow, y_m):
        """predict -> projectmask -> update (reference kalman.py:676-700) without the host between them: the state
        prediction started ahead on the device is still in flight or has just ended, projectmask of its result is queued
        behind it (hm_chain_project), the update takes the projected state from device memory as its prior mean and reports
        what the two kernels found (predicted / projected state, Newton iterations) with its own results.  The covariance
        half of predict() is what it always is here (predict_take / cov_predict).  Same numbers as the three steps; taken
        when every precondition of theirs holds: fused update, prediction started ahead on the device from exactly the
        current state, springs and parameters, mask of the observation resident on the device.  chain = False switches it
        off."""
        st = self.state
        r = st.renderer
        if not (self.chain and self.fused_update and self.predict_ahead and self.newton_on_device and not self.device_predict
                and self._ahead is not None and self._worker_dev and isinstance(y_m, DeviceObservation)
                and hasattr(r, "chain_project") and isinstance(st._W, DeviceCovariance)):
            return False
        X0, bars0, l00, par0 = self._ahead
        if not (np.array_equal(X0, np.asarray(st.X, np.float64).reshape(-1)) and np.array_equal(bars0, self._bars)
                and np.array_equal(l00, st.l0[:, 0])
                and par0 == (float(self.kappa), float(self.M), float(self.deltat), int(self.maxiter), float(self.tol))):
            return False
        t0 = time.time()
        if not r.chain_project():            # nothing in flight after all
            return False
        self.orig_x = st.X.copy()
        # covariance half of predict(), as there
        taken = None
        if self._cov_armed and hasattr(r, "predict_take"):
            taken = r.predict_take(st._W, st.X, self._bars, st.l0[:, 0], self.kappa, self.deltat, self.deltat / self.M, st.eps_F)
        self._cov_armed = False
        if taken is not None:
            st.W = taken
        else:
            st.W = r.cov_predict(st._W, self._bars, self._spring_blocks(), self.deltat, self.deltat / self.M, st.eps_F, fetch=False)
            if hasattr(r, "update_prefactor"):
                r.update_prefactor(st._W)
        self._ahead = None
        self._chain_predict_s = time.time() - t0
        stats.statepredtime[0] += self._chain_predict_s
        from .renderer import ChainedPredictionFailed
        try:
            self.update(y_im, y_flow, y_m)
        except ChainedPredictionFailed:
            # the prediction's inner solve gave up on the device (never observed; test knob newton_fail): on the host,
            # then the three steps; the covariance prediction above stands
            self._newton()
            self.pred_x = st.X.copy()
            self.projectmask(y_m)
            self.update(y_im, y_flow, y_m)
            return True
        pred, proj, its, moved = r.chain_states()
        self.pred_x = pred.reshape(-1, 1)
        self.proj_x = proj.reshape(-1, 1)
        self.newton_iterations = its
        self.moved = moved
        return True

    def _ahead_inputs(self):
        st = self.state
        return (np.ascontiguousarray(self._bars, np.int32).copy(), np.ascontiguousarray(st.l0[:, 0], np.float64).copy(),
                (float(self.kappa), float(self.M), float(self.deltat), int(self.maxiter), float(self.tol)))

    def _get_worker(self):
        if self._worker is None:
            w = ctypes.c_void_p()
            _lib.check(_lib.lib().hm_ms_worker_create(ctypes.byref(w)), "hm_ms_worker_create")
            self._worker = w
            self._worker_dev = None
        dev = bool(self.newton_on_device) and hasattr(self.state.renderer, "attach_worker")
        if dev != self._worker_dev:
            if hasattr(self.state.renderer, "attach_worker"):
                self.state.renderer.attach_worker(self._worker, dev)
            self._worker_dev = dev
        return self._worker

    def _update_fused(self, y_im, y_flow, y_m):
        """hm_update_run, armed (hm_update_arm_newton) to start the next frame's state prediction on the worker thread
        the moment the state it ends with is known -- before the covariance of that state is formed and fetched."""
        r = self.state.renderer
        if self.predict_ahead and not self.device_predict and hasattr(r, "arm_newton"):
            bars, l0, par = self._ahead_inputs()
            r.arm_newton(self._get_worker(), bars, l0, *par)
            self._armed = (bars, l0, par)
            if self.cov_ahead and hasattr(r, "arm_cov"):
                r.arm_cov(self.state.eps_F)
                self._cov_armed = True
        if getattr(y_m, "next_mask", None) and hasattr(r, "arm_mask"):
            r.arm_mask(y_m.next_mask)        # the next frame's outline: queued when this update has its final state
        try:
            IteratedKalmanFilter._update_fused(self, y_im, y_flow, y_m)
        except Exception:
            self._cov_armed = False
            if self._armed is not None:          # whether the job was started is not known: wait for it if it was
                self._armed = None
                tmp = np.empty(4 * self.state.N)
                _lib.lib().hm_ms_newton_finish(self._worker, _lib.ptr(tmp), None)
            raise

    def _after_update(self):
        """The state the update ended with is what the next frame's _newton starts from: it runs on the worker thread
        (started by hm_update_run itself when armed, else here by hm_ms_newton_start) beside the end of this frame
        on the device and the caller's work between frames.  predict() takes the result if the state is still the
        one it was started from."""
        if not self.predict_ahead or (self.device_predict and self._armed is None):
            return
        st = self.state
        X = np.ascontiguousarray(st.X.reshape(-1), np.float64).copy()
        if self._armed is not None:
            bars, l0, par = self._armed
            self._armed = None
        else:
            bars, l0, par = self._ahead_inputs()
            _lib.check(_lib.lib().hm_ms_newton_start(self._get_worker(), int(st.N), int(bars.shape[0]), _lib.ptr(bars),
                                                     _lib.ptr(l0), par[0], par[1], par[2], par[3], par[4], _lib.ptr(X)),
                       "hm_ms_newton_start")
        self._ahead = (X, bars, l0, par)

    def _take_ahead(self):
        """The prediction started by _after_update, if its inputs are what _newton would use now; else None (its
        result is waited for and dropped)."""
        if self._ahead is None:
            return None
        X0, bars0, l00, par0 = self._ahead
        self._ahead = None
        st = self.state
        out = np.empty_like(X0)
        its = ctypes.c_int()
        _lib.check(_lib.lib().hm_ms_newton_finish(self._worker, _lib.ptr(out), ctypes.byref(its)), "hm_ms_newton_finish")
        same = (np.array_equal(X0, np.asarray(st.X, np.float64).reshape(-1)) and np.array_equal(bars0, self._bars)
                and np.array_equal(l00, st.l0[:, 0])
                and par0 == (float(self.kappa), float(self.M), float(self.deltat), int(self.maxiter), float(self.tol)))
        return (out, its.value) if same else None

    def close(self):
        if getattr(self, "_worker", None) is not None:
            r = getattr(getattr(self, "state", None), "renderer", None)
            if r is not None and hasattr(r, "detach_worker"):
                r.detach_worker(self._worker)
            _lib.lib().hm_ms_worker_destroy(self._worker)
            self._worker = None
            self._ahead = None
        KalmanFilter.close(self)

    def __del__(self):
        try:
            self.close()
        except Exception:               # noqa: BLE001 -- interpreter shutdown
            pass

    def _newton(self):
        """20 implicit-Euler sub-steps, each solved by Newton's method (:923-960) -- native host
        code in libhydra_mi.so (csrc/predict.cpp: block-eliminated 2N x 2N system, spring operator
        applied bar by bar, conjugate gradients)."""
        st = self.state
        ahead = self._take_ahead()
        if ahead is not None:
            self.newton_iterations = ahead[1]
            st.X = ahead[0].reshape(-1, 1)
            return
        X = np.ascontiguousarray(st.X.reshape(-1), np.float64).copy()
        bars = np.ascontiguousarray(self._bars, np.int32)
        l0 = np.ascontiguousarray(st.l0[:, 0], np.float64)
        its = ctypes.c_int()
        if self.newton_on_device and hasattr(st.renderer, "attach_worker"):
            # the same launch a prediction started ahead would have been: the same bits with and without predict_ahead
            w = self._get_worker()
            L = _lib.lib()
            _lib.check(L.hm_ms_newton_start(w, int(st.N), int(bars.shape[0]), _lib.ptr(bars), _lib.ptr(l0), float(self.kappa),
                                            float(self.M), float(self.deltat), int(self.maxiter), float(self.tol), _lib.ptr(X)),
                       "hm_ms_newton_start")
            _lib.check(L.hm_ms_newton_finish(w, _lib.ptr(X), ctypes.byref(its)), "hm_ms_newton_finish")
            self.newton_iterations = its.value
            st.X = X.reshape(-1, 1)
            return
        _lib.check(_lib.lib().hm_ms_newton(int(st.N), int(bars.shape[0]), _lib.ptr(bars), _lib.ptr(l0),
                                           float(self.kappa), float(self.M), float(self.deltat), int(self.maxiter),
                                           float(self.tol), _lib.ptr(X), ctypes.byref(its)), "hm_ms_newton")
        self.newton_iterations = its.value
        st.X = X.reshape(-1, 1)


def _solve_near_identity(S, b):
    """Solve S s = b for S = I - dt^2/M dfdy: symmetric, eigenvalues within a few percent of 1
    (dt^2 = 0.0025 times a spring Jacobian of norm ~ 2 x vertex degree), so conjugate gradients
    reach the rounding floor in about ten sparse products; a sparse direct solve takes over if
    they do not."""
    b = np.asarray(b, np.float64).reshape(-1)
    bn = np.linalg.norm(b)
    if bn == 0.0:
        return np.zeros((b.size, 1))
    x = b.copy()                                   # S ~ I: b is already a good guess
    r = b - S.dot(x)
    p = r.copy()
    rs = float(r.dot(r))
    for _ in range(60):
        if np.sqrt(rs) <= 1e-15 * bn:
            return x.reshape(-1, 1)
        Sp = S.dot(p)
        alpha = rs / float(p.dot(Sp))
        x += alpha * p
        r -= alpha * Sp
        rs_new = float(r.dot(r))
        p = r + (rs_new / rs) * p
        rs = rs_new
    if np.sqrt(rs) <= 1e-13 * bn:
        return x.reshape(-1, 1)
    return spla.spsolve(S.tocsc(), b).reshape(-1, 1)


def _fwft(A, a, W):
    """F W F^T for F = [[I, a I], [A, I]] with A sparse (2N x 2N), by blocks."""
    n2 = A.shape[0]
    W11, W12, W21, W22 = W[:n2, :n2], W[:n2, n2:], W[n2:, :n2], W[n2:, n2:]
    P11 = W11 + a * W21
    P12 = W12 + a * W22
    P21 = A.dot(np.ascontiguousarray(W11)) + W21
    P22 = A.dot(np.ascontiguousarray(W12)) + W22
    out = np.empty_like(W)
    out[:n2, :n2] = P11 + a * P12
    out[:n2, n2:] = A.dot(np.ascontiguousarray(P11.T)).T + P12
    out[n2:, :n2] = P21 + a * P22
    out[n2:, n2:] = A.dot(np.ascontiguousarray(P21.T)).T + P22
    return out


def _inside(y_m, p):
    """Per vertex: do the four pixels around it all belong to the mask (then its distance is <= 0)."""
    m = np.asarray(y_m)
    H, W = m.shape
    x0 = np.floor(p[:, 0]).astype(int)
    y0 = np.floor(p[:, 1]).astype(int)
    ok = (x0 >= 0) & (y0 >= 0) & (x0 + 1 < W) & (y0 + 1 < H)
    xs, ys = np.clip(x0, 0, W - 2), np.clip(y0, 0, H - 2)
    return ok & (m[ys, xs] > 0.5) & (m[ys, xs + 1] > 0.5) & (m[ys + 1, xs] > 0.5) & (m[ys + 1, xs + 1] > 0.5)


def _mask_distance(y_m):
    """Signed distance to the object's outline: positive outside the mask, negative inside -- the reference's fd
    (imgproc.py:195-235: -cv2.pointPolygonTest of the contour through the object's border pixels), restated in
    imgproc.outline_distance; what hm_project_mask evaluates on the device."""
    from .imgproc import outline_distance
    return outline_distance(y_m)
