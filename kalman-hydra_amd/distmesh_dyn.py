"""DistMesh triangulation of the object interior -- reference distmesh_dyn.py:11-222 restated on NumPy / SciPy
(the reference uses PyDistMesh's helpers ``dm.huniform``, ``ml.dense``, ``ml.unique_rows``; OpenCV only to
draw).  One-off initialisation of the tracker (reference run_kalmanfilter.py:58-63), not on the per-frame
path; the filter reads ``.p .t .bars .L .h0 .size()``.

``createMesh(ctrs, fd, frame)`` follows the reference step by step (:42-139):
 1. hexagonal seed grid of spacing h0 over the frame;
 2. points with fd(p) < geps are kept (the rejection step keeps every point: the size function is uniform,
    r0 / r0.max() = 1 > random());
 3. Delaunay whenever a point has moved more than ttol h0; triangles whose centroid is inside (fd < -geps);
 4. bars = unique edges;
 6. bar forces F = k (1.5 h0 - L), repulsive only (negative forces are zeroed); p += deltat Ftot;
 7. points outside are walked back onto the outline: ten steps p -= d grad fd / |grad fd|^2 with forward
    differences of 0.1 px, d and the set of outside points not refreshed inside the loop;
 8. stop when every interior point moves less than dptol h0 (or after maxiter rounds).
"""
import pickle

import numpy as np
from scipy.spatial import Delaunay


def _unique_rows(a):
    return np.unique(a, axis=0)


def _bar_forces(bars, Fvec, N):
    """ml.dense(bars[:, [0,0,1,1]], [[0,1,0,1]], [Fvec, -Fvec], (N, 2)): the force of every bar added to its first
    vertex and subtracted from its second, in bar order."""
    Ftot = np.zeros((N, 2))
    np.add.at(Ftot, bars[:, 0], Fvec)
    np.add.at(Ftot, bars[:, 1], -Fvec)
    return Ftot


class DistMesh:
    def __init__(self, frame, h0=35, dptol=0.01):
        self.bars = None
        self.frame = frame
        self.dptol = dptol
        self.h0 = h0
        self.N = 0
        nx, ny = np.shape(frame)[0:2]
        self.nx, self.ny = nx, ny
        self.bbox = (0, 0, ny, nx)
        self.ttol = .1
        self.Fscale = 1.2
        self.deltat = .2
        self.geps = .001 * h0
        self.deps = np.sqrt(np.finfo(np.double).eps) * h0
        self.densityctrlfreq = 1
        self.k = 1.5
        self.maxiter = 500
        self.F = lambda L: -self.k * (L - h0)
        self.iterations = 0

    def _triangulate(self, p, fd):
        t = Delaunay(p).simplices
        pmid = p[t].sum(1) / 3
        t = t[fd(pmid) < -self.geps]
        bars = np.vstack((t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]))
        bars.sort(axis=1)
        return t, _unique_rows(bars)

    def _project(self, p, fd):
        """step 7: outside points back to the outline (d and ix are those before the first step, :122-129)"""
        d = fd(p)
        ix = d > 0
        ddeps = 1e-1
        for _ in range(10):
            if ix.any():
                gx = (fd(p[ix] + [ddeps, 0]) - d[ix]) / ddeps
                gy = (fd(p[ix] + [0, ddeps]) - d[ix]) / ddeps
                g2 = gx ** 2 + gy ** 2
                with np.errstate(divide="ignore", invalid="ignore"):
                    step = np.where(g2 > 0, d[ix] / g2, 0.0)
                p[ix] -= (step * np.vstack((gx, gy))).T
        return d

    def createMesh(self, ctrs, fd, frame, plot=False):
        self.frame = frame
        xmin, ymin, xmax, ymax = self.bbox
        h0 = self.h0
        x, y = np.mgrid[xmin:(xmax + h0):h0, ymin:(ymax + h0 * np.sqrt(3) / 2):h0 * np.sqrt(3) / 2]
        x[:, 1::2] += h0 / 2
        p = np.vstack((x.flat, y.flat)).T
        p = p[np.where(fd(p) < self.geps)]
        N = p.shape[0]
        if N < 3:
            raise ValueError("DistMesh.createMesh: fewer than 3 seed points inside the object (h0 = %g too large?)" % h0)
        self.N = N
        pold = np.full_like(p, np.inf)
        t = bars = L = None
        count = 0
        while count < self.maxiter:
            count += 1
            if (np.sqrt(((p - pold) ** 2).sum(1)) / h0).max() > self.ttol:
                pold = p.copy()
                t, bars = self._triangulate(p, fd)
            barvec = p[bars[:, 0]] - p[bars[:, 1]]
            L = np.sqrt((barvec ** 2).sum(1))
            L0 = 1.5 * h0 * np.ones_like(L)
            F = self.k * (L0 - L)
            F[F < 0] = 0
            Fvec = F[:, None] / L[:, None].dot([[1, 1]]) * barvec
            Ftot = _bar_forces(bars, Fvec, N)
            p += self.deltat * Ftot
            d = self._project(p, fd)
            inner = d < -self.geps
            if not inner.any() or (np.sqrt((self.deltat * Ftot[inner] ** 2).sum(1)) / h0).max() < self.dptol:
                break
        self.iterations = count
        self.p, self.t, self.bars, self.L = p, t, bars, L
        self._drop_unused()

    def _drop_unused(self):
        """Vertices no interior triangle uses would be rendered by nothing and carry no measurement: remove them
        (the reference keeps them in p; KFState.__init__ then still counts them in N -- with the uniform seed grid
        of a compact object there are none)."""
        used = np.unique(self.t)
        if used.size == self.p.shape[0]:
            return
        remap = -np.ones(self.p.shape[0], np.int64)
        remap[used] = np.arange(used.size)
        self.p = self.p[used]
        self.t = remap[self.t]
        self.bars = remap[self.bars]
        keep = (self.bars >= 0).all(axis=1)
        self.bars, self.L = self.bars[keep], self.L[keep]
        self.N = self.p.shape[0]

    def updateMesh(self, ctrs, fd, frame_orig, pfix=None, n_iter=20):
        """reference distmesh_dyn.py:141-200: relax the existing mesh towards a new outline (fixed topology)."""
        deltat = 0.1
        N = self.p.shape[0]
        pold = np.full_like(self.p, np.inf)
        bars, L = self.bars, self.L
        for _ in range(n_iter):
            if (np.sqrt(((self.p - pold) ** 2).sum(1)) / self.h0).max() > self.ttol:
                pold = self.p.copy()
                pmid = self.p[self.t].sum(1) / 3
                self.t = self.t[fd(pmid) < -self.geps]
                bars = np.vstack((self.t[:, [0, 1]], self.t[:, [1, 2]], self.t[:, [2, 0]]))
                bars.sort(axis=1)
                bars = _unique_rows(bars)
            barvec = self.p[bars[:, 0]] - self.p[bars[:, 1]]
            L = np.sqrt((barvec ** 2).sum(1))
            F = self.F(L)
            Fvec = F[:, None] / L[:, None].dot([[1, 1]]) * barvec
            self.p += deltat * _bar_forces(bars, Fvec, N)
            self._project(self.p, fd)
        self.bars, self.L = bars, L

    def size(self):
        return self.N

    _FIELDS = ("N", "bars", "frame", "dptol", "nx", "ny", "bbox", "ttol", "Fscale", "deltat", "geps", "deps",
               "densityctrlfreq", "k", "maxiter", "p", "t", "bars", "L")

    def save(self, fn_out):
        """the 19 pickled fields of reference distmesh_dyn.py:205-211, in its order"""
        with open(fn_out, "wb") as f:
            for name in self._FIELDS:
                pickle.dump(getattr(self, name), f, protocol=pickle.HIGHEST_PROTOCOL)

    def load(self, fn_in):
        with open(fn_in, "rb") as f:
            for name in self._FIELDS:
                setattr(self, name, pickle.load(f))
