"""Triangle meshes with the attributes the filter reads from a DistMesh object.

The reference builds its mesh with PyDistMesh inside ``DistMesh.createMesh``
(reference distmesh_dyn.py:42-139), a one-off initialisation that is outside the
hot path; the filter only reads ``.p`` (N x 2 vertices), ``.t`` (T x 3
triangles), ``.bars`` (I x 2 edges), ``.L`` (I edge lengths), ``.h0`` and
``.size()`` (reference kalman.py:112,139,142,172-175,396).  ``Mesh`` carries
exactly those; the builders make the fixed meshes of the BASELINE configs
(hexagonal seed grid clipped to the object, Delaunay, as DistMesh seeds its own
iteration: distmesh_dyn.py:70-85).
"""
import numpy as np
from scipy.spatial import Delaunay


class Mesh:
    def __init__(self, p, t, h0=0.0):
        self.p = np.asarray(p, np.float64)
        self.t = np.asarray(t, np.int64)
        self.h0 = h0
        self._rebuild_bars()

    def _rebuild_bars(self):
        t = self.t
        e = np.vstack((t[:, [0, 1]], t[:, [1, 2]], t[:, [0, 2]]))
        e = np.unique(np.sort(e, axis=1), axis=0)
        self.bars = e
        d = self.p[e[:, 0]] - self.p[e[:, 1]]
        self.L = np.sqrt((d * d).sum(1))

    def size(self):
        return self.p.shape[0]


def _triangulate(pts, inside, h0):
    tri = Delaunay(pts).simplices
    cen = pts[tri].mean(axis=1)
    tri = tri[inside(cen)]
    # drop vertices no triangle uses and renumber
    used = np.unique(tri)
    remap = -np.ones(len(pts), np.int64)
    remap[used] = np.arange(len(used))
    return Mesh(pts[used], remap[tri], h0)


def _hexgrid(x0, y0, x1, y1, h0):
    xs = np.arange(x0, x1 + 1e-9, h0)
    ys = np.arange(y0, y1 + 1e-9, h0 * np.sqrt(3) / 2)
    gx, gy = np.meshgrid(xs, ys)
    gx = gx.copy()
    gx[1::2] += h0 / 2
    return np.column_stack((gx.ravel(), gy.ravel()))


def box_mesh(x0, y0, x1, y1, h0):
    """Mesh of the axis-aligned box [x0,x1] x [y0,y1]: hex interior + points on the outline."""
    pts = _hexgrid(x0 + h0 / 2, y0 + h0 / 2, x1 - h0 / 2, y1 - h0 / 2, h0)
    pts = pts[(pts[:, 0] < x1 - h0 / 4) & (pts[:, 1] < y1 - h0 / 4)]
    nx = max(2, int(round((x1 - x0) / h0)) + 1)
    ny = max(2, int(round((y1 - y0) / h0)) + 1)
    ex = np.linspace(x0, x1, nx)
    ey = np.linspace(y0, y1, ny)
    border = np.vstack((np.column_stack((ex, np.full(nx, y0))), np.column_stack((ex, np.full(nx, y1))),
                        np.column_stack((np.full(ny - 2, x0), ey[1:-1])),
                        np.column_stack((np.full(ny - 2, x1), ey[1:-1]))))
    pts = np.vstack((border, pts))
    inside = lambda c: (c[:, 0] > x0) & (c[:, 0] < x1) & (c[:, 1] > y0) & (c[:, 1] < y1)
    return _triangulate(pts, inside, h0)


def disk_mesh(cx, cy, r, h0):
    """Mesh of a disk: hex interior + equally spaced points on the circle."""
    pts = _hexgrid(cx - r, cy - r, cx + r, cy + r, h0)
    d = np.hypot(pts[:, 0] - cx, pts[:, 1] - cy)
    pts = pts[d < r - 0.45 * h0]
    nb = max(6, int(round(2 * np.pi * r / h0)))
    a = 2 * np.pi * np.arange(nb) / nb
    ring = np.column_stack((cx + r * np.cos(a), cy + r * np.sin(a)))
    pts = np.vstack((ring, pts))
    inside = lambda c: np.hypot(c[:, 0] - cx, c[:, 1] - cy) < r
    return _triangulate(pts, inside, h0)


def square4_mesh(start, end):
    """The 4-vertex, 2-triangle square of the reference's unit-test fixture
    (reference test/createtestdata_kalmanfilter.py:37-41)."""
    p = np.array([[start, start], [end, start], [start, end], [end, end]], np.float64)
    t = np.array([[0, 1, 2], [1, 3, 2]], np.int64)
    return Mesh(p, t, float(end - start))


def mask_mesh(mask, h0):
    """Mesh of an arbitrary object mask: hex grid points inside the mask (eroded by h0/4) plus
    Delaunay, triangles with centroid outside dropped.  A stand-in for the DistMesh initialisation
    of the reference CLI (run_kalmanfilter.py:58-63), which is outside the hot path."""
    from scipy import ndimage
    m = np.asarray(mask) > 0
    ys, xs = np.nonzero(m)
    if len(xs) == 0:
        raise ValueError("empty object mask")
    dist = ndimage.distance_transform_edt(m)
    pts = _hexgrid(xs.min(), ys.min(), xs.max(), ys.max(), h0)
    inside_px = lambda q: ndimage.map_coordinates(dist, [q[:, 1], q[:, 0]], order=1, mode="constant", cval=0.0)
    pts = pts[inside_px(pts) > 0.25 * h0]
    # boundary points: contour pixels thinned to spacing ~h0
    edge = np.column_stack(np.nonzero(m & (dist <= 1.0)))[:, ::-1].astype(float)
    keep = []
    for q in edge:
        if all((q[0] - k[0]) ** 2 + (q[1] - k[1]) ** 2 >= (0.8 * h0) ** 2 for k in keep):
            keep.append(q)
    pts = np.vstack((np.array(keep).reshape(-1, 2), pts))
    return _triangulate(pts, lambda c: inside_px(c) > 0.0, h0)
