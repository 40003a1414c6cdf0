"""Object segmentation and its signed distance function -- reference imgproc.py:175-248 without OpenCV.

``findObjectThreshold(img, threshold)`` returns ``(mask, ctrs, fd)`` like the reference:

* ``mask``  1 where the (gray) frame is above the threshold (cv2.threshold ... THRESH_TOZERO, then != 0,
  imgproc.py:195-197);
* ``ctrs``  the contours that survive the reference's pruning (:205-228): the outer contour of the largest
  object, and of its holes those of at least 40 square pixels -- everything else (smaller objects, small
  holes, anything nested deeper) is dropped.  Here a ``Contours`` object holding, per kept contour, the
  boundary pixels in image order (not a traced polyline; ``fd`` does not need the order);
* ``fd``    the signed distance to the object outline, negative inside, positive outside, for (x, y)
  points: ``ddiff(outer, dunion(holes))`` of ``-cv2.pointPolygonTest(contour, p, True)`` (:232-235), i.e. the
  distance to the polygon through the centres of the boundary pixels.  Evaluated exactly for points outside
  the object (the nearest point of the outline lies on a segment between two 8-adjacent boundary pixels; a
  k-d tree of segment midpoints finds the candidates) and in sign everywhere (a point is inside the polygon
  through the pixel centres iff its grid cell has four object corners, or three and the point on their
  side of the diagonal).  Inside the object the magnitude can be up to a pixel short where diagonal links
  between boundary pixels that the traced contour would not use are nearer -- the mesher and the filter only
  ever use the sign there (distmesh_dyn.py:63, 91, 122; kalman.py:728).
"""
import numpy as np
from scipy import ndimage
from scipy.spatial import cKDTree


def to_gray(img):
    img = np.asarray(img)
    if img.ndim == 3:                                     # cvtColor(BGR2GRAY), imgproc.py:190-191
        return np.rint(0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).astype(np.uint8)
    return img


class Contours:
    """The kept contours: boundary pixels (x, y) of the object's outer outline and of its holes."""

    def __init__(self, contours, levels):
        self.contours = contours
        self.levels = levels
        self.nC = len(contours)

    def traverse(self):
        for ct, level in zip(self.contours, self.levels):
            yield ct, level


_EIGHT = ndimage.generate_binary_structure(2, 2)


def _touch4(region):
    """pixels with a 4-neighbour in `region`"""
    pad = np.pad(region, 1, constant_values=False)
    return pad[:-2, 1:-1] | pad[2:, 1:-1] | pad[1:-1, :-2] | pad[1:-1, 2:]


def _object_and_holes(mask, min_area=40.0):
    """The region the reference's contour pruning keeps (imgproc.py:205-228): inside the outer contour of the level-0
    object of largest cv2.contourArea, outside its holes of area >= 40 (smaller holes count as object; objects inside
    a kept hole go with 'level > 1').  contourArea by Pick's theorem from pixel counts -- outer contour: pixels inside
    or on it minus half the object's pixels next to the outside minus 1; hole: pixels inside it plus half the
    enclosing object's pixels next to it minus 1 -- the rule csrc/project_kernels.h applies per frame on the device.
    Nothing is kept when even the largest object has area < 40 (the reference fails there)."""
    mask = np.asarray(mask, bool)
    keep = np.zeros(mask.shape, bool)
    lab, n = ndimage.label(mask, structure=_EIGHT)
    if n == 0:
        return keep
    bg, nb = ndimage.label(~mask)                          # 4-connected background regions
    edge = set(np.unique(np.concatenate((bg[0, :], bg[-1, :], bg[:, 0], bg[:, -1])))) - {0}
    outside = np.isin(bg, list(edge))                      # background that reaches the frame edge
    frame = np.zeros(mask.shape, bool)
    frame[0, :] = frame[-1, :] = frame[:, 0] = frame[:, -1] = True
    next_out = _touch4(outside) | frame
    best, best_a2 = 0, -1
    fills = {}
    for k in np.unique(lab[mask & next_out]):              # level-0 objects: they touch the outside
        obj = lab == k
        fill = ndimage.binary_fill_holes(obj)
        a2 = 2 * int(fill.sum()) - int((obj & next_out).sum()) - 2
        fills[k] = fill
        if a2 > best_a2:                                   # (ascending labels: ties keep the first in raster order)
            best, best_a2 = k, a2
    if best == 0 or best_a2 < 2 * min_area:
        return keep
    obj = lab == best
    keep = fills[best].copy()
    holes, nh = ndimage.label(keep & ~mask)
    near_obj = _touch4(obj)
    for c in range(1, nh + 1):
        hole = holes == c
        if not (hole & near_obj).any():
            continue                                        # a hole of something nested in a hole of the object
        inside = ndimage.binary_fill_holes(hole)
        if 2 * int(inside.sum()) + int((obj & _touch4(hole)).sum()) - 2 >= 2 * min_area:
            keep &= ~inside
    return keep


def _boundary(region):
    """Pixels of `region` with a 4-neighbour outside it (or the frame edge): the pixels a contour runs through."""
    pad = np.pad(region, 1, constant_values=False)
    inner = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    return region & ~inner


class SignedDistance:
    """fd(p) for an object region (bool HxW): see the module docstring."""

    def __init__(self, region):
        self.region = np.asarray(region, bool)
        self.H, self.W = self.region.shape
        b = _boundary(self.region)
        ys, xs = np.nonzero(b)
        self.empty = xs.size == 0
        if self.empty:
            return
        idx = -np.ones((self.H + 2, self.W + 2), np.int64)
        idx[ys + 1, xs + 1] = np.arange(xs.size)
        seg_a, seg_b = [], []
        for dy, dx in ((0, 1), (1, -1), (1, 0), (1, 1)):    # each 8-adjacent pair once
            j = idx[ys + 1 + dy, xs + 1 + dx]
            ok = j >= 0
            seg_a.append(np.flatnonzero(ok))
            seg_b.append(j[ok])
        a, bb = np.concatenate(seg_a), np.concatenate(seg_b)
        pts = np.column_stack((xs, ys)).astype(np.float64)
        if a.size == 0:                                     # a single pixel
            a = bb = np.array([0])
        self.A, self.B = pts[a], pts[bb]
        self.tree = cKDTree(0.5 * (self.A + self.B))
        self.k = int(min(16, self.A.shape[0]))

    def _inside(self, p):
        x, y = p[:, 0], p[:, 1]
        x0 = np.clip(np.floor(x).astype(np.int64), 0, self.W - 1)
        y0 = np.clip(np.floor(y).astype(np.int64), 0, self.H - 1)
        x1, y1 = np.minimum(x0 + 1, self.W - 1), np.minimum(y0 + 1, self.H - 1)
        fx, fy = np.clip(x - x0, 0.0, 1.0), np.clip(y - y0, 0.0, 1.0)
        c00, c10 = self.region[y0, x0], self.region[y0, x1]
        c01, c11 = self.region[y1, x0], self.region[y1, x1]
        n = c00.astype(int) + c10 + c01 + c11
        ins = n == 4
        three = n == 3                                       # on the side of the diagonal away from the missing corner
        ins |= three & ~c00 & (fx + fy >= 1.0)
        ins |= three & ~c11 & (fx + fy <= 1.0)
        ins |= three & ~c10 & (fy >= fx)
        ins |= three & ~c01 & (fy <= fx)
        off = (x < 0) | (y < 0) | (x > self.W - 1) | (y > self.H - 1)
        return ins & ~off

    def __call__(self, p):
        p = np.atleast_2d(np.asarray(p, np.float64))
        if self.empty:
            return np.full(len(p), np.inf)
        _, near = self.tree.query(p, k=self.k)
        near = near.reshape(len(p), -1)
        A, B = self.A[near], self.B[near]                    # (n, k, 2)
        ab = B - A
        ap = p[:, None, :] - A
        den = (ab * ab).sum(-1)
        t = np.where(den > 0, (ap * ab).sum(-1) / np.where(den > 0, den, 1.0), 0.0)
        t = np.clip(t, 0.0, 1.0)
        d = np.sqrt(((ap - t[..., None] * ab) ** 2).sum(-1)).min(axis=1)
        return np.where(self._inside(p), -d, d)


def outline_distance(mask):
    """fd of the per-frame form KalmanFilter.projectmask uses (reference kalman.py:725: findObjectThreshold(y_m, 0.5)[2]):
    the mask pruned as the reference prunes its contours (_object_and_holes), then the signed distance to the polygon
    through the centres of the border pixels of what is left (object pixels with a 4-neighbour that is background or off
    the frame), sides between 8-adjacent border pixels, exact minimum over every side in binary64 (no k-d tree: the same
    numbers hm_project_mask computes on the device); sign as SignedDistance; a blank mask gives 0."""
    m = _object_and_holes(np.asarray(mask) > 0.5)
    sd = SignedDistance(m)
    if sd.empty:
        return lambda p: np.zeros(len(np.atleast_2d(p)))
    b = _boundary(m)
    ys, xs = np.nonzero(b)
    bp = np.pad(b, 1, constant_values=False)
    A, B = [], []
    for dx, dy in ((1, 0), (-1, 1), (0, 1), (1, 1)):
        ok = bp[ys + 1 + dy, xs + 1 + dx]
        A.append(np.column_stack((xs[ok], ys[ok])))
        B.append(np.column_stack((xs[ok] + dx, ys[ok] + dy)))
    A, B = np.concatenate(A).astype(np.float64), np.concatenate(B).astype(np.float64)
    P = np.column_stack((xs, ys)).astype(np.float64)
    abx, aby = B[:, 0] - A[:, 0], B[:, 1] - A[:, 1]
    den = abx * abx + aby * aby

    def fd(p):
        p = np.atleast_2d(np.asarray(p, np.float64))
        d = np.empty(len(p))
        for i, (x, y) in enumerate(p):
            if len(A):
                apx, apy = x - A[:, 0], y - A[:, 1]
                t = np.minimum(np.maximum((apx * abx + apy * aby) / den, 0.0), 1.0)
                ex, ey = apx - t * abx, apy - t * aby
                d[i] = np.sqrt((ex * ex + ey * ey).min())
            else:
                apx, apy = x - P[:, 0], y - P[:, 1]
                d[i] = np.sqrt((apx * apx + apy * apy).min())
        return np.where(sd._inside(p), -d, d)
    return fd


def findObjectThreshold(img, threshold=7):
    """reference imgproc.py:175-248 -> (mask u8 HxW, Contours, fd)."""
    gray = to_gray(img)
    mask = (np.asarray(gray) > threshold).astype(np.uint8)
    region = _object_and_holes(mask > 0)
    contours, levels = [], []
    outer = ndimage.binary_fill_holes(region)
    for reg, level in ((outer, 0),):
        ys, xs = np.nonzero(_boundary(reg))
        contours.append(np.column_stack((xs, ys)))
        levels.append(level)
    holes, nh = ndimage.label(outer & ~region)
    for k in range(1, nh + 1):
        ys, xs = np.nonzero(_boundary(ndimage.binary_dilation(holes == k, structure=_EIGHT) & region))
        contours.append(np.column_stack((xs, ys)))
        levels.append(1)
    return mask, Contours(contours, levels), SignedDistance(region)
