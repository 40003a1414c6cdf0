"""Masks shared by the CPU and GPU tests of the reference's contour pruning (imgproc.py:198-228)."""
import numpy as np


def blobs(rng, H, W, k):
    """a random mask of k soft blobs with pinholes and specks"""
    yy, xx = np.mgrid[:H, :W]
    f = np.zeros((H, W))
    for _ in range(k):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(4, 0.3 * min(H, W))
        f += np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)) * rng.uniform(0.5, 1.5)
    m = f > rng.uniform(0.5, 0.9)
    m ^= rng.random((H, W)) < 0.01                                  # pinholes in the objects, specks outside
    for _ in range(3):                                              # a few larger holes, some with something inside
        cx, cy, a, b = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(3, 10)), int(rng.integers(3, 10))
        m[max(0, cy - b):cy + b, max(0, cx - a):cx + a] = False
        if rng.random() < 0.5:
            m[cy:cy + 2, cx:cx + 2] = True
    return m
