"""Synthetic inputs for the BASELINE configs (input generators only).

* ``test_data``     -- the translating two-tone square of the reference
                       (reference synth.py:10-42), including its quirk that the
                       box is flipped vertically but the flow rectangle is not.
* ``flowfields``    -- the analytic velocity fields of the reference
                       (reference synthetic/flowfields.py:1-7), defined on a
                       600-px canvas; ``scaled_field`` maps them to any size.
* ``noise_texture`` / ``warp_pair`` / ``disk_video`` -- the smooth random
                       texture, warped frame pairs and advected-disk videos the
                       1-GPU and 8-GPU bench configurations run on.
Nothing here is on the timed path.
"""
import numpy as np
from scipy import ndimage


def test_data(nx, ny):
    """10 frames of a 128/255 two-tone box moving up-left by 3 px per frame.

    Returns (video[nx,ny,10] u8, flow[nx,ny,2,10] f32).  reference synth.py:10-42.
    """
    nframes, speed = 10, 3
    start, end = nx // 3, 2 * nx // 3
    side = end - start
    rows, cols = np.mgrid[0:nx, 0:ny]
    inside = (rows >= start) & (rows < end) & (cols >= start) & (cols < end)
    box = np.where(inside, np.where(rows > cols, 128.0, 255.0), 0.0)
    box = box[::-1, :]                      # flipud of the image only (synth.py:29)
    video = np.zeros((nx, ny, nframes), np.uint8)
    flow = np.zeros((nx, ny, 2, nframes), np.float32)
    for k in range(nframes):
        s = start - speed * k
        flow[s:s + side, s:s + side, :, k] = -speed
        shifted = box[speed * k:, speed * k:]
        video[:shifted.shape[0], :shifted.shape[1], k] = shifted
    return video, flow


# reference synthetic/flowfields.py:1-7 -- pt = (x, y) on a 600-px canvas, returns (vx, vy)
flowfields = {
    "translate_leftup": lambda pt, t: (-1.5 + 0.0 * pt[0], -1.5 + 0.0 * pt[1]),
    "translate_leftup_stretch": lambda pt, t: (-1.0 + pt[0] / 300.0, -1.0 + pt[1] / 300.0),
    "rotate": lambda pt, t: (-(pt[1] - 300.0) / 25.0, (pt[0] - 300.0) / 25.0),
    "warp": lambda pt, t: (-(pt[0] / 300.0 - 1.0) * pt[0] * (pt[0] / 800.0 - 1.0) / 250.0,
                           (pt[1] / 300.0 - 1.0) * pt[1] * (pt[1] / 800.0 - 1.0) / 300.0),
}


def scaled_field(name, n):
    """The named field with coordinates and velocities scaled from 600 px to n px."""
    f = flowfields[name]
    k = 600.0 / float(n)

    def field(x, y):
        vx, vy = f((x * k, y * k), 0)
        return vx / k, vy / k
    return field


def noise_texture(n, seed=0, sigma=3.0):
    """Gaussian-filtered uniform noise stretched to 0..255 (float64, n x n)."""
    rng = np.random.default_rng(seed)
    img = ndimage.gaussian_filter(rng.random((n, n)), sigma, mode="reflect")
    img = (img - img.min()) / (img.max() - img.min())
    return 255.0 * img


def _backward_map(field, n, iters=30):
    """q(p) with q + field(q) = p, by fixed-point iteration (fields are smooth and small)."""
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    qx, qy = xx.copy(), yy.copy()
    for _ in range(iters):
        vx, vy = field(qx, qy)
        qx, qy = xx - vx, yy - vy
    return qx, qy


def warp_pair(n, name="translate_leftup_stretch", seed=0):
    """(frame0 u8, frame1 u8, true_u f32, true_v f32): frame1(x+u, y+v) = frame0(x, y)."""
    tex = noise_texture(n, seed)
    field = scaled_field(name, n)
    qx, qy = _backward_map(field, n)
    f1 = ndimage.map_coordinates(tex, [qy, qx], order=3, mode="reflect")
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    tu, tv = field(xx, yy)
    tu = np.broadcast_to(tu, (n, n)).astype(np.float32)
    tv = np.broadcast_to(tv, (n, n)).astype(np.float32)
    to8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return to8(tex), to8(f1), tu, tv


def disk_video(n, nframes=10, name="translate_leftup", seed=0, radius=0.33):
    """Textured disk on black, advected frame to frame by the named field.

    Returns (video[nframes,n,n] u8, masks[nframes,n,n] u8, centre, radius_px).
    Frame k+1 is frame k pulled back through the field, so the Brox flow of
    (k, k+1) approximates the field inside the object.
    """
    tex = noise_texture(n, seed)
    # keep the object clearly above the background threshold
    tex = 40.0 + tex * (215.0 / 255.0)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    c = (n - 1) / 2.0
    r = radius * n
    obj = np.where((xx - c) ** 2 + (yy - c) ** 2 <= r * r, tex, 0.0)
    field = scaled_field(name, n)
    qx, qy = _backward_map(field, n)
    frames = [obj]
    for _ in range(nframes - 1):
        frames.append(ndimage.map_coordinates(frames[-1], [qy, qx], order=1, mode="constant", cval=0.0))
    video = np.clip(np.rint(np.stack(frames)), 0, 255).astype(np.uint8)
    masks = (video > 20).astype(np.uint8)
    video = video * masks
    return video, masks, (c, c), r
