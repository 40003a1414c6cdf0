"""oracle/ekf_ref.py -- CPU restatement of the reference's EKF measurement path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.

What it restates (reference file:line):
  * the measurement model = four off-screen renders of the textured mesh
    (renderer.py:310-325; shaders :22-48,70-113; vertex mapping :503-524;
    texture coordinates :579) -- here a software rasteriser, see `render`;
  * the NumPy "CPU path" of the reduction kernels, which is the semantic
    reference for them: initjacobian_CPU (cuda.py:940-950), jz_CPU (:972-980),
    j_CPU (:982-1010);
  * finite-difference assembly: _jacobian (kalman.py:491-518), _hessian_sparse
    (:583-606), i.e. multi=False, sparse=True, cuda=False -- the only CPU
    configuration of the reference that is not broken;
  * Renderer.error (renderer.py:485-501) including its uint8 wrap-around;
  * the filter: state / covariances (kalman.py:178-186), spring incidence
    (:209-217), predict (:703-718 and :850-960), IEKF update (:774-831),
    non-iterated update (:745-761), compute (:676-700).

Pinning: the reference's own known-answer test for this path
(test/test_cuda.py:198-266, the "ones" case) is reproduced in
tests/test_oracle_ekf.py; its pickled fixture is missing from the reference
(.MISSING_LARGE_BLOBS), so the inputs are regenerated from
test/createtestdata_kalmanfilter.py:37-54.

Raster semantics that OpenGL leaves to the driver are fixed here, and the HIP
rasteriser follows them exactly:
  * vertex positions are snapped to 1/256 pixel (8 sub-pixel bits, what GPUs do)
    and coverage is decided by exact integer edge functions with a top-left
    tie-break, so a pixel centre on a shared edge belongs to exactly one triangle;
  * pixel (row r, col c) is covered iff its centre (c+.5, r+.5) is inside
    (NDC mapping renderer.py:509-510 + read_pixels' vertical flip cuda.py:929-938);
  * attributes are interpolated in binary32 in plane-equation form
    a0 + l1 (a1 - a0) + l2 (a2 - a0), l_i = E_i / (2 area), which returns a
    constant attribute exactly (test/test_cuda.py:256-266 relies on that);
  * the texture is sampled at the nearest texel of the INITIAL frame (vispy
    Texture2D default filter) at the interpolated initial vertex position;
  * blending is additive (gloo.set_state('additive'), renderer.py:346, stays on
    for the FBO passes): overlapping triangles add, 8-bit targets saturate;
  * both triangle orientations are drawn (face culling is off).
"""
import numpy as np

SUB = 256          # sub-pixel grid


def snap(P):
    """(N,2) float64 pixel coordinates -> int64 on the 1/256 grid (round half to even)."""
    return np.rint(np.asarray(P, np.float64) * SUB).astype(np.int64)


def _topleft(dx, dy):
    return (dy > 0) | ((dy == 0) & (dx < 0))


def render(X, N, tri, uv, tex, W, H, labels=None):
    """Render state X -> (im u8, fx f32, fy f32, m u8), each HxW.

    X: 4N doubles [x0,y0,...,vx0,vy0,...] (kalman.py:178).  fy is the render of
    -v_y (renderer.py:513).  m is 255 where any triangle covers the pixel.

    labels (one per triangle, -1 = none): also return the id image of the mask render's G and B channels,
    id = 256 G + B with the primitive's colour (255, label // 256, label % 256), label -1 -> (255, 255, 255)
    (renderer.py:610-614, shader :90-101), blended additively with 8-bit saturation like the other targets;
    uncovered pixels have id 0.  This is what cuda_multi.py:132-143 reads its `face` from.
    """
    X = np.asarray(X, np.float64).reshape(-1)
    P = snap(X[:2 * N].reshape(N, 2))
    vel = X[2 * N:].reshape(N, 2)
    ax = vel[:, 0].astype(np.float32)
    ay = (-vel[:, 1]).astype(np.float32)
    uvf = np.asarray(uv, np.float32)
    acc_im = np.zeros((H, W), np.int64)
    fx = np.zeros((H, W), np.float32)
    fy = np.zeros((H, W), np.float32)
    cnt = np.zeros((H, W), np.int64)
    if labels is not None:
        labels = np.asarray(labels).astype(np.int64)
        sum_g = np.zeros((H, W), np.int64)
        sum_b = np.zeros((H, W), np.int64)
    for tidx, t in enumerate(np.asarray(tri)):
        i0, i1, i2 = int(t[0]), int(t[1]), int(t[2])
        (x0, y0), (x1, y1), (x2, y2) = P[i0], P[i1], P[i2]
        area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
        if area == 0:
            continue
        if area < 0:
            i1, i2 = i2, i1
            (x1, y1), (x2, y2) = (x2, y2), (x1, y1)
            area = -area
        c_lo = max(0, int((min(x0, x1, x2) - 128) // SUB))
        c_hi = min(W - 1, int((max(x0, x1, x2) - 128) // SUB) + 1)
        r_lo = max(0, int((min(y0, y1, y2) - 128) // SUB))
        r_hi = min(H - 1, int((max(y0, y1, y2) - 128) // SUB) + 1)
        if c_lo > c_hi or r_lo > r_hi:
            continue
        px = (np.arange(c_lo, c_hi + 1, dtype=np.int64) * SUB + 128)[None, :]
        py = (np.arange(r_lo, r_hi + 1, dtype=np.int64) * SUB + 128)[:, None]
        e0 = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
        e1 = (x0 - x2) * (py - y2) - (y0 - y2) * (px - x2)
        e2 = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0)
        ins = ((e0 > 0) | ((e0 == 0) & _topleft(x2 - x1, y2 - y1))) & \
              ((e1 > 0) | ((e1 == 0) & _topleft(x0 - x2, y0 - y2))) & \
              ((e2 > 0) | ((e2 == 0) & _topleft(x1 - x0, y1 - y0)))
        if not ins.any():
            continue
        inv = np.float32(1.0) / np.float32(area)
        l1 = e1.astype(np.float32) * inv
        l2 = e2.astype(np.float32) * inv

        def lerp(a):        # plane-equation form: a constant attribute is reproduced exactly
            return (a[i0] + l1 * (a[i1] - a[i0])) + l2 * (a[i2] - a[i0])

        tx = np.clip(np.floor(lerp(uvf[:, 0])).astype(np.int64), 0, W - 1)
        ty = np.clip(np.floor(lerp(uvf[:, 1])).astype(np.int64), 0, H - 1)
        sl = (slice(r_lo, r_hi + 1), slice(c_lo, c_hi + 1))
        acc_im[sl] += np.where(ins, tex[ty, tx].astype(np.int64), 0)
        fx[sl] = np.where(ins, fx[sl] + lerp(ax), fx[sl])
        fy[sl] = np.where(ins, fy[sl] + lerp(ay), fy[sl])
        cnt[sl] += ins
        if labels is not None:
            lab = int(labels[tidx])
            g_, b_ = (255, 255) if lab < 0 else (lab // 256, lab % 256)
            sum_g[sl] += np.where(ins, g_, 0)
            sum_b[sl] += np.where(ins, b_, 0)
    im = np.minimum(acc_im, 255).astype(np.uint8)
    m = np.where(cnt > 0, 255, 0).astype(np.uint8)
    if labels is not None:
        return im, fx, fy, m, 256 * np.minimum(sum_g, 255) + np.minimum(sum_b, 255)
    return im, fx, fy, m


class Measurement:
    """The CPU twin of CUDAGL (cuda.py:929-1010) on top of `render`."""

    def __init__(self, N, tri, uv, tex, eps_Z, eps_J, eps_M):
        self.N = int(N)
        self.tri = np.asarray(tri, np.int64)
        self.uv = np.asarray(uv, np.float32)
        tex = np.asarray(tex)
        self.tex = tex if tex.ndim == 2 else tex[:, :, 0]     # read_pixels()[:,:,0]
        self.H, self.W = self.tex.shape
        self.eps_Z, self.eps_J, self.eps_M = float(eps_Z), float(eps_J), float(eps_M)

    def render(self, X):
        return render(X, self.N, self.tri, self.uv, self.tex, self.W, self.H)

    def initjacobian(self, X, y_im, y_flow, y_m):
        """cuda.py:940-950 (called with 255*y_m, renderer.py:679)."""
        yt, yfx, yfy, ym = self.render(X)
        self.X0 = np.array(X, np.float64).reshape(-1)
        self.ref = (yt, yfx, yfy, ym)
        self.z = (np.asarray(y_im, np.float64) - yt.astype(np.float64)) / 255.0
        self.zfx = np.asarray(y_flow[:, :, 0], np.float32) - yfx
        self.zfy = np.asarray(y_flow[:, :, 1], np.float32) + yfy
        self.zm = (255.0 * np.asarray(y_m, np.float64) - ym.astype(np.float64)) / 255.0

    def _diff(self, Xp):
        yt, yfx, yfy, ym = self.ref
        pt, pfx, pfy, pm = self.render(Xp)
        return ((pt.astype(np.float64) - yt.astype(np.float64)) / 255.0, pfx - yfx, pfy - yfy,
                (pm.astype(np.float64) - ym.astype(np.float64)) / 255.0)

    def jz(self, Xp):
        """cuda.py:972-980 -> (total, [im, fx, fy, m])."""
        d, dfx, dfy, dm = self._diff(Xp)
        c = np.array([np.sum(d * self.z) / self.eps_Z,
                      np.sum(dfx.astype(np.float64) * self.zfx.astype(np.float64)) / self.eps_J,
                      -np.sum(dfy.astype(np.float64) * self.zfy.astype(np.float64)) / self.eps_J,
                      np.sum(dm * self.zm) / self.eps_M])
        return c.sum(), c

    def j(self, deltaX, i, jdx):
        """cuda.py:982-1010: perturbations of the initjacobian state."""
        Xp = self.X0.copy(); Xp[i] += deltaX
        Xq = self.X0.copy(); Xq[jdx] += deltaX
        a = self._diff(Xp)
        b = self._diff(Xq)
        return (np.sum(a[0] * b[0]) / self.eps_Z
                + np.sum(a[1].astype(np.float64) * b[1].astype(np.float64)) / self.eps_J
                + np.sum(a[2].astype(np.float64) * b[2].astype(np.float64)) / self.eps_J
                + np.sum(a[3] * b[3]) / self.eps_M)

    # -- the label-segmented ("multi-perturbation") variants, cuda_multi.py:81-248 ------------------------
    def _face(self, labels, *states):
        """cuda_multi.py:132-143 / :215-235: the id of the first render that covers the pixel, in the order
        reference render, first perturbed render, second perturbed render; 65535 where none does."""
        face = np.full((self.H, self.W), 65535, np.int64)
        done = np.zeros((self.H, self.W), bool)
        for X in states:
            _, _, _, m, ids = render(X, self.N, self.tri, self.uv, self.tex, self.W, self.H, labels)
            take = (m == 255) & ~done
            face[take] = ids[take]
            done |= m == 255
        return face

    def jz_multi(self, Xp, labels, n_labels):
        """histogram_jz (cuda_multi.py:81-157): the per-pixel terms of jz summed per label -> (hz [n], hzc [n,4]).
        Sums in binary64 like jz (the reference's kernel adds binary32 terms with atomics)."""
        d, dfx, dfy, dm = self._diff(Xp)
        face = self._face(labels, self.X0, Xp)
        terms = (d * self.z / self.eps_Z, dfx.astype(np.float64) * self.zfx.astype(np.float64) / self.eps_J,
                 -dfy.astype(np.float64) * self.zfy.astype(np.float64) / self.eps_J, dm * self.zm / self.eps_M)
        hzc = np.zeros((n_labels, 4))
        ok = face < min(n_labels, 65535)
        for k, t in enumerate(terms):
            hzc[:, k] = np.bincount(face[ok], weights=t[ok], minlength=n_labels)[:n_labels]
        return hzc.sum(axis=1), hzc

    def j_multi(self, deltaX, ee, labels, n_labels):
        """histogram_j (cuda_multi.py:159-248) with the two renders of CUDAGL_multi.j_multi (:979-1027): X0 with
        ee[:,0] perturbed, X0 with ee[:,1] perturbed -> (h [n], nz [n], hcomp [n,4])."""
        ee = np.asarray(ee).reshape(-1, 2)
        Xp = self.X0.copy(); Xp[ee[:, 0]] += deltaX
        Xq = self.X0.copy(); Xq[ee[:, 1]] += deltaX
        a, b = self._diff(Xp), self._diff(Xq)
        face = self._face(labels, self.X0, Xp, Xq)
        terms = (a[0] * b[0] / self.eps_Z, a[1].astype(np.float64) * b[1].astype(np.float64) / self.eps_J,
                 a[2].astype(np.float64) * b[2].astype(np.float64) / self.eps_J, a[3] * b[3] / self.eps_M)
        hc = np.zeros((n_labels, 4))
        ok = face < min(n_labels, 65535)
        for k, t in enumerate(terms):
            hc[:, k] = np.bincount(face[ok], weights=t[ok], minlength=n_labels)[:n_labels]
        nz = np.bincount(face[ok], minlength=n_labels)[:n_labels].astype(np.float64)
        return hc.sum(axis=1), nz, hc

    def error(self, X, y_im, y_flow, y_m):
        """renderer.py:485-501.  y_im / y_m / the renders are uint8 there, so the
        image and mask differences and their squares wrap modulo 256 before the sum."""
        pt, pfx, pfy, pm = self.render(X)
        y_im = np.asarray(y_im)
        y_m = np.asarray(y_m)
        if y_im.dtype == np.uint8:
            d = (y_im - pt).astype(np.uint8)
            e_im = int(np.sum((d * d).astype(np.uint8), dtype=np.uint64))
        else:
            d = y_im.astype(np.float64) - pt
            e_im = float(np.sum(d * d))
        if y_m.dtype == np.uint8:
            d = ((255 * y_m.astype(np.int64)).astype(np.uint8) - pm).astype(np.uint8)
            e_m = int(np.sum((d * d).astype(np.uint8), dtype=np.uint64))
        else:
            d = 255.0 * y_m.astype(np.float64) - pm
            e_m = float(np.sum(d * d))
        dfx = np.asarray(y_flow[:, :, 0], np.float32) - pfx
        dfy = np.asarray(y_flow[:, :, 1], np.float32) + pfy
        e_fx = float(np.sum(dfx.astype(np.float64) ** 2))
        e_fy = float(np.sum(dfy.astype(np.float64) ** 2))
        return e_im, e_fx, e_fy, e_m, pfx, pfy


def adjacency(N, tri):
    """Jv (kalman.py:189-198) and the 4N x 4N pattern J = kron(ones(2,2), kron(Jv, ones(2,2))) (:202-205)."""
    Jv = np.eye(N)
    for t in tri:
        for a in t:
            for b in t:
                Jv[a, b] = 1
    J = np.kron(np.ones((2, 2)), np.kron(Jv, np.ones((2, 2))))
    return Jv, J


def jacobian(meas, X, y_im, y_flow, y_m, deltaX=2.0):
    """_jacobian, kalman.py:491-518: central differences of jz."""
    if hasattr(meas, "jacobian_all"):          # the C twin (oracle/ekf_c.py): the same loop, OpenMP-parallel
        return meas.jacobian_all(X, y_im, y_flow, y_m, deltaX)
    n = X.size
    X = np.array(X, np.float64).reshape(-1)
    meas.initjacobian(X, y_im, y_flow, y_m)
    Hz = np.zeros((n, 1))
    Hzc = np.zeros((n, 4))
    for k in range(n):
        Xp = X.copy(); Xp[k] += deltaX
        hp, cp = meas.jz(Xp)
        Xm = X.copy(); Xm[k] -= deltaX
        hm_, cm = meas.jz(Xm)
        Hz[k, 0] = (hp / deltaX - hm_ / deltaX) / 2
        Hzc[k] = (cp / deltaX - cm / deltaX) / 2
    return Hz, Hzc


def hessian_sparse(meas, X, J, deltaX=2.0):
    """_hessian_sparse, kalman.py:583-606 (initjacobian must have been called at X)."""
    if hasattr(meas, "hessian_all"):
        return meas.hessian_all(X, J, deltaX)
    n = X.size
    HTH = np.zeros((n, n))
    for i in range(n):
        for k in range(i, n):
            if J[i, k] == 1:
                HTH[i, k] = meas.j(deltaX, i, k) / deltaX / deltaX
                HTH[k, i] = HTH[i, k]
    return HTH


def jacobian_multi(meas, X, E, labels, y_im, y_flow, y_m, deltaX=2.0):
    """_jacobian_multi, kalman.py:452-489: every partition's vertices perturbed together, label-segmented sums."""
    N = meas.N
    X = np.array(X, np.float64).reshape(-1)
    n = X.size
    Hz = np.zeros((n, 1))
    Hzc = np.zeros((n, 4))
    for idx, e in enumerate(E):
        e = np.asarray(e, np.int64)
        meas.initjacobian(X, y_im, y_flow, y_m)
        lab = labels[:, idx]
        for i in range(2):
            for j in range(2):
                ee = i + 2 * N * j + 2 * e
                Xp = X.copy(); Xp[ee] += deltaX
                hz, hzc = meas.jz_multi(Xp, lab, N)
                Hz[ee, 0] = hz[e] / deltaX
                Hzc[ee, :] = hzc[e, :] / deltaX
                Xm = X.copy(); Xm[ee] -= deltaX
                hz, hzc = meas.jz_multi(Xm, lab, N)
                Hz[ee, 0] = Hz[ee, 0] - hz[e] / deltaX
                Hzc[ee, :] -= hzc[e, :] / deltaX
                Hz[ee, 0] = Hz[ee, 0] / 2
                Hzc[ee, :] = Hzc[ee, :] / 2
    return Hz, Hzc


def hessian_sparse_multi(meas, X, Q, E_hessian, E_hessian_idx, labels_hess, y_im, y_flow, y_m, deltaX=2.0):
    """_hessian_sparse_multi, kalman.py:539-581."""
    N = meas.N
    X = np.array(X, np.float64).reshape(-1)
    n = X.size
    HTH = np.zeros((n, n))
    nq = len(Q)
    for idx, e in enumerate(E_hessian):
        e = np.asarray(e, np.int64).reshape(-1, 2)
        meas.initjacobian(X, y_im, y_flow, y_m)
        lab = labels_hess[:, idx]
        for i1 in range(2):
            for j1 in range(2):
                for i2 in range(2):
                    for j2 in range(2):
                        ee = np.column_stack((2 * e[:, 0] + i1 + 2 * N * j1, 2 * e[:, 1] + i2 + 2 * N * j2))
                        h, nz, hc = meas.j_multi(deltaX, ee, lab, nq)
                        for qi in np.flatnonzero(nz > 0):
                            q = Q[qi]
                            q1 = 2 * q[0] + i1 + 2 * N * j1
                            q2 = 2 * q[1] + i2 + 2 * N * j2
                            HTH[q1, q2] = h[qi] / deltaX / deltaX
                            HTH[q2, q1] = HTH[q1, q2]
    return HTH


# ---------------------------------------------------------------------------------
# the filter (host algebra), restated for the oracle
# ---------------------------------------------------------------------------------
def initial_covariances(N, eps_F):
    """kalman.py:179-186."""
    e = np.eye(2 * N)
    z = np.zeros((2 * N, 2 * N))
    F = np.block([[e, e], [z, e]])
    Weps = eps_F * np.block([[e / 4, e / 2], [e / 2, e]])
    W = np.block([[1e-2 * e, z], [z, e]])
    return F, Weps, W


def incidence(N, bars):
    """K = kron(Kp, I2), kalman.py:209-214."""
    Kp = np.zeros((N, len(bars)))
    for idx, (i1, i2) in enumerate(bars):
        Kp[i1, idx] = 1
        Kp[i2, idx] = -1
    return np.kron(Kp, np.eye(2))


def bar_lengths(K, y):
    d = (K.T @ y.reshape(-1, 1)).reshape(-1, 2)
    return np.sqrt((d * d).sum(1))


def ms_dfdy(K, l0, y, kappa):
    """IteratedMSKalmanFilter._jacobian, kalman.py:865-902."""
    y = y.reshape(-1, 1)
    I = l0.size
    d = K.T @ y
    l = bar_lengths(K, y)
    k = np.kron(np.diag(kappa * (1 - l0 / l)), np.eye(2))
    dk = np.zeros((I, y.size))
    for i in range(I):
        KTi = K.T[2 * i:2 * i + 2, :]
        dk[i, :] = kappa * l0[i] / l[i] ** 3 * (y.T @ (KTi.T @ KTi))
    dk = np.kron(dk, np.ones((2, 1)))
    return -K @ (k @ K.T) - K @ (np.diagflat(d) @ dk)


def ms_predict(X, W, Weps, K, l0, kappa=-1.0, M=1.0, deltat=0.05, maxiter=1000, tol=1e-4):
    """IteratedMSKalmanFilter.predict, kalman.py:850-863 with _dfdx :904-912 and _newton :923-960."""
    X = np.array(X, np.float64).reshape(-1, 1)
    n2 = X.size // 2
    e = np.eye(n2)
    dfdy = ms_dfdy(K, l0, X[:n2], kappa)
    F = np.block([[e, deltat * e], [deltat * dfdy / M, e]])
    for _ in range(int(np.ceil(1 / deltat))):
        x = X.copy()
        xp = x.copy()
        xo = np.zeros_like(x)
        it = 0
        while it < maxiter and np.linalg.norm(xo - xp) > tol * np.linalg.norm(xp):
            xo = xp.copy()
            v = X[n2:]
            y = X[:n2]
            d = K.T @ y
            l = bar_lengths(K, y)
            k = np.kron(np.diag(kappa * (1 - l0 / l)), np.eye(2))
            f = K @ (k @ d)
            g = xp - x - deltat * np.vstack((v, f / M))
            dfdy_k = ms_dfdy(K, l0, y, kappa)
            G = np.block([[e, -deltat * e], [-deltat * dfdy_k / M, e]])
            xp = xp - np.linalg.inv(G) @ g
            X = xp
            it += 1
    Wn = F @ (W @ F.T) + Weps
    return X, Wn


def orientation(X, N, tri):
    """update_orientation, kalman.py:410-414."""
    ver = X[:2 * N].reshape(-1, 2)
    a = ver[tri[:, 1]] - ver[tri[:, 0]]
    b = ver[tri[:, 2]] - ver[tri[:, 0]]
    return np.sign(a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0])


def iekf_update(meas, X, W, J, tri, y_im, y_flow, y_m, nI=10, reltol=1e-4, deltaX=2.0):
    """IteratedKalmanFilter.update, kalman.py:774-831.  Returns (X, W, iterations, trace)."""
    N = X.size // 4
    X = np.array(X, np.float64).reshape(-1, 1)
    X_orig, X_old = X.copy(), X.copy()
    W = np.array(W, np.float64)
    W_old = W.copy()
    invW_orig = np.linalg.inv(W)
    eold = 0.0
    trace = []
    for it in range(nI):
        Hz, Hzc = jacobian(meas, X.reshape(-1), y_im, y_flow, y_m, deltaX)
        HTH = hessian_sparse(meas, X.reshape(-1), J, deltaX)
        W = np.linalg.inv(invW_orig + HTH)
        X = X_orig + W @ Hz - W @ (HTH @ (X_orig - X))
        if np.any(orientation(X.reshape(-1), N, tri) < 0):
            X, W = X_old, W_old
            trace.append(("reverted",))
            break
        e_im, e_fx, e_fy, e_m, _, _ = meas.error(X.reshape(-1), y_im, y_flow, y_m)
        enew = float(np.sqrt(float(e_im) ** 2 + e_fx ** 2 + e_fy ** 2 + float(e_m) ** 2))
        trace.append((e_im, e_fx, e_fy, e_m))
        if abs(enew - eold) / enew < reltol:
            break
        eold = enew
        X_old, W_old = X.copy(), W.copy()
    return X, W, len(trace), trace


def kf_update(meas, X, W, J, y_im, y_flow, y_m, deltaX=2.0):
    """KalmanFilter.update, kalman.py:745-761."""
    X = np.array(X, np.float64).reshape(-1, 1)
    Hz, Hzc = jacobian(meas, X.reshape(-1), y_im, y_flow, y_m, deltaX)
    HTH = hessian_sparse(meas, X.reshape(-1), J, deltaX)
    Wn = np.linalg.inv(np.linalg.inv(W) + HTH)
    return X + Wn @ Hz, Wn


def mask_flow(y_flow, y_m):
    """compute(): the observed flow is multiplied by the mask first, kalman.py:679-682."""
    return np.dstack((y_m * y_flow[:, :, 0], y_m * y_flow[:, :, 1])).astype(np.float32)


def pruned_object(mask, min_area=40.0):
    """The region whose outline the reference's fd describes: imgproc.py:198-228 restated without OpenCV.

    cv2.findContours(mask, RETR_TREE) gives the outer contour of every 8-connected object (level 0), the contour of every
    hole in it (4-connected background enclosed by it, level 1), of objects inside holes (level 2), ...; the frame counts
    as surrounded by background, so background that reaches the frame edge is outside, not a hole.  The reference then
    removes (:205-228) every level-0 contour whose cv2.contourArea is below the largest area (with everything inside it),
    every contour of area < 40 (with everything inside it) and every contour deeper than level 1, and builds
    fd = ddiff(outer, dunion(holes)) (:232-235) from what is left: the largest object with its holes of area >= 40 --
    smaller holes count as object (with whatever was inside them), and whatever lies inside a kept hole belongs to the
    hole (objects inside it went with 'level > 1').

    cv2.contourArea is the area of the polygon through the centres of the pixels the contour visits.  By Pick's theorem
    (area = interior lattice points + boundary lattice points / 2 - 1) for a contour that visits no pixel twice:
      outer contour of an object:  (pixels inside or on it: the object, its holes, everything in them)
                                   - (object pixels 4-adjacent to the outside or on the frame edge) / 2 - 1
      contour of a hole:           (pixels inside it: the hole's background pixels and every object nested in it)
                                   + (pixels of the enclosing object 4-adjacent to the hole) / 2 - 1
    (a hole's contour runs through the pixels of the enclosing object around it).  Those are the areas used here; where a
    contour visits a pixel twice (one-pixel-wide bridges) OpenCV's value differs by the doubly counted boundary points.
    Ties between level-0 areas: the object whose first pixel in raster order comes first (the reference would keep both
    and treat the second as a hole -- not restated).  If even the largest object has area < 40 nothing is left (the
    reference then fails with an IndexError): the result is empty.

    -> bool HxW: inside the kept outer contour and not inside a kept hole."""
    m = np.asarray(mask) > 0.5
    H, W = m.shape
    out = np.zeros((H, W), bool)
    if not m.any():
        return out
    from scipy import ndimage
    eight = np.ones((3, 3), bool)
    lab, n = ndimage.label(m, structure=eight)

    def touches4(region, of):
        """pixels of `of` with a 4-neighbour in `region`"""
        pad = np.pad(region, 1, constant_values=False)
        nb = pad[:-2, 1:-1] | pad[2:, 1:-1] | pad[1:-1, :-2] | pad[1:-1, 2:]
        return of & nb

    filled = []
    for k in range(1, n + 1):
        filled.append(ndimage.binary_fill_holes(lab == k))          # 4-connected background that cannot reach the frame edge
    best, best_area = -1, -1.0
    for k in range(1, n + 1):
        obj = lab == k
        nested = any(j != k and filled[j - 1][obj].all() for j in range(1, n + 1))
        if nested:                                                   # level >= 2: inside a hole of another object
            continue
        outside = np.pad(~filled[k - 1], 1, constant_values=True)    # ... including beyond the frame edge
        nb = outside[:-2, 1:-1] | outside[2:, 1:-1] | outside[1:-1, :-2] | outside[1:-1, 2:]
        b_outer = int((obj & nb).sum())
        area = float(filled[k - 1].sum()) - b_outer / 2.0 - 1.0
        if area > best_area:                                         # (ascending labels = ascending first pixels: ties keep the first)
            best, best_area = k, area
    if best < 0 or best_area < min_area:
        return out
    obj = lab == best
    out = filled[best - 1].copy()
    holes, nh = ndimage.label(out & ~m)                              # background inside the outer contour, 4-connected
    for c in range(1, nh + 1):
        hole = holes == c
        ring = touches4(hole, obj)
        if not ring.any():
            continue                                                 # a hole of an object nested deeper: goes with its level-1 ancestor
        inside = ndimage.binary_fill_holes(hole)                     # the hole and everything nested in it
        area = float(inside.sum()) + int(ring.sum()) / 2.0 - 1.0
        if area >= min_area:
            out &= ~inside
    return out


def outline_distance(y_m, prune=True):
    """fd of reference imgproc.py:195-235 for a mask: -cv2.pointPolygonTest(contour, p, True) with the contour that
    cv2.findContours traces through the border pixels of the object, i.e. the signed distance (negative inside) to the
    polygon through the centres of the object's border pixels.  Restated without OpenCV:

    * border pixels: object pixels with a 4-neighbour that is background or outside the frame (findContours treats the
      frame as surrounded by background);
    * sides of the polygon: the segments between 8-adjacent border pixels.  A traced contour uses a subset of them; the
      rest are diagonal short cuts on the object's side of the contour, so outside the object (where projectmask
      evaluates fd, kalman.py:728: d > 1) the nearest point of the outline is the same.  Exact minimum over all of them
      in binary64; a mask whose border pixels have no 8-adjacent pair at all (a lone pixel) is its own degenerate
      polygon (nearest border pixel); no border pixel at all (blank mask): 0;
    * sign: a point is inside the polygon through the pixel centres iff the four pixels around it are all object, or
      three are and it lies on their side of the diagonal; points off the frame are outside.

    The reference's pruning of contours (imgproc.py:205-228: the largest object and its holes of area >= 40 only) is
    applied first (pruned_object; prune=False: every border pixel of the raw mask counts -- round 3's form, kept for the
    comparison in the tests)."""
    m = np.asarray(y_m) > 0.5
    if prune:
        m = pruned_object(m)
    H, W = m.shape
    pad = np.pad(m, 1, constant_values=False)
    inner = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    border = m & ~inner
    ys, xs = np.nonzero(border)
    A, B = [], []
    bp = np.pad(border, 1, constant_values=False)
    for dx, dy in ((1, 0), (-1, 1), (0, 1), (1, 1)):            # each 8-adjacent pair once
        ok = bp[ys + 1 + dy, xs + 1 + dx]
        A.append(np.column_stack((xs[ok], ys[ok])))
        B.append(np.column_stack((xs[ok] + dx, ys[ok] + dy)))
    A = np.concatenate(A).astype(np.float64) if len(xs) else np.zeros((0, 2))
    B = np.concatenate(B).astype(np.float64) if len(xs) else np.zeros((0, 2))
    P = np.column_stack((xs, ys)).astype(np.float64)

    def inside(q):
        x, y = q[:, 0], q[:, 1]
        x0 = np.clip(np.floor(x).astype(np.int64), 0, W - 1)
        y0 = np.clip(np.floor(y).astype(np.int64), 0, H - 1)
        x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
        fx, fy = np.clip(x - x0, 0.0, 1.0), np.clip(y - y0, 0.0, 1.0)
        c00, c10, c01, c11 = m[y0, x0], m[y0, x1], m[y1, x0], m[y1, x1]
        n = c00.astype(int) + c10 + c01 + c11
        ins = n == 4
        three = n == 3
        ins = ins | (three & ((~c00 & (fx + fy >= 1.0)) | (~c11 & (fx + fy <= 1.0)) | (~c10 & (fy >= fx)) | (~c01 & (fy <= fx))))
        off = (x < 0) | (y < 0) | (x > W - 1) | (y > H - 1)
        return ins & ~off

    def fd(q):
        q = np.atleast_2d(np.asarray(q, np.float64))
        if len(P) == 0:
            return np.zeros(len(q))
        out = np.empty(len(q))
        for i, (x, y) in enumerate(q):
            if len(A):
                apx, apy = x - A[:, 0], y - A[:, 1]
                abx, aby = B[:, 0] - A[:, 0], B[:, 1] - A[:, 1]
                den = abx * abx + aby * aby
                t = np.minimum(np.maximum((apx * abx + apy * aby) / den, 0.0), 1.0)
                ex, ey = apx - t * abx, apy - t * aby
                out[i] = np.sqrt((ex * ex + ey * ey).min())
            else:
                apx, apy = x - P[:, 0], y - P[:, 1]
                out[i] = np.sqrt((apx * apx + apy * apy).min())
        return np.where(inside(q), -out, out)
    return fd


def project_mask(X, N, y_m, steps=10, ddeps=1e-1):
    """KalmanFilter.projectmask, kalman.py:724-742, with fd = outline_distance(y_m) (the reference takes it from
    findObjectThreshold(y_m, 0.5), :725).  Loop structure (10 steps, forward difference 0.1, d and the index set not
    refreshed inside the loop, displacement also added to the velocities) follows the reference."""
    X = np.array(X, np.float64).reshape(-1, 1)
    m = np.asarray(y_m) > 0.5
    if not m.any():
        return X
    fd = outline_distance(m)
    p = X[:2 * N].reshape(-1, 2).copy()
    p0 = p.copy()
    d = fd(p)
    ix = d > 1
    for _ in range(steps):
        if ix.any():
            gx = (fd(p[ix] + [ddeps, 0]) - d[ix]) / ddeps
            gy = (fd(p[ix] + [0, ddeps]) - d[ix]) / ddeps
            g2 = gx ** 2 + gy ** 2
            with np.errstate(divide="ignore", invalid="ignore"):
                s = np.where(g2 > 0, d[ix] / g2, 0.0)
            p[ix] -= (s * np.vstack((gx, gy))).T
    X[:2 * N] = p.reshape(-1, 1)
    X[2 * N:] += (p - p0).reshape(-1, 1)
    return X


def remove_flat_faces(p, t, bars, L):
    """KFState.__init__, kalman.py:148-175: drop faces with |sin| <= 0.06 and orphaned bars."""
    p = np.asarray(p, np.float32)
    a = p[t[:, 1]] - p[t[:, 0]]
    b = p[t[:, 2]] - p[t[:, 0]]
    cr = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    keep = np.abs(cr / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))) > 0.06
    t = t[keep]
    used = set()
    for v1, v2, v3 in t:
        used |= {(min(v1, v2), max(v1, v2)), (min(v2, v3), max(v2, v3)), (min(v1, v3), max(v1, v3))}
    sel = np.array([(int(b0), int(b1)) in used for b0, b1 in bars], bool)
    return t, bars[sel], L[sel]


class Tracker:
    """IteratedMSKalmanFilter driven frame by frame (kalman.py:676-700 + :834-960), oracle side."""

    def __init__(self, p, t, bars, L, im, eps_F=1e-1, eps_Z=1e-3, eps_J=1.0, eps_M=1.0, nI=10, vel=None,
                 measurement=None):
        """measurement: the class that evaluates the measurement model -- Measurement (NumPy, default) or
        its C twin oracle.ekf_c.Measurement (same numbers to rounding, usable at 512^2 and above)."""
        t, bars, L = remove_flat_faces(p, np.asarray(t), np.asarray(bars), np.asarray(L))
        self.N = len(p)
        self.tri = t
        ver = np.asarray(p, np.float32).astype(np.float64)       # KFState keeps float32 vertices (:112)
        v0 = np.zeros_like(ver) if vel is None else np.asarray(vel, np.float64).reshape(ver.shape)
        self.X = np.concatenate((ver.reshape(-1), v0.reshape(-1))).reshape(-1, 1)
        self.F, self.Weps, self.W = initial_covariances(self.N, eps_F)
        self.Jv, self.J = adjacency(self.N, t)
        self.K = incidence(self.N, bars)
        self.l0 = bar_lengths(self.K, self.X[:2 * self.N])
        self.meas = (measurement or Measurement)(self.N, t, ver, im, eps_Z, eps_J, eps_M)
        self.nI = nI

    def compute(self, y_im, y_flow, y_m, dynamics="ms"):
        if dynamics == "ms":
            self.X, self.W = ms_predict(self.X, self.W, self.Weps, self.K, self.l0)
        else:                                                     # KalmanFilter.predict, kalman.py:703-718
            self.X = self.F @ self.X
            self.W = self.F @ (self.W @ self.F.T) + self.Weps
        self.X = project_mask(self.X, self.N, y_m)
        fm = mask_flow(y_flow, y_m)
        self.X, self.W, self.niter, self.trace = iekf_update(self.meas, self.X, self.W, self.J, self.tri, y_im, fm,
                                                             y_m, nI=self.nI)
        return self.meas.error(self.X.reshape(-1), y_im, y_flow, y_m)
