/*
 * oracle/brox_ref.c -- CPU restatement of the Brox variational optical flow.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kalman-hydra_amd/ may import, link
 * or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED against the reference's own arithmetic.  The reference does
 * not contain the solver: its only compute line is
 *     cv::cuda::BroxOpticalFlow::create(alpha,gamma,scale,inner,outer,solver)
 *         ->calc(frame0f, frame1f, flow)
 * (reference src/optical_flow_ext.cpp:310,317), i.e. OpenCV's opencv_contrib
 * `cudaoptflow` module (NCVBroxOpticalFlow), un-vendored and un-pinned
 * (reference makefile:2-3 uses `pkg-config opencv`).  OpenCV is absent from the
 * build image, and the reference holds no stored flow vectors
 * (test_flow.py:120-138 compares against analytic fields only).  This file
 * therefore restates the PUBLISHED algorithm -- Brox, Bruhn, Papenberg,
 * Weickert, "High accuracy optical flow estimation based on a theory for
 * warping", ECCV 2004 (reference README.md:48) -- in the variant the build
 * contract fixes (BASELINE.json north_star): Gaussian pyramid, bilinear warp,
 * red-black SOR fixed-point solve, with the parameter meaning the reference
 * documents at src/optical_flow_ext.cpp:300-308 and its defaults
 * (alpha .197, gamma 50, scale .8, inner 10, outer 77, solver 10; :453-488).
 * The call-site contract that IS pinned by the reference and is followed here:
 *   - inputs are 8-bit gray frames converted with x * (1/255) (:314-315);
 *   - output is two row-major f32 planes flowx, flowy of the frame size
 *     (:322-328), such that frame1(x+u, y+v) ~ frame0(x, y).
 *
 * All arithmetic is IEEE binary32 with a fixed operation order, no fused
 * multiply-add (build with -ffp-contract=off) and correctly rounded sqrt and
 * division, so that the HIP kernels -- built the same way -- can be compared
 * with this file bit for bit, not just to a tolerance.
 *
 * Algorithm (each step is one function below):
 *   pyramid   level k has size ceil(W*s^k) x ceil(H*s^k); levels are added
 *             while the previous one is larger than 15 px on both sides and
 *             fewer than `outer` levels exist.  Level k is level k-1 blurred by
 *             a separable Gaussian (sigma = 0.6*sqrt(1/s^2-1), radius
 *             ceil(3 sigma), mirrored border) and resampled bilinearly.
 *   deriv     5-tap (1,-8,0,8,-1)/12 with mirrored border: Ix0,Iy0 of frame 0;
 *             Ix,Iy,Ixx,Ixy,Iyy of frame 1.
 *   warp      bilinear sampling of frame 1 and its five derivative images at
 *             (x+u, y+v); forms Iz, Ixz, Iyz.  Where (x+u, y+v) falls outside
 *             frame 1 all eight warped fields are zero (no data term there);
 *             within 2 px of a border (pixel or any of its bilinear taps) the
 *             five gradient-constancy fields are zero, because the mirrored
 *             5-tap stencils of the two frames disagree there.
 *   prepare   robust data / gradient / smoothness weights
 *             psi'(s2) = 1/(2 sqrt(s2 + 1e-6)) at the current (du,dv); writes
 *             the 2x2-block linear system: num_u, num_v, a12, inv_den_u,
 *             inv_den_v and the edge diffusivities sx, sy.
 *   sor       `solver` red-black SOR iterations (omega 1.99; within a pixel du is
 *             relaxed first and dv then uses the new du) on (du,dv).
 *   per level `inner` x (prepare + sor); u += du, v += dv; bilinear
 *             prolongation to the next finer level, scaled by the size ratio.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BROX_EPS2 1e-6f
static float BROX_OMEGA = 1.99f;
void brox_ref_set_omega(float w) { BROX_OMEGA = w; }
#define BROX_MAX_LEVELS 128
#define BROX_MAX_RADIUS 16

static int mirror(int i, int n)
{
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        else i = 2 * n - i - 1;
    }
    return i;
}

static int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

/* threads used by the OpenMP loops (rows of one colour are independent, so the
 * result does not depend on the thread count) */
void brox_ref_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

/* ---- pyramid geometry ------------------------------------------------- */
int brox_ref_levels(int W, int H, float scale, int outer, int *ws, int *hs)
{
    int n = 1;
    float sc = 1.0f;
    ws[0] = W; hs[0] = H;
    while (ws[n - 1] > 15 && hs[n - 1] > 15 && n < outer && n < BROX_MAX_LEVELS) {
        sc = sc * scale;
        int w = (int)ceilf((float)W * sc);
        int h = (int)ceilf((float)H * sc);
        if (w < 1) w = 1;
        if (h < 1) h = 1;
        ws[n] = w; hs[n] = h;
        n++;
    }
    return n;
}

/* Gaussian taps shared with the product's host code by FORMULA, not by code:
 * sigma = 0.6*sqrt(1/s^2 - 1), R = ceil(3 sigma) (>=1), g[i] = exp(-i^2/(2 sigma^2))
 * normalised in double, rounded once to float. */
int brox_ref_gauss(float scale, float *g)
{
    double sigma = 0.6 * sqrt(1.0 / ((double)scale * (double)scale) - 1.0);
    int R = (int)ceil(3.0 * sigma);
    if (R < 1) R = 1;
    if (R > BROX_MAX_RADIUS) R = BROX_MAX_RADIUS;
    double tmp[2 * BROX_MAX_RADIUS + 1], sum = 0.0;
    for (int i = -R; i <= R; i++) {
        tmp[i + R] = exp(-(double)(i * i) / (2.0 * sigma * sigma));
        sum += tmp[i + R];
    }
    for (int i = 0; i <= 2 * R; i++) g[i] = (float)(tmp[i] / sum);
    return R;
}

/* ---- image operators --------------------------------------------------- */
static void blur(const float *src, float *tmp, float *dst, int w, int h, const float *g, int R)
{
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0.0f;
            for (int i = -R; i <= R; i++)
                acc = acc + g[i + R] * src[y * w + mirror(x + i, w)];
            tmp[y * w + x] = acc;
        }
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0.0f;
            for (int i = -R; i <= R; i++)
                acc = acc + g[i + R] * tmp[mirror(y + i, h) * w + x];
            dst[y * w + x] = acc;
        }
}

/* bilinear sample with coordinates clamped to the image */
static float bilin(const float *img, int w, int h, float px, float py)
{
    if (px < 0.0f) px = 0.0f;
    if (py < 0.0f) py = 0.0f;
    if (px > (float)(w - 1)) px = (float)(w - 1);
    if (py > (float)(h - 1)) py = (float)(h - 1);
    float fx0 = floorf(px), fy0 = floorf(py);
    int x0 = (int)fx0, y0 = (int)fy0;
    int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
    float ax = px - fx0, ay = py - fy0;
    float a = img[y0 * w + x0], b = img[y0 * w + x1];
    float c = img[y1 * w + x0], d = img[y1 * w + x1];
    float top = (1.0f - ax) * a + ax * b;
    float bot = (1.0f - ax) * c + ax * d;
    return (1.0f - ay) * top + ay * bot;
}

/* resample src (ws x hs) onto dst (wd x hd), value scaled by `mul` */
static void resample(const float *src, int ws, int hs, float *dst, int wd, int hd, float mul)
{
    float rx = (float)ws / (float)wd, ry = (float)hs / (float)hd;
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < hd; y++)
        for (int x = 0; x < wd; x++) {
            float sx = ((float)x + 0.5f) * rx - 0.5f;
            float sy = ((float)y + 0.5f) * ry - 0.5f;
            dst[y * wd + x] = bilin(src, ws, hs, sx, sy) * mul;
        }
}

static float d5(float m2, float m1, float p1, float p2)
{
    return (8.0f * (p1 - m1) - (p2 - m2)) * (1.0f / 12.0f);
}

static void deriv_x(const float *f, float *d, int w, int h)
{
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            d[y * w + x] = d5(f[y * w + mirror(x - 2, w)], f[y * w + mirror(x - 1, w)],
                              f[y * w + mirror(x + 1, w)], f[y * w + mirror(x + 2, w)]);
}

static void deriv_y(const float *f, float *d, int w, int h)
{
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            d[y * w + x] = d5(f[mirror(y - 2, h) * w + x], f[mirror(y - 1, h) * w + x],
                              f[mirror(y + 1, h) * w + x], f[mirror(y + 2, h) * w + x]);
}

/* ---- one level --------------------------------------------------------- */
typedef struct {
    float *Iz, *Ix, *Iy, *Ixz, *Iyz, *Ixx, *Ixy, *Iyy;          /* warped data */
    float *nu, *nv, *a12, *idu, *idv, *sx, *sy;                  /* linear system */
} level_ws;

static float psi_half_rsqrt(float s2) { return 0.5f / sqrtf(s2 + BROX_EPS2); }

void brox_ref_prepare(const float *u, const float *v, const float *du, const float *dv,
                      const float *Iz, const float *Ix, const float *Iy,
                      const float *Ixz, const float *Iyz,
                      const float *Ixx, const float *Ixy, const float *Iyy,
                      float *nu, float *nv, float *a12, float *idu, float *idv,
                      float *sx, float *sy, int w, int h, float alpha, float gamma)
{
#define UU(X, Y) (u[(Y) * w + (X)] + du[(Y) * w + (X)])
#define VV(X, Y) (v[(Y) * w + (X)] + dv[(Y) * w + (X)])
    /* edge diffusivities: sx on the edge (x,y)-(x+1,y), sy on (x,y)-(x,y+1) */
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int ym = clampi(y - 1, 0, h - 1), yp = clampi(y + 1, 0, h - 1);
            int xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
            if (x + 1 < w) {
                float ux = UU(x + 1, y) - UU(x, y);
                float vx = VV(x + 1, y) - VV(x, y);
                float uy = 0.25f * ((UU(x, yp) - UU(x, ym)) + (UU(x + 1, yp) - UU(x + 1, ym)));
                float vy = 0.25f * ((VV(x, yp) - VV(x, ym)) + (VV(x + 1, yp) - VV(x + 1, ym)));
                sx[y * w + x] = alpha * psi_half_rsqrt(((ux * ux + uy * uy) + vx * vx) + vy * vy);
            } else sx[y * w + x] = 0.0f;
            if (y + 1 < h) {
                float uy = UU(x, y + 1) - UU(x, y);
                float vy = VV(x, y + 1) - VV(x, y);
                float ux = 0.25f * ((UU(xp, y) - UU(xm, y)) + (UU(xp, y + 1) - UU(xm, y + 1)));
                float vx = 0.25f * ((VV(xp, y) - VV(xm, y)) + (VV(xp, y + 1) - VV(xm, y + 1)));
                sy[y * w + x] = alpha * psi_half_rsqrt(((ux * ux + uy * uy) + vx * vx) + vy * vy);
            } else sy[y * w + x] = 0.0f;
        }
#undef UU
#undef VV
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int p = y * w + x;
            float ddu = du[p], ddv = dv[p];
            float ix = Ix[p], iy = Iy[p], iz = Iz[p];
            float ixx = Ixx[p], ixy = Ixy[p], iyy = Iyy[p], ixz = Ixz[p], iyz = Iyz[p];
            float q0 = (iz + ix * ddu) + iy * ddv;
            float pd = psi_half_rsqrt(q0 * q0);
            float q1 = (ixz + ixx * ddu) + ixy * ddv;
            float q2 = (iyz + ixy * ddu) + iyy * ddv;
            float pg = gamma * psi_half_rsqrt(q1 * q1 + q2 * q2);
            float A11 = pd * (ix * ix) + pg * (ixx * ixx + ixy * ixy);
            float A12 = pd * (ix * iy) + pg * (ixx * ixy + ixy * iyy);
            float A22 = pd * (iy * iy) + pg * (ixy * ixy + iyy * iyy);
            float b1 = -(pd * (ix * iz) + pg * (ixx * ixz + ixy * iyz));
            float b2 = -(pd * (iy * iz) + pg * (ixy * ixz + iyy * iyz));
            float sl = x > 0 ? sx[p - 1] : 0.0f, sr = sx[p];
            float st = y > 0 ? sy[p - w] : 0.0f, sb = sy[p];
            int pl = x > 0 ? p - 1 : p, pr = x + 1 < w ? p + 1 : p;
            int pt = y > 0 ? p - w : p, pb = y + 1 < h ? p + w : p;
            float uc = u[p], vc = v[p];
            float su = ((sl * (u[pl] - uc) + sr * (u[pr] - uc)) + st * (u[pt] - uc)) + sb * (u[pb] - uc);
            float sv = ((sl * (v[pl] - vc) + sr * (v[pr] - vc)) + st * (v[pt] - vc)) + sb * (v[pb] - vc);
            float ssum = ((sl + sr) + st) + sb;
            nu[p] = b1 + su;
            nv[p] = b2 + sv;
            a12[p] = A12;
            idu[p] = 1.0f / (A11 + ssum);
            idv[p] = 1.0f / (A22 + ssum);
        }
}

/* one full red+black iteration; colour 0 = (x+y) even first */
void brox_ref_sor(float *du, float *dv, const float *nu, const float *nv, const float *a12,
                  const float *idu, const float *idv, const float *sx, const float *sy,
                  int w, int h, int iters)
{
    const float om = BROX_OMEGA, om1 = 1.0f - BROX_OMEGA;
    for (int it = 0; it < iters; it++)
        for (int col = 0; col < 2; col++)
            _Pragma("omp parallel for schedule(static)")
            for (int y = 0; y < h; y++)
                for (int x = (y + col) & 1; x < w; x += 2) {
                    int p = y * w + x;
                    float sl = x > 0 ? sx[p - 1] : 0.0f, sr = sx[p];
                    float st = y > 0 ? sy[p - w] : 0.0f, sb = sy[p];
                    int pl = x > 0 ? p - 1 : p, pr = x + 1 < w ? p + 1 : p;
                    int pt = y > 0 ? p - w : p, pb = y + 1 < h ? p + w : p;
                    float su = ((sl * du[pl] + sr * du[pr]) + st * du[pt]) + sb * du[pb];
                    float sv = ((sl * dv[pl] + sr * dv[pr]) + st * dv[pt]) + sb * dv[pb];
                    float dun = om1 * du[p] + om * (((nu[p] - a12[p] * dv[p]) + su) * idu[p]);
                    float dvn = om1 * dv[p] + om * (((nv[p] - a12[p] * dun) + sv) * idv[p]);
                    du[p] = dun;
                    dv[p] = dvn;
                }
}

void brox_ref_warp(const float *I0, const float *Ix0, const float *Iy0,
                   const float *I1, const float *I1x, const float *I1y,
                   const float *I1xx, const float *I1xy, const float *I1yy,
                   const float *u, const float *v, int w, int h,
                   float *Iz, float *Ix, float *Iy, float *Ixz, float *Iyz,
                   float *Ixx, float *Ixy, float *Iyy)
{
    _Pragma("omp parallel for schedule(static)")
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int p = y * w + x;
            float px = (float)x + u[p], py = (float)y + v[p];
            /* a pixel whose correspondence leaves frame 1 carries no data term */
            if (px < 0.0f || py < 0.0f || px > (float)(w - 1) || py > (float)(h - 1)) {
                Iz[p] = 0.0f; Ix[p] = 0.0f; Iy[p] = 0.0f; Ixz[p] = 0.0f; Iyz[p] = 0.0f;
                Ixx[p] = 0.0f; Ixy[p] = 0.0f; Iyy[p] = 0.0f;
                continue;
            }
            float i1 = bilin(I1, w, h, px, py);
            float ix = bilin(I1x, w, h, px, py), iy = bilin(I1y, w, h, px, py);
            Iz[p] = i1 - I0[p];
            Ix[p] = ix; Iy[p] = iy;
            /* gradient constancy only where neither 5-tap stencil (at the pixel
             * in frame 0, at the four bilinear taps in frame 1) reaches across
             * the mirrored border; elsewhere the term is switched off */
            if (x < 2 || y < 2 || x > w - 3 || y > h - 3 ||
                px < 2.0f || py < 2.0f || px > (float)(w - 4) || py > (float)(h - 4)) {
                Ixz[p] = 0.0f; Iyz[p] = 0.0f; Ixx[p] = 0.0f; Ixy[p] = 0.0f; Iyy[p] = 0.0f;
                continue;
            }
            Ixz[p] = ix - Ix0[p]; Iyz[p] = iy - Iy0[p];
            Ixx[p] = bilin(I1xx, w, h, px, py);
            Ixy[p] = bilin(I1xy, w, h, px, py);
            Iyy[p] = bilin(I1yy, w, h, px, py);
        }
}

/* ---- driver ------------------------------------------------------------ */
/* Returns 0 on success.  f0/f1: W*H float in [0,1]; u/v: W*H float out. */
int brox_ref_calc_f32(const float *f0, const float *f1, int W, int H,
                      float alpha, float gamma, float scale, int inner, int outer, int solver,
                      float *u_out, float *v_out)
{
    int ws[BROX_MAX_LEVELS], hs[BROX_MAX_LEVELS];
    if (W < 1 || H < 1 || !(scale > 0.0f && scale < 1.0f) || outer < 1) return -1;
    int nl = brox_ref_levels(W, H, scale, outer, ws, hs);
    float g[2 * BROX_MAX_RADIUS + 1];
    int R = brox_ref_gauss(scale, g);
    size_t n0 = (size_t)W * H;

    float **P0 = calloc(nl, sizeof(float *)), **P1 = calloc(nl, sizeof(float *));
    float *tmp = malloc(n0 * sizeof(float)), *tmp2 = malloc(n0 * sizeof(float));
    for (int k = 0; k < nl; k++) {
        P0[k] = malloc((size_t)ws[k] * hs[k] * sizeof(float));
        P1[k] = malloc((size_t)ws[k] * hs[k] * sizeof(float));
    }
    memcpy(P0[0], f0, n0 * sizeof(float));
    memcpy(P1[0], f1, n0 * sizeof(float));
    for (int k = 1; k < nl; k++) {
        blur(P0[k - 1], tmp, tmp2, ws[k - 1], hs[k - 1], g, R);
        resample(tmp2, ws[k - 1], hs[k - 1], P0[k], ws[k], hs[k], 1.0f);
        blur(P1[k - 1], tmp, tmp2, ws[k - 1], hs[k - 1], g, R);
        resample(tmp2, ws[k - 1], hs[k - 1], P1[k], ws[k], hs[k], 1.0f);
    }

    float *buf[27];
    for (int i = 0; i < 27; i++) buf[i] = malloc(n0 * sizeof(float));
    float *Ix0 = buf[0], *Iy0 = buf[1], *I1x = buf[2], *I1y = buf[3], *I1xx = buf[4], *I1xy = buf[5],
          *I1yy = buf[6], *Iz = buf[7], *Ix = buf[8], *Iy = buf[9], *Ixz = buf[10], *Iyz = buf[11],
          *Ixx = buf[12], *Ixy = buf[13], *Iyy = buf[14], *nu = buf[15], *nv = buf[16], *a12 = buf[17],
          *idu = buf[18], *idv = buf[19], *sx = buf[20], *sy = buf[21], *u = buf[22], *v = buf[23],
          *du = buf[24], *dv = buf[25], *up = buf[26];

    int wc = ws[nl - 1], hc = hs[nl - 1];
    memset(u, 0, (size_t)wc * hc * sizeof(float));
    memset(v, 0, (size_t)wc * hc * sizeof(float));
    for (int k = nl - 1; k >= 0; k--) {
        int w = ws[k], h = hs[k];
        size_t n = (size_t)w * h;
        deriv_x(P0[k], Ix0, w, h); deriv_y(P0[k], Iy0, w, h);
        deriv_x(P1[k], I1x, w, h); deriv_y(P1[k], I1y, w, h);
        deriv_x(I1x, I1xx, w, h); deriv_y(I1x, I1xy, w, h); deriv_y(I1y, I1yy, w, h);
        brox_ref_warp(P0[k], Ix0, Iy0, P1[k], I1x, I1y, I1xx, I1xy, I1yy, u, v, w, h,
                      Iz, Ix, Iy, Ixz, Iyz, Ixx, Ixy, Iyy);
        memset(du, 0, n * sizeof(float));
        memset(dv, 0, n * sizeof(float));
        for (int it = 0; it < inner; it++) {
            brox_ref_prepare(u, v, du, dv, Iz, Ix, Iy, Ixz, Iyz, Ixx, Ixy, Iyy,
                             nu, nv, a12, idu, idv, sx, sy, w, h, alpha, gamma);
            brox_ref_sor(du, dv, nu, nv, a12, idu, idv, sx, sy, w, h, solver);
        }
        for (size_t i = 0; i < n; i++) { u[i] = u[i] + du[i]; v[i] = v[i] + dv[i]; }
        if (k > 0) {
            int wf = ws[k - 1], hf = hs[k - 1];
            resample(u, w, h, up, wf, hf, (float)wf / (float)w);
            memcpy(tmp, up, (size_t)wf * hf * sizeof(float));
            resample(v, w, h, up, wf, hf, (float)hf / (float)h);
            memcpy(u, tmp, (size_t)wf * hf * sizeof(float));
            memcpy(v, up, (size_t)wf * hf * sizeof(float));
        }
    }
    memcpy(u_out, u, n0 * sizeof(float));
    memcpy(v_out, v, n0 * sizeof(float));

    for (int i = 0; i < 27; i++) free(buf[i]);
    for (int k = 0; k < nl; k++) { free(P0[k]); free(P1[k]); }
    free(P0); free(P1); free(tmp); free(tmp2);
    return 0;
}

/* u8 entry point: the conversion the reference applies before calc()
 * (src/optical_flow_ext.cpp:314-315: convertTo(CV_32F, 1.0/255.0)). */
int brox_ref_calc_u8(const uint8_t *f0, const uint8_t *f1, int W, int H,
                     float alpha, float gamma, float scale, int inner, int outer, int solver,
                     float *u_out, float *v_out)
{
    size_t n = (size_t)W * H;
    float *a = malloc(n * sizeof(float)), *b = malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) {
        a[i] = (float)f0[i] * (1.0f / 255.0f);
        b[i] = (float)f1[i] * (1.0f / 255.0f);
    }
    int rc = brox_ref_calc_f32(a, b, W, H, alpha, gamma, scale, inner, outer, solver, u_out, v_out);
    free(a); free(b);
    return rc;
}

/* stand-alone pieces exposed for kernel-level parity tests */
void brox_ref_blur(const float *src, float *dst, int w, int h, float scale)
{
    float g[2 * BROX_MAX_RADIUS + 1];
    int R = brox_ref_gauss(scale, g);
    float *tmp = malloc((size_t)w * h * sizeof(float));
    blur(src, tmp, dst, w, h, g, R);
    free(tmp);
}
void brox_ref_resample(const float *src, int ws, int hs, float *dst, int wd, int hd, float mul)
{
    resample(src, ws, hs, dst, wd, hd, mul);
}
void brox_ref_deriv(const float *f, float *dx, float *dy, int w, int h)
{
    deriv_x(f, dx, w, h);
    deriv_y(f, dy, w, h);
}
