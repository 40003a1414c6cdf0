/*
 * oracle/ekf_ref_c.c -- C/OpenMP twin of oracle/ekf_ref.py's measurement model.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by oracle/ekf_c.py, which only tests/,
 * tools/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * import.  The product package never links or loads this file.
 *
 * It restates, operation for operation, what oracle/ekf_ref.py does in NumPy (and
 * through it the reference's CPU path):
 *   render          the four off-screen renders of the textured mesh
 *                   (reference renderer.py:310-325; shaders :22-48,70-113; vertex
 *                   mapping :503-524; texture coordinates :579) with the raster
 *                   rules ekf_ref.py's header fixes;
 *   initjacobian    cuda.py:940-950 (initjacobian_CPU);
 *   jz              cuda.py:972-980 (jz_CPU): one full-frame render of the perturbed
 *                   state and four full-frame sums;
 *   j               cuda.py:982-1010 (j_CPU): two full-frame renders, four sums;
 *   error           renderer.py:485-501, with the uint8 wrap-around;
 *   jacobian        kalman.py:491-518 (_jacobian): 2 * 4N jz evaluations;
 *   hessian_sparse  kalman.py:583-606 (_hessian_sparse): one j per non-zero of the
 *                   upper triangle of the pattern J (kalman.py:202-205).
 * Every evaluation renders the whole frame and sums the whole frame, as the
 * reference does: this is the CPU baseline bench.py times, not a shortcut.  The
 * loops over evaluations are OpenMP-parallel (each thread has its own render
 * targets); the pixel sums of one evaluation run sequentially in row-major order,
 * so a result does not depend on the number of threads.  (NumPy sums pairwise:
 * the two agree to rounding, tests/test_oracle_ekf_c.py holds them to 1e-12.)
 * The render is integer / binary32 arithmetic in a fixed order and is bit-identical
 * to ekf_ref.render (same test file).  Build with -ffp-contract=off.
 *
 * Pinning: through ekf_ref.py -- the reference's own known answers
 * (test/test_cuda.py:198-266) are reproduced with this implementation as well
 * (tests/test_oracle_ekf_c.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SUB 256

typedef struct {
    int N, T, W, H;
    int64_t *tri;             /* T x 3 */
    float *uv;                /* N x 2 */
    uint8_t *tex;             /* H x W */
    double eps_Z, eps_J, eps_M;
    /* initjacobian state */
    double *X0;               /* 4N */
    uint8_t *r_im, *r_m;      /* reference render */
    float *r_fx, *r_fy;
    double *z, *zm;           /* residual images (cuda.py:943-950) */
    float *zfx, *zfy;
    int have_ref;
    int threads;
} ekf_meas;

typedef struct {              /* one set of render targets */
    int64_t *acc, *cnt;
    float *fx, *fy;
    uint8_t *im, *m;
} targets;

static int targets_alloc(targets *t, size_t n)
{
    t->acc = (int64_t *)malloc(n * sizeof(int64_t));
    t->cnt = (int64_t *)malloc(n * sizeof(int64_t));
    t->fx = (float *)malloc(n * sizeof(float));
    t->fy = (float *)malloc(n * sizeof(float));
    t->im = (uint8_t *)malloc(n);
    t->m = (uint8_t *)malloc(n);
    return t->acc && t->cnt && t->fx && t->fy && t->im && t->m;
}

static void targets_free(targets *t)
{
    free(t->acc); free(t->cnt); free(t->fx); free(t->fy); free(t->im); free(t->m);
}

static int64_t floordiv(int64_t a, int64_t b)      /* Python's // for b > 0 */
{
    int64_t q = a / b;
    if ((a % b != 0) && (a < 0)) q--;
    return q;
}

static int topleft(int64_t dx, int64_t dy) { return (dy > 0) || (dy == 0 && dx < 0); }

static int64_t min3(int64_t a, int64_t b, int64_t c) { int64_t m = a < b ? a : b; return m < c ? m : c; }
static int64_t max3(int64_t a, int64_t b, int64_t c) { int64_t m = a > b ? a : b; return m > c ? m : c; }

/* ekf_ref.render, line by line */
static void render_into(const ekf_meas *h, const double *X, targets *o)
{
    const int N = h->N, W = h->W, H = h->H;
    const size_t n = (size_t)W * H;
    memset(o->acc, 0, n * sizeof(int64_t));
    memset(o->cnt, 0, n * sizeof(int64_t));
    memset(o->fx, 0, n * sizeof(float));
    memset(o->fy, 0, n * sizeof(float));
    for (int t = 0; t < h->T; t++) {
        int i0 = (int)h->tri[3 * t], i1 = (int)h->tri[3 * t + 1], i2 = (int)h->tri[3 * t + 2];
        /* snap: rint(x * 256), round half to even (the default rounding mode) */
        int64_t x0 = (int64_t)rint(X[2 * i0] * SUB), y0 = (int64_t)rint(X[2 * i0 + 1] * SUB);
        int64_t x1 = (int64_t)rint(X[2 * i1] * SUB), y1 = (int64_t)rint(X[2 * i1 + 1] * SUB);
        int64_t x2 = (int64_t)rint(X[2 * i2] * SUB), y2 = (int64_t)rint(X[2 * i2 + 1] * SUB);
        int64_t area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
        if (area == 0) continue;
        if (area < 0) {
            int ti = i1; i1 = i2; i2 = ti;
            int64_t tx = x1, ty = y1;
            x1 = x2; y1 = y2; x2 = tx; y2 = ty;
            area = -area;
        }
        int64_t c_lo = floordiv(min3(x0, x1, x2) - 128, SUB), c_hi = floordiv(max3(x0, x1, x2) - 128, SUB) + 1;
        int64_t r_lo = floordiv(min3(y0, y1, y2) - 128, SUB), r_hi = floordiv(max3(y0, y1, y2) - 128, SUB) + 1;
        if (c_lo < 0) c_lo = 0;
        if (r_lo < 0) r_lo = 0;
        if (c_hi > W - 1) c_hi = W - 1;
        if (r_hi > H - 1) r_hi = H - 1;
        if (c_lo > c_hi || r_lo > r_hi) continue;
        const int tl0 = topleft(x2 - x1, y2 - y1), tl1 = topleft(x0 - x2, y0 - y2), tl2 = topleft(x1 - x0, y1 - y0);
        const float inv = 1.0f / (float)area;
        const float ux0 = h->uv[2 * i0], ux1 = h->uv[2 * i1], ux2 = h->uv[2 * i2];
        const float uy0 = h->uv[2 * i0 + 1], uy1 = h->uv[2 * i1 + 1], uy2 = h->uv[2 * i2 + 1];
        const float ax0 = (float)X[2 * N + 2 * i0], ax1 = (float)X[2 * N + 2 * i1], ax2 = (float)X[2 * N + 2 * i2];
        const float ay0 = (float)(-X[2 * N + 2 * i0 + 1]), ay1 = (float)(-X[2 * N + 2 * i1 + 1]),
                    ay2 = (float)(-X[2 * N + 2 * i2 + 1]);
        for (int64_t r = r_lo; r <= r_hi; r++) {
            const int64_t py = r * SUB + 128;
            for (int64_t c = c_lo; c <= c_hi; c++) {
                const int64_t px = c * SUB + 128;
                const int64_t e0 = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1);
                const int64_t e1 = (x0 - x2) * (py - y2) - (y0 - y2) * (px - x2);
                const int64_t e2 = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0);
                const int ins = (e0 > 0 || (e0 == 0 && tl0)) && (e1 > 0 || (e1 == 0 && tl1)) && (e2 > 0 || (e2 == 0 && tl2));
                if (!ins) continue;
                const float l1 = (float)e1 * inv, l2 = (float)e2 * inv;
                /* plane-equation form: (a0 + l1 (a1 - a0)) + l2 (a2 - a0) */
                const float tx = (ux0 + l1 * (ux1 - ux0)) + l2 * (ux2 - ux0);
                const float ty = (uy0 + l1 * (uy1 - uy0)) + l2 * (uy2 - uy0);
                int64_t cx = (int64_t)floorf(tx), cy = (int64_t)floorf(ty);
                if (cx < 0) cx = 0;
                if (cx > W - 1) cx = W - 1;
                if (cy < 0) cy = 0;
                if (cy > H - 1) cy = H - 1;
                const size_t p = (size_t)r * W + (size_t)c;
                o->acc[p] += h->tex[(size_t)cy * W + (size_t)cx];
                o->fx[p] = o->fx[p] + ((ax0 + l1 * (ax1 - ax0)) + l2 * (ax2 - ax0));
                o->fy[p] = o->fy[p] + ((ay0 + l1 * (ay1 - ay0)) + l2 * (ay2 - ay0));
                o->cnt[p] += 1;
            }
        }
    }
    for (size_t p = 0; p < n; p++) {
        o->im[p] = (uint8_t)(o->acc[p] > 255 ? 255 : o->acc[p]);
        o->m[p] = o->cnt[p] > 0 ? 255 : 0;
    }
}

/* ---- handle ------------------------------------------------------------------------------ */
ekf_meas *ekf_c_create(int N, int T, const int64_t *tri, const float *uv, const uint8_t *tex, int W, int H,
                       double eps_Z, double eps_J, double eps_M)
{
    ekf_meas *h = (ekf_meas *)calloc(1, sizeof(ekf_meas));
    if (!h) return NULL;
    const size_t n = (size_t)W * H;
    h->N = N; h->T = T; h->W = W; h->H = H;
    h->eps_Z = eps_Z; h->eps_J = eps_J; h->eps_M = eps_M;
    h->tri = (int64_t *)malloc((size_t)3 * T * sizeof(int64_t));
    h->uv = (float *)malloc((size_t)2 * N * sizeof(float));
    h->tex = (uint8_t *)malloc(n);
    h->X0 = (double *)malloc((size_t)4 * N * sizeof(double));
    h->r_im = (uint8_t *)malloc(n); h->r_m = (uint8_t *)malloc(n);
    h->r_fx = (float *)malloc(n * sizeof(float)); h->r_fy = (float *)malloc(n * sizeof(float));
    h->z = (double *)malloc(n * sizeof(double)); h->zm = (double *)malloc(n * sizeof(double));
    h->zfx = (float *)malloc(n * sizeof(float)); h->zfy = (float *)malloc(n * sizeof(float));
    memcpy(h->tri, tri, (size_t)3 * T * sizeof(int64_t));
    memcpy(h->uv, uv, (size_t)2 * N * sizeof(float));
    memcpy(h->tex, tex, n);
    h->threads = 1;
    return h;
}

void ekf_c_destroy(ekf_meas *h)
{
    if (!h) return;
    free(h->tri); free(h->uv); free(h->tex); free(h->X0); free(h->r_im); free(h->r_m); free(h->r_fx); free(h->r_fy);
    free(h->z); free(h->zm); free(h->zfx); free(h->zfy);
    free(h);
}

void ekf_c_set_threads(ekf_meas *h, int n) { h->threads = n < 1 ? 1 : n; }

int ekf_c_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* render(X) -> im, fx, fy, m (any may be NULL) */
int ekf_c_render(const ekf_meas *h, const double *X, uint8_t *im, float *fx, float *fy, uint8_t *m)
{
    const size_t n = (size_t)h->W * h->H;
    targets t;
    if (!targets_alloc(&t, n)) { targets_free(&t); return -1; }
    render_into(h, X, &t);
    if (im) memcpy(im, t.im, n);
    if (m) memcpy(m, t.m, n);
    if (fx) memcpy(fx, t.fx, n * sizeof(float));
    if (fy) memcpy(fy, t.fy, n * sizeof(float));
    targets_free(&t);
    return 0;
}

/* initjacobian_CPU (cuda.py:940-950); y_m in {0,1}, multiplied by 255 as renderer.py:679 does */
int ekf_c_initjacobian(ekf_meas *h, const double *X, const uint8_t *y_im, const float *y_fx, const float *y_fy,
                       const uint8_t *y_m)
{
    const size_t n = (size_t)h->W * h->H;
    targets t;
    if (!targets_alloc(&t, n)) { targets_free(&t); return -1; }
    render_into(h, X, &t);
    memcpy(h->X0, X, (size_t)4 * h->N * sizeof(double));
    memcpy(h->r_im, t.im, n); memcpy(h->r_m, t.m, n);
    memcpy(h->r_fx, t.fx, n * sizeof(float)); memcpy(h->r_fy, t.fy, n * sizeof(float));
    for (size_t p = 0; p < n; p++) {
        h->z[p] = ((double)y_im[p] - (double)t.im[p]) / 255.0;
        h->zfx[p] = y_fx[p] - t.fx[p];
        h->zfy[p] = y_fy[p] + t.fy[p];
        h->zm[p] = (255.0 * (double)y_m[p] - (double)t.m[p]) / 255.0;
    }
    h->have_ref = 1;
    targets_free(&t);
    return 0;
}

/* the four sums of jz_CPU (cuda.py:972-980) for the render in t; c = components already divided by eps */
static double jz_sums(const ekf_meas *h, const targets *t, double c[4])
{
    const size_t n = (size_t)h->W * h->H;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (size_t p = 0; p < n; p++) {
        const double d = ((double)t->im[p] - (double)h->r_im[p]) / 255.0;
        const float dfx = t->fx[p] - h->r_fx[p], dfy = t->fy[p] - h->r_fy[p];
        const double dm = ((double)t->m[p] - (double)h->r_m[p]) / 255.0;
        s0 += d * h->z[p];
        s1 += (double)dfx * (double)h->zfx[p];
        s2 += (double)dfy * (double)h->zfy[p];
        s3 += dm * h->zm[p];
    }
    c[0] = s0 / h->eps_Z; c[1] = s1 / h->eps_J; c[2] = -s2 / h->eps_J; c[3] = s3 / h->eps_M;
    return ((c[0] + c[1]) + c[2]) + c[3];
}

int ekf_c_jz(const ekf_meas *h, const double *Xp, double *total, double c[4])
{
    if (!h->have_ref) return -2;
    targets t;
    if (!targets_alloc(&t, (size_t)h->W * h->H)) { targets_free(&t); return -1; }
    render_into(h, Xp, &t);
    *total = jz_sums(h, &t, c);
    targets_free(&t);
    return 0;
}

/* j_CPU (cuda.py:982-1010) for the two renders a, b */
static double j_sums(const ekf_meas *h, const targets *a, const targets *b)
{
    const size_t n = (size_t)h->W * h->H;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (size_t p = 0; p < n; p++) {
        const double ri = (double)h->r_im[p], rm = (double)h->r_m[p];
        const float rfx = h->r_fx[p], rfy = h->r_fy[p];
        s0 += (((double)a->im[p] - ri) / 255.0) * (((double)b->im[p] - ri) / 255.0);
        s1 += (double)(a->fx[p] - rfx) * (double)(b->fx[p] - rfx);
        s2 += (double)(a->fy[p] - rfy) * (double)(b->fy[p] - rfy);
        s3 += (((double)a->m[p] - rm) / 255.0) * (((double)b->m[p] - rm) / 255.0);
    }
    return ((s0 / h->eps_Z + s1 / h->eps_J) + s2 / h->eps_J) + s3 / h->eps_M;
}

int ekf_c_j(const ekf_meas *h, double deltaX, int i, int j, double *out)
{
    if (!h->have_ref) return -2;
    const size_t n = (size_t)h->W * h->H, n4 = (size_t)4 * h->N;
    targets a, b;
    double *Xp = (double *)malloc(n4 * sizeof(double)), *Xq = (double *)malloc(n4 * sizeof(double));
    int ok = targets_alloc(&a, n);
    ok = targets_alloc(&b, n) && ok && Xp && Xq;
    if (ok) {
        memcpy(Xp, h->X0, n4 * sizeof(double)); Xp[i] += deltaX;
        memcpy(Xq, h->X0, n4 * sizeof(double)); Xq[j] += deltaX;
        render_into(h, Xp, &a);
        render_into(h, Xq, &b);
        *out = j_sums(h, &a, &b);
    }
    targets_free(&a); targets_free(&b); free(Xp); free(Xq);
    return ok ? 0 : -1;
}

/* Renderer.error (renderer.py:485-501) for uint8 y_im / y_m: differences and squares wrap modulo 256 */
int ekf_c_error(const ekf_meas *h, const double *X, const uint8_t *y_im, const float *y_fx, const float *y_fy,
                const uint8_t *y_m, double err[4], float *fx, float *fy)
{
    const size_t n = (size_t)h->W * h->H;
    targets t;
    if (!targets_alloc(&t, n)) { targets_free(&t); return -1; }
    render_into(h, X, &t);
    uint64_t e_im = 0, e_m = 0;
    double e_fx = 0.0, e_fy = 0.0;
    for (size_t p = 0; p < n; p++) {
        const uint8_t d = (uint8_t)(y_im[p] - t.im[p]);
        e_im += (uint8_t)(d * d);
        const uint8_t dm = (uint8_t)((uint8_t)(255 * y_m[p]) - t.m[p]);
        e_m += (uint8_t)(dm * dm);
        const float dfx = y_fx[p] - t.fx[p], dfy = y_fy[p] + t.fy[p];
        e_fx += (double)dfx * (double)dfx;
        e_fy += (double)dfy * (double)dfy;
    }
    err[0] = (double)e_im; err[1] = e_fx; err[2] = e_fy; err[3] = (double)e_m;
    if (fx) memcpy(fx, t.fx, n * sizeof(float));
    if (fy) memcpy(fy, t.fy, n * sizeof(float));
    targets_free(&t);
    return 0;
}

/* _jacobian (kalman.py:491-518): central differences of jz for the state indices idx[0..cnt)
 * (cnt = 4N, idx = 0..4N-1 for the whole Jacobian).  initjacobian must have been called at X.
 * Hz[k], Hzc[k*4..] receive the entries of idx[k]. */
int ekf_c_jacobian(const ekf_meas *h, double deltaX, int cnt, const int *idx, double *Hz, double *Hzc)
{
    if (!h->have_ref) return -2;
    const size_t n = (size_t)h->W * h->H, n4 = (size_t)4 * h->N;
    int fail = 0;
#pragma omp parallel num_threads(h->threads)
    {
        targets t;
        double *Xp = (double *)malloc(n4 * sizeof(double));
        const int ok = targets_alloc(&t, n) && Xp;
        if (!ok) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (int k = 0; k < cnt; k++) {
            if (!ok) continue;
            double cp[4], cm[4];
            memcpy(Xp, h->X0, n4 * sizeof(double)); Xp[idx[k]] += deltaX;
            render_into(h, Xp, &t);
            const double hp = jz_sums(h, &t, cp);
            memcpy(Xp, h->X0, n4 * sizeof(double)); Xp[idx[k]] -= deltaX;
            render_into(h, Xp, &t);
            const double hm = jz_sums(h, &t, cm);
            Hz[k] = (hp / deltaX - hm / deltaX) / 2;
            for (int q = 0; q < 4; q++) Hzc[4 * k + q] = (cp[q] / deltaX - cm[q] / deltaX) / 2;
        }
        targets_free(&t);
        free(Xp);
    }
    return fail ? -1 : 0;
}

/* _hessian_sparse (kalman.py:583-606): out[k] = j(deltaX, pi[k], pj[k]) / deltaX / deltaX for the listed pairs */
int ekf_c_hessian_pairs(const ekf_meas *h, double deltaX, int cnt, const int *pi, const int *pj, double *out)
{
    if (!h->have_ref) return -2;
    const size_t n = (size_t)h->W * h->H, n4 = (size_t)4 * h->N;
    int fail = 0;
#pragma omp parallel num_threads(h->threads)
    {
        targets a, b;
        double *Xp = (double *)malloc(n4 * sizeof(double));
        int ok = targets_alloc(&a, n);
        ok = targets_alloc(&b, n) && ok && Xp;
        if (!ok) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (int k = 0; k < cnt; k++) {
            if (!ok) continue;
            memcpy(Xp, h->X0, n4 * sizeof(double)); Xp[pi[k]] += deltaX;
            render_into(h, Xp, &a);
            memcpy(Xp, h->X0, n4 * sizeof(double)); Xp[pj[k]] += deltaX;
            render_into(h, Xp, &b);
            out[k] = j_sums(h, &a, &b) / deltaX / deltaX;
        }
        targets_free(&a); targets_free(&b);
        free(Xp);
    }
    return fail ? -1 : 0;
}
