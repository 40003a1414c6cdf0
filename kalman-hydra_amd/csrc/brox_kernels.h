// Device kernels of the Brox optical-flow path (gfx950).  All arithmetic is
// binary32 in a fixed order with contraction off (-ffp-contract=off) and
// correctly rounded sqrt / divide, so results are reproducible bit for bit.
//
// Layout: every field of one pyramid level is a pitched plane, `pitch` floats
// per row (multiple of 16, so rows start 64-byte aligned and pixel pairs can be
// moved as float2), `plane` floats per batch item; blockIdx.z = batch item.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define BROX_EPS2 1e-6f
#define BROX_MAX_TAPS 33

struct Geo {      // one pyramid level
    int w, h, pitch, plane;
};

struct Taps {
    float g[BROX_MAX_TAPS];
    int R;
};

__device__ __forceinline__ int d_mirror(int i, int n)
{
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        else i = 2 * n - i - 1;
    }
    return i;
}

__device__ __forceinline__ int d_clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

// ---- frame conversion: x * (1/255), as the reference does before calc() -----
// blockIdx.z < nb: frame 0 of pair z -> dst0, else frame 1 of pair z - nb -> dst1 (one launch for both)
__global__ void k_u8_to_f32(const uint8_t *__restrict__ src0, const uint8_t *__restrict__ src1, int src_pitch, int src_plane,
                            float *__restrict__ dst0, float *__restrict__ dst1, Geo g, int nb)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const int second = (int)blockIdx.z >= nb, b = second ? blockIdx.z - nb : blockIdx.z;
    const uint8_t *s = (second ? src1 : src0) + (size_t)b * src_plane;
    float *d = (second ? dst1 : dst0) + (size_t)b * g.plane;
    d[y * g.pitch + x] = (float)s[y * src_pitch + x] * (1.0f / 255.0f);
}

// ---- separable Gaussian, mirrored border ---------------------------------------
// Two images per launch (the two frames of the pairs): blockIdx.z < nb works on set 0, the rest on set 1.
template <bool VERT>
__global__ void k_blur(const float *__restrict__ src0, float *__restrict__ dst0, const float *__restrict__ src1,
                       float *__restrict__ dst1, Geo g, Taps t, int nb)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const int second = (int)blockIdx.z >= nb, b = second ? blockIdx.z - nb : blockIdx.z;
    const float *s = (second ? src1 : src0) + (size_t)b * g.plane;
    float acc = 0.0f;
    for (int i = -t.R; i <= t.R; i++) {
        float val = VERT ? s[d_mirror(y + i, g.h) * g.pitch + x] : s[y * g.pitch + d_mirror(x + i, g.w)];
        acc = acc + t.g[i + t.R] * val;
    }
    (second ? dst1 : dst0)[(size_t)b * g.plane + y * g.pitch + x] = acc;
}

// ---- bilinear sampling, coordinates clamped to the image -----------------------
__device__ __forceinline__ float d_bilin(const float *__restrict__ img, int w, int h, int pitch,
                                         float px, float py)
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
    float a = img[y0 * pitch + x0], b = img[y0 * pitch + x1];
    float c = img[y1 * pitch + x0], d = img[y1 * pitch + x1];
    float top = (1.0f - ax) * a + ax * b;
    float bot = (1.0f - ax) * c + ax * d;
    return (1.0f - ay) * top + ay * bot;
}

// resample src level onto dst level, value scaled by mul (pyramid and prolongation); two images per launch as k_blur
__global__ void k_resample(const float *__restrict__ src0, float *__restrict__ dst0, const float *__restrict__ src1,
                           float *__restrict__ dst1, Geo gs, Geo gd, float mul, int nb)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= gd.w || y >= gd.h) return;
    float rx = (float)gs.w / (float)gd.w, ry = (float)gs.h / (float)gd.h;
    float sx = ((float)x + 0.5f) * rx - 0.5f;
    float sy = ((float)y + 0.5f) * ry - 0.5f;
    const int second = (int)blockIdx.z >= nb, b = second ? blockIdx.z - nb : blockIdx.z;
    const float *s = (second ? src1 : src0) + (size_t)b * gs.plane;
    (second ? dst1 : dst0)[(size_t)b * gd.plane + y * gd.pitch + x] = d_bilin(s, gs.w, gs.h, gs.pitch, sx, sy) * mul;
}

// ---- 5-tap derivatives ------------------------------------------------------------
__device__ __forceinline__ float d_d5(float m2, float m1, float p1, float p2)
{
    return (8.0f * (p1 - m1) - (p2 - m2)) * (1.0f / 12.0f);
}

// dx and/or dy of an image (either output may be null); two images per launch as k_blur
__global__ void k_deriv(const float *__restrict__ src0, float *__restrict__ dx0, float *__restrict__ dy0,
                        const float *__restrict__ src1, float *__restrict__ dx1, float *__restrict__ dy1, Geo g, int nb)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const int second = (int)blockIdx.z >= nb, b = second ? blockIdx.z - nb : blockIdx.z;
    size_t off = (size_t)b * g.plane;
    const float *s = (second ? src1 : src0) + off;
    float *dx = second ? dx1 : dx0, *dy = second ? dy1 : dy0;
    if (dx) {
        const float *r = s + y * g.pitch;
        dx[off + y * g.pitch + x] = d_d5(r[d_mirror(x - 2, g.w)], r[d_mirror(x - 1, g.w)],
                                         r[d_mirror(x + 1, g.w)], r[d_mirror(x + 2, g.w)]);
    }
    if (dy) {
        dy[off + y * g.pitch + x] = d_d5(s[d_mirror(y - 2, g.h) * g.pitch + x], s[d_mirror(y - 1, g.h) * g.pitch + x],
                                         s[d_mirror(y + 1, g.h) * g.pitch + x], s[d_mirror(y + 2, g.h) * g.pitch + x]);
    }
}

// ---- one pyramid level in one launch: Gaussian (rows, then columns) and resampling, both frames ------
// The value of every tap is computed the way the separate kernels compute it (k_blur<false>, k_blur<true>,
// k_resample: the same operations in the same order, so the same bits); the blurred image is never
// stored.  A level costs one launch instead of six; the ~100 reads per output pixel come out of L1 / L2.
// blockIdx.z < nb: frame 0, else frame 1.
__device__ __forceinline__ float d_blur_at(const float *__restrict__ s, const Geo &g, const Taps &t, int x, int y)
{
    float acc = 0.0f;                                   // k_blur<true> over rows y - R .. y + R of ...
    for (int j = -t.R; j <= t.R; j++) {
        const float *row = s + d_mirror(y + j, g.h) * g.pitch;
        float h = 0.0f;                                 // ... k_blur<false> of that row at x
        for (int i = -t.R; i <= t.R; i++) h = h + t.g[i + t.R] * row[d_mirror(x + i, g.w)];
        acc = acc + t.g[j + t.R] * h;
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_pyr_down(const float *__restrict__ src0, const float *__restrict__ src1, Geo gs,
                                                   float *__restrict__ dst0, float *__restrict__ dst1, Geo gd, Taps t, int nb)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= gd.w || y >= gd.h) return;
    const int second = (int)blockIdx.z >= nb, b = second ? blockIdx.z - nb : blockIdx.z;
    const float *s = (second ? src1 : src0) + (size_t)b * gs.plane;
    float *d = (second ? dst1 : dst0) + (size_t)b * gd.plane;
    // k_resample / d_bilin on the blurred image
    const float rx = (float)gs.w / (float)gd.w, ry = (float)gs.h / (float)gd.h;
    float px = ((float)x + 0.5f) * rx - 0.5f;
    float py = ((float)y + 0.5f) * ry - 0.5f;
    const int w = gs.w, h = gs.h;
    if (px < 0.0f) px = 0.0f;
    if (py < 0.0f) py = 0.0f;
    if (px > (float)(w - 1)) px = (float)(w - 1);
    if (py > (float)(h - 1)) py = (float)(h - 1);
    const float fx0 = floorf(px), fy0 = floorf(py);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    const int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
    const float ax = px - fx0, ay = py - fy0;
    const float a = d_blur_at(s, gs, t, x0, y0), bq = d_blur_at(s, gs, t, x1, y0);
    const float c = d_blur_at(s, gs, t, x0, y1), dq = d_blur_at(s, gs, t, x1, y1);
    const float top = (1.0f - ax) * a + ax * bq;
    const float bot = (1.0f - ax) * c + ax * dq;
    d[y * gd.pitch + x] = ((1.0f - ay) * top + ay * bot) * 1.0f;
}

// ---- all derivative images of a level in one launch ----------------------------------------------------
// Ix0, Iy0 of frame 0; I1x, I1y, I1xx, I1xy, I1yy of frame 1 -- what four k_deriv launches produce
// (second derivatives = the first-derivative operator applied to the first-derivative image, evaluated
// here from the image itself: the same operations on the same values).
struct DerivOut {
    float *Ix0, *Iy0, *I1x, *I1y, *I1xx, *I1xy, *I1yy;
};
__device__ __forceinline__ float d_dx_at(const float *__restrict__ s, const Geo &g, int x, int y)
{
    const float *r = s + y * g.pitch;
    return d_d5(r[d_mirror(x - 2, g.w)], r[d_mirror(x - 1, g.w)], r[d_mirror(x + 1, g.w)], r[d_mirror(x + 2, g.w)]);
}
__device__ __forceinline__ float d_dy_at(const float *__restrict__ s, const Geo &g, int x, int y)
{
    return d_d5(s[d_mirror(y - 2, g.h) * g.pitch + x], s[d_mirror(y - 1, g.h) * g.pitch + x],
                s[d_mirror(y + 1, g.h) * g.pitch + x], s[d_mirror(y + 2, g.h) * g.pitch + x]);
}
__global__ __launch_bounds__(256) void k_deriv_all(const float *__restrict__ I0, const float *__restrict__ I1, DerivOut o, Geo g)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t off = (size_t)blockIdx.z * g.plane;
    const float *a = I0 + off, *b = I1 + off;
    const size_t p = off + y * g.pitch + x;
    o.Ix0[p] = d_dx_at(a, g, x, y);
    o.Iy0[p] = d_dy_at(a, g, x, y);
    o.I1x[p] = d_dx_at(b, g, x, y);
    o.I1y[p] = d_dy_at(b, g, x, y);
    const int xm2 = d_mirror(x - 2, g.w), xm1 = d_mirror(x - 1, g.w), xp1 = d_mirror(x + 1, g.w), xp2 = d_mirror(x + 2, g.w);
    const int ym2 = d_mirror(y - 2, g.h), ym1 = d_mirror(y - 1, g.h), yp1 = d_mirror(y + 1, g.h), yp2 = d_mirror(y + 2, g.h);
    o.I1xx[p] = d_d5(d_dx_at(b, g, xm2, y), d_dx_at(b, g, xm1, y), d_dx_at(b, g, xp1, y), d_dx_at(b, g, xp2, y));
    o.I1xy[p] = d_d5(d_dx_at(b, g, x, ym2), d_dx_at(b, g, x, ym1), d_dx_at(b, g, x, yp1), d_dx_at(b, g, x, yp2));
    o.I1yy[p] = d_d5(d_dy_at(b, g, x, ym2), d_dy_at(b, g, x, ym1), d_dy_at(b, g, x, yp1), d_dy_at(b, g, x, yp2));
}

// ---- u + du, v + dv carried to the next finer level (or, at level 0, to the caller's arrays) in one launch ----
// What k_add followed by two k_resample launches (and, at the end, two strided copies per pair) did: the taps
// of the bilinear prolongation are the sums u + du themselves.
__device__ __forceinline__ float d_bilin_sum(const float *__restrict__ a, const float *__restrict__ b, int w, int h, int pitch,
                                             float px, float py)
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
    float q00 = a[y0 * pitch + x0] + b[y0 * pitch + x0], q01 = a[y0 * pitch + x1] + b[y0 * pitch + x1];
    float q10 = a[y1 * pitch + x0] + b[y1 * pitch + x0], q11 = a[y1 * pitch + x1] + b[y1 * pitch + x1];
    float top = (1.0f - ax) * q00 + ax * q01;
    float bot = (1.0f - ax) * q10 + ax * q11;
    return (1.0f - ay) * top + ay * bot;
}

__global__ __launch_bounds__(256) void k_add_prolong(const float *__restrict__ u, const float *__restrict__ v,
                                                      const float *__restrict__ du, const float *__restrict__ dv, Geo gs,
                                                      float *__restrict__ u2, float *__restrict__ v2, Geo gd, float mulx, float muly)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= gd.w || y >= gd.h) return;
    const float rx = (float)gs.w / (float)gd.w, ry = (float)gs.h / (float)gd.h;
    const float sx = ((float)x + 0.5f) * rx - 0.5f;
    const float sy = ((float)y + 0.5f) * ry - 0.5f;
    const size_t so = (size_t)blockIdx.z * gs.plane, q = (size_t)blockIdx.z * gd.plane + y * gd.pitch + x;
    u2[q] = d_bilin_sum(u + so, du + so, gs.w, gs.h, gs.pitch, sx, sy) * mulx;
    v2[q] = d_bilin_sum(v + so, dv + so, gs.w, gs.h, gs.pitch, sx, sy) * muly;
}

// level 0: the sums go straight into the caller's tight W x H planes
__global__ void k_add_out(const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ du,
                          const float *__restrict__ dv, Geo g, float *__restrict__ ox, float *__restrict__ oy)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t p = (size_t)blockIdx.z * g.plane + y * g.pitch + x;
    const size_t q = ((size_t)blockIdx.z * g.h + y) * g.w + x;
    ox[q] = u[p] + du[p];
    oy[q] = v[p] + dv[p];
}

// ---- warp: frame 1 and its derivative images sampled at (x+u, y+v) ---------------
struct WarpIn {
    const float *I0, *Ix0, *Iy0, *I1, *I1x, *I1y, *I1xx, *I1xy, *I1yy, *u, *v;
};
struct WarpOut {
    float *Iz, *Ix, *Iy, *Ixz, *Iyz, *Ixx, *Ixy, *Iyy;
};

// One 64x4 block, two ways of getting at the taps (same values, same arithmetic):
//   WINDOW = false: every thread reads its 2x2 taps from memory; neighbouring lanes share cache lines
//       and the vector L1 serves the re-use.
//   WINDOW = true: the taps of a block's samples lie in a window only a little larger than the block
//       wherever the flow is smooth; the block finds the bounding box of its taps (wave min/max, then
//       LDS atomics), stages that window of the six sampled fields in LDS with row-wise coalesced
//       reads and samples from there.  A block whose taps do not fit (a discontinuity, a very large
//       divergence) falls back to direct reads.
// Measured at 8 x 1024^2 (profiles/): direct 126 us (5.0 TB/s of algorithmic traffic), window 173 us
// with an 80x12 window and 280 us with 88x20 -- the staging costs two barriers and LDS space
// (occupancy) and buys nothing the L1 does not already give.  Direct is the default
// (hm_brox_tune "warp_window").
#define WARP_WX 80            // 64 + 2 * 8
#define WARP_WY 12            // 4 + 2 * 4
__device__ __forceinline__ float d_bilin_win(const float (*win)[WARP_WX], int w, int h, int wx0, int wy0, float px, float py)
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
    float a = win[y0 - wy0][x0 - wx0], b = win[y0 - wy0][x1 - wx0];
    float c = win[y1 - wy0][x0 - wx0], d = win[y1 - wy0][x1 - wx0];
    float top = (1.0f - ax) * a + ax * b;
    float bot = (1.0f - ax) * c + ax * d;
    return (1.0f - ay) * top + ay * bot;
}

template <bool WINDOW>
__global__ __launch_bounds__(256) void k_warp(WarpIn in, WarpOut out, Geo g)
{
    __shared__ float win[WINDOW ? 6 : 1][WINDOW ? WARP_WY : 1][WARP_WX];
    __shared__ int s_box[4];                      // min x, max x, min y, max y over the block's taps
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int w = g.w, h = g.h, pitch = g.pitch;
    const size_t off = (size_t)blockIdx.z * g.plane;
    const bool inside = x < w && y < h;
    const size_t p = off + (inside ? y * pitch + x : 0);
    float px = 0.0f, py = 0.0f;
    bool sampled = false;
    if (inside) {
        px = (float)x + in.u[p];
        py = (float)y + in.v[p];
        sampled = !(px < 0.0f || py < 0.0f || px > (float)(w - 1) || py > (float)(h - 1));
    }
    bool staged = false;
    int wx0 = 0, wy0 = 0;
    if (WINDOW) {
    if (tid == 0) { s_box[0] = w; s_box[1] = -1; s_box[2] = h; s_box[3] = -1; }
    __syncthreads();
    {
        int bx0 = w, bx1 = -1, by0 = h, by1 = -1;
        if (sampled) {
            bx0 = (int)floorf(px); by0 = (int)floorf(py);
            bx1 = bx0 + 1 < w ? bx0 + 1 : w - 1; by1 = by0 + 1 < h ? by0 + 1 : h - 1;
        }
        for (int o = 32; o > 0; o >>= 1) {          // per wave first: four LDS atomics per wave, not per thread
            bx0 = min(bx0, __shfl_xor(bx0, o, 64)); bx1 = max(bx1, __shfl_xor(bx1, o, 64));
            by0 = min(by0, __shfl_xor(by0, o, 64)); by1 = max(by1, __shfl_xor(by1, o, 64));
        }
        if ((tid & 63) == 0) {
            atomicMin(&s_box[0], bx0); atomicMax(&s_box[1], bx1);
            atomicMin(&s_box[2], by0); atomicMax(&s_box[3], by1);
        }
    }
    __syncthreads();
    const int wx1 = s_box[1], wy1 = s_box[3];
    wx0 = s_box[0]; wy0 = s_box[2];
    staged = wx1 >= wx0 && wx1 - wx0 < WARP_WX && wy1 - wy0 < WARP_WY;
    if (staged) {
        const float *src[6] = {in.I1 + off, in.I1x + off, in.I1y + off, in.I1xx + off, in.I1xy + off, in.I1yy + off};
        const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
        for (int i = tid; i < ww * wh; i += 256) {
            const int ly = i / ww, lx = i - ly * ww;
            const int q = (wy0 + ly) * pitch + wx0 + lx;
#pragma unroll
            for (int f = 0; f < 6; f++) win[WINDOW ? f : 0][ly][lx] = src[f][q];
        }
    }
    __syncthreads();
    }
    if (!inside) return;
    if (!sampled) {
        out.Iz[p] = 0.0f; out.Ix[p] = 0.0f; out.Iy[p] = 0.0f; out.Ixz[p] = 0.0f; out.Iyz[p] = 0.0f;
        out.Ixx[p] = 0.0f; out.Ixy[p] = 0.0f; out.Iyy[p] = 0.0f;
        return;
    }
    float i1, ix, iy;
    if (staged) {
        i1 = d_bilin_win(win[WINDOW ? 0 : 0], w, h, wx0, wy0, px, py);
        ix = d_bilin_win(win[WINDOW ? 1 : 0], w, h, wx0, wy0, px, py);
        iy = d_bilin_win(win[WINDOW ? 2 : 0], w, h, wx0, wy0, px, py);
    } else {
        i1 = d_bilin(in.I1 + off, w, h, pitch, px, py);
        ix = d_bilin(in.I1x + off, w, h, pitch, px, py);
        iy = d_bilin(in.I1y + off, w, h, pitch, px, py);
    }
    out.Iz[p] = i1 - in.I0[p];
    out.Ix[p] = ix;
    out.Iy[p] = iy;
    if (x < 2 || y < 2 || x > w - 3 || y > h - 3 ||
        px < 2.0f || py < 2.0f || px > (float)(w - 4) || py > (float)(h - 4)) {
        out.Ixz[p] = 0.0f; out.Iyz[p] = 0.0f; out.Ixx[p] = 0.0f; out.Ixy[p] = 0.0f; out.Iyy[p] = 0.0f;
        return;
    }
    out.Ixz[p] = ix - in.Ix0[p];
    out.Iyz[p] = iy - in.Iy0[p];
    if (staged) {
        out.Ixx[p] = d_bilin_win(win[WINDOW ? 3 : 0], w, h, wx0, wy0, px, py);
        out.Ixy[p] = d_bilin_win(win[WINDOW ? 4 : 0], w, h, wx0, wy0, px, py);
        out.Iyy[p] = d_bilin_win(win[WINDOW ? 5 : 0], w, h, wx0, wy0, px, py);
    } else {
        out.Ixx[p] = d_bilin(in.I1xx + off, w, h, pitch, px, py);
        out.Ixy[p] = d_bilin(in.I1xy + off, w, h, pitch, px, py);
        out.Iyy[p] = d_bilin(in.I1yy + off, w, h, pitch, px, py);
    }
}

// ---- linear system of one lagged-nonlinearity step ---------------------------------
__device__ __forceinline__ float d_psi(float s2) { return 0.5f / sqrtf(s2 + BROX_EPS2); }

struct PrepIn {
    const float *u, *v, *du, *dv, *Iz, *Ix, *Iy, *Ixz, *Iyz, *Ixx, *Ixy, *Iyy;
};
struct Coef {
    float *nu, *nv, *a12, *idu, *idv, *sx, *sy;
};

// One launch per lagged-nonlinearity step: the edge diffusivities of the smoothness term at the
// current u+du, v+dv, then the 2x2 system of every pixel.  One 64x4 block; u, v, u+du and v+dv of the
// block plus a 1-px ring are staged in LDS once (ring filled with clamped reads, so clamped
// neighbours coincide with it); every thread computes the diffusivities of its own right and lower
// edge, the block's first column / first row of threads also those of the edges towards the
// neighbouring blocks (recomputed there with the same operations: the same bits), and the system is
// assembled from LDS.  sx, sy are written for the SOR sweeps; nothing is re-read from memory.
#define PREP_BX 64
#define PREP_BY 4
__device__ __forceinline__ float d_edge_x(const float (*sU)[PREP_BX + 2], const float (*sV)[PREP_BX + 2], int lx, int ly,
                                          float alpha)
{
    float ux = sU[ly][lx + 1] - sU[ly][lx];
    float vx = sV[ly][lx + 1] - sV[ly][lx];
    float uy = 0.25f * ((sU[ly + 1][lx] - sU[ly - 1][lx]) + (sU[ly + 1][lx + 1] - sU[ly - 1][lx + 1]));
    float vy = 0.25f * ((sV[ly + 1][lx] - sV[ly - 1][lx]) + (sV[ly + 1][lx + 1] - sV[ly - 1][lx + 1]));
    return alpha * d_psi(((ux * ux + uy * uy) + vx * vx) + vy * vy);
}
__device__ __forceinline__ float d_edge_y(const float (*sU)[PREP_BX + 2], const float (*sV)[PREP_BX + 2], int lx, int ly,
                                          float alpha)
{
    float uy = sU[ly + 1][lx] - sU[ly][lx];
    float vy = sV[ly + 1][lx] - sV[ly][lx];
    float ux = 0.25f * ((sU[ly][lx + 1] - sU[ly][lx - 1]) + (sU[ly + 1][lx + 1] - sU[ly + 1][lx - 1]));
    float vx = 0.25f * ((sV[ly][lx + 1] - sV[ly][lx - 1]) + (sV[ly + 1][lx + 1] - sV[ly + 1][lx - 1]));
    return alpha * d_psi(((ux * ux + uy * uy) + vx * vx) + vy * vy);
}

__global__ __launch_bounds__(PREP_BX * PREP_BY) void k_prepare(PrepIn in, Coef c, Geo g, float alpha, float gamma)
{
    __shared__ float sU[PREP_BY + 2][PREP_BX + 2];      // u + du
    __shared__ float sV[PREP_BY + 2][PREP_BX + 2];
    __shared__ float su0[PREP_BY + 2][PREP_BX + 2];     // u
    __shared__ float sv0[PREP_BY + 2][PREP_BX + 2];
    __shared__ float sSX[PREP_BY][PREP_BX + 1];         // sx of columns bx0-1 .. bx0+63
    __shared__ float sSY[PREP_BY + 1][PREP_BX];         // sy of rows by0-1 .. by0+3
    const int w = g.w, h = g.h, pitch = g.pitch;
    const size_t off = (size_t)blockIdx.z * g.plane;
    const int bx0 = blockIdx.x * PREP_BX, by0 = blockIdx.y * PREP_BY;
    const int tid = threadIdx.y * PREP_BX + threadIdx.x;
    // Every load of the workgroup is issued before the first wait: the staged ring first (two rounds of the 256 threads
    // over its 6 x 66 pixels), then the ten fields and du, dv of the thread's own pixel, which are only needed after
    // the diffusivities -- one round trip to memory per workgroup instead of three (a clamped address keeps the loads
    // of a thread outside the image unconditional; nothing of theirs is used).
    constexpr int NS = (PREP_BY + 2) * (PREP_BX + 2), NTH = PREP_BX * PREP_BY, NR = (NS + NTH - 1) / NTH;
    float r_u[NR], r_v[NR], r_du[NR], r_dv[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int i = min(tid + k * NTH, NS - 1);
        const int ly = i / (PREP_BX + 2), lx = i - ly * (PREP_BX + 2);
        const int gx = d_clampi(bx0 + lx - 1, 0, w - 1), gy = d_clampi(by0 + ly - 1, 0, h - 1);
        const size_t p = off + gy * pitch + gx;
        r_u[k] = in.u[p]; r_v[k] = in.v[p]; r_du[k] = in.du[p]; r_dv[k] = in.dv[p];
    }
    const int x = bx0 + threadIdx.x, y = by0 + threadIdx.y;
    const bool inside = x < w && y < h;
    const size_t p = off + min(y, h - 1) * pitch + min(x, w - 1);
    const float ddu = in.du[p], ddv = in.dv[p];
    const float ix = in.Ix[p], iy = in.Iy[p], iz = in.Iz[p];
    const float ixx = in.Ixx[p], ixy = in.Ixy[p], iyy = in.Iyy[p], ixz = in.Ixz[p], iyz = in.Iyz[p];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int i = tid + k * NTH;
        if (i < NS) {
            const int ly = i / (PREP_BX + 2), lx = i - ly * (PREP_BX + 2);
            su0[ly][lx] = r_u[k];
            sv0[ly][lx] = r_v[k];
            sU[ly][lx] = r_u[k] + r_du[k];
            sV[ly][lx] = r_v[k] + r_dv[k];
        }
    }
    __syncthreads();
    const int lx = threadIdx.x + 1, ly = threadIdx.y + 1;
    // diffusivities of the own right / lower edge (zero across the image border) ...
    float rsx = 0.0f, rsy = 0.0f;
    if (inside) {
        if (x + 1 < w) rsx = d_edge_x(sU, sV, lx, ly, alpha);
        if (y + 1 < h) rsy = d_edge_y(sU, sV, lx, ly, alpha);
    }
    sSX[threadIdx.y][threadIdx.x + 1] = rsx;
    sSY[threadIdx.y + 1][threadIdx.x] = rsy;
    // ... and of the edges towards the block on the left (first column of threads) and above (first row)
    if (threadIdx.x == 0) {
        float v = 0.0f;
        if (bx0 > 0 && y < h) v = d_edge_x(sU, sV, 0, ly, alpha);      // pixel (bx0-1, y): x + 1 = bx0 < w
        sSX[threadIdx.y][0] = v;
    }
    if (threadIdx.y == 0) {
        float v = 0.0f;
        if (by0 > 0 && x < w) v = d_edge_y(sU, sV, lx, 0, alpha);      // pixel (x, by0-1): y + 1 = by0 < h
        sSY[0][threadIdx.x] = v;
    }
    __syncthreads();
    if (!inside) return;
    c.sx[p] = rsx;
    c.sy[p] = rsy;
    float q0 = (iz + ix * ddu) + iy * ddv;
    float pd = d_psi(q0 * q0);
    float q1 = (ixz + ixx * ddu) + ixy * ddv;
    float q2 = (iyz + ixy * ddu) + iyy * ddv;
    float pg = gamma * d_psi(q1 * q1 + q2 * q2);
    float A11 = pd * (ix * ix) + pg * (ixx * ixx + ixy * ixy);
    float A12 = pd * (ix * iy) + pg * (ixx * ixy + ixy * iyy);
    float A22 = pd * (iy * iy) + pg * (ixy * ixy + iyy * iyy);
    float b1 = -(pd * (ix * iz) + pg * (ixx * ixz + ixy * iyz));
    float b2 = -(pd * (iy * iz) + pg * (ixy * ixz + iyy * iyz));
    float sl = x > 0 ? sSX[threadIdx.y][threadIdx.x] : 0.0f, sr = rsx;
    float st = y > 0 ? sSY[threadIdx.y][threadIdx.x] : 0.0f, sb = rsy;
    // clamped neighbours of u, v: the ring holds them
    float uc = su0[ly][lx], vc = sv0[ly][lx];
    float su = ((sl * (su0[ly][lx - 1] - uc) + sr * (su0[ly][lx + 1] - uc)) + st * (su0[ly - 1][lx] - uc)) + sb * (su0[ly + 1][lx] - uc);
    float sv = ((sl * (sv0[ly][lx - 1] - vc) + sr * (sv0[ly][lx + 1] - vc)) + st * (sv0[ly - 1][lx] - vc)) + sb * (sv0[ly + 1][lx] - vc);
    float ssum = ((sl + sr) + st) + sb;
    c.nu[p] = b1 + su;
    c.nv[p] = b2 + sv;
    c.a12[p] = A12;
    c.idu[p] = 1.0f / (A11 + ssum);
    c.idv[p] = 1.0f / (A22 + ssum);
}

// ---- u += du, v += dv ------------------------------------------------------------------
__global__ void k_add(float *__restrict__ u, float *__restrict__ v, const float *__restrict__ du,
                      const float *__restrict__ dv, Geo g)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    size_t p = (size_t)blockIdx.z * g.plane + y * g.pitch + x;
    u[p] = u[p] + du[p];
    v[p] = v[p] + dv[p];
}

// ---- red-black SOR, K iterations per launch (K is a launch argument) -----------------------------------------------
// One workgroup relaxes a TW x TH tile K times without going back to memory
// (overlapped temporal tiling: the tile carries a halo of 2K pixels per side,
// one ring of which becomes stale per half-sweep; only the interior is written
// back).  Red-black relaxation does not depend on the visiting order inside a
// colour, so the result is bit-identical to a whole-image sweep.
//
//  * the seven coefficient fields of a thread's pixels live in registers for all K
//    iterations (loaded once, coalesced float2 per pixel pair);
//  * du / dv live in LDS (interleaved as float2) in a checkerboard-compressed layout -- in each row the
//    pixels of colour 0 come first, then colour 1 -- so every access of a
//    half-sweep (own pixel and its four neighbours, which all have the other
//    colour) is unit-stride across lanes: no bank conflicts, all lanes active;
//  * du / dv are double-buffered in memory (a neighbouring tile reads the halo
//    this tile would otherwise overwrite).
struct SorArgs {
    const float *du_in, *dv_in;
    float *du_out, *dv_out;
    const float *nu, *nv, *a12, *idu, *idv, *sx, *sy;
    Geo g;
    int tiles_x, tiles_y;   // tile grid
    int step_x, step_y;     // interior size of a tile
    int halo_x, halo_y;     // 2K, or 0 when the image fits the tile on that axis
    float om, om1;          // omega, 1 - omega
};

// What a thread holds of the system of its pixels for all sweeps.  [row][pixel of the pair]; the weight towards the
// upper neighbour is the lower weight of the row above (own register except for the first row), the weight
// towards the left neighbour of the odd pixel is the right weight of the even one.
template <int RPT>
struct SorRegs {
    float nu[RPT][2], nv[RPT][2], a12[RPT][2], idu[RPT][2], idv[RPT][2];
    float sr[RPT][2], sb[RPT][2], sl0[RPT], st0[2];
};

// K red-black iterations of a tile held in s_uv (checkerboard-compressed rows: the pixels of colour 0 first,
// then colour 1; (du, dv) side by side) by NT threads, thread (grp, i) owning rows grp * RPT .. and the pixel
// pair 2i, 2i + 1 of each.  (gx0, gy0) = image coordinates of the tile's first pixel.  BORDER (uniform over the
// workgroup; BORDER_T its default) = false skips the image-border selects: a tile that does not touch the border never
// takes them.
//
// `own`: the thread's own pixels, [row][pixel of the pair], the values s_uv holds for them on entry and on exit.  A
// relaxed pixel's own value, its neighbour in the same pair (one of left / right is always the other pixel of the
// pair) and its neighbours in the thread's other rows come from these registers; LDS is read for the pixels of OTHER
// threads only -- per half-sweep and thread RPT reads to the side and two up / down instead of 5 RPT -- and written
// for every relaxed pixel as before.  (The sweeps of a tile are bound by what the workgroup puts through the CU's LDS
// pipe and the barrier behind it: 24 LDS operations per thread and half-sweep at four rows per thread, now 10.)  Same
// values, same arithmetic: at the edge of a tile the clamped LDS slots the reads fell back on are this thread's own
// pixels too, so they are read as before.
template <int TW, int TH, int NT, bool BORDER_T>
__device__ __forceinline__ void d_sor_sweeps(const SorRegs<TH / (NT / (TW / 2))> &R, float2 *s_uv, int K, int w, int h,
                                             int gx0, int gy0, float om, float om1, float2 (&own)[TH / (NT / (TW / 2))][2],
                                             const bool BORDER = BORDER_T)
{
    constexpr int HALF = TW / 2, NG = NT / HALF, RPT = TH / NG;
    const int grp = threadIdx.x / HALF, i = threadIdx.x - grp * HALF;
    const int r0 = grp * RPT, gxa = gx0 + 2 * i;
#pragma unroll 1
    for (int it = 0; it < K; it++) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            // the reads of other threads' pixels first (all of a half-sweep's in flight together), then the arithmetic
            float2 side[RPT], up, down;
#pragma unroll
            for (int j = 0; j < RPT; j++) {
                const int q = (j + c) & 1;
                const int r = r0 + j, oc = (1 - c) * HALF;
                int is = q ? i + 1 : i - 1;              // the neighbour outside the pair: right of the odd, left of the even pixel
                is = is < 0 ? 0 : (is > HALF - 1 ? HALF - 1 : is);
                side[j] = s_uv[r * TW + oc + is];
            }
            {
                const int oc = (1 - c) * HALF;
                const int ru = r0 > 0 ? r0 - 1 : 0, rd = r0 + RPT - 1 < TH - 1 ? r0 + RPT : TH - 1;
                up = s_uv[ru * TW + oc + i];
                down = s_uv[rd * TW + oc + i];
            }
#pragma unroll
            for (int j = 0; j < RPT; j++) {
                // colour c in row r: pixel x = 2i + q with q = (r + c) & 1 = (j + c) & 1
                const int q = (j + c) & 1;
                const int r = r0 + j, gy = gy0 + r, gx = gxa + q;
                const float2 pc = own[j][q];
                const float2 pl = q ? own[j][0] : side[j], pr = q ? side[j] : own[j][1];
                const float2 pt = j > 0 ? own[j > 0 ? j - 1 : 0][q] : up;
                const float2 pb = j < RPT - 1 ? own[j < RPT - 1 ? j + 1 : 0][q] : down;
                const float cu = pc.x, cv = pc.y;
                float ul = pl.x, ur = pr.x, ut = pt.x, ub = pb.x;
                float vl = pl.y, vr = pr.y, vt = pt.y, vb = pb.y;
                if (BORDER) {
                    // at the image border the oracle pairs the (zero) weight with the pixel itself
                    if (gx <= 0) { ul = cu; vl = cv; }
                    if (gx >= w - 1) { ur = cu; vr = cv; }
                    if (gy <= 0) { ut = cu; vt = cv; }
                    if (gy >= h - 1) { ub = cu; vb = cv; }
                }
                const float wl = q ? R.sr[j][0] : R.sl0[j];
                const float wt = j > 0 ? R.sb[j > 0 ? j - 1 : 0][q] : R.st0[q];
                const float su = ((wl * ul + R.sr[j][q] * ur) + wt * ut) + R.sb[j][q] * ub;
                const float sv = ((wl * vl + R.sr[j][q] * vr) + wt * vt) + R.sb[j][q] * vb;
                const float dun = om1 * cu + om * (((R.nu[j][q] - R.a12[j][q] * cv) + su) * R.idu[j][q]);
                const float dvn = om1 * cv + om * (((R.nv[j][q] - R.a12[j][q] * dun) + sv) * R.idv[j][q]);
                own[j][q] = make_float2(dun, dvn);
                s_uv[r * TW + c * HALF + i] = make_float2(dun, dvn);
            }
            __syncthreads();
        }
    }
}

// the tile `bid` of batch item `z`, relaxed K times by the calling workgroup (NT threads; s_uv: TH * TW float2 of LDS)
template <int TW, int TH, int NT>
__device__ __forceinline__ void d_sor_tile(const SorArgs &a, int K, int bid, int z, float2 *s_uv)
{
    constexpr int HALF = TW / 2;             // pixels of one colour per row
    constexpr int NG = NT / HALF;            // lane groups, each owns whole rows
    constexpr int RPT = TH / NG;             // rows per thread
    static_assert(TH % NG == 0 && RPT % 2 == 0, "rows per thread must be even");
    const int ty = bid / a.tiles_x, tx = bid - ty * a.tiles_x;
    const int w = a.g.w, h = a.g.h, pitch = a.g.pitch;
    const size_t off = (size_t)z * a.g.plane;
    const int gx0 = tx * a.step_x - a.halo_x;      // even
    const int gy0 = ty * a.step_y - a.halo_y;      // even

    const int grp = threadIdx.x / HALF, i = threadIdx.x - grp * HALF;
    const int r0 = grp * RPT;                      // even
    const int gxa = gx0 + 2 * i;                   // global x of the even pixel of the pair

    SorRegs<RPT> R;
    R.st0[0] = 0.0f; R.st0[1] = 0.0f;
    float2 own[RPT][2];                            // (du, dv) of the thread's pixels: d_sor_sweeps

#pragma unroll
    for (int j = 0; j < RPT; j++) {
        const int r = r0 + j, gy = gy0 + r;
        const bool rowok = gy >= 0 && gy < h;
        const bool ok0 = rowok && gxa >= 0 && gxa < w;      // gxa is even: ok1 implies ok0
        const bool ok1 = rowok && gxa + 1 >= 0 && gxa + 1 < w;
        float2 z = make_float2(0.0f, 0.0f);
        float2 f_nu = z, f_nv = z, f_a = z, f_iu = z, f_iv = z, f_sx = z, f_sy = z, f_du = z, f_dv = z;
        float f_sl = 0.0f;
        if (ok0) {   // gxa even, pitch even: the pair is 8-byte aligned and inside the row's pitch
            const size_t p = off + (size_t)gy * pitch + gxa;
            f_nu = *(const float2 *)(a.nu + p);
            f_nv = *(const float2 *)(a.nv + p);
            f_a = *(const float2 *)(a.a12 + p);
            f_iu = *(const float2 *)(a.idu + p);
            f_iv = *(const float2 *)(a.idv + p);
            f_sx = *(const float2 *)(a.sx + p);
            f_sy = *(const float2 *)(a.sy + p);
            f_du = *(const float2 *)(a.du_in + p);
            f_dv = *(const float2 *)(a.dv_in + p);
            if (gxa > 0) f_sl = a.sx[p - 1];
            if (j == 0 && gy > 0) {
                float2 t = *(const float2 *)(a.sy + p - pitch);
                R.st0[0] = t.x;
                R.st0[1] = ok1 ? t.y : 0.0f;
            }
        }
        // a pixel outside the image gets an all-zero system: it stays finite and is never used
        R.nu[j][0] = f_nu.x; R.nv[j][0] = f_nv.x; R.a12[j][0] = f_a.x; R.idu[j][0] = f_iu.x; R.idv[j][0] = f_iv.x;
        R.sr[j][0] = f_sx.x; R.sb[j][0] = f_sy.x; R.sl0[j] = f_sl;
        R.nu[j][1] = ok1 ? f_nu.y : 0.0f; R.nv[j][1] = ok1 ? f_nv.y : 0.0f; R.a12[j][1] = ok1 ? f_a.y : 0.0f;
        R.idu[j][1] = ok1 ? f_iu.y : 0.0f; R.idv[j][1] = ok1 ? f_iv.y : 0.0f;
        R.sr[j][1] = ok1 ? f_sx.y : 0.0f; R.sb[j][1] = ok1 ? f_sy.y : 0.0f;
        // checkerboard-compressed LDS row: colour of the even pixel is (r & 1) == (j & 1)
        const int c0 = j & 1;
        own[j][0] = make_float2(f_du.x, f_dv.x);
        own[j][1] = ok1 ? make_float2(f_du.y, f_dv.y) : make_float2(0.0f, 0.0f);
        s_uv[r * TW + c0 * HALF + i] = own[j][0];
        s_uv[r * TW + (1 - c0) * HALF + i] = own[j][1];
    }
    __syncthreads();

    // A tile that does not touch the image border needs none of the border selects (they never fire).
    const bool touches = gx0 <= 0 || gy0 <= 0 || gx0 + TW >= w || gy0 + TH >= h;
    // write back the interior (from the registers the sweeps leave the thread's pixels in)
    const int lx = 2 * i;
    const bool colin = lx >= a.halo_x && lx < a.halo_x + a.step_x;
    auto write_back = [&]() {
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const int r = r0 + j, gy = gy0 + r;
            if (!colin || r < a.halo_y || r >= a.halo_y + a.step_y || gy >= h || gxa >= w) continue;
            const size_t p = off + (size_t)gy * pitch + gxa;
            const float2 q0 = own[j][0], q1 = own[j][1];
            const float u0 = q0.x, u1 = q1.x, v0 = q0.y, v1 = q1.y;
            if (gxa + 1 < w) {
                *(float2 *)(a.du_out + p) = make_float2(u0, u1);
                *(float2 *)(a.dv_out + p) = make_float2(v0, v1);
            } else {
                a.du_out[p] = u0;
                a.dv_out[p] = v0;
            }
        }
    };
    // ONE loop, the border selects behind a branch that is uniform over the workgroup (an interior tile skips them: a
    // quarter fewer vector instructions).  As two instantiations of the loop, one per kind of tile, the kernel needed 210
    // registers once the thread's pixels were carried through them (what either loop keeps invariant is hoisted above the
    // branch between them) -- 52 to 80 spills at the 128 of two workgroups per CU; this form takes 114.
    d_sor_sweeps<TW, TH, NT, true>(R, s_uv, K, w, h, gx0, gy0, a.om, a.om1, own, __builtin_amdgcn_readfirstlane(touches));
    write_back();
}

// (the 1 024-thread form of the 64 x 64 tile is held to the 64 registers of two workgroups per CU: 66 -> 64, two spills)
template <int TW, int TH, int NT>
__global__ __launch_bounds__(NT, (NT == 1024 && TW == 64) ? 8 : 1) void k_sor(SorArgs a, int K)
{
    __shared__ float2 s_uv[TH * TW];          // (du, dv) of a pixel side by side: one 8-byte LDS access each
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs, so
    // give each XCD a contiguous run of tiles (neighbouring tiles share halo lines
    // in that XCD's L2).  Speed only.
    const int nt = a.tiles_x * a.tiles_y;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, k = bid >> 3, q = nt >> 3, rem = nt & 7;
        bid = xcd * q + (xcd < rem ? xcd : rem) + k;   // bijection on [0, nt)
    }
    d_sor_tile<TW, TH, NT>(a, K, bid, blockIdx.z, s_uv);
}

// ---- the coarse end of the pyramid in one launch -----------------------------------------------------------------
// A level of at most 64 x 64 pixels is one SOR tile: one workgroup can take a pair through the whole level --
// derivatives, warp, `inner` x (linear system, `solver` red-black iterations), prolongation -- and through every
// further level that still fits, without going back to the host.  Launched per level that is 2 + 2 * inner + 1
// launches of ~3 us each for microseconds of work (7 levels of a 1024^2 pyramid: 161 of ~650 launches).
//
// One workgroup of T x T / 4 threads per pair (T = 32 for levels up to 32 x 32 px, else 64); a thread owns the
// 2 x 2 pixels (2i + q, 2 grp + j) -- the SOR tile's own layout, so the system goes from the registers that assemble it straight into the sweeps.  u, v, u + du,
// v + dv and the edge diffusivities live in LDS as T x T planes, du / dv in the sweeps' checkerboard layout;
// the warped data terms stay in registers for the whole level.  Global memory sees, per level, the reads of the
// two pyramid images and one write + read of the derivative images (the warp samples those of frame 1 at
// arbitrary positions).  Every value is computed with the operations, in the order, of k_deriv_all, k_warp,
// k_prepare, k_sor and k_add_prolong / k_add_out: the same bits.
#define COARSE_MAX 32
struct CoarseArgs {
    int nlev;                                   // levels of this launch, coarsest first
    Geo g[COARSE_MAX];
    const float *I0[COARSE_MAX], *I1[COARSE_MAX];
    const float *u_in, *v_in;                   // u, v of g[0] (pitched planes)
    float *Ix0, *Iy0, *I1x, *I1y, *I1xx, *I1xy, *I1yy;      // scratch planes,
    int scratch_plane;                                        // ... floats per pair in them (>= every g[].plane)
    int stagger;                                // tests: pairs 8.. wait ~0.3 ms between writing and reading their scratch
    Geo gout;                                   // the next finer level; w == 0: g[nlev - 1] is the image itself
    float *u_out, *v_out;                       // pitched planes of gout, or the caller's tight W x H arrays
    int inner, solver;
    float alpha, gamma, om, om1;
};

template <int T>
__device__ __forceinline__ int d_cb(int x, int y) { return y * T + ((x + y) & 1) * (T / 2) + (x >> 1); }   // checkerboard slot

template <int T>
__device__ __forceinline__ float d_co_edge_x(const float (*U)[T], const float (*V)[T], int x, int y, int h, float alpha)
{
    const int ym = y > 0 ? y - 1 : 0, yp = y + 1 < h ? y + 1 : h - 1;
    float ux = U[y][x + 1] - U[y][x];
    float vx = V[y][x + 1] - V[y][x];
    float uy = 0.25f * ((U[yp][x] - U[ym][x]) + (U[yp][x + 1] - U[ym][x + 1]));
    float vy = 0.25f * ((V[yp][x] - V[ym][x]) + (V[yp][x + 1] - V[ym][x + 1]));
    return alpha * d_psi(((ux * ux + uy * uy) + vx * vx) + vy * vy);
}
template <int T>
__device__ __forceinline__ float d_co_edge_y(const float (*U)[T], const float (*V)[T], int x, int y, int w, float alpha)
{
    const int xm = x > 0 ? x - 1 : 0, xp = x + 1 < w ? x + 1 : w - 1;
    float uy = U[y + 1][x] - U[y][x];
    float vy = V[y + 1][x] - V[y][x];
    float ux = 0.25f * ((U[y][xp] - U[y][xm]) + (U[y + 1][xp] - U[y + 1][xm]));
    float vx = 0.25f * ((V[y][xp] - V[y][xm]) + (V[y + 1][xp] - V[y + 1][xm]));
    return alpha * d_psi(((ux * ux + uy * uy) + vx * vx) + vy * vy);
}

// d_bilin_sum with u in an LDS plane and du / dv in the checkerboard tile
template <int T, int COMP>
__device__ __forceinline__ float d_co_bilin_sum(const float (*a)[T], const float2 *s_uv, int w, int h, float px, float py)
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
    const float2 d00 = s_uv[d_cb<T>(x0, y0)], d01 = s_uv[d_cb<T>(x1, y0)], d10 = s_uv[d_cb<T>(x0, y1)], d11 = s_uv[d_cb<T>(x1, y1)];
    float q00 = a[y0][x0] + (COMP ? d00.y : d00.x), q01 = a[y0][x1] + (COMP ? d01.y : d01.x);
    float q10 = a[y1][x0] + (COMP ? d10.y : d10.x), q11 = a[y1][x1] + (COMP ? d11.y : d11.x);
    float top = (1.0f - ax) * q00 + ax * q01;
    float bot = (1.0f - ax) * q10 + ax * q11;
    return (1.0f - ay) * top + ay * bot;
}

template <int T>
__global__ __launch_bounds__(T * T / 4) void k_coarse(CoarseArgs a)
{
    constexpr int NT = T * T / 4, RPT = 2, HALF = T / 2;
    __shared__ float2 s_uv[T * T];
    __shared__ float su0[T][T], sv0[T][T];          // u, v
    __shared__ float sU[T][T], sV[T][T];            // u + du, v + dv
    __shared__ float sSX[T][T], sSY[T][T];          // diffusivity of the right / lower edge of a pixel
    const int z = blockIdx.x;
    const int grp = threadIdx.x / HALF, i = threadIdx.x - grp * HALF;
    const int xa = 2 * i, ya = RPT * grp;
    {
        const Geo g = a.g[0];
        const size_t off = (size_t)z * g.plane;
#pragma unroll
        for (int j = 0; j < RPT; j++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int x = xa + q, y = ya + j;
                const bool in = x < g.w && y < g.h;
                su0[y][x] = in ? a.u_in[off + y * g.pitch + x] : 0.0f;
                sv0[y][x] = in ? a.v_in[off + y * g.pitch + x] : 0.0f;
            }
    }
    for (int lv = 0; lv < a.nlev; lv++) {
        const Geo g = a.g[lv];
        const int w = g.w, h = g.h, pitch = g.pitch;
        const size_t off = (size_t)z * g.plane;
        // The derivative planes are this workgroup's own scratch: one fixed slab per pair.  (The per-level offset
        // z * plane that the launch-per-operator path uses -- every pair at the same level at any time -- would let
        // the slab of pair z at one level overlap that of pair z + 1 at the next while the two workgroups are at
        // different levels.)
        const size_t offs = (size_t)z * a.scratch_plane;
        const float *I0 = a.I0[lv] + off, *I1 = a.I1[lv] + off;
        // derivative images (k_deriv_all), pixel by pixel: nothing of this phase stays in registers
#pragma unroll 1
        for (int pp = 0; pp < 2 * RPT; pp++) {
            const int x = xa + (pp & 1), y = ya + (pp >> 1);
            s_uv[d_cb<T>(x, y)] = make_float2(0.0f, 0.0f);          // du = dv = 0 at the start of a level
            if (x >= w || y >= h) continue;
            const size_t p = offs + y * pitch + x;
            a.Ix0[p] = d_dx_at(I0, g, x, y);
            a.Iy0[p] = d_dy_at(I0, g, x, y);
            a.I1x[p] = d_dx_at(I1, g, x, y);
            a.I1y[p] = d_dy_at(I1, g, x, y);
            const int xm2 = d_mirror(x - 2, w), xm1 = d_mirror(x - 1, w), xp1 = d_mirror(x + 1, w), xp2 = d_mirror(x + 2, w);
            const int ym2 = d_mirror(y - 2, h), ym1 = d_mirror(y - 1, h), yp1 = d_mirror(y + 1, h), yp2 = d_mirror(y + 2, h);
            a.I1xx[p] = d_d5(d_dx_at(I1, g, xm2, y), d_dx_at(I1, g, xm1, y), d_dx_at(I1, g, xp1, y), d_dx_at(I1, g, xp2, y));
            a.I1xy[p] = d_d5(d_dx_at(I1, g, x, ym2), d_dx_at(I1, g, x, ym1), d_dx_at(I1, g, x, yp1), d_dx_at(I1, g, x, yp2));
            a.I1yy[p] = d_d5(d_dy_at(I1, g, x, ym2), d_dy_at(I1, g, x, ym1), d_dy_at(I1, g, x, yp1), d_dy_at(I1, g, x, yp2));
        }
        __syncthreads();                 // the derivative images are this workgroup's own writes: visible after the barrier
        if (a.stagger && z >= 8) {       // tests: pairs 8.. linger here while pairs 0..7 run levels ahead, then
            for (int k = 0; k < 96; k++) __builtin_amdgcn_s_sleep(127);
            asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");   // ... read what memory holds, not their L1
            __syncthreads();
        }
        // warp (k_warp): the data terms of the own pixels, kept in registers for the level
        float Iz[RPT][2], Ix[RPT][2], Iy[RPT][2], Ixz[RPT][2], Iyz[RPT][2], Ixx[RPT][2], Ixy[RPT][2], Iyy[RPT][2];
#pragma unroll
        for (int j = 0; j < RPT; j++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int x = xa + q, y = ya + j;
                Iz[j][q] = 0.0f; Ix[j][q] = 0.0f; Iy[j][q] = 0.0f; Ixz[j][q] = 0.0f; Iyz[j][q] = 0.0f;
                Ixx[j][q] = 0.0f; Ixy[j][q] = 0.0f; Iyy[j][q] = 0.0f;
                if (x >= w || y >= h) continue;
                const float px = (float)x + su0[y][x], py = (float)y + sv0[y][x];
                if (px < 0.0f || py < 0.0f || px > (float)(w - 1) || py > (float)(h - 1)) continue;
                const float i1 = d_bilin(I1, w, h, pitch, px, py);
                const float ix = d_bilin(a.I1x + offs, w, h, pitch, px, py);
                const float iy = d_bilin(a.I1y + offs, w, h, pitch, px, py);
                Iz[j][q] = i1 - I0[y * pitch + x];
                Ix[j][q] = ix;
                Iy[j][q] = iy;
                if (x < 2 || y < 2 || x > w - 3 || y > h - 3 ||
                    px < 2.0f || py < 2.0f || px > (float)(w - 4) || py > (float)(h - 4)) continue;
                Ixz[j][q] = ix - a.Ix0[offs + y * pitch + x];
                Iyz[j][q] = iy - a.Iy0[offs + y * pitch + x];
                Ixx[j][q] = d_bilin(a.I1xx + offs, w, h, pitch, px, py);
                Ixy[j][q] = d_bilin(a.I1xy + offs, w, h, pitch, px, py);
                Iyy[j][q] = d_bilin(a.I1yy + offs, w, h, pitch, px, py);
            }
        for (int it = 0; it < a.inner; it++) {
            // u + du, v + dv (k_prepare's staging)
#pragma unroll
            for (int j = 0; j < RPT; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int x = xa + q, y = ya + j;
                    if (x >= w || y >= h) continue;
                    const float2 d = s_uv[d_cb<T>(x, y)];
                    sU[y][x] = su0[y][x] + d.x;
                    sV[y][x] = sv0[y][x] + d.y;
                }
            __syncthreads();
            SorRegs<RPT> R;
#pragma unroll
            for (int j = 0; j < RPT; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int x = xa + q, y = ya + j;
                    float rsx = 0.0f, rsy = 0.0f;
                    if (x < w && y < h) {
                        if (x + 1 < w) rsx = d_co_edge_x<T>(sU, sV, x, y, h, a.alpha);
                        if (y + 1 < h) rsy = d_co_edge_y<T>(sU, sV, x, y, w, a.alpha);
                    }
                    sSX[y][x] = rsx;
                    sSY[y][x] = rsy;
                    R.sr[j][q] = rsx;
                    R.sb[j][q] = rsy;
                }
            __syncthreads();
            // the 2 x 2 system of every own pixel (k_prepare), into the registers of the sweeps (k_sor's loads)
#pragma unroll
            for (int j = 0; j < RPT; j++) {
                const int y = ya + j;
                R.sl0[j] = (xa < w && y < h && xa > 0) ? sSX[y][xa - 1] : 0.0f;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int x = xa + q;
                    if (j == 0) R.st0[q] = (x < w && y < h && y > 0) ? sSY[y - 1][x] : 0.0f;
                    R.nu[j][q] = 0.0f; R.nv[j][q] = 0.0f; R.a12[j][q] = 0.0f; R.idu[j][q] = 0.0f; R.idv[j][q] = 0.0f;
                    if (x >= w || y >= h) continue;
                    const float2 d = s_uv[d_cb<T>(x, y)];
                    const float ddu = d.x, ddv = d.y;
                    const float ix = Ix[j][q], iy = Iy[j][q], iz = Iz[j][q];
                    const float ixx = Ixx[j][q], ixy = Ixy[j][q], iyy = Iyy[j][q], ixz = Ixz[j][q], iyz = Iyz[j][q];
                    float q0 = (iz + ix * ddu) + iy * ddv;
                    float pd = d_psi(q0 * q0);
                    float q1 = (ixz + ixx * ddu) + ixy * ddv;
                    float q2 = (iyz + ixy * ddu) + iyy * ddv;
                    float pg = a.gamma * d_psi(q1 * q1 + q2 * q2);
                    float A11 = pd * (ix * ix) + pg * (ixx * ixx + ixy * ixy);
                    float A12 = pd * (ix * iy) + pg * (ixx * ixy + ixy * iyy);
                    float A22 = pd * (iy * iy) + pg * (ixy * ixy + iyy * iyy);
                    float b1 = -(pd * (ix * iz) + pg * (ixx * ixz + ixy * iyz));
                    float b2 = -(pd * (iy * iz) + pg * (ixy * ixz + iyy * iyz));
                    const int xm = x > 0 ? x - 1 : 0, xp = x + 1 < w ? x + 1 : w - 1;
                    const int ym = y > 0 ? y - 1 : 0, yp = y + 1 < h ? y + 1 : h - 1;
                    float sl = x > 0 ? sSX[y][x - 1] : 0.0f, sr = R.sr[j][q];
                    float st = y > 0 ? sSY[y - 1][x] : 0.0f, sb = R.sb[j][q];
                    float uc = su0[y][x], vc = sv0[y][x];
                    float su = ((sl * (su0[y][xm] - uc) + sr * (su0[y][xp] - uc)) + st * (su0[ym][x] - uc)) + sb * (su0[yp][x] - uc);
                    float sv = ((sl * (sv0[y][xm] - vc) + sr * (sv0[y][xp] - vc)) + st * (sv0[ym][x] - vc)) + sb * (sv0[yp][x] - vc);
                    float ssum = ((sl + sr) + st) + sb;
                    R.nu[j][q] = b1 + su;
                    R.nv[j][q] = b2 + sv;
                    R.a12[j][q] = A12;
                    R.idu[j][q] = 1.0f / (A11 + ssum);
                    R.idv[j][q] = 1.0f / (A22 + ssum);
                }
            }
            float2 own[RPT][2];                    // the thread's pixels as the planes hold them (all threads are past the
#pragma unroll                                     // barrier behind the last writes of s_uv)
            for (int j = 0; j < RPT; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) own[j][q] = s_uv[d_cb<T>(xa + q, ya + j)];
            d_sor_sweeps<T, T, NT, true>(R, s_uv, a.solver, w, h, 0, 0, a.om, a.om1, own);   // ends on a barrier
        }
        // u + du, v + dv carried to the next finer level (k_add_prolong) or out (k_add_out)
        if (lv + 1 < a.nlev) {
            const Geo gd = a.g[lv + 1];
            const float rx = (float)w / (float)gd.w, ry = (float)h / (float)gd.h;
            const float mulx = (float)gd.w / (float)w, muly = (float)gd.h / (float)h;
            float nu_[RPT][2], nv_[RPT][2];
#pragma unroll
            for (int j = 0; j < RPT; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int x = xa + q, y = ya + j;
                    nu_[j][q] = 0.0f; nv_[j][q] = 0.0f;
                    if (x >= gd.w || y >= gd.h) continue;
                    const float sx = ((float)x + 0.5f) * rx - 0.5f;
                    const float sy = ((float)y + 0.5f) * ry - 0.5f;
                    nu_[j][q] = d_co_bilin_sum<T, 0>(su0, s_uv, w, h, sx, sy) * mulx;
                    nv_[j][q] = d_co_bilin_sum<T, 1>(sv0, s_uv, w, h, sx, sy) * muly;
                }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < RPT; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    su0[ya + j][xa + q] = nu_[j][q];
                    sv0[ya + j][xa + q] = nv_[j][q];
                }
            __syncthreads();
        } else if (a.gout.w > 0) {
            const Geo gd = a.gout;
            const float rx = (float)w / (float)gd.w, ry = (float)h / (float)gd.h;
            const float mulx = (float)gd.w / (float)w, muly = (float)gd.h / (float)h;
            const size_t offd = (size_t)z * gd.plane;
            for (int idx = threadIdx.x; idx < gd.w * gd.h; idx += NT) {
                const int y = idx / gd.w, x = idx - y * gd.w;
                const float sx = ((float)x + 0.5f) * rx - 0.5f;
                const float sy = ((float)y + 0.5f) * ry - 0.5f;
                a.u_out[offd + y * gd.pitch + x] = d_co_bilin_sum<T, 0>(su0, s_uv, w, h, sx, sy) * mulx;
                a.v_out[offd + y * gd.pitch + x] = d_co_bilin_sum<T, 1>(sv0, s_uv, w, h, sx, sy) * muly;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPT; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int x = xa + q, y = ya + j;
                    if (x >= w || y >= h) continue;
                    const float2 d = s_uv[d_cb<T>(x, y)];
                    const size_t o = ((size_t)z * h + y) * w + x;
                    a.u_out[o] = su0[y][x] + d.x;
                    a.v_out[o] = sv0[y][x] + d.y;
                }
        }
    }
}
