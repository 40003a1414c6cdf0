// Device kernels of the EKF measurement model (gfx950): software rasteriser of the
// textured mesh (replaces the OpenGL passes of reference renderer.py:310-325) and
// the fused perturb-and-reduce kernels (replace reference cuda.py / cuda_multi.py).
//
// Raster rules (identical to oracle/ekf_ref.py): vertices snapped to 1/256 px,
// exact integer edge functions with a top-left tie-break, pixel centre sampling,
// binary32 plane-equation interpolation a0 + l1 (a1-a0) + l2 (a2-a0), nearest
// texel of the initial frame, additive blending with 8-bit saturation, no culling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EKF_SUB 256
#define EKF_MAX_STAR 24        // triangles around one vertex
#define EKF_TILE 16
#define EKF_MAX_TRI 4096

struct TriSetup {              // one triangle in one configuration
    // edge function E_k = ea*c + eb*r + ec at pixel (col c, row r).  Whole numbers kept as doubles:
    // for coordinates within +-2^24 px every product and sum stays below 2^53, so binary64 evaluates
    // them exactly -- the same values as 64-bit integers at the cost of two full-rate v_fma_f64
    // (64-bit integer multiplies are built from quarter-rate 32-bit ones on gfx950).
    double ea[3], eb[3], ec[3];
    // ec + (1 if a zero edge value counts as inside, the top-left rule): E is a whole number, so
    // "E > 0 or (E == 0 and top-left)" is the single comparison ea*c + eb*r + ecb > 0
    double ecb[3];
    int cmin, cmax, rmin, rmax;  // pixel bounding box (inclusive), empty if cmin > cmax (16-byte aligned: one scalar load)
    int tl[3];                 // 1 if a zero edge value counts as inside (top-left edge)
    float inv;                 // 1 / (2 area)
    int i0, i1, i2;            // vertex ids after orientation normalisation
    // attributes of the three vertices in that order: texture coordinates (pixels of the initial
    // frame) and the two velocity render attributes vx, -vy.  Kept here so that a covered pixel
    // needs no dependent global loads besides its texel.
    float ux[3], uy[3], ax[3], ay[3];
};

__device__ __forceinline__ void d_tri_attr(TriSetup &s, const float *__restrict__ uv, const double *__restrict__ X, int N)
{
    const int id[3] = {s.i0, s.i1, s.i2};
    for (int k = 0; k < 3; k++) {
        s.ux[k] = uv[2 * id[k]];
        s.uy[k] = uv[2 * id[k] + 1];
        s.ax[k] = (float)X[2 * N + 2 * id[k]];
        s.ay[k] = (float)(-X[2 * N + 2 * id[k] + 1]);
    }
}

__device__ __forceinline__ long long d_snap(double x) { return (long long)rint(x * (double)EKF_SUB); }

// Pixel bounding box (inclusive) of the triangle with snapped integer positions, as d_tri_setup keeps it: empty
// (cmin > cmax) for a degenerate or out-of-range triangle.  One function for every caller: a tile that asks "can
// this triangle reach me" gets the answer the setup itself would give.
__device__ __forceinline__ bool d_tri_sane(long long x0, long long y0, long long x1, long long y1, long long x2, long long y2)
{
    const long long lim = (long long)1 << 32;     // 2^24 px in 1/256 px units: the exact range of the edge functions
    return x0 > -lim && x0 < lim && y0 > -lim && y0 < lim && x1 > -lim && x1 < lim && y1 > -lim && y1 < lim &&
           x2 > -lim && x2 < lim && y2 > -lim && y2 < lim;
}
__device__ __forceinline__ void d_tri_bbox(long long x0, long long y0, long long x1, long long y1, long long x2, long long y2,
                                           int W, int H, int &cmin, int &cmax, int &rmin, int &rmax)
{
    cmin = 1; cmax = 0; rmin = 1; rmax = 0;
    const long long area = d_tri_sane(x0, y0, x1, y1, x2, y2) ? (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0) : 0;
    if (area == 0) return;
    long long xmin = x0 < x1 ? (x0 < x2 ? x0 : x2) : (x1 < x2 ? x1 : x2);
    long long xmax = x0 > x1 ? (x0 > x2 ? x0 : x2) : (x1 > x2 ? x1 : x2);
    long long ymin = y0 < y1 ? (y0 < y2 ? y0 : y2) : (y1 < y2 ? y1 : y2);
    long long ymax = y0 > y1 ? (y0 > y2 ? y0 : y2) : (y1 > y2 ? y1 : y2);
    // floor division by 256 (arithmetic shift), as in the oracle
    long long cl = (xmin - 128) >> 8, ch = ((xmax - 128) >> 8) + 1;
    long long rl = (ymin - 128) >> 8, rh = ((ymax - 128) >> 8) + 1;
    if (cl < 0) cl = 0;
    if (rl < 0) rl = 0;
    if (ch > W - 1) ch = W - 1;
    if (rh > H - 1) rh = H - 1;
    cmin = (int)cl; cmax = (int)ch; rmin = (int)rl; rmax = (int)rh;
}

// Build the setup of triangle (v0,v1,v2) with snapped integer positions p[3][2].
__device__ inline void d_tri_setup(TriSetup &s, int v0, int v1, int v2, long long x0, long long y0,
                                   long long x1, long long y1, long long x2, long long y2, int W, int H)
{
    s.cmin = 1; s.cmax = 0; s.rmin = 1; s.rmax = 0;
    const bool sane = d_tri_sane(x0, y0, x1, y1, x2, y2);
    long long area = sane ? (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0) : 0;
    s.i0 = v0; s.i1 = v1; s.i2 = v2;
    s.inv = 0.0f;
    for (int k = 0; k < 3; k++) { s.ea[k] = 0.0; s.eb[k] = 0.0; s.ec[k] = -1.0; s.ecb[k] = -1.0; s.tl[k] = 0; }
    if (area == 0) return;
    d_tri_bbox(x0, y0, x1, y1, x2, y2, W, H, s.cmin, s.cmax, s.rmin, s.rmax);      // (the box does not depend on the orientation)
    if (area < 0) {
        long long t;
        t = x1; x1 = x2; x2 = t;
        t = y1; y1 = y2; y2 = t;
        s.i1 = v2; s.i2 = v1;
        area = -area;
    }
    // E0: edge 1->2 (weight of vertex 0), E1: edge 2->0, E2: edge 0->1
    const long long ex[3] = {x2 - x1, x0 - x2, x1 - x0};
    const long long ey[3] = {y2 - y1, y0 - y2, y1 - y0};
    const long long ox[3] = {x1, x2, x0};
    const long long oy[3] = {y1, y2, y0};
    for (int k = 0; k < 3; k++) {
        // E(px,py) = ex*(py-oy) - ey*(px-ox), px = 256 c + 128, py = 256 r + 128
        s.ea[k] = (double)(-ey[k] * EKF_SUB);
        s.eb[k] = (double)(ex[k] * EKF_SUB);
        s.ec[k] = (double)(ex[k] * (128 - oy[k]) - ey[k] * (128 - ox[k]));
        s.tl[k] = (ey[k] > 0) || (ey[k] == 0 && ex[k] < 0);
        s.ecb[k] = s.ec[k] + (double)s.tl[k];
    }
    s.inv = 1.0f / (float)area;
}

// coverage of the pixel centre (dc, dr) without the bounding-box shortcut and without branches (for
// pixels inside the frame the three edge tests imply the box); barycentrics separately
__device__ __forceinline__ bool d_tri_cover(const TriSetup &s, double dc, double dr)
{
    const double e0 = fma(s.ea[0], dc, fma(s.eb[0], dr, s.ecb[0]));
    const double e1 = fma(s.ea[1], dc, fma(s.eb[1], dr, s.ecb[1]));
    const double e2 = fma(s.ea[2], dc, fma(s.eb[2], dr, s.ecb[2]));
    return (e0 > 0.0) & (e1 > 0.0) & (e2 > 0.0);
}
// the same, handing back the values of edges 1 and 2 (with the top-left bias in: ecb): the barycentrics of a covered pixel
// follow from them by taking the bias out again -- whole numbers below 2^53, so e - bias IS ea c + eb r + ec, exactly
__device__ __forceinline__ bool d_tri_cover2(const TriSetup &s, double dc, double dr, double &e1, double &e2)
{
    const double e0 = fma(s.ea[0], dc, fma(s.eb[0], dr, s.ecb[0]));
    e1 = fma(s.ea[1], dc, fma(s.eb[1], dr, s.ecb[1]));
    e2 = fma(s.ea[2], dc, fma(s.eb[2], dr, s.ecb[2]));
    return (e0 > 0.0) & (e1 > 0.0) & (e2 > 0.0);
}
__device__ __forceinline__ void d_tri_bary2(const TriSetup &s, double e1, double e2, float &l1, float &l2)
{
    l1 = (float)(e1 - (s.tl[1] ? 1.0 : 0.0)) * s.inv;
    l2 = (float)(e2 - (s.tl[2] ? 1.0 : 0.0)) * s.inv;
}
__device__ __forceinline__ void d_tri_bary(const TriSetup &s, double dc, double dr, float &l1, float &l2)
{
    l1 = (float)fma(s.ea[1], dc, fma(s.eb[1], dr, s.ec[1])) * s.inv;
    l2 = (float)fma(s.ea[2], dc, fma(s.eb[2], dr, s.ec[2])) * s.inv;
}

// coverage + barycentrics of pixel (c, r)
__device__ __forceinline__ bool d_tri_eval(const TriSetup &s, int c, int r, float &l1, float &l2)
{
    if (c < s.cmin || c > s.cmax || r < s.rmin || r > s.rmax) return false;
    const double dc = (double)c, dr = (double)r;
    const double e0 = fma(s.ea[0], dc, fma(s.eb[0], dr, s.ec[0]));      // exact: see TriSetup
    const double e1 = fma(s.ea[1], dc, fma(s.eb[1], dr, s.ec[1]));
    const double e2 = fma(s.ea[2], dc, fma(s.eb[2], dr, s.ec[2]));
    bool in = (e0 > 0 || (e0 == 0 && s.tl[0])) && (e1 > 0 || (e1 == 0 && s.tl[1])) &&
              (e2 > 0 || (e2 == 0 && s.tl[2]));
    if (!in) return false;
    l1 = (float)e1 * s.inv;
    l2 = (float)e2 * s.inv;
    return true;
}

__device__ __forceinline__ float d_lerp(float a0, float a1, float a2, float l1, float l2)
{
    return (a0 + l1 * (a1 - a0)) + l2 * (a2 - a0);
}

// offset of the nearest texel in the initial frame
__device__ __forceinline__ int d_texel_at(const TriSetup &s, float l1, float l2, int W, int H)
{
    float tx = d_lerp(s.ux[0], s.ux[1], s.ux[2], l1, l2);
    float ty = d_lerp(s.uy[0], s.uy[1], s.uy[2], l1, l2);
    int cx = (int)floorf(tx), cy = (int)floorf(ty);
    cx = cx < 0 ? 0 : (cx > W - 1 ? W - 1 : cx);
    cy = cy < 0 ? 0 : (cy > H - 1 ? H - 1 : cy);
    return cy * W + cx;
}
__device__ __forceinline__ int d_texel(const uint8_t *__restrict__ tex, const TriSetup &s, float l1, float l2, int W,
                                       int H)
{
    return tex[d_texel_at(s, l1, l2, W, H)];
}

struct Mesh {
    int W, H, N, T;
    const int *tri;           // T*3
    const float *uv;          // N*2
    const uint8_t *tex;       // W*H
};

// unclamped render targets: im = min(255, acc), m = cnt > 0 ? 255 : 0
struct Targets {
    int *acc;                 // sum of texels
    float *fx, *fy;           // sums of interpolated vx, -vy
    int *cnt;                 // covering triangles
};

// ---- per-triangle setup for a full render of state X --------------------------------
__global__ void k_setup_all(Mesh m, const double *__restrict__ X, TriSetup *__restrict__ out)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.T) return;
    int v0 = m.tri[3 * t], v1 = m.tri[3 * t + 1], v2 = m.tri[3 * t + 2];
    TriSetup s;
    d_tri_setup(s, v0, v1, v2, d_snap(X[2 * v0]), d_snap(X[2 * v0 + 1]), d_snap(X[2 * v1]), d_snap(X[2 * v1 + 1]),
                d_snap(X[2 * v2]), d_snap(X[2 * v2 + 1]), m.W, m.H);
    d_tri_attr(s, m.uv, X, m.N);
    out[t] = s;
}

// ---- full-frame render: one 16x16 tile per workgroup -----------------------------------
// The triangles whose bounding box meets the tile are marked in an LDS bit set and
// visited in ascending index order by every pixel (the order the oracle adds in).
// MODE 0: the four targets; 1: targets and the id image of the label palette; 2: the id image only.
// The id image is the G and B channel of the reference's mask render (renderer.py:90-101, 610-614): a
// primitive's colour is (255, label / 256, label % 256), (255, 255, 255) for label -1, blended additively
// with 8-bit saturation; id = 256 G + B (cuda_multi.py:137-143), 0 where nothing is drawn.
template <int MODE>
__global__ __launch_bounds__(EKF_TILE *EKF_TILE) void k_render(Mesh m, const double *__restrict__ X,
                                                                const TriSetup *__restrict__ setup, Targets out,
                                                                const int *__restrict__ labels, int *__restrict__ ids)
{
    __shared__ unsigned s_mask[EKF_MAX_TRI / 32];
    const int tid = threadIdx.y * EKF_TILE + threadIdx.x;
    const int words = (m.T + 31) / 32;
    for (int i = tid; i < words; i += EKF_TILE * EKF_TILE) s_mask[i] = 0;
    __syncthreads();
    const int c0 = blockIdx.x * EKF_TILE, r0 = blockIdx.y * EKF_TILE;
    for (int t = tid; t < m.T; t += EKF_TILE * EKF_TILE) {
        const TriSetup &s = setup[t];
        if (s.cmin <= s.cmax && s.cmax >= c0 && s.cmin < c0 + EKF_TILE && s.rmax >= r0 && s.rmin < r0 + EKF_TILE)
            atomicOr(&s_mask[t >> 5], 1u << (t & 31));
    }
    __syncthreads();
    const int c = c0 + threadIdx.x, r = r0 + threadIdx.y;
    if (c >= m.W || r >= m.H) return;
    int acc = 0, cnt = 0, sg = 0, sb = 0;
    float fx = 0.0f, fy = 0.0f;
    for (int wd = 0; wd < words; wd++) {
        unsigned bits = s_mask[wd];
        while (bits) {
            int b = __ffs(bits) - 1;
            bits &= bits - 1;
            const TriSetup &s = setup[wd * 32 + b];
            float l1, l2;
            if (!d_tri_eval(s, c, r, l1, l2)) continue;
            if (MODE != 2) {
                acc += d_texel(m.tex, s, l1, l2, m.W, m.H);
                fx = fx + d_lerp(s.ax[0], s.ax[1], s.ax[2], l1, l2);
                fy = fy + d_lerp(s.ay[0], s.ay[1], s.ay[2], l1, l2);
                cnt++;
            }
            if (MODE != 0) {
                const int lab = labels[wd * 32 + b];
                sg += lab < 0 ? 255 : lab / 256;
                sb += lab < 0 ? 255 : lab % 256;
            }
        }
    }
    const int p = r * m.W + c;
    if (MODE != 2) { out.acc[p] = acc; out.fx[p] = fx; out.fy[p] = fy; out.cnt[p] = cnt; }
    if (MODE != 0) ids[p] = 256 * min(sg, 255) + min(sb, 255);
}

// unclamped targets -> the 8-bit images a caller sees
__global__ void k_resolve(Targets t, uint8_t *__restrict__ im, uint8_t *__restrict__ mk, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (im) im[p] = (uint8_t)(t.acc[p] > 255 ? 255 : t.acc[p]);
    if (mk) mk[p] = t.cnt[p] > 0 ? 255 : 0;
}

// y_m * flow (kalman.py:679-682)
__global__ void k_mask_flow(const uint8_t *__restrict__ ym, const float *__restrict__ fx, const float *__restrict__ fy,
                            float *__restrict__ fxm, float *__restrict__ fym, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float mm = (float)ym[p];
    fxm[p] = mm * fx[p];
    fym[p] = mm * fy[p];
}

// ---- block reduction of NACC doubles per thread (fixed order: deterministic) -----------------
template <int NACC, int NT>
__device__ inline void d_block_reduce(double (&a)[NACC], double *s_red /* [NT/64][NACC] */, double *out)
{
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        double v = a[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        a[k] = v;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < NACC; k++) s_red[wv * NACC + k] = a[k];
    __syncthreads();
    if (threadIdx.x < NACC) {
        double v = 0.0;
        for (int w = 0; w < NT / 64; w++) v += s_red[w * NACC + threadIdx.x];
        out[threadIdx.x] = v;
    }
}

struct Obs {
    const uint8_t *yim;       // observed frame
    const float *yfx, *yfy;   // observed flow (masked or raw, chosen by the caller)
    const uint8_t *ym;        // mask in {0,1}
};

// ---- whole-image reductions of the fine-grained API -------------------------------------------
// jz (cuda.py:972-980): sums of (render_p - render_ref) * residual per channel.  One partial
// per workgroup, summed in index order by the host.
#define RED_NT 256
__global__ __launch_bounds__(RED_NT) void k_jz(Targets ref, Targets p, Obs o, int n, double *__restrict__ partial)
{
    __shared__ double s_red[(RED_NT / 64) * 4];
    double a[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * RED_NT + threadIdx.x; i < n; i += gridDim.x * RED_NT) {
        int rim = ref.acc[i] > 255 ? 255 : ref.acc[i], pim = p.acc[i] > 255 ? 255 : p.acc[i];
        int rm = ref.cnt[i] > 0 ? 255 : 0, pm = p.cnt[i] > 0 ? 255 : 0;
        double z = ((double)o.yim[i] - (double)rim) / 255.0;
        double zm = (255.0 * (double)o.ym[i] - (double)rm) / 255.0;
        float zfx = o.yfx[i] - ref.fx[i];
        float zfy = o.yfy[i] + ref.fy[i];
        a[0] += ((double)(pim - rim) / 255.0) * z;
        a[1] += (double)(p.fx[i] - ref.fx[i]) * (double)zfx;
        a[2] += (double)(p.fy[i] - ref.fy[i]) * (double)zfy;
        a[3] += ((double)(pm - rm) / 255.0) * zm;
    }
    d_block_reduce<4, RED_NT>(a, s_red, partial + 4 * blockIdx.x);
}

// j (cuda.py:982-1010): sums of (render_p - ref)(render_q - ref) per channel
__global__ __launch_bounds__(RED_NT) void k_j(Targets ref, Targets p, Targets q, int n, double *__restrict__ partial)
{
    __shared__ double s_red[(RED_NT / 64) * 4];
    double a[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * RED_NT + threadIdx.x; i < n; i += gridDim.x * RED_NT) {
        int rim = ref.acc[i] > 255 ? 255 : ref.acc[i];
        int pim = p.acc[i] > 255 ? 255 : p.acc[i], qim = q.acc[i] > 255 ? 255 : q.acc[i];
        int rm = ref.cnt[i] > 0 ? 255 : 0, pm = p.cnt[i] > 0 ? 255 : 0, qm = q.cnt[i] > 0 ? 255 : 0;
        a[0] += ((double)(pim - rim) / 255.0) * ((double)(qim - rim) / 255.0);
        a[1] += (double)(p.fx[i] - ref.fx[i]) * (double)(q.fx[i] - ref.fx[i]);
        a[2] += (double)(p.fy[i] - ref.fy[i]) * (double)(q.fy[i] - ref.fy[i]);
        a[3] += ((double)(pm - rm) / 255.0) * ((double)(qm - rm) / 255.0);
    }
    d_block_reduce<4, RED_NT>(a, s_red, partial + 4 * blockIdx.x);
}

// ---- label-segmented reductions: the reference's multi-perturbation kernels ---------------------------------
// histogram_jz / histogram_j (cuda_multi.py:81-248) attribute every pixel to a label -- the id the mask
// render shows there, taken from the reference render where it covers the pixel, else from the (first, then
// second) perturbed render (:132-143, :215-235) -- and add the pixel's terms into that label's bin with float
// atomics.  Here one workgroup owns one label: it scans the bounding box of the triangles that carry the label
// (in any of the renders involved; the host knows the states) and adds the pixels whose id is its own, in a
// fixed order -- the same sums, reproducible, binary64 like the other reductions.
struct MultiArgs {
    Targets ref, p, q;        // q unused by jz_multi
    const int *idr, *idp, *idq;
    Obs o;
    int W;
    const int4 *box;          // per label: cmin, cmax, rmin, rmax (empty if cmin > cmax)
    double *out;              // per label: 4 sums (jz_multi) or 4 sums + pixel count (j_multi)
};

__global__ __launch_bounds__(RED_NT) void k_jz_multi(MultiArgs a)
{
    __shared__ double s_red[(RED_NT / 64) * 4];
    const int lab = blockIdx.x;
    const int4 b = a.box[lab];
    double s[4] = {0, 0, 0, 0};
    const int bw = b.y - b.x + 1, bh = b.w - b.z + 1;
    const int npx = (bw > 0 && bh > 0) ? bw * bh : 0;
    for (int k = threadIdx.x; k < npx; k += RED_NT) {
        const int i = (b.z + k / bw) * a.W + b.x + k % bw;
        const bool m = a.ref.cnt[i] > 0, mp = a.p.cnt[i] > 0;
        const int face = m ? a.idr[i] : (mp ? a.idp[i] : 65535);
        if (face != lab || face >= 65535) continue;
        const int rim = a.ref.acc[i] > 255 ? 255 : a.ref.acc[i], pim = a.p.acc[i] > 255 ? 255 : a.p.acc[i];
        const int rm = m ? 255 : 0, pm = mp ? 255 : 0;
        const double z = ((double)a.o.yim[i] - (double)rim) / 255.0;
        const double zm = (255.0 * (double)a.o.ym[i] - (double)rm) / 255.0;
        const float zfx = a.o.yfx[i] - a.ref.fx[i];
        const float zfy = a.o.yfy[i] + a.ref.fy[i];
        s[0] += ((double)(pim - rim) / 255.0) * z;
        s[1] += (double)(a.p.fx[i] - a.ref.fx[i]) * (double)zfx;
        s[2] += (double)(a.p.fy[i] - a.ref.fy[i]) * (double)zfy;
        s[3] += ((double)(pm - rm) / 255.0) * zm;
    }
    d_block_reduce<4, RED_NT>(s, s_red, a.out + 4 * lab);
}

__global__ __launch_bounds__(RED_NT) void k_j_multi(MultiArgs a)
{
    __shared__ double s_red[(RED_NT / 64) * 5];
    const int lab = blockIdx.x;
    const int4 b = a.box[lab];
    double s[5] = {0, 0, 0, 0, 0};
    const int bw = b.y - b.x + 1, bh = b.w - b.z + 1;
    const int npx = (bw > 0 && bh > 0) ? bw * bh : 0;
    for (int k = threadIdx.x; k < npx; k += RED_NT) {
        const int i = (b.z + k / bw) * a.W + b.x + k % bw;
        const bool m = a.ref.cnt[i] > 0, mp = a.p.cnt[i] > 0, mq = a.q.cnt[i] > 0;
        const int face = m ? a.idr[i] : (mp ? a.idp[i] : (mq ? a.idq[i] : 65535));
        if (face != lab || face >= 65535) continue;
        const int rim = a.ref.acc[i] > 255 ? 255 : a.ref.acc[i];
        const int pim = a.p.acc[i] > 255 ? 255 : a.p.acc[i], qim = a.q.acc[i] > 255 ? 255 : a.q.acc[i];
        const int rm = m ? 255 : 0, pm = mp ? 255 : 0, qm = mq ? 255 : 0;
        s[0] += ((double)(pim - rim) / 255.0) * ((double)(qim - rim) / 255.0);
        s[1] += (double)(a.p.fx[i] - a.ref.fx[i]) * (double)(a.q.fx[i] - a.ref.fx[i]);
        s[2] += (double)(a.p.fy[i] - a.ref.fy[i]) * (double)(a.q.fy[i] - a.ref.fy[i]);
        s[3] += ((double)(pm - rm) / 255.0) * ((double)(qm - rm) / 255.0);
        s[4] += 1.0;
    }
    d_block_reduce<5, RED_NT>(s, s_red, a.out + 5 * lab);
}

// ---- fused perturb-and-reduce (KFState.update in one launch) --------------------------------------
// One workgroup per job.  Job v < N: vertex v -- the central differences of jz for its four
// state components and the 4x4 diagonal block of HTH.  Job N + e: edge e = (v, w) -- the 4x4
// off-diagonal block HTH[(v,.),(w,.)].  A perturbation of vertex v changes the render only inside
// the triangles around v (its star), so every sum runs over the bounding box of that star only;
// the perturbed renders are never materialised.
struct StarTopo {
    const int *star_off;      // N+1, CSR offsets into star_tri
    const int *star_tri;      // triangle ids around each vertex
    const int *edges;         // E*2 vertex pairs (v < w)
    int E;
};

// value of one pixel of the star of vertex `v` in one configuration: sum over the star's
// triangles of texel / vx / -vy / coverage.
struct StarVal {
    int acc, cnt;
    float fx, fy;
};
// The texels of the first two covering triangles are not fetched inside the loop over the star but
// handed back as offsets (-1: none): the caller fetches the texels of all its configurations in one
// go, so that their memory latencies overlap instead of adding up (a wave would otherwise wait for
// every texel where it is found; with ~155 VGPRs there are not enough waves to hide that).
struct StarTex {
    int t0, t1;
};
__device__ __forceinline__ int d_star_texels(const uint8_t *__restrict__ tex, const StarTex &q)
{
    const int a = tex[q.t0 < 0 ? 0 : q.t0], b = tex[q.t1 < 0 ? 0 : q.t1];
    return (q.t0 < 0 ? 0 : a) + (q.t1 < 0 ? 0 : b);
}

// In the reference configuration the four velocity perturbations of v (vx +- d, vy +- d) leave the
// geometry alone: the same covering triangles, the same barycentrics, only the attribute of v differs.
// They are therefore evaluated in the same pass (fxp/fxm: star sum of vx with vx_v +- d, fyp/fym
// likewise for -vy), with the very expression a full render would use.
struct StarVel {
    float fxp, fxm, fyp, fym;
};

// The setups are the same for every lane (scalar loads when cfg is a read-only kernel argument), and
// so is the 8x8 tile the wave works on: `mask` has a bit for every triangle of the star whose
// bounding box -- over all five configurations, k_star_regions keeps that union -- meets the tile, one
// scalar test per triangle and tile.  About two thirds of the star's triangles at ~48 px edge length
// drop out and cost no vector instruction.  (The box contains every covered pixel, so the result is
// the same; the triangles are still taken in ascending order.)
template <bool VEL>
__device__ __forceinline__ StarVal d_star_eval(const TriSetup *__restrict__ cfg, unsigned mask, int c, int r,
                                               const Mesh &m, int v, float vxp, float vxm, float nvyp, float nvym,
                                               StarVel &vel, StarTex &q)
{
    StarVal s = {0, 0, 0.0f, 0.0f};
    q.t0 = -1; q.t1 = -1;
    if (VEL) { vel.fxp = 0.0f; vel.fxm = 0.0f; vel.fyp = 0.0f; vel.fym = 0.0f; }
    const double dc = (double)c, dr = (double)r;
    auto add = [&](const TriSetup &t, double e1, double e2) {
        float l1, l2;
        d_tri_bary2(t, e1, e2, l1, l2);            // (round 4: not evaluated a second time -- 319 -> 315.5 us per iteration)
        const int at = d_texel_at(t, l1, l2, m.W, m.H);
        if (s.cnt == 0) q.t0 = at;
        else if (s.cnt == 1) q.t1 = at;
        else s.acc += m.tex[at];                   // three triangles over one pixel: a folded mesh
        const float a0 = t.ax[0], a1 = t.ax[1], a2 = t.ax[2];
        const float b0 = t.ay[0], b1 = t.ay[1], b2 = t.ay[2];
        s.fx = s.fx + d_lerp(a0, a1, a2, l1, l2);
        s.fy = s.fy + d_lerp(b0, b1, b2, l1, l2);
        s.cnt++;
        if (VEL) {
            const bool v0 = t.i0 == v, v1 = t.i1 == v, v2 = t.i2 == v;
            vel.fxp = vel.fxp + d_lerp(v0 ? vxp : a0, v1 ? vxp : a1, v2 ? vxp : a2, l1, l2);
            vel.fxm = vel.fxm + d_lerp(v0 ? vxm : a0, v1 ? vxm : a1, v2 ? vxm : a2, l1, l2);
            vel.fyp = vel.fyp + d_lerp(v0 ? nvyp : b0, v1 ? nvyp : b1, v2 ? nvyp : b2, l1, l2);
            vel.fym = vel.fym + d_lerp(v0 ? nvym : b0, v1 ? nvym : b1, v2 ? nvym : b2, l1, l2);
        }
    };
    for (unsigned mm = mask; mm != 0; mm &= mm - 1) {
        const TriSetup &t = cfg[__builtin_ctz(mm)];
        double e1, e2;
        if (d_tri_cover2(t, dc, dr, e1, e2)) add(t, e1, e2);
    }
    return s;
}

// numerator of a difference that is a multiple of 1/255 (image and mask channels)
__device__ __forceinline__ int d_i255(double x) { return (int)rint(x * 255.0); }

// Difference images D = render(perturbed) - render(ref) at one pixel, for the four channels,
// given the reference accumulators of the pixel, the star's contribution to them (sref) and the
// star's contribution in the perturbed configuration (sp).
struct Diff {
    double im, m;             // already / 255
    float fx, fy;
};

// k255[x] = (double)x / 255.0 for x in -255..255 (a table in LDS, filled with real divisions: the
// image and mask differences are such quotients and a binary64 division costs ~35 instructions)
__device__ __forceinline__ void d_fill_k255(double *tab, int nthreads)
{
    for (int i = threadIdx.x; i < 511; i += nthreads) tab[i] = (double)(i - 255) / 255.0;
}
__device__ __forceinline__ double d_q255(const double *k255, int x)
{
    return (x >= -255 && x <= 255) ? k255[x] : (double)x / 255.0;
}

__device__ __forceinline__ Diff d_diff(const double *k255, int racc, int rcnt, float rfx, float rfy, const StarVal &sref,
                                       const StarVal &sp)
{
    int rim = racc > 255 ? 255 : racc;
    int pacc = racc - sref.acc + sp.acc;
    int pim = pacc > 255 ? 255 : pacc;
    int rm = rcnt > 0 ? 255 : 0;
    int pm = (rcnt - sref.cnt + sp.cnt) > 0 ? 255 : 0;
    Diff d;
    d.im = k255[pim - rim];                        // both in 0..255
    d.m = (double)((pm - rm) / 255);               // -1, 0 or 1: the quotient is exact
    float pfx = (rfx - sref.fx) + sp.fx;
    float pfy = (rfy - sref.fy) + sp.fy;
    d.fx = pfx - rfx;
    d.fy = pfy - rfy;
    return d;
}

// star setups of vertex v: cfg[0] = reference positions, cfg[1..] = v moved by dx / dy
__device__ inline void d_star_setup_one(TriSetup &out, int t, const Mesh &m, const double *X, int v, double dx, double dy)
{
    int v0 = m.tri[3 * t], v1 = m.tri[3 * t + 1], v2 = m.tri[3 * t + 2];
    double px[3] = {X[2 * v0], X[2 * v1], X[2 * v2]}, py[3] = {X[2 * v0 + 1], X[2 * v1 + 1], X[2 * v2 + 1]};
    int vs[3] = {v0, v1, v2};
    for (int q = 0; q < 3; q++)
        if (vs[q] == v) { px[q] += dx; py[q] += dy; }
    d_tri_setup(out, v0, v1, v2, d_snap(px[0]), d_snap(py[0]), d_snap(px[1]), d_snap(py[1]), d_snap(px[2]),
                d_snap(py[2]), m.W, m.H);
    d_tri_attr(out, m.uv, X, m.N);
}
__device__ inline void d_star_setup_pad(TriSetup &e)     // a triangle that covers nothing (pads a star to an even count)
{
    e.cmin = 1; e.cmax = 0; e.rmin = 1; e.rmax = 0;
    for (int k = 0; k < 3; k++) { e.ea[k] = 0.0; e.eb[k] = 0.0; e.ec[k] = -1.0; e.ecb[k] = -1.0; e.tl[k] = 0; }
}
__device__ inline void d_star_setups(TriSetup *dst, int ns, const int *tris, const Mesh &m, const double *X, int v,
                                     double dx, double dy, int lane, int stride)
{
    for (int k = lane; k < ns; k += stride) d_star_setup_one(dst[k], tris[k], m, X, v, dx, dy);
    if ((ns & 1) && lane == 0) d_star_setup_pad(dst[ns]);
}

// Difference images of the forward position perturbations, kept for the edge jobs.  For every
// vertex v its star region (bounding box over all configurations) is a slab of the pool starting at
// pixel offset off[v]; per pixel: D_{v,x} and D_{v,y} in the four channels (image and mask as 16-bit
// integers in 1/255 units, flow as f32) and the flow channel of the two velocity perturbations.
// Layout for whole sectors: a region starts at a column that is a multiple of 8 and is a multiple of 8
// columns wide, so that the 8-pixel rows of the tiles k_measure_vertex works on are aligned 32-byte pieces
// of every plane (the two 16-bit channels of a perturbation share one 32-bit plane) -- and of the reference
// render and the observation, whose rows the same tiles read.  Unaligned, an 8-pixel row of a 16-bit plane
// is a quarter of a sector and an f32 row straddles two: the kernel moved 3.5x its payload
// (profiles/r01_ekf_traffic.csv).
// A pixel of a region that no configuration of the star covers (about a third of a bounding box, and most of the
// intersection of two) has all-zero differences: its first plane holds POOL_EMPTY instead of a numerator, and the
// edge jobs -- every term of which has a factor from each vertex -- skip the other seven planes of both vertices
// when they meet it (182 -> 132 MB read per launch, 36 -> 33 us).  The zeros are stored all the same: leaving the
// seven planes of such pixels unwritten saves 16 MB of writes and costs 59 MB of reads (lines written in part are
// read back and merged by the memory side), no time gained; and skipping a tile no triangle reaches before its
// loads pushes the vertex kernel over its 128 registers (253 spilled, 85 -> 400 us).
// TILE-MAJOR: a region is a grid of 8x8-pixel tiles (c0, r0, rw, rh all multiples of 8, so the tiles of all regions lie
// on one grid of the frame) and a tile's 64 pixels are contiguous in every plane: a wave parks a tile with one
// 256-byte store per plane and an edge job reads it back the same way.  Row-major regions had every 8-pixel tile row
// a separate 32-byte piece: 150 MB counted at the memory side for 50 MB of payload (profiles/r02_ekf_traffic.csv).
// live[] has one word per tile: 0 when no pixel of the tile is covered in any configuration -- an edge job looks at
// the two words before it touches the tile's planes.  Pixels of a padded tile that lie outside the frame are parked
// as empty ones.
#define POOL_EMPTY 0x7FFF
struct DPool {
    int *hdr;                 // N x 4: c0, r0, rw, rh of the region (all multiples of 8)
    int *live;                // one word per tile of the pool (cap / 64)
    const int *area;          // N region areas (k_star_regions); a region's offset is the sum of those before it
    short2 *xi, *yi;          // (image, mask) numerators of D_{v,x} and D_{v,y}
    float *xfx, *xfy, *yfx, *yfy, *vxfx, *vyfy;
    long long cap;            // pixels in the pool
    int *overflow;            // set to 1 if the regions do not fit
};

__device__ __forceinline__ void d_park_empty(const DPool &P, long long pp)
{
    P.xi[pp] = make_short2(POOL_EMPTY, 0); P.yi[pp] = make_short2(0, 0);
    P.xfx[pp] = 0.0f; P.xfy[pp] = 0.0f; P.yfx[pp] = 0.0f; P.yfy[pp] = 0.0f;
    P.vxfx[pp] = 0.0f; P.vyfy[pp] = 0.0f;
}

struct MeasureArgs {
    Mesh m;
    StarTopo topo;
    Targets ref;
    Obs obs;
    const double *X;
    double delta;
    double *out;              // njobs * MEAS_VSPLIT_MAX * MEAS_OUT doubles
    DPool pool;
    int vsplit;               // workgroups per vertex job (gridDim.y of k_measure_vertex)
    int esplit;               // workgroups per edge job (gridDim.y of k_measure_edge)
    double iZ, iJ, iM;        // 1 / eps_Z, 1 / eps_J, 1 / eps_M
    TriSetup *cfgs;           // N x MEAS_NCFG x (EKF_MAX_STAR + 1): the star setups, written by k_star_regions
    int4 *ubox;               // N x UBOX_STRIDE: per triangle of the star, its pixel box over all configurations
    unsigned *tmask;          // N x tmask_stride: per tile of the region, the triangles of the star that can reach it
    int *tlist, *tcount;      // N x tmask_stride: the tiles with a non-zero word, ascending; N: how many (-1: the region has more tiles than tmask_stride)
    int tmask_stride;
};

#define MEAS_NT 256
#define MEAS_OUT 40
#define MEAS_VSPLIT_MAX 16     // most workgroups per vertex job (their partial sums are added in order)
// vertex job output layout (doubles): only the non-zero terms are accumulated, and only in the
// combinations KFState.update uses -- 19 sums per thread instead of 38 keep the kernel under 128
// VGPRs (four waves per SIMD):
enum {
    // jz, plus minus minus (the central difference of kalman.py:499-515 is linear in the sums): x and y
    // per channel (Hzc wants the components), vx only has an fx term, vy only an fy term
    A_X = 0, A_Y = 4, A_VX = 8, A_VY = 9,
    // self block of HTH (forward differences): the four channels already weighted by 1/eps and added
    A_XX = 10, A_XY = 11, A_YY = 12,
    A_XVX = 13, A_YVX = 14,               // fx channel only (not weighted)
    A_XVY = 15, A_YVY = 16,               // fy channel only
    A_VXVX = 17, A_VYVY = 18,
    A_NV = 19
};
enum {
    // edge job (v,w): D_v,a * D_w,b
    B_XX = 0, B_XY = 1, B_YX = 2, B_YY = 3,        // geometry x geometry: channels weighted by 1/eps and added
    B_XVX = 4, B_YVX = 5, B_VXX = 6, B_VXY = 7, B_VXVX = 8,        // fx channel (not weighted)
    B_XVY = 9, B_YVY = 10, B_VYX = 11, B_VYY = 12, B_VYVY = 13,    // fy channel
    B_NV = 14
};

#define MEAS_NCFG 5            // reference, +x, -x, +y, -y of the vertex
#define UBOX_STRIDE (EKF_MAX_STAR + 4)
#define TMASK_STRIDE 1024       // tiles of a star region with a triangle word each (a region of up to 256 x 256 px)

__device__ inline void d_vertex_cfgs(TriSetup (*cfg)[EKF_MAX_STAR + 1], int nsv, const int *trv, const Mesh &m,
                                     const double *X, int v, double d, int nthreads)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = nthreads >> 6;
    const double dxs[MEAS_NCFG] = {0, d, -d, 0, 0}, dys[MEAS_NCFG] = {0, 0, 0, d, -d};
    for (int c = wv; c < MEAS_NCFG; c += nw) d_star_setups(cfg[c], nsv, trv, m, X, v, dxs[c], dys[c], lane, 64);
}

// ---- pass 0: star regions and their places in the pool -------------------------------------------------
// One wave per configuration where there are enough (MEAS_NCFG waves per vertex in a launch of its own, four when
// the regions ride along with the render of the iterate, k_render_iter): setups in parallel, then the bounding box
// of the star over all configurations by a lane-parallel min/max.
#define REGION_NT (64 * MEAS_NCFG)
struct RegionShared {
    int4 tbox[MEAS_NCFG][EKF_MAX_STAR + 1];      // pixel box of every star triangle in every configuration (cmin, cmax, rmin, rmax)
    double ea[MEAS_NCFG][EKF_MAX_STAR + 1][3], eb[MEAS_NCFG][EKF_MAX_STAR + 1][3], ecb[MEAS_NCFG][EKF_MAX_STAR + 1][3];   // their edge functions
    int box[MEAS_NCFG][4];
    int reg[4];                                  // the region: c0, r0, rw, rh
    int wsum[8];                                 // live tiles per wave of a round
};
template <int NT>
__device__ inline void d_star_regions(const MeasureArgs &a, int *__restrict__ area, int v, RegionShared &sh)
{
    const Mesh &m = a.m;
    const int nsv = a.topo.star_off[v + 1] - a.topo.star_off[v];
    const int *trv = a.topo.star_tri + a.topo.star_off[v];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NW = NT / 64;
    const double d = a.delta;
    const double dxs[MEAS_NCFG] = {0, d, -d, 0, 0}, dys[MEAS_NCFG] = {0, 0, 0, d, -d};
    for (int cf = wv; cf < MEAS_NCFG; cf += NW) {
        // lane k sets up triangle k of the star in configuration cf and stores it where k_measure_vertex reads it
        // (the padding entry behind an odd count covers nothing); the box stays in LDS for the region
        TriSetup *dst = a.cfgs + ((size_t)v * MEAS_NCFG + cf) * (EKF_MAX_STAR + 1);
        if (lane < nsv) {
            TriSetup su;
            d_star_setup_one(su, trv[lane], m, a.X, v, dxs[cf], dys[cf]);
            dst[lane] = su;
            sh.tbox[cf][lane] = make_int4(su.cmin, su.cmax, su.rmin, su.rmax);
            for (int k = 0; k < 3; k++) { sh.ea[cf][lane][k] = su.ea[k]; sh.eb[cf][lane][k] = su.eb[k]; sh.ecb[cf][lane][k] = su.ecb[k]; }
        }
        if ((nsv & 1) && lane == 0) {
            TriSetup pad;
            d_star_setup_pad(pad);
            pad.inv = 0.0f; pad.i0 = pad.i1 = pad.i2 = 0;
            for (int k = 0; k < 3; k++) { pad.ux[k] = pad.uy[k] = pad.ax[k] = pad.ay[k] = 0.0f; }
            dst[nsv] = pad;
        }
    }
    __syncthreads();
    if (wv == 0 && lane < UBOX_STRIDE) {   // box of triangle `lane` over all configurations (cmin, cmax, rmin, rmax)
        int4 u = make_int4(1, 0, 1, 0);
        if (lane < nsv)
            for (int cfg = 0; cfg < MEAS_NCFG; cfg++) {
                const int4 s = sh.tbox[cfg][lane];
                if (s.x > s.y) continue;
                if (u.x > u.y) u = s;
                else u = make_int4(min(u.x, s.x), max(u.y, s.y), min(u.z, s.z), max(u.w, s.w));
            }
        a.ubox[(size_t)v * UBOX_STRIDE + lane] = u;
    }
    for (int cf = wv; cf < MEAS_NCFG; cf += NW) {   // bounding box of configuration cf
        int c0 = m.W, c1 = -1, r0 = m.H, r1 = -1;
        if (lane < nsv) {
            const int4 s = sh.tbox[cf][lane];
            if (s.x <= s.y) { c0 = s.x; c1 = s.y; r0 = s.z; r1 = s.w; }
        }
        for (int o = 16; o > 0; o >>= 1) {          // EKF_MAX_STAR <= 32 lanes carry values
            c0 = min(c0, __shfl_down(c0, o, 64)); c1 = max(c1, __shfl_down(c1, o, 64));
            r0 = min(r0, __shfl_down(r0, o, 64)); r1 = max(r1, __shfl_down(r1, o, 64));
        }
        if (lane == 0) { sh.box[cf][0] = c0; sh.box[cf][1] = c1; sh.box[cf][2] = r0; sh.box[cf][3] = r1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c0 = m.W, c1 = -1, r0 = m.H, r1 = -1;
        for (int cfg = 0; cfg < MEAS_NCFG; cfg++) {
            c0 = min(c0, sh.box[cfg][0]); c1 = max(c1, sh.box[cfg][1]);
            r0 = min(r0, sh.box[cfg][2]); r1 = max(r1, sh.box[cfg][3]);
        }
        if (c1 >= c0) { c0 &= ~7; r0 &= ~7; }                         // whole tiles on the frame's 8x8 grid: see DPool
        const int rw = c1 >= c0 ? (c1 - c0 + 8) & ~7 : 0, rh = (c1 >= c0 && r1 >= r0) ? (r1 - r0 + 8) & ~7 : 0;
        a.pool.hdr[4 * v] = c0; a.pool.hdr[4 * v + 1] = r0; a.pool.hdr[4 * v + 2] = rw; a.pool.hdr[4 * v + 3] = rh;
        area[v] = rw * rh;
        sh.reg[0] = c0; sh.reg[1] = r0; sh.reg[2] = rw; sh.reg[3] = rh;
    }
    __syncthreads();
    // Which triangles of the star can cover a pixel of which 8x8 tile of the region, in any configuration: bit k of the
    // tile's word.  A triangle cannot if its box misses the tile or if one of its edge functions is <= 0 (with the
    // top-left rule folded in: ecb) at all four corner pixels of the tile -- E is linear, whole numbers evaluated exactly
    // in binary64 like the coverage test itself, so the word is a superset of the truth and a tight one: about a third
    // of a region's tiles lie outside the star and get 0 -- k_measure_vertex evaluates nothing there.  (A region of
    // more than tmask_stride tiles keeps the per-triangle boxes: k_measure_vertex tests those instead.)
    const int ntx = sh.reg[2] >> 3, ntiles = ntx * (sh.reg[3] >> 3);
    if (ntiles > a.tmask_stride) {                 // (uniform: the whole workgroup leaves)
        if (threadIdx.x == 0) a.tcount[v] = -1;
        return;
    }
    // ... and the list of the tiles with a non-zero word, in ascending order (a ballot per wave, wave totals through LDS):
    // k_measure_vertex walks that list and never touches the others
    int listed = 0;
    for (int t0 = 0; t0 < ntiles; t0 += NT) {
        const int t = t0 + threadIdx.x;
        unsigned word = 0;
        if (t < ntiles) {
            const int tc0 = sh.reg[0] + 8 * (t % ntx), tr0 = sh.reg[1] + 8 * (t / ntx);
            const double xa = (double)tc0, xb = (double)min(tc0 + 7, m.W - 1), ya = (double)tr0, yb = (double)min(tr0 + 7, m.H - 1);
#pragma unroll 1
            for (int k = 0; k < nsv; k++) {
                bool reach = false;
#pragma unroll 1
                for (int cf = 0; cf < MEAS_NCFG && !reach; cf++) {
                    const int4 b = sh.tbox[cf][k];
                    if ((b.x > b.y) | (b.y < tc0) | (b.x > tc0 + 7) | (b.w < tr0) | (b.z > tr0 + 7)) continue;
                    bool in = true;
#pragma unroll 1
                    for (int e = 0; e < 3; e++) {
                        const double A = sh.ea[cf][k][e], B = sh.eb[cf][k][e], C = sh.ecb[cf][k][e];
                        const double e00 = fma(A, xa, fma(B, ya, C)), e10 = fma(A, xb, fma(B, ya, C));
                        const double e01 = fma(A, xa, fma(B, yb, C)), e11 = fma(A, xb, fma(B, yb, C));
                        in &= (e00 > 0.0) | (e10 > 0.0) | (e01 > 0.0) | (e11 > 0.0);
                    }
                    reach = in;
                }
                if (reach) word |= 1u << k;
            }
            a.tmask[(size_t)v * a.tmask_stride + t] = word;
        }
        const unsigned long long bal = __ballot(word != 0);
        __syncthreads();                           // sh.wsum of the previous round has been read
        if (lane == 0) sh.wsum[wv] = __popcll(bal);
        __syncthreads();
        int before = listed;
        for (int w = 0; w < wv; w++) before += sh.wsum[w];
        if (word != 0) a.tlist[(size_t)v * a.tmask_stride + before + __popcll(bal & ((1ull << lane) - 1ull))] = t;
        for (int w = 0; w < NW; w++) listed += sh.wsum[w];
    }
    if (threadIdx.x == 0) a.tcount[v] = listed;
}

__global__ __launch_bounds__(REGION_NT) void k_star_regions(MeasureArgs a, int *__restrict__ area)
{
    __shared__ RegionShared sh;
    d_star_regions<REGION_NT>(a, area, blockIdx.x, sh);
}

// Places in the pool: the region of vertex v starts at pixel offset sum_{u < v} area[u].  Every
// workgroup adds that up for itself (a few hundred integers from L2, exact in any order) rather than
// waiting for a scan kernel between k_star_regions and the jobs; `tot` = all areas, for the capacity check.
template <int NT>
__device__ inline void d_region_sums(const int *__restrict__ area, int N, int v, int w, long long &bv, long long &bw,
                                     long long &tot, long long *s /* [NT / 64][3] */)
{
    long long a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < N; i += NT) {
        const long long x = area[i];
        c += x;
        if (i < v) a += x;
        if (i < w) b += x;
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s[wv * 3] = a; s[wv * 3 + 1] = b; s[wv * 3 + 2] = c; }
    __syncthreads();
    bv = 0; bw = 0; tot = 0;
    for (int k = 0; k < NT / 64; k++) { bv += s[k * 3]; bw += s[k * 3 + 1]; tot += s[k * 3 + 2]; }
}

// ---- pass 1: vertex jobs ----------------------------------------------------------------------------------
// One launch = every jz evaluation of KFState.update and the diagonal blocks of HTH.  gridDim.y
// workgroups per vertex; a perturbation of vertex v changes the render only inside the triangles
// around v (its star), so every sum runs over the bounding box of that star; the perturbed renders
// are never materialised.  The forward difference images are parked in the pool for pass 2.
#ifndef MEAS_OCC
#define MEAS_OCC 4        // waves per SIMD the kernel is compiled for (experiments: -DMEAS_OCC=5 caps it at 96 VGPRs, with spills)
#endif
__global__ __launch_bounds__(MEAS_NT, MEAS_OCC) void k_measure_vertex(MeasureArgs a, const TriSetup *__restrict__ cfgs,
                                                                const int4 *__restrict__ ubox, const unsigned *__restrict__ tmask)
{
    __shared__ double s_red[(MEAS_NT / 64) * MEAS_OUT];
    __shared__ double s_k255[511];
    const double *k255 = s_k255 + 255;
    d_fill_k255(s_k255, MEAS_NT);
    const Mesh &m = a.m;
    const int N = m.N, W = m.W;
    const int v = blockIdx.x, job = v;
    const double *X = a.X;
    const double d = a.delta;
    const int nsv = a.topo.star_off[v + 1] - a.topo.star_off[v];
    // The star setups k_star_regions computed (cfgs == a.cfgs, as a read-only kernel argument): the same
    // for every lane, so they are read with scalar loads into SGPRs where the edge functions take them
    // as operands -- staged in LDS, their broadcast reads (12 doubles per triangle and configuration,
    // per tile) kept the LDS pipe busier than the vector ALUs.
    const TriSetup *__restrict__ s_cfg[MEAS_NCFG];
#pragma unroll
    for (int c = 0; c < MEAS_NCFG; c++) s_cfg[c] = cfgs + ((size_t)v * MEAS_NCFG + c) * (EKF_MAX_STAR + 1);
    __syncthreads();
    const int c0 = a.pool.hdr[4 * v], r0 = a.pool.hdr[4 * v + 1], rw = a.pool.hdr[4 * v + 2], rh = a.pool.hdr[4 * v + 3];
    __shared__ long long s_sum[(MEAS_NT / 64) * 3];
    long long base, unused, total;
    d_region_sums<MEAS_NT>(a.pool.area, N, v, 0, base, unused, total, s_sum);
    const bool park = total <= a.pool.cap;
    if (v == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.pool.overflow = park ? 0 : 1;
    double acc[A_NV];
#pragma unroll
    for (int k = 0; k < A_NV; k++) acc[k] = 0.0;
    // A wave works on 8x8 pixel tiles of the region (a compact tile meets one or two triangles of
    // the star, a 64x1 strip three or four); the tiles are dealt to the waves of the vertex's
    // workgroups round robin.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lx = threadIdx.x & 7, ly = (threadIdx.x & 63) >> 3;
    const int ntx = (rw + 7) >> 3, ntiles = ntx * ((rh + 7) >> 3);
    const int nwaves = (MEAS_NT / 64) * gridDim.y;
    // The tiles some triangle of the star can reach (the region pass listed them); the others -- about a third of a
    // region -- are not evaluated and not parked: their live word is cleared below and nobody reads their planes.
    const int listed = a.tcount[v];
    const int nwalk = listed >= 0 ? listed : ntiles;
    for (int j = blockIdx.y * (MEAS_NT / 64) + wave; j < nwalk; j += nwaves) {
        const int tile = listed >= 0 ? a.tlist[(size_t)v * a.tmask_stride + j] : j;
        const int tr0 = r0 + 8 * (tile / ntx), tc0 = c0 + 8 * (tile % ntx);
        const int r = tr0 + ly, c = tc0 + lx;
        unsigned mask = 0;                         // the triangles of the star that can reach this tile
        if (listed >= 0) {
            mask = tmask[(size_t)v * a.tmask_stride + tile];          // from the region pass: box and corner tests
        } else {
            for (int k = 0; k < nsv; k++) {
                const int4 b = ubox[(size_t)v * UBOX_STRIDE + k];
                if (!((b.y < tc0) | (b.x > tc0 + 7) | (b.w < tr0) | (b.z > tr0 + 7))) mask |= 1u << k;
            }
        }
        const long long pp = base + (long long)tile * 64 + (threadIdx.x & 63);      // tile-major (DPool)
        if (c >= W || r >= m.H) {                  // the padding of a region may leave the frame
            if (park) d_park_empty(a.pool, pp);
            continue;
        }
        const int p = r * W + c;
        const int racc = a.ref.acc[p], rcnt = a.ref.cnt[p];
        const float rfx = a.ref.fx[p], rfy = a.ref.fy[p];
        StarVel vel, none;
        StarTex q0, q1, q2, q3, q4;
        StarVal sref = d_star_eval<true>(s_cfg[0], mask, c, r, m, v, (float)(X[2 * N + 2 * v] + d),
                                         (float)(X[2 * N + 2 * v] - d), (float)(-(X[2 * N + 2 * v + 1] + d)),
                                         (float)(-(X[2 * N + 2 * v + 1] - d)), vel, q0);
        StarVal sxp = d_star_eval<false>(s_cfg[1], mask, c, r, m, v, 0, 0, 0, 0, none, q1);
        StarVal sxm = d_star_eval<false>(s_cfg[2], mask, c, r, m, v, 0, 0, 0, 0, none, q2);
        StarVal syp = d_star_eval<false>(s_cfg[3], mask, c, r, m, v, 0, 0, 0, 0, none, q3);
        StarVal sym = d_star_eval<false>(s_cfg[4], mask, c, r, m, v, 0, 0, 0, 0, none, q4);
        {   // all texels at once
            const int e0 = d_star_texels(m.tex, q0), e1 = d_star_texels(m.tex, q1), e2 = d_star_texels(m.tex, q2);
            const int e3 = d_star_texels(m.tex, q3), e4 = d_star_texels(m.tex, q4);
            sref.acc += e0; sxp.acc += e1; sxm.acc += e2; syp.acc += e3; sym.acc += e4;
        }
        const bool covered = sref.cnt + sxp.cnt + sxm.cnt + syp.cnt + sym.cnt != 0;
        const bool tile_live = __any(covered);     // (over the lanes inside the frame: at least one, the regions are clipped)
        if (park && (threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_read_exec()))      // one lane of those still here
            a.pool.live[base / 64 + tile] = tile_live ? 1 : 0;
        if (!covered) {
            if (park) d_park_empty(a.pool, pp);
            continue;
        }
        // velocity perturbations keep the geometry: same coverage and texels as the reference
        const StarVal svxp = {sref.acc, sref.cnt, vel.fxp, sref.fy}, svxm = {sref.acc, sref.cnt, vel.fxm, sref.fy};
        const StarVal svyp = {sref.acc, sref.cnt, sref.fx, vel.fyp}, svym = {sref.acc, sref.cnt, sref.fx, vel.fym};
        const Diff dxp = d_diff(k255, racc, rcnt, rfx, rfy, sref, sxp), dxm = d_diff(k255, racc, rcnt, rfx, rfy, sref, sxm);
        const Diff dyp = d_diff(k255, racc, rcnt, rfx, rfy, sref, syp), dym = d_diff(k255, racc, rcnt, rfx, rfy, sref, sym);
        const Diff dvxp = d_diff(k255, racc, rcnt, rfx, rfy, sref, svxp), dvxm = d_diff(k255, racc, rcnt, rfx, rfy, sref, svxm);
        const Diff dvyp = d_diff(k255, racc, rcnt, rfx, rfy, sref, svyp), dvym = d_diff(k255, racc, rcnt, rfx, rfy, sref, svym);
        // residuals (cuda.py:943-950); the sums below are accumulated with fused multiply-adds (binary64:
        // half the instructions of separate multiplies and adds, and no less accurate)
        const int rim = racc > 255 ? 255 : racc, rm = rcnt > 0 ? 255 : 0;
        const double z = k255[(int)a.obs.yim[p] - rim];
        const double zm = d_q255(k255, 255 * (int)a.obs.ym[p] - rm);
        const double zfx = (double)(a.obs.yfx[p] - rfx), zfy = (double)(a.obs.yfy[p] + rfy);
        {   // jz: z . (D+ - D-) per channel
            acc[A_X + 0] = fma(dxp.im - dxm.im, z, acc[A_X + 0]); acc[A_X + 1] = fma((double)dxp.fx - (double)dxm.fx, zfx, acc[A_X + 1]);
            acc[A_X + 2] = fma((double)dxp.fy - (double)dxm.fy, zfy, acc[A_X + 2]); acc[A_X + 3] = fma(dxp.m - dxm.m, zm, acc[A_X + 3]);
            acc[A_Y + 0] = fma(dyp.im - dym.im, z, acc[A_Y + 0]); acc[A_Y + 1] = fma((double)dyp.fx - (double)dym.fx, zfx, acc[A_Y + 1]);
            acc[A_Y + 2] = fma((double)dyp.fy - (double)dym.fy, zfy, acc[A_Y + 2]); acc[A_Y + 3] = fma(dyp.m - dym.m, zm, acc[A_Y + 3]);
            acc[A_VX] = fma((double)dvxp.fx - (double)dvxm.fx, zfx, acc[A_VX]);
            acc[A_VY] = fma((double)dvyp.fy - (double)dvym.fy, zfy, acc[A_VY]);
        }
        {   // HTH diagonal block (forward differences, cuda.py:993-996): sum over channels of D_a D_b / eps
            const double xi = dxp.im, xf = (double)dxp.fx, xg = (double)dxp.fy, xm = dxp.m;
            const double yi = dyp.im, yf = (double)dyp.fx, yg = (double)dyp.fy, ym = dyp.m;
            const double wxi = xi * a.iZ, wxf = xf * a.iJ, wxg = xg * a.iJ, wxm = xm * a.iM;
            acc[A_XX] = fma(wxm, xm, fma(wxg, xg, fma(wxf, xf, fma(wxi, xi, acc[A_XX]))));
            acc[A_XY] = fma(wxm, ym, fma(wxg, yg, fma(wxf, yf, fma(wxi, yi, acc[A_XY]))));
            acc[A_YY] = fma(ym * a.iM, ym, fma(yg * a.iJ, yg, fma(yf * a.iJ, yf, fma(yi * a.iZ, yi, acc[A_YY]))));
            const double vf = (double)dvxp.fx, vg = (double)dvyp.fy;
            acc[A_XVX] = fma(xf, vf, acc[A_XVX]); acc[A_YVX] = fma(yf, vf, acc[A_YVX]);
            acc[A_XVY] = fma(xg, vg, acc[A_XVY]); acc[A_YVY] = fma(yg, vg, acc[A_YVY]);
            acc[A_VXVX] = fma(vf, vf, acc[A_VXVX]); acc[A_VYVY] = fma(vg, vg, acc[A_VYVY]);
        }
        if (park) {
            a.pool.xi[pp] = make_short2((short)d_i255(dxp.im), (short)d_i255(dxp.m));
            a.pool.yi[pp] = make_short2((short)d_i255(dyp.im), (short)d_i255(dyp.m));
            a.pool.xfx[pp] = dxp.fx; a.pool.xfy[pp] = dxp.fy; a.pool.yfx[pp] = dyp.fx; a.pool.yfy[pp] = dyp.fy;
            a.pool.vxfx[pp] = dvxp.fx; a.pool.vyfy[pp] = dvyp.fy;
        }
    }
    if (park && listed >= 0)                       // the tiles nobody walked hold nothing
        for (int t = blockIdx.y * MEAS_NT + threadIdx.x; t < ntiles; t += MEAS_NT * gridDim.y)
            if (tmask[(size_t)v * a.tmask_stride + t] == 0) a.pool.live[base / 64 + t] = 0;
    d_block_reduce<A_NV, MEAS_NT>(acc, s_red, a.out + ((size_t)job * MEAS_VSPLIT_MAX + blockIdx.y) * MEAS_OUT);
}

// ---- pass 2: edge jobs ----------------------------------------------------------------------------------------
// HTH[(v,.),(w,.)] for adjacent vertices: sums of products of the parked difference images over the
// intersection of the two star regions (outside its own star a difference image is zero).  The regions lie on one
// grid of 8x8 tiles (DPool): a wave takes whole tiles of the intersection.  First every lane looks up the live words of
// one candidate tile for both vertices (one round trip for up to 64 tiles); the tiles live for both -- those the two
// shared triangles and their fringes reach -- are then read plane by plane, 256 contiguous bytes per plane and
// vertex, the loads of the next tile issued before the sums of the current one.
struct EdgeTile {
    short2 ax, ay, bx, by;
    float axfx, axfy, ayfx, ayfy, avx, avy, bxfx, bxfy, byfx, byfy, bvx, bvy;
};
__device__ __forceinline__ EdgeTile d_edge_load(const DPool &P, long long pv, long long pw)
{
    EdgeTile t;
    t.ax = P.xi[pv]; t.bx = P.xi[pw]; t.ay = P.yi[pv]; t.by = P.yi[pw];
    t.axfx = P.xfx[pv]; t.axfy = P.xfy[pv]; t.ayfx = P.yfx[pv]; t.ayfy = P.yfy[pv]; t.avx = P.vxfx[pv]; t.avy = P.vyfy[pv];
    t.bxfx = P.xfx[pw]; t.bxfy = P.xfy[pw]; t.byfx = P.yfx[pw]; t.byfy = P.yfy[pw]; t.bvx = P.vxfx[pw]; t.bvy = P.vyfy[pw];
    return t;
}

__global__ __launch_bounds__(MEAS_NT) void k_measure_edge(MeasureArgs a)
{
    __shared__ double s_red[(MEAS_NT / 64) * MEAS_OUT];
    __shared__ double s_k255[511];
    const double *k255 = s_k255 + 255;
    d_fill_k255(s_k255, MEAS_NT);
    __syncthreads();
    const int N = a.m.N;
    const int e = blockIdx.x;
    const int v = a.topo.edges[2 * e], w = a.topo.edges[2 * e + 1];
    const int *hv = a.pool.hdr + 4 * v, *hw = a.pool.hdr + 4 * w;
    // the intersection of the two regions in tiles of the frame's grid
    const int tx0 = max(hv[0], hw[0]) >> 3, ty0 = max(hv[1], hw[1]) >> 3;
    const int tx1 = min(hv[0] + hv[2], hw[0] + hw[2]) >> 3, ty1 = min(hv[1] + hv[3], hw[1] + hw[3]) >> 3;
    const int ntx = tx1 - tx0, nty = ty1 - ty0;
    const int vx0 = hv[0] >> 3, vy0 = hv[1] >> 3, vnx = hv[2] >> 3;
    const int wx0 = hw[0] >> 3, wy0 = hw[1] >> 3, wnx = hw[2] >> 3;
    __shared__ long long s_sum[(MEAS_NT / 64) * 3];
    long long bv, bw, total;
    d_region_sums<MEAS_NT>(a.pool.area, N, v, w, bv, bw, total, s_sum);
    // regions that do not fit the pool were not parked (k_measure_vertex reported the overflow; the caller grows the pool
    // and measures again): nothing of the pool is looked at, the places of these regions lie beyond its end
    const int nt = (total <= a.pool.cap && ntx > 0 && nty > 0) ? ntx * nty : 0;
    double acc[B_NV];
#pragma unroll
    for (int k = 0; k < B_NV; k++) acc[k] = 0.0;
    const DPool &P = a.pool;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.y * (MEAS_NT / 64) + (threadIdx.x >> 6), nwaves = (MEAS_NT / 64) * gridDim.y;
    const double iZ = a.iZ, iJ = a.iJ, iM = a.iM;
    const int ntxd = ntx > 0 ? ntx : 1;
    for (int first = 0; first < nt; first += 64 * nwaves) {
        // lane i: candidate tile first + i * nwaves + gw of this wave
        const int cand = first + lane * nwaves + gw;
        int tv = 0, tw = 0;
        bool go = false;
        if (cand < nt) {
            const int tx = tx0 + cand % ntxd, ty = ty0 + cand / ntxd;
            tv = (ty - vy0) * vnx + (tx - vx0);
            tw = (ty - wy0) * wnx + (tx - wx0);
            go = (P.live[bv / 64 + tv] != 0) & (P.live[bw / 64 + tw] != 0);
        }
        unsigned long long todo = __ballot(go);
        if (todo == 0) continue;
        int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        EdgeTile nx = d_edge_load(P, bv + (long long)__shfl(tv, l, 64) * 64 + lane, bw + (long long)__shfl(tw, l, 64) * 64 + lane);
        for (;;) {
            const EdgeTile t = nx;
            const bool more = todo != 0;
            if (more) {
                l = __builtin_ctzll(todo);
                todo &= todo - 1;
                nx = d_edge_load(P, bv + (long long)__shfl(tv, l, 64) * 64 + lane, bw + (long long)__shfl(tw, l, 64) * 64 + lane);
            }
            // numerators in -255..255, or the mark of a pixel outside the star (all its other planes are zero: every
            // product below has a factor from each vertex)
            if (t.ax.x != POOL_EMPTY && t.bx.x != POOL_EMPTY) {
                const double axim = k255[t.ax.x], axm = k255[t.ax.y];
                const double ayim = k255[t.ay.x], aym = k255[t.ay.y];
                const double bxim = k255[t.bx.x], bxm = k255[t.bx.y];
                const double byim = k255[t.by.x], bym = k255[t.by.y];
                const double axfx = t.axfx, axfy = t.axfy, ayfx = t.ayfx, ayfy = t.ayfy;
                const double bxfx = t.bxfx, bxfy = t.bxfy, byfx = t.byfx, byfy = t.byfy;
                const double avx = t.avx, avy = t.avy, bvx = t.bvx, bvy = t.bvy;
                const double wxi = axim * iZ, wxf = axfx * iJ, wxg = axfy * iJ, wxm = axm * iM;    // D_v,x / eps
                const double wyi = ayim * iZ, wyf = ayfx * iJ, wyg = ayfy * iJ, wym = aym * iM;    // D_v,y / eps
                acc[B_XX] = fma(wxm, bxm, fma(wxg, bxfy, fma(wxf, bxfx, fma(wxi, bxim, acc[B_XX]))));
                acc[B_XY] = fma(wxm, bym, fma(wxg, byfy, fma(wxf, byfx, fma(wxi, byim, acc[B_XY]))));
                acc[B_YX] = fma(wym, bxm, fma(wyg, bxfy, fma(wyf, bxfx, fma(wyi, bxim, acc[B_YX]))));
                acc[B_YY] = fma(wym, bym, fma(wyg, byfy, fma(wyf, byfx, fma(wyi, byim, acc[B_YY]))));
                acc[B_XVX] = fma(axfx, bvx, acc[B_XVX]); acc[B_YVX] = fma(ayfx, bvx, acc[B_YVX]);
                acc[B_VXX] = fma(avx, bxfx, acc[B_VXX]); acc[B_VXY] = fma(avx, byfx, acc[B_VXY]);
                acc[B_VXVX] = fma(avx, bvx, acc[B_VXVX]);
                acc[B_XVY] = fma(axfy, bvy, acc[B_XVY]); acc[B_YVY] = fma(ayfy, bvy, acc[B_YVY]);
                acc[B_VYX] = fma(avy, bxfy, acc[B_VYX]); acc[B_VYY] = fma(avy, byfy, acc[B_VYY]);
                acc[B_VYVY] = fma(avy, bvy, acc[B_VYVY]);
            }
            if (!more) break;
        }
    }
    d_block_reduce<B_NV, MEAS_NT>(acc, s_red, a.out + ((size_t)(N + e) * MEAS_VSPLIT_MAX + blockIdx.y) * MEAS_OUT);
}

// ---- the render of an iterate, with everything else that needs nothing but the iterate -----------------------
// One launch per IEKF iteration in place of four (round 2: k_setup_all, k_render<0>, k_error, k_star_regions):
//   blocks [0, n_regions)   the star regions of the measurement at this state (d_star_regions: they read the state only);
//   the other blocks        one RI_W x RI_H strip of the render each, four pixels per thread (a wave covers a 64-pixel
//       row: whole 256-byte lines of every target).  The triangle setups are not read from memory (69 KB per 16x16
//       tile for a 360-triangle mesh in k_render) but made on the spot: every thread tests its share of the triangles
//       against the strip -- first with the plain extent of the three vertices widened by two pixels (a superset of
//       the box d_tri_setup keeps, a few instructions), then with that box itself (d_tri_bbox) -- the few that meet it
//       are set up by one thread each into LDS, RI_CHUNK at a time, and visited by every pixel in ascending index
//       order: the values of k_setup_all + k_render<0>, bit for bit.  With `with_err` the pixels' four terms of
//       Renderer.error (renderer.py:485-501: image and mask terms in uint8, difference and square both wrapping
//       modulo 256 before the sum) are added up over the strip and stored as partial[strip]; the final sums
//       are formed in a fixed two-level order (d_tile_partial_sums; hm_tile_partial_sums on the host).
#define RI_CHUNK 32
#define RI_GROUPS 256
#define RI_W 64
#define RI_H 16                        // strip height of the default launch (hm_ctx_tune "render_rows": 16 or 8)
struct IterRenderArgs {
    Mesh m;
    const double *X;
    Targets out;
    Obs o;
    const float *raw_fx, *raw_fy;   // the observed flow before the mask was multiplied in (o.yfx / o.yfy may be either)
    double *partial;          // strips x RI_NV
    int tiles_x, tiles_y, with_err, n_regions;
};
#define RI_NV 6               // image, flow x, flow y, mask terms against `o`, then flow x, flow y against the raw flow
struct RenderShared {
    unsigned mask[EKF_MAX_TRI / 32];
    TriSetup cand[RI_CHUNK];
    double red[(256 / 64) * RI_NV];
    unsigned band[RI_W / 8];  // per 8-column band: the candidates of the chunk that can reach it
};

#ifdef HM_STAMP
__device__ long long *g_stamp;          // development builds only (tools/stamp_render.py): per-workgroup wall-clock stamps
#define HM_STAMP_AT(k) do { if (g_stamp && threadIdx.x == 0) g_stamp[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define HM_STAMP_AT(k) do { } while (0)
#endif
template <int RIH>
__global__ __launch_bounds__(256, 4) void k_render_iter(IterRenderArgs r, MeasureArgs a, int *__restrict__ area)
{
    HM_STAMP_AT(0);
    constexpr int RI_PX = RI_W * RIH / 256;      // pixels per thread: rows r0 + (tid >> 6) + 4 q
    __shared__ union {
        RegionShared reg;
        RenderShared t;
    } sh;
    if ((int)blockIdx.x < r.n_regions) {
        d_star_regions<256>(a, area, blockIdx.x, sh.reg);
        HM_STAMP_AT(4);
        return;
    }
    const Mesh &m = r.m;
    const int tile = blockIdx.x - r.n_regions;
    const int tid = threadIdx.x;
    const int words = (m.T + 31) / 32;
    for (int i = tid; i < words; i += 256) sh.t.mask[i] = 0;
    __syncthreads();
    const int c0 = (tile % r.tiles_x) * RI_W, r0 = (tile / r.tiles_x) * RIH;
    const double *__restrict__ X = r.X;
    // two triangles per thread and round, their loads issued together (tri -> X is a dependent pair of round trips)
    for (int tb = tid; tb < m.T; tb += 512) {
        const int ta = tb, tc = tb + 256;
        const bool has2 = tc < m.T;
        const int a0 = m.tri[3 * ta], a1 = m.tri[3 * ta + 1], a2 = m.tri[3 * ta + 2];
        const int b0 = has2 ? m.tri[3 * tc] : a0, b1 = has2 ? m.tri[3 * tc + 1] : a1, b2 = has2 ? m.tri[3 * tc + 2] : a2;
        const double ax0 = X[2 * a0], ay0 = X[2 * a0 + 1], ax1 = X[2 * a1], ay1 = X[2 * a1 + 1], ax2 = X[2 * a2], ay2 = X[2 * a2 + 1];
        const double bx0 = X[2 * b0], by0 = X[2 * b0 + 1], bx1 = X[2 * b1], by1 = X[2 * b1 + 1], bx2 = X[2 * b2], by2 = X[2 * b2 + 1];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (k == 1 && !has2) break;
            const int t = k ? tc : ta;
            const double x0 = k ? bx0 : ax0, y0 = k ? by0 : ay0, x1 = k ? bx1 : ax1, y1 = k ? by1 : ay1, x2 = k ? bx2 : ax2, y2 = k ? by2 : ay2;
            // pixel (c, r) can only be covered if its centre lies within the extent of the vertices (snapped to 1/256 px:
            // half a pixel of slack would do); anything non-finite takes the exact test
            const double lo_x = fmin(x0, fmin(x1, x2)) - 2.0, hi_x = fmax(x0, fmax(x1, x2)) + 2.0;
            const double lo_y = fmin(y0, fmin(y1, y2)) - 2.0, hi_y = fmax(y0, fmax(y1, y2)) + 2.0;
            if (hi_x < (double)c0 || lo_x > (double)(c0 + RI_W) || hi_y < (double)r0 || lo_y > (double)(r0 + RIH)) continue;
            int cmin, cmax, rmin, rmax;
            d_tri_bbox(d_snap(x0), d_snap(y0), d_snap(x1), d_snap(y1), d_snap(x2), d_snap(y2), m.W, m.H, cmin, cmax, rmin, rmax);
            if (cmin <= cmax && cmax >= c0 && cmin < c0 + RI_W && rmax >= r0 && rmin < r0 + RIH)
                atomicOr(&sh.t.mask[t >> 5], 1u << (t & 31));
        }
    }
    __syncthreads();
    HM_STAMP_AT(1);
    int total = 0;
    for (int w = 0; w < words; w++) total += __popc(sh.t.mask[w]);
    const int c = c0 + (tid & 63), rb = r0 + (tid >> 6);
    int acc[RI_PX], cnt[RI_PX], tadr[RI_PX];
    float fx[RI_PX], fy[RI_PX];
#pragma unroll
    for (int q = 0; q < RI_PX; q++) { acc[q] = 0; cnt[q] = 0; fx[q] = 0.0f; fy[q] = 0.0f; tadr[q] = -1; }
    for (int base = 0; base < total; base += RI_CHUNK) {
        const int nch = min(RI_CHUNK, total - base);
        // The candidates base .. base + nch - 1 (in ascending triangle order: the rank of a triangle among the set bits
        // is its slot) are set up by the threads that tested them -- spread over all waves, each with its own triangles
        // -- not by the lanes of one wave one after the other.  A candidate no pixel of the strip can lie in (E_k <= 0
        // at all four corner pixels for one edge k: E is linear) is left with an empty box: the pixel loop passes it
        // at its first comparison.
        for (int tb = tid; tb < m.T; tb += 256) {
            const unsigned word = sh.t.mask[tb >> 5], bit = 1u << (tb & 31);
            if (!(word & bit)) continue;
            int rank = __popc(word & (bit - 1u));
            for (int w = 0; w < (tb >> 5); w++) rank += __popc(sh.t.mask[w]);
            if (rank < base || rank >= base + nch) continue;
            const int v0 = m.tri[3 * tb], v1 = m.tri[3 * tb + 1], v2 = m.tri[3 * tb + 2];
            TriSetup su;
            d_tri_setup(su, v0, v1, v2, d_snap(X[2 * v0]), d_snap(X[2 * v0 + 1]), d_snap(X[2 * v1]), d_snap(X[2 * v1 + 1]),
                        d_snap(X[2 * v2]), d_snap(X[2 * v2 + 1]), m.W, m.H);
            d_tri_attr(su, m.uv, X, m.N);
            {
                const double xa = (double)c0, xb = (double)min(c0 + RI_W - 1, m.W - 1);
                const double ya = (double)r0, yb = (double)min(r0 + RIH - 1, m.H - 1);
                bool out = false;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double e00 = fma(su.ea[k], xa, fma(su.eb[k], ya, su.ecb[k])), e10 = fma(su.ea[k], xb, fma(su.eb[k], ya, su.ecb[k]));
                    const double e01 = fma(su.ea[k], xa, fma(su.eb[k], yb, su.ecb[k])), e11 = fma(su.ea[k], xb, fma(su.eb[k], yb, su.ecb[k]));
                    out |= !(e00 > 0.0 || e10 > 0.0 || e01 > 0.0 || e11 > 0.0);
                }
                if (out) { su.cmin = 1; su.cmax = 0; }
            }
            sh.t.cand[rank - base] = su;
        }
        __syncthreads();
        {   // which candidates can reach which 8-column band of the strip (the same corner test, one (band, candidate)
            // pair per thread): a pixel then looks at the one or two triangles of its band, not at all of the strip's
            const int band = tid >> 5, q = tid & 31;
            bool reach = false;
            if (q < nch) {
                const TriSetup &su = sh.t.cand[q];
                const int xa_i = c0 + 8 * band, xb_i = min(xa_i + 7, m.W - 1), ya_i = r0, yb_i = min(r0 + RIH - 1, m.H - 1);
                reach = su.cmin <= su.cmax && su.cmax >= xa_i && su.cmin <= xb_i && xa_i < m.W;
                if (reach) {
                    const double xa = (double)xa_i, xb = (double)xb_i, ya = (double)ya_i, yb = (double)yb_i;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const double e00 = fma(su.ea[k], xa, fma(su.eb[k], ya, su.ecb[k])), e10 = fma(su.ea[k], xb, fma(su.eb[k], ya, su.ecb[k]));
                        const double e01 = fma(su.ea[k], xa, fma(su.eb[k], yb, su.ecb[k])), e11 = fma(su.ea[k], xb, fma(su.eb[k], yb, su.ecb[k]));
                        reach &= (e00 > 0.0 || e10 > 0.0 || e01 > 0.0 || e11 > 0.0);
                    }
                }
            }
            const unsigned long long b = __ballot(reach);          // lanes 0..31: band 2 wave, lanes 32..63: band 2 wave + 1
            if ((tid & 63) == 0) { sh.t.band[2 * (tid >> 6)] = (unsigned)b; sh.t.band[2 * (tid >> 6) + 1] = (unsigned)(b >> 32); }
        }
        __syncthreads();
        if (c < m.W)
            for (unsigned bm = sh.t.band[(tid & 63) >> 3]; bm != 0; bm &= bm - 1) {      // ascending triangle order
                const TriSetup &su = sh.t.cand[__builtin_ctz(bm)];
#pragma unroll
                for (int j = 0; j < RI_PX; j++) {
                    float l1, l2;
                    if (!d_tri_eval(su, c, rb + 4 * j, l1, l2)) continue;    // (rows past the frame are outside the box)
                    // the texel of the first covering triangle is fetched after the loop, all pixels' together (a
                    // fetch here would be waited for before the next triangle is looked at); further ones (a folded
                    // mesh) at once -- integer sums: the order does not matter
                    const int at = d_texel_at(su, l1, l2, m.W, m.H);
                    if (tadr[j] < 0) tadr[j] = at;
                    else acc[j] += m.tex[at];
                    fx[j] = fx[j] + d_lerp(su.ax[0], su.ax[1], su.ax[2], l1, l2);
                    fy[j] = fy[j] + d_lerp(su.ay[0], su.ay[1], su.ay[2], l1, l2);
                    cnt[j]++;
                }
            }
        __syncthreads();
    }
    HM_STAMP_AT(2);
    {
        int tx[RI_PX];
#pragma unroll
        for (int j = 0; j < RI_PX; j++) tx[j] = m.tex[tadr[j] < 0 ? 0 : tadr[j]];
#pragma unroll
        for (int j = 0; j < RI_PX; j++) acc[j] += tadr[j] < 0 ? 0 : tx[j];
    }
    double e[RI_NV] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < RI_PX; j++) {
        const int rr = rb + 4 * j;
        if (c >= m.W || rr >= m.H) continue;
        const int p = rr * m.W + c;
        r.out.acc[p] = acc[j]; r.out.fx[p] = fx[j]; r.out.fy[p] = fy[j]; r.out.cnt[p] = cnt[j];
        if (r.with_err) {                          // Renderer.error (renderer.py:485-501): uint8 wrap-around of the image / mask terms
            const unsigned rim = acc[j] > 255 ? 255 : acc[j];
            const unsigned rm = cnt[j] > 0 ? 255 : 0;
            const unsigned d = ((unsigned)r.o.yim[p] - rim) & 255u;
            const unsigned dm = ((255u * (unsigned)r.o.ym[p]) - rm) & 255u;
            const float dfx = r.o.yfx[p] - fx[j];
            const float dfy = r.o.yfy[p] + fy[j];
            e[0] += (double)((d * d) & 255u);
            e[1] += (double)dfx * (double)dfx;
            e[2] += (double)dfy * (double)dfy;
            e[3] += (double)((dm * dm) & 255u);
            // the same two terms against the flow as observed (KalmanFilter.compute updates on the masked flow and
            // reports Renderer.error of the final state against the raw one, kalman.py:679-700): the report of the
            // last iterate comes out of its own render
            const float gfx = r.raw_fx[p] - fx[j];
            const float gfy = r.raw_fy[p] + fy[j];
            e[4] += (double)gfx * (double)gfx;
            e[5] += (double)gfy * (double)gfy;
        }
    }
    HM_STAMP_AT(3);
    if (!r.with_err) return;
    d_block_reduce<RI_NV, 256>(e, sh.t.red, r.partial + RI_NV * (size_t)tile);
    HM_STAMP_AT(4);
}

// The RI_NV error sums from the per-strip partials, in a fixed order: thread g of RI_GROUPS adds the partials g,
// g + RI_GROUPS, ... in ascending order, then the group sums are added in ascending order (hm_tile_partial_sums on
// the host does the same additions).  sp: RI_GROUPS x RI_NV doubles of LDS; out[0..RI_NV-1] written by threads
// 0..RI_NV-1 after a barrier inside.
__device__ __forceinline__ void d_tile_partial_sums(const double *__restrict__ partial, int ntiles, double *sp, double *out)
{
    const int t = threadIdx.x;
    if (t < RI_GROUPS) {
        double s[RI_NV];
#pragma unroll
        for (int k = 0; k < RI_NV; k++) s[k] = 0.0;
#pragma unroll 4
        for (int i = t; i < ntiles; i += RI_GROUPS) {
            const double2 *q = (const double2 *)(partial + RI_NV * (size_t)i);
#pragma unroll
            for (int k = 0; k < RI_NV / 2; k++) { const double2 v = q[k]; s[2 * k] += v.x; s[2 * k + 1] += v.y; }
        }
#pragma unroll
        for (int k = 0; k < RI_NV; k++) sp[RI_NV * t + k] = s[k];
    }
    __syncthreads();
    if (t < RI_NV) {
        double s = 0.0;
        for (int g = 0; g < RI_GROUPS; g++) s += sp[RI_NV * g + t];
        out[t] = s;
    }
}

// ---- job sums -> Hz, Hz components, dense HTH (device twin of the unpacking in KFState.update) -------
// Every entry of H is written by exactly one job (vertex jobs own the diagonal 4x4 blocks, edge jobs
// the two mirrored off-diagonal ones), always the same set, so plain stores suffice and H has to be
// zero outside that set only once.
struct ScatterArgs {
    const double *out;        // njobs * MEAS_VSPLIT_MAX * MEAS_OUT
    const int *edges;
    int N, E, vsplit, esplit;
    double eZ, eJ, eM, d;
    double *H, *Hz, *Hzc;     // 4N x 4N, 4N, 4N x 4
};

__device__ __forceinline__ void d_put(double *H, int n4, int p, int q, double val, double d)
{
    const double v = val / d / d;
    H[(size_t)p * n4 + q] = v;
    H[(size_t)q * n4 + p] = v;
}

// One wave per job (four jobs per workgroup): lane k < MEAS_OUT adds up value k of the job's
// partial sums (in order), then every entry the job owns is formed by a lane of its own -- the
// divisions of a job run side by side instead of one after the other in a single thread.
__global__ __launch_bounds__(256) void k_hth_scatter(ScatterArgs a)
{
    __shared__ double so[4][MEAS_OUT];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + wv;
    const int N = a.N, n4 = 4 * N;
    const bool on = job < N + a.E;
    if (on && lane < MEAS_OUT) {
        const double *src = a.out + (size_t)job * MEAS_VSPLIT_MAX * MEAS_OUT;
        const int parts = job < N ? a.vsplit : a.esplit;
        double v = src[lane];
        for (int q = 1; q < parts; q++) v += src[q * MEAS_OUT + lane];
        so[wv][lane] = v;
    }
    __syncthreads();
    if (!on) return;
    const double *o = so[wv];
    const double eZ = a.eZ, eJ = a.eJ, eM = a.eM, d = a.d;
    if (job < N) {
        const int v = job;
        const int idx[4] = {2 * v, 2 * v + 1, 2 * N + 2 * v, 2 * N + 2 * v + 1};
        if (lane < 4) {
            // central differences of jz (kalman.py:499-515); component sums carry the sign of jz_CPU
            const int k = lane;
            double c[4] = {0.0, 0.0, 0.0, 0.0};
            if (k < 2) {
                const double *p = o + (k == 0 ? A_X : A_Y);
                c[0] = p[0] / eZ; c[1] = p[1] / eJ; c[2] = -p[2] / eJ; c[3] = p[3] / eM;
            } else if (k == 2) {
                c[1] = o[A_VX] / eJ;
            } else {
                c[2] = -o[A_VY] / eJ;
            }
            a.Hz[idx[k]] = (((c[0] + c[1]) + c[2]) + c[3]) / d / 2;
            for (int ch = 0; ch < 4; ch++) a.Hzc[(size_t)idx[k] * 4 + ch] = c[ch] / d / 2;
        } else if (lane >= 8 && lane < 17) {
            const int ix = idx[0], iy = idx[1], ivx = idx[2], ivy = idx[3];
            switch (lane - 8) {
            case 0: d_put(a.H, n4, ix, ix, o[A_XX], d); break;
            case 1: d_put(a.H, n4, ix, iy, o[A_XY], d); break;
            case 2: d_put(a.H, n4, iy, iy, o[A_YY], d); break;
            case 3: d_put(a.H, n4, ix, ivx, o[A_XVX] / eJ, d); break;
            case 4: d_put(a.H, n4, iy, ivx, o[A_YVX] / eJ, d); break;
            case 5: d_put(a.H, n4, ix, ivy, o[A_XVY] / eJ, d); break;
            case 6: d_put(a.H, n4, iy, ivy, o[A_YVY] / eJ, d); break;
            case 7: d_put(a.H, n4, ivx, ivx, o[A_VXVX] / eJ, d); break;
            default: d_put(a.H, n4, ivy, ivy, o[A_VYVY] / eJ, d); break;
            }
        }
    } else if (lane < 14) {
        const int e = job - N;
        const int v = a.edges[2 * e], w = a.edges[2 * e + 1];
        const int vx_ = 2 * v, vy_ = 2 * v + 1, vvx = 2 * N + 2 * v, vvy = 2 * N + 2 * v + 1;
        const int wx_ = 2 * w, wy_ = 2 * w + 1, wvx = 2 * N + 2 * w, wvy = 2 * N + 2 * w + 1;
        switch (lane) {
        case 0: d_put(a.H, n4, vx_, wx_, o[B_XX], d); break;
        case 1: d_put(a.H, n4, vx_, wy_, o[B_XY], d); break;
        case 2: d_put(a.H, n4, vy_, wx_, o[B_YX], d); break;
        case 3: d_put(a.H, n4, vy_, wy_, o[B_YY], d); break;
        case 4: d_put(a.H, n4, vx_, wvx, o[B_XVX] / eJ, d); break;
        case 5: d_put(a.H, n4, vy_, wvx, o[B_YVX] / eJ, d); break;
        case 6: d_put(a.H, n4, vvx, wx_, o[B_VXX] / eJ, d); break;
        case 7: d_put(a.H, n4, vvx, wy_, o[B_VXY] / eJ, d); break;
        case 8: d_put(a.H, n4, vvx, wvx, o[B_VXVX] / eJ, d); break;
        case 9: d_put(a.H, n4, vx_, wvy, o[B_XVY] / eJ, d); break;
        case 10: d_put(a.H, n4, vy_, wvy, o[B_YVY] / eJ, d); break;
        case 11: d_put(a.H, n4, vvy, wx_, o[B_VYX] / eJ, d); break;
        case 12: d_put(a.H, n4, vvy, wy_, o[B_VYY] / eJ, d); break;
        default: d_put(a.H, n4, vvy, wvy, o[B_VYVY] / eJ, d); break;
        }
    }
}
