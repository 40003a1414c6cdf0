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
    int a256[3], b256[3];      // edge function E_k = a256*c + b256*r + c0 at pixel (col c, row r)
    long long c0[3];
    int tl[3];                 // 1 if a zero edge value counts as inside (top-left edge)
    float inv;                 // 1 / (2 area)
    int i0, i1, i2;            // vertex ids after orientation normalisation
    int cmin, cmax, rmin, rmax;  // pixel bounding box (inclusive), empty if cmin > cmax
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

// Build the setup of triangle (v0,v1,v2) with snapped integer positions p[3][2].
__device__ inline void d_tri_setup(TriSetup &s, int v0, int v1, int v2, long long x0, long long y0,
                                   long long x1, long long y1, long long x2, long long y2, int W, int H)
{
    long long area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
    s.cmin = 1; s.cmax = 0; s.rmin = 1; s.rmax = 0;
    s.i0 = v0; s.i1 = v1; s.i2 = v2;
    s.inv = 0.0f;
    for (int k = 0; k < 3; k++) { s.a256[k] = 0; s.b256[k] = 0; s.c0[k] = -1; s.tl[k] = 0; }
    if (area == 0) return;
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
        s.a256[k] = (int)(-ey[k] * EKF_SUB);
        s.b256[k] = (int)(ex[k] * EKF_SUB);
        s.c0[k] = ex[k] * (128 - oy[k]) - ey[k] * (128 - ox[k]);
        s.tl[k] = (ey[k] > 0) || (ey[k] == 0 && ex[k] < 0);
    }
    s.inv = 1.0f / (float)area;
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
    s.cmin = (int)cl; s.cmax = (int)ch; s.rmin = (int)rl; s.rmax = (int)rh;
}

// coverage + barycentrics of pixel (c, r)
__device__ __forceinline__ bool d_tri_eval(const TriSetup &s, int c, int r, float &l1, float &l2)
{
    if (c < s.cmin || c > s.cmax || r < s.rmin || r > s.rmax) return false;
    long long e0 = (long long)s.a256[0] * c + (long long)s.b256[0] * r + s.c0[0];
    long long e1 = (long long)s.a256[1] * c + (long long)s.b256[1] * r + s.c0[1];
    long long e2 = (long long)s.a256[2] * c + (long long)s.b256[2] * r + s.c0[2];
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

__device__ __forceinline__ int d_texel(const uint8_t *__restrict__ tex, const TriSetup &s, float l1, float l2, int W,
                                       int H)
{
    float tx = d_lerp(s.ux[0], s.ux[1], s.ux[2], l1, l2);
    float ty = d_lerp(s.uy[0], s.uy[1], s.uy[2], l1, l2);
    int cx = (int)floorf(tx), cy = (int)floorf(ty);
    cx = cx < 0 ? 0 : (cx > W - 1 ? W - 1 : cx);
    cy = cy < 0 ? 0 : (cy > H - 1 ? H - 1 : cy);
    return tex[cy * W + cx];
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
__global__ __launch_bounds__(EKF_TILE *EKF_TILE) void k_render(Mesh m, const double *__restrict__ X,
                                                                const TriSetup *__restrict__ setup, Targets out)
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
    int acc = 0, cnt = 0;
    float fx = 0.0f, fy = 0.0f;
    for (int wd = 0; wd < words; wd++) {
        unsigned bits = s_mask[wd];
        while (bits) {
            int b = __ffs(bits) - 1;
            bits &= bits - 1;
            const TriSetup &s = setup[wd * 32 + b];
            float l1, l2;
            if (!d_tri_eval(s, c, r, l1, l2)) continue;
            acc += d_texel(m.tex, s, l1, l2, m.W, m.H);
            fx = fx + d_lerp(s.ax[0], s.ax[1], s.ax[2], l1, l2);
            fy = fy + d_lerp(s.ay[0], s.ay[1], s.ay[2], l1, l2);
            cnt++;
        }
    }
    const int p = r * m.W + c;
    out.acc[p] = acc; out.fx[p] = fx; out.fy[p] = fy; out.cnt[p] = cnt;
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

// Renderer.error (renderer.py:485-501).  The image and mask terms are computed in
// uint8 there: difference and square both wrap modulo 256 before the sum.
__global__ __launch_bounds__(RED_NT) void k_error(Targets ref, Obs o, int n, double *__restrict__ partial)
{
    __shared__ double s_red[(RED_NT / 64) * 4];
    double a[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * RED_NT + threadIdx.x; i < n; i += gridDim.x * RED_NT) {
        unsigned rim = ref.acc[i] > 255 ? 255 : ref.acc[i];
        unsigned rm = ref.cnt[i] > 0 ? 255 : 0;
        unsigned d = ((unsigned)o.yim[i] - rim) & 255u;
        unsigned dm = ((255u * (unsigned)o.ym[i]) - rm) & 255u;
        float dfx = o.yfx[i] - ref.fx[i];
        float dfy = o.yfy[i] + ref.fy[i];
        a[0] += (double)((d * d) & 255u);
        a[1] += (double)dfx * (double)dfx;
        a[2] += (double)dfy * (double)dfy;
        a[3] += (double)((dm * dm) & 255u);
    }
    d_block_reduce<4, RED_NT>(a, s_red, partial + 4 * blockIdx.x);
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

// In the reference configuration the four velocity perturbations of v (vx +- d, vy +- d) leave the
// geometry alone: the same covering triangles, the same barycentrics, only the attribute of v differs.
// They are therefore evaluated in the same pass (fxp/fxm: star sum of vx with vx_v +- d, fyp/fym
// likewise for -vy), with the very expression a full render would use.
struct StarVel {
    float fxp, fxm, fyp, fym;
};

template <bool VEL>
__device__ __forceinline__ StarVal d_star_eval(const TriSetup *__restrict__ cfg, int ns, int c, int r, const Mesh &m,
                                               const double *__restrict__ X, int v, float vxp, float vxm, float nvyp,
                                               float nvym, StarVel &vel)
{
    StarVal s = {0, 0, 0.0f, 0.0f};
    if (VEL) { vel.fxp = 0.0f; vel.fxm = 0.0f; vel.fyp = 0.0f; vel.fym = 0.0f; }
    for (int k = 0; k < ns; k++) {
        float l1, l2;
        if (!d_tri_eval(cfg[k], c, r, l1, l2)) continue;
        const TriSetup &t = cfg[k];
        s.acc += d_texel(m.tex, t, l1, l2, m.W, m.H);
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
    }
    return s;
}

// Difference images D = render(perturbed) - render(ref) at one pixel, for the four channels,
// given the reference accumulators of the pixel, the star's contribution to them (sref) and the
// star's contribution in the perturbed configuration (sp).
struct Diff {
    double im, m;             // already / 255
    float fx, fy;
};

__device__ __forceinline__ Diff d_diff(int racc, int rcnt, float rfx, float rfy, const StarVal &sref, const StarVal &sp)
{
    int rim = racc > 255 ? 255 : racc;
    int pacc = racc - sref.acc + sp.acc;
    int pim = pacc > 255 ? 255 : pacc;
    int rm = rcnt > 0 ? 255 : 0;
    int pm = (rcnt - sref.cnt + sp.cnt) > 0 ? 255 : 0;
    Diff d;
    d.im = (double)(pim - rim) / 255.0;
    d.m = (double)(pm - rm) / 255.0;
    float pfx = (rfx - sref.fx) + sp.fx;
    float pfy = (rfy - sref.fy) + sp.fy;
    d.fx = pfx - rfx;
    d.fy = pfy - rfy;
    return d;
}

// star setups of vertex v: cfg[0] = reference positions, cfg[1..] = v moved by dx / dy
__device__ inline void d_star_setups(TriSetup *dst, int ns, const int *tris, const Mesh &m, const double *X, int v,
                                     double dx, double dy, int lane, int stride)
{
    for (int k = lane; k < ns; k += stride) {
        int t = tris[k];
        int v0 = m.tri[3 * t], v1 = m.tri[3 * t + 1], v2 = m.tri[3 * t + 2];
        double px[3] = {X[2 * v0], X[2 * v1], X[2 * v2]}, py[3] = {X[2 * v0 + 1], X[2 * v1 + 1], X[2 * v2 + 1]};
        int vs[3] = {v0, v1, v2};
        for (int q = 0; q < 3; q++)
            if (vs[q] == v) { px[q] += dx; py[q] += dy; }
        d_tri_setup(dst[k], v0, v1, v2, d_snap(px[0]), d_snap(py[0]), d_snap(px[1]), d_snap(py[1]), d_snap(px[2]),
                    d_snap(py[2]), m.W, m.H);
        d_tri_attr(dst[k], m.uv, X, m.N);
    }
}

struct MeasureArgs {
    Mesh m;
    StarTopo topo;
    Targets ref;
    Obs obs;
    const double *X;
    double delta;
    double *out;              // njobs * MEAS_OUT doubles
};

#define MEAS_NT 256
#define MEAS_OUT 40
#define MEAS_VSPLIT 3          // workgroups per vertex job (their partial sums are added in order)
// vertex job output layout (doubles):
//   [0..15]  jz(+d) per (component x,y,vx,vy) x (channel im,fx,fy,m)        (sums, not yet / eps)
//   [16..31] jz(-d) likewise
//   [32..35] self block, image channel: (x,x) (x,y) (y,y) and unused
//   ... see ekf.hip for the exact unpacking
// To keep the register budget the kernel accumulates exactly the non-zero terms:
enum {
    // jz sums: plus then minus; x and y have 4 channels, vx only fx, vy only fy
    A_XP = 0, A_YP = 4, A_VXP = 8, A_VYP = 9, A_XM = 10, A_YM = 14, A_VXM = 18, A_VYM = 19,
    // self block of HTH (forward differences): per channel sums
    A_XX = 20, A_XY = 24, A_YY = 28,      // 4 channels each
    A_XVX = 32, A_YVX = 33,               // fx channel
    A_XVY = 34, A_YVY = 35,               // fy channel
    A_VXVX = 36, A_VYVY = 37,
    A_NV = 38
};
enum {
    // edge job (v,w): D_v,a * D_w,b
    B_XX = 0, B_XY = 4, B_YX = 8, B_YY = 12,       // geometry x geometry, 4 channels each
    B_XVX = 16, B_YVX = 17, B_VXX = 18, B_VXY = 19, B_VXVX = 20,   // fx channel
    B_XVY = 21, B_YVY = 22, B_VYX = 23, B_VYY = 24, B_VYVY = 25,   // fy channel
    B_NV = 26
};

// VERTEX selects the job kind at compile time: two kernels, each with only its own accumulators live
// (38 / 26 doubles), so that more workgroups fit a CU.
template <bool VERTEX>
__global__ __launch_bounds__(MEAS_NT) void k_measure(MeasureArgs a)
{
    __shared__ TriSetup s_cfg[6][EKF_MAX_STAR];
    __shared__ double s_red[(MEAS_NT / 64) * MEAS_OUT];
    const Mesh &m = a.m;
    const int N = m.N, W = m.W, H = m.H;
    const int job = VERTEX ? blockIdx.x : N + blockIdx.x;
    const double *X = a.X;
    const double d = a.delta;
    constexpr bool isv = VERTEX;
    const int v = isv ? job : a.topo.edges[2 * (job - N)];
    const int w = isv ? -1 : a.topo.edges[2 * (job - N) + 1];
    const int nsv = a.topo.star_off[v + 1] - a.topo.star_off[v];
    const int *trv = a.topo.star_tri + a.topo.star_off[v];
    int nsw = 0;
    const int *trw = nullptr;
    if (!isv) {
        nsw = a.topo.star_off[w + 1] - a.topo.star_off[w];
        trw = a.topo.star_tri + a.topo.star_off[w];
    }
    // configurations: vertex job: ref, +x, -x, +y, -y of v.   edge job: ref v, +x v, +y v, ref w, +x w, +y w
    {
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (isv) {
            const double dxs[5] = {0, d, -d, 0, 0}, dys[5] = {0, 0, 0, d, -d};
            for (int cfg = wv; cfg < 5; cfg += MEAS_NT / 64)
                d_star_setups(s_cfg[cfg], nsv, trv, m, X, v, dxs[cfg], dys[cfg], lane, 64);
        } else {
            const double dxs[3] = {0, d, 0}, dys[3] = {0, 0, d};
            for (int cfg = wv; cfg < 6; cfg += MEAS_NT / 64) {
                if (cfg < 3) d_star_setups(s_cfg[cfg], nsv, trv, m, X, v, dxs[cfg], dys[cfg], lane, 64);
                else d_star_setups(s_cfg[cfg], nsw, trw, m, X, w, dxs[cfg - 3], dys[cfg - 3], lane, 64);
            }
        }
    }
    __syncthreads();
    // region: vertex job -- bounding boxes of the star of v in all its configurations; edge job --
    // only the triangles that contain BOTH v and w (a product of two difference images vanishes
    // wherever one of the stars does not reach), again in all configurations of either vertex
    int c0 = W, c1 = -1, r0 = H, r1 = -1;
    {
        const int ncfg = isv ? 5 : 6;
        for (int cfg = 0; cfg < ncfg; cfg++) {
            const int ns = (isv || cfg < 3) ? nsv : nsw;
            const int other = isv ? -1 : (cfg < 3 ? w : v);
            for (int k = 0; k < ns; k++) {
                const TriSetup &s = s_cfg[cfg][k];
                if (s.cmin > s.cmax) continue;
                if (!isv && s.i0 != other && s.i1 != other && s.i2 != other) continue;
                c0 = min(c0, s.cmin); c1 = max(c1, s.cmax); r0 = min(r0, s.rmin); r1 = max(r1, s.rmax);
            }
        }
    }
    constexpr int NACC = VERTEX ? (int)A_NV : (int)B_NV;
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;

    const int rw = c1 - c0 + 1, rh = r1 - r0 + 1;
    const int npx = (rw > 0 && rh > 0) ? rw * rh : 0;
    // a vertex job is shared by MEAS_VSPLIT workgroups (blockIdx.y), pixels dealt round-robin in chunks of 256
    const int nsplit = VERTEX ? MEAS_VSPLIT : 1;
    for (int i = threadIdx.x + MEAS_NT * blockIdx.y; i < npx; i += MEAS_NT * nsplit) {
        const int r = r0 + i / rw, c = c0 + i % rw;
        const int p = r * W + c;
        const int racc = a.ref.acc[p], rcnt = a.ref.cnt[p];
        const float rfx = a.ref.fx[p], rfy = a.ref.fy[p];
        if (isv) {
            StarVel vel, none;
            const StarVal sref = d_star_eval<true>(s_cfg[0], nsv, c, r, m, X, v, (float)(X[2 * N + 2 * v] + d),
                                                   (float)(X[2 * N + 2 * v] - d), (float)(-(X[2 * N + 2 * v + 1] + d)),
                                                   (float)(-(X[2 * N + 2 * v + 1] - d)), vel);
            const StarVal sxp = d_star_eval<false>(s_cfg[1], nsv, c, r, m, X, v, 0, 0, 0, 0, none);
            const StarVal sxm = d_star_eval<false>(s_cfg[2], nsv, c, r, m, X, v, 0, 0, 0, 0, none);
            const StarVal syp = d_star_eval<false>(s_cfg[3], nsv, c, r, m, X, v, 0, 0, 0, 0, none);
            const StarVal sym = d_star_eval<false>(s_cfg[4], nsv, c, r, m, X, v, 0, 0, 0, 0, none);
            if (sref.cnt + sxp.cnt + sxm.cnt + syp.cnt + sym.cnt == 0) continue;
            // velocity perturbations keep the geometry: same coverage and texels as the reference
            const StarVal svxp = {sref.acc, sref.cnt, vel.fxp, sref.fy}, svxm = {sref.acc, sref.cnt, vel.fxm, sref.fy};
            const StarVal svyp = {sref.acc, sref.cnt, sref.fx, vel.fyp}, svym = {sref.acc, sref.cnt, sref.fx, vel.fym};
            const Diff dxp = d_diff(racc, rcnt, rfx, rfy, sref, sxp), dxm = d_diff(racc, rcnt, rfx, rfy, sref, sxm);
            const Diff dyp = d_diff(racc, rcnt, rfx, rfy, sref, syp), dym = d_diff(racc, rcnt, rfx, rfy, sref, sym);
            const Diff dvxp = d_diff(racc, rcnt, rfx, rfy, sref, svxp), dvxm = d_diff(racc, rcnt, rfx, rfy, sref, svxm);
            const Diff dvyp = d_diff(racc, rcnt, rfx, rfy, sref, svyp), dvym = d_diff(racc, rcnt, rfx, rfy, sref, svym);
            // residuals (cuda.py:943-950)
            const int rim = racc > 255 ? 255 : racc, rm = rcnt > 0 ? 255 : 0;
            const double z = ((double)a.obs.yim[p] - (double)rim) / 255.0;
            const double zm = (255.0 * (double)a.obs.ym[p] - (double)rm) / 255.0;
            const double zfx = (double)(a.obs.yfx[p] - rfx), zfy = (double)(a.obs.yfy[p] + rfy);
            acc[A_XP + 0] += dxp.im * z; acc[A_XP + 1] += (double)dxp.fx * zfx; acc[A_XP + 2] += (double)dxp.fy * zfy; acc[A_XP + 3] += dxp.m * zm;
            acc[A_YP + 0] += dyp.im * z; acc[A_YP + 1] += (double)dyp.fx * zfx; acc[A_YP + 2] += (double)dyp.fy * zfy; acc[A_YP + 3] += dyp.m * zm;
            acc[A_XM + 0] += dxm.im * z; acc[A_XM + 1] += (double)dxm.fx * zfx; acc[A_XM + 2] += (double)dxm.fy * zfy; acc[A_XM + 3] += dxm.m * zm;
            acc[A_YM + 0] += dym.im * z; acc[A_YM + 1] += (double)dym.fx * zfx; acc[A_YM + 2] += (double)dym.fy * zfy; acc[A_YM + 3] += dym.m * zm;
            acc[A_VXP] += (double)dvxp.fx * zfx; acc[A_VXM] += (double)dvxm.fx * zfx;
            acc[A_VYP] += (double)dvyp.fy * zfy; acc[A_VYM] += (double)dvym.fy * zfy;
            // HTH diagonal block (forward differences, cuda.py:993-996)
            acc[A_XX + 0] += dxp.im * dxp.im; acc[A_XX + 1] += (double)dxp.fx * (double)dxp.fx;
            acc[A_XX + 2] += (double)dxp.fy * (double)dxp.fy; acc[A_XX + 3] += dxp.m * dxp.m;
            acc[A_XY + 0] += dxp.im * dyp.im; acc[A_XY + 1] += (double)dxp.fx * (double)dyp.fx;
            acc[A_XY + 2] += (double)dxp.fy * (double)dyp.fy; acc[A_XY + 3] += dxp.m * dyp.m;
            acc[A_YY + 0] += dyp.im * dyp.im; acc[A_YY + 1] += (double)dyp.fx * (double)dyp.fx;
            acc[A_YY + 2] += (double)dyp.fy * (double)dyp.fy; acc[A_YY + 3] += dyp.m * dyp.m;
            acc[A_XVX] += (double)dxp.fx * (double)dvxp.fx; acc[A_YVX] += (double)dyp.fx * (double)dvxp.fx;
            acc[A_XVY] += (double)dxp.fy * (double)dvyp.fy; acc[A_YVY] += (double)dyp.fy * (double)dvyp.fy;
            acc[A_VXVX] += (double)dvxp.fx * (double)dvxp.fx; acc[A_VYVY] += (double)dvyp.fy * (double)dvyp.fy;
        } else {
            StarVel velv, velw, none;
            const StarVal vref = d_star_eval<true>(s_cfg[0], nsv, c, r, m, X, v, (float)(X[2 * N + 2 * v] + d), 0.0f,
                                                   (float)(-(X[2 * N + 2 * v + 1] + d)), 0.0f, velv);
            const StarVal vxp = d_star_eval<false>(s_cfg[1], nsv, c, r, m, X, v, 0, 0, 0, 0, none);
            const StarVal vyp = d_star_eval<false>(s_cfg[2], nsv, c, r, m, X, v, 0, 0, 0, 0, none);
            if (vref.cnt + vxp.cnt + vyp.cnt == 0) continue;
            const StarVal wref = d_star_eval<true>(s_cfg[3], nsw, c, r, m, X, w, (float)(X[2 * N + 2 * w] + d), 0.0f,
                                                   (float)(-(X[2 * N + 2 * w + 1] + d)), 0.0f, velw);
            const StarVal wxp = d_star_eval<false>(s_cfg[4], nsw, c, r, m, X, w, 0, 0, 0, 0, none);
            const StarVal wyp = d_star_eval<false>(s_cfg[5], nsw, c, r, m, X, w, 0, 0, 0, 0, none);
            if (wref.cnt + wxp.cnt + wyp.cnt == 0) continue;
            const StarVal vvx = {vref.acc, vref.cnt, velv.fxp, vref.fy}, vvy = {vref.acc, vref.cnt, vref.fx, velv.fyp};
            const StarVal wvx = {wref.acc, wref.cnt, velw.fxp, wref.fy}, wvy = {wref.acc, wref.cnt, wref.fx, velw.fyp};
            const Diff ax = d_diff(racc, rcnt, rfx, rfy, vref, vxp), ay = d_diff(racc, rcnt, rfx, rfy, vref, vyp);
            const Diff avx = d_diff(racc, rcnt, rfx, rfy, vref, vvx), avy = d_diff(racc, rcnt, rfx, rfy, vref, vvy);
            const Diff bx = d_diff(racc, rcnt, rfx, rfy, wref, wxp), by = d_diff(racc, rcnt, rfx, rfy, wref, wyp);
            const Diff bvx = d_diff(racc, rcnt, rfx, rfy, wref, wvx), bvy = d_diff(racc, rcnt, rfx, rfy, wref, wvy);
#define CH4(base, P, Q)                                                                           \
    acc[base + 0] += P.im * Q.im; acc[base + 1] += (double)P.fx * (double)Q.fx;                   \
    acc[base + 2] += (double)P.fy * (double)Q.fy; acc[base + 3] += P.m * Q.m;
            CH4(B_XX, ax, bx) CH4(B_XY, ax, by) CH4(B_YX, ay, bx) CH4(B_YY, ay, by)
#undef CH4
            acc[B_XVX] += (double)ax.fx * (double)bvx.fx; acc[B_YVX] += (double)ay.fx * (double)bvx.fx;
            acc[B_VXX] += (double)avx.fx * (double)bx.fx; acc[B_VXY] += (double)avx.fx * (double)by.fx;
            acc[B_VXVX] += (double)avx.fx * (double)bvx.fx;
            acc[B_XVY] += (double)ax.fy * (double)bvy.fy; acc[B_YVY] += (double)ay.fy * (double)bvy.fy;
            acc[B_VYX] += (double)avy.fy * (double)bx.fy; acc[B_VYY] += (double)avy.fy * (double)by.fy;
            acc[B_VYVY] += (double)avy.fy * (double)bvy.fy;
        }
    }
    d_block_reduce<NACC, MEAS_NT>(acc, s_red, a.out + ((size_t)job * MEAS_VSPLIT + blockIdx.y) * MEAS_OUT);
}


// ---- job sums -> Hz, Hz components, dense HTH (device twin of the unpacking in KFState.update) -------
// One thread per job; every entry of H is written by exactly one job (vertex jobs own the diagonal
// 4x4 blocks, edge jobs the two mirrored off-diagonal ones), so plain stores suffice.  H must be zero.
struct ScatterArgs {
    const double *out;        // njobs * MEAS_OUT
    const int *edges;
    int N, E;
    double eZ, eJ, eM, d;
    double *H, *Hz, *Hzc;     // 4N x 4N, 4N, 4N x 4
};

__device__ __forceinline__ void d_put(double *H, int n4, int p, int q, double val, double d)
{
    const double v = val / d / d;
    H[(size_t)p * n4 + q] = v;
    H[(size_t)q * n4 + p] = v;
}

__global__ void k_hth_scatter(ScatterArgs a)
{
    const int job = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.N, n4 = 4 * N;
    if (job >= N + a.E) return;
    double o[MEAS_OUT];
    {
        const double *src = a.out + (size_t)job * MEAS_VSPLIT * MEAS_OUT;
        const int parts = job < N ? MEAS_VSPLIT : 1;
        for (int k = 0; k < MEAS_OUT; k++) {
            double v = src[k];
            for (int q = 1; q < parts; q++) v += src[q * MEAS_OUT + k];
            o[k] = v;
        }
    }
    const double eZ = a.eZ, eJ = a.eJ, eM = a.eM, d = a.d;
#define SUM4(s) ((((s)[0] / eZ + (s)[1] / eJ) + (s)[2] / eJ) + (s)[3] / eM)
    if (job < N) {
        const int v = job;
        const int idx[4] = {2 * v, 2 * v + 1, 2 * N + 2 * v, 2 * N + 2 * v + 1};
        // central differences of jz (kalman.py:499-515); component sums carry the sign of jz_CPU
        const double cp[4][4] = {{o[A_XP] / eZ, o[A_XP + 1] / eJ, -o[A_XP + 2] / eJ, o[A_XP + 3] / eM},
                                 {o[A_YP] / eZ, o[A_YP + 1] / eJ, -o[A_YP + 2] / eJ, o[A_YP + 3] / eM},
                                 {0.0, o[A_VXP] / eJ, 0.0, 0.0},
                                 {0.0, 0.0, -o[A_VYP] / eJ, 0.0}};
        const double cm[4][4] = {{o[A_XM] / eZ, o[A_XM + 1] / eJ, -o[A_XM + 2] / eJ, o[A_XM + 3] / eM},
                                 {o[A_YM] / eZ, o[A_YM + 1] / eJ, -o[A_YM + 2] / eJ, o[A_YM + 3] / eM},
                                 {0.0, o[A_VXM] / eJ, 0.0, 0.0},
                                 {0.0, 0.0, -o[A_VYM] / eJ, 0.0}};
        for (int k = 0; k < 4; k++) {
            const double hp = ((cp[k][0] + cp[k][1]) + cp[k][2]) + cp[k][3];
            const double hm_ = ((cm[k][0] + cm[k][1]) + cm[k][2]) + cm[k][3];
            a.Hz[idx[k]] = (hp / d - hm_ / d) / 2;
            for (int ch = 0; ch < 4; ch++) a.Hzc[(size_t)idx[k] * 4 + ch] = (cp[k][ch] / d - cm[k][ch] / d) / 2;
        }
        const int ix = idx[0], iy = idx[1], ivx = idx[2], ivy = idx[3];
        d_put(a.H, n4, ix, ix, SUM4(o + A_XX), d);
        d_put(a.H, n4, ix, iy, SUM4(o + A_XY), d);
        d_put(a.H, n4, iy, iy, SUM4(o + A_YY), d);
        d_put(a.H, n4, ix, ivx, o[A_XVX] / eJ, d);
        d_put(a.H, n4, iy, ivx, o[A_YVX] / eJ, d);
        d_put(a.H, n4, ix, ivy, o[A_XVY] / eJ, d);
        d_put(a.H, n4, iy, ivy, o[A_YVY] / eJ, d);
        d_put(a.H, n4, ivx, ivx, o[A_VXVX] / eJ, d);
        d_put(a.H, n4, ivy, ivy, o[A_VYVY] / eJ, d);
    } else {
        const int e = job - N;
        const int v = a.edges[2 * e], w = a.edges[2 * e + 1];
        const int vx_ = 2 * v, vy_ = 2 * v + 1, vvx = 2 * N + 2 * v, vvy = 2 * N + 2 * v + 1;
        const int wx_ = 2 * w, wy_ = 2 * w + 1, wvx = 2 * N + 2 * w, wvy = 2 * N + 2 * w + 1;
        d_put(a.H, n4, vx_, wx_, SUM4(o + B_XX), d);
        d_put(a.H, n4, vx_, wy_, SUM4(o + B_XY), d);
        d_put(a.H, n4, vy_, wx_, SUM4(o + B_YX), d);
        d_put(a.H, n4, vy_, wy_, SUM4(o + B_YY), d);
        d_put(a.H, n4, vx_, wvx, o[B_XVX] / eJ, d);
        d_put(a.H, n4, vy_, wvx, o[B_YVX] / eJ, d);
        d_put(a.H, n4, vvx, wx_, o[B_VXX] / eJ, d);
        d_put(a.H, n4, vvx, wy_, o[B_VXY] / eJ, d);
        d_put(a.H, n4, vvx, wvx, o[B_VXVX] / eJ, d);
        d_put(a.H, n4, vx_, wvy, o[B_XVY] / eJ, d);
        d_put(a.H, n4, vy_, wvy, o[B_YVY] / eJ, d);
        d_put(a.H, n4, vvy, wx_, o[B_VYX] / eJ, d);
        d_put(a.H, n4, vvy, wy_, o[B_VYY] / eJ, d);
        d_put(a.H, n4, vvy, wvy, o[B_VYVY] / eJ, d);
    }
#undef SUM4
}
