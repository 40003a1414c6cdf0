// KalmanFilter.projectmask (reference kalman.py:724-742) on the device (gfx950).
//
// The reference asks OpenCV for the signed distance of a vertex to the object contour
// (imgproc.py:175-248, outside this path); the product defines it as the Euclidean distance
// transform of the mask (object pixel -> nearest background pixel, negative; background pixel
// -> nearest object pixel, positive), sampled bilinearly -- see oracle/ekf_ref.py:project_mask.
// The nearest pixel of the other kind always touches the outline (has a 4-neighbour of the
// other kind), so the transform is never formed: k_outline compacts the outline pixels of
// either kind (a few thousand at 1024^2) and k_project_mask, one workgroup per vertex, takes
// the exact integer minimum of dx^2+dy^2 over them for the four pixels around a sample point.
// The arithmetic after that is the host's, operation by operation (no contraction), so the
// projected state is the same f64 numbers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct Outline {
    int2 *pts;       // capacity cap = W*H: object pixels from the front, background pixels from the back
    int *count;      // [0] object outline pixels, [1] background outline pixels, [2] vertices moved
    int cap;
};

#define OUTLINE_NT 256
// one wave per 64-pixel row segment; one atomic per wave and kind
__global__ __launch_bounds__(OUTLINE_NT) void k_outline(const uint8_t *__restrict__ ym, int W, int H, Outline o)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * (OUTLINE_NT / 64) + (threadIdx.x >> 6);
    bool on = false, obj = false;
    if (x < W && y < H) {
        const uint8_t *row = ym + (size_t)y * W;
        obj = row[x] > 0;
        if (x > 0) on |= (row[x - 1] > 0) != obj;
        if (x + 1 < W) on |= (row[x + 1] > 0) != obj;
        if (y > 0) on |= (row[x - W] > 0) != obj;
        if (y + 1 < H) on |= (row[x + W] > 0) != obj;
    }
    const unsigned long long bo = __ballot(on && obj), bb = __ballot(on && !obj);
    if (!(bo | bb)) return;
    int base_o = 0, base_b = 0;
    if (lane == 0) {
        if (bo) base_o = atomicAdd(&o.count[0], __popcll(bo));
        if (bb) base_b = atomicAdd(&o.count[1], __popcll(bb));
    }
    base_o = __shfl(base_o, 0);
    base_b = __shfl(base_b, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (on && obj) o.pts[base_o + __popcll(bo & below)] = make_int2(x, y);
    if (on && !obj) o.pts[o.cap - 1 - (base_b + __popcll(bb & below))] = make_int2(x, y);
}

struct ProjArgs {
    const uint8_t *ym;
    int W, H, N;
    Outline o;
    double *X;       // 4N, in place
};

#define PROJ_NT 256
#define PROJ_NONE 0x7fffffffffffffffLL

// signed distance at (x, y), uniform over the workgroup; red = PROJ_NT/64 * 4 LDS words
static __device__ double d_mask_distance(const ProjArgs &a, double x, double y, long long *red)
{
#pragma clang fp contract(off)
    const int W = a.W, H = a.H;
    const double xc = fmin(fmax(x, 0.0), (double)(W - 1)), yc = fmin(fmax(y, 0.0), (double)(H - 1));
    const int px = (int)floor(xc), py = (int)floor(yc);
    int qx[4], qy[4];
    bool in[4];
    for (int k = 0; k < 4; k++) {
        qx[k] = min(px + (k & 1), W - 1);
        qy[k] = min(py + (k >> 1), H - 1);
        in[k] = a.ym[(size_t)qy[k] * W + qx[k]] > 0;
    }
    long long best[4] = {PROJ_NONE, PROJ_NONE, PROJ_NONE, PROJ_NONE};
    const int n_obj = a.o.count[0], n_bg = a.o.count[1];
    if (!(in[0] && in[1] && in[2] && in[3]))
        for (int i = threadIdx.x; i < n_obj; i += PROJ_NT) {
            const int2 p = a.o.pts[i];
            for (int k = 0; k < 4; k++) {
                const long long dx = p.x - qx[k], dy = p.y - qy[k];
                if (!in[k]) best[k] = min(best[k], dx * dx + dy * dy);
            }
        }
    if (in[0] || in[1] || in[2] || in[3])
        for (int i = threadIdx.x; i < n_bg; i += PROJ_NT) {
            const int2 p = a.o.pts[a.o.cap - 1 - i];
            for (int k = 0; k < 4; k++) {
                const long long dx = p.x - qx[k], dy = p.y - qy[k];
                if (in[k]) best[k] = min(best[k], dx * dx + dy * dy);
            }
        }
    for (int k = 0; k < 4; k++)
        for (int s = 32; s > 0; s >>= 1) best[k] = min(best[k], __shfl_xor(best[k], s));
    const int wv = threadIdx.x >> 6;
    __syncthreads();                                   // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 4; k++) red[wv * 4 + k] = best[k];
    __syncthreads();
    double q[4];
    for (int k = 0; k < 4; k++) {
        long long b = red[k];
        for (int w = 1; w < PROJ_NT / 64; w++) b = min(b, red[w * 4 + k]);
        const double r = b == PROJ_NONE ? 0.0 : sqrt((double)b);
        q[k] = in[k] ? -r : r;
    }
    const double ax = xc - (double)px, ay = yc - (double)py;
    return (1.0 - ay) * ((1.0 - ax) * q[0] + ax * q[1]) + ay * ((1.0 - ax) * q[2] + ax * q[3]);
}

// one workgroup per vertex: 10 steps p -= d g / |g|^2 with forward differences of 0.1 px; d is
// the distance before the first step throughout and only vertices with d > 1 move, as in the
// reference; the displacement is added to the velocity
__global__ __launch_bounds__(PROJ_NT) void k_project_mask(ProjArgs a)
{
#pragma clang fp contract(off)
    __shared__ long long red[PROJ_NT / 64 * 4];
    const int v = blockIdx.x;
    const double x0 = a.X[2 * v], y0 = a.X[2 * v + 1];
    const double fx = floor(x0), fy = floor(y0);
    if (fx >= 0.0 && fy >= 0.0 && fx + 1.0 < (double)a.W && fy + 1.0 < (double)a.H) {
        const uint8_t *r = a.ym + (size_t)(int)fy * a.W + (int)fx;
        if (r[0] > 0 && r[1] > 0 && r[a.W] > 0 && r[a.W + 1] > 0) return;      // d <= 0
    }
    const double d = d_mask_distance(a, x0, y0, red);
    if (!(d > 1.0)) return;
    const double eps = 1e-1;
    double x = x0, y = y0;
    for (int it = 0; it < 10; it++) {
        const double gx = (d_mask_distance(a, x + eps, y + 0.0, red) - d) / eps;
        const double gy = (d_mask_distance(a, x + 0.0, y + eps, red) - d) / eps;
        const double g2 = gx * gx + gy * gy;
        const double step = g2 > 0.0 ? d / g2 : 0.0;
        x = x - step * gx;
        y = y - step * gy;
    }
    if (threadIdx.x == 0) {
        a.X[2 * v] = x;
        a.X[2 * v + 1] = y;
        a.X[2 * a.N + 2 * v] += x - x0;
        a.X[2 * a.N + 2 * v + 1] += y - y0;
        atomicAdd(&a.o.count[2], 1);
    }
}
