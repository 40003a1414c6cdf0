// KalmanFilter.projectmask (reference kalman.py:724-742) on the device (gfx950).
//
// The reference asks OpenCV for the signed distance of a vertex to the object's contour:
// fd(p) = -cv2.pointPolygonTest(contour, p, True) with the contour cv2.findContours traces through
// the border pixels of the mask (imgproc.py:195-235) -- the distance to the POLYGON THROUGH THE
// CENTRES OF THE OBJECT'S BORDER PIXELS, negative inside.  Here:
//   * k_outline marks and compacts the border pixels of the object: object pixels with a 4-neighbour
//     that is background or lies outside the frame (findContours treats the frame as surrounded by
//     background);
//   * the polygon's sides are the segments between 8-adjacent border pixels (a traced contour uses a
//     subset of them; the others are diagonal short cuts that lie on the object's side of the contour,
//     so for a point outside the object -- the only place the walk below evaluates it, d > 1 -- the
//     nearest point is the same).  k_project_mask, one workgroup per vertex, takes the exact minimum of
//     the squared point-to-segment distance over all of them in binary64 (min does not depend on the
//     order) and the sign from the 2x2 pixels around the point: inside iff all four are object, or
//     three and the point lies on their side of the diagonal -- inside the polygon through the centres.
// What is NOT applied per frame is the reference's contour pruning (imgproc.py:205-228: only the largest
// object and its holes of at least 40 px): every border pixel of the mask counts.  For a mask with one object
// and no small holes the numbers are those of imgproc.findObjectThreshold(mask).fd (tests).
// The arithmetic is the oracle's (oracle/ekf_ref.py:outline_distance, project_mask), operation by
// operation with contraction off, so the projected state is the same f64 numbers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct Outline {
    int2 *pts;       // capacity W*H: border pixels of the object
    int *count;      // [0] border pixels, [1] (unused), [2] vertices moved
    int cap;
    uint8_t *flag;   // W*H: 1 at border pixels, else 0
};

#define OUTLINE_NT 256
// one wave per 64-pixel row segment; one atomic per wave
__global__ __launch_bounds__(OUTLINE_NT) void k_outline(const uint8_t *__restrict__ ym, int W, int H, Outline o)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * (OUTLINE_NT / 64) + (threadIdx.x >> 6);
    bool on = false;
    if (x < W && y < H) {
        const uint8_t *row = ym + (size_t)y * W;
        if (row[x] > 0) {
            on = x == 0 || x == W - 1 || y == 0 || y == H - 1;
            if (!on) on = !(row[x - 1] > 0) || !(row[x + 1] > 0) || !(row[x - W] > 0) || !(row[x + W] > 0);
        }
        o.flag[(size_t)y * W + x] = on ? 1 : 0;
    }
    const unsigned long long bo = __ballot(on);
    if (!bo) return;
    int base_o = 0;
    if (lane == 0) base_o = atomicAdd(&o.count[0], __popcll(bo));
    base_o = __shfl(base_o, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (on) o.pts[base_o + __popcll(bo & below)] = make_int2(x, y);
}

struct ProjArgs {
    const uint8_t *ym;
    int W, H, N;
    Outline o;
    double *X;       // 4N, in place
};

#define PROJ_NT 256

// signed distance at (x, y), uniform over the workgroup; red = PROJ_NT/64 * 3 doubles of LDS
static __device__ double d_mask_distance(const ProjArgs &a, double x, double y, double *red)
{
#pragma clang fp contract(off)
    const int W = a.W, H = a.H;
    const int n_obj = a.o.count[0];
    if (n_obj == 0) return 0.0;                        // a blank mask has no outline to be pulled to
    double best_seg = 1e300, best_pt = 1e300, nseg = 0.0;
    for (int i = threadIdx.x; i < n_obj; i += PROJ_NT) {
        const int2 p = a.o.pts[i];
        const double apx = x - (double)p.x, apy = y - (double)p.y;
        best_pt = fmin(best_pt, apx * apx + apy * apy);
        // each 8-adjacent pair once: right, lower left, lower, lower right
        const int ddx[4] = {1, -1, 0, 1}, ddy[4] = {0, 1, 1, 1};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int qx = p.x + ddx[k], qy = p.y + ddy[k];
            if (qx < 0 || qx >= W || qy >= H || !a.o.flag[(size_t)qy * W + qx]) continue;
            const double abx = (double)ddx[k], aby = (double)ddy[k], den = abx * abx + aby * aby;
            double t = (apx * abx + apy * aby) / den;
            t = fmin(fmax(t, 0.0), 1.0);
            const double ex = apx - t * abx, ey = apy - t * aby;
            best_seg = fmin(best_seg, ex * ex + ey * ey);
            nseg += 1.0;
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        best_seg = fmin(best_seg, __shfl_xor(best_seg, s));
        best_pt = fmin(best_pt, __shfl_xor(best_pt, s));
        nseg += __shfl_xor(nseg, s);
    }
    const int wv = threadIdx.x >> 6;
    __syncthreads();                                   // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) { red[wv * 3] = best_seg; red[wv * 3 + 1] = best_pt; red[wv * 3 + 2] = nseg; }
    __syncthreads();
    double bs = red[0], bp = red[1], ns = red[2];
    for (int w = 1; w < PROJ_NT / 64; w++) { bs = fmin(bs, red[w * 3]); bp = fmin(bp, red[w * 3 + 1]); ns += red[w * 3 + 2]; }
    const double dist = sqrt(ns > 0.0 ? bs : bp);      // a lone pixel is its own (degenerate) polygon
    // inside the polygon through the pixel centres: the corners of the point's grid cell
    bool ins = false;
    if (!(x < 0.0 || y < 0.0 || x > (double)(W - 1) || y > (double)(H - 1))) {
        const int x0 = min(max((int)floor(x), 0), W - 1), y0 = min(max((int)floor(y), 0), H - 1);
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const double fx = fmin(fmax(x - (double)x0, 0.0), 1.0), fy = fmin(fmax(y - (double)y0, 0.0), 1.0);
        const bool c00 = a.ym[(size_t)y0 * W + x0] > 0, c10 = a.ym[(size_t)y0 * W + x1] > 0;
        const bool c01 = a.ym[(size_t)y1 * W + x0] > 0, c11 = a.ym[(size_t)y1 * W + x1] > 0;
        const int n = (int)c00 + (int)c10 + (int)c01 + (int)c11;
        ins = n == 4;
        if (n == 3) {                                   // on the side of the diagonal away from the missing corner
            ins = (!c00 && fx + fy >= 1.0) || (!c11 && fx + fy <= 1.0) || (!c10 && fy >= fx) || (!c01 && fy <= fx);
        }
    }
    return ins ? -dist : dist;
}

// one workgroup per vertex: 10 steps p -= d g / |g|^2 with forward differences of 0.1 px; d is
// the distance before the first step throughout and only vertices with d > 1 move, as in the
// reference; the displacement is added to the velocity.  Uniform over the workgroup: false = the vertex stays.
static __device__ bool d_project_vertex(const ProjArgs &a, double x0, double y0, double *red, double &x, double &y)
{
#pragma clang fp contract(off)
    x = x0; y = y0;
    const double fx = floor(x0), fy = floor(y0);
    if (fx >= 0.0 && fy >= 0.0 && fx + 1.0 < (double)a.W && fy + 1.0 < (double)a.H) {
        const uint8_t *r = a.ym + (size_t)(int)fy * a.W + (int)fx;
        if (r[0] > 0 && r[1] > 0 && r[a.W] > 0 && r[a.W + 1] > 0) return false;      // d <= 0
    }
    const double d = d_mask_distance(a, x0, y0, red);
    if (!(d > 1.0)) return false;
    const double eps = 1e-1;
    for (int it = 0; it < 10; it++) {
        const double gx = (d_mask_distance(a, x + eps, y + 0.0, red) - d) / eps;
        const double gy = (d_mask_distance(a, x + 0.0, y + eps, red) - d) / eps;
        const double g2 = gx * gx + gy * gy;
        const double step = g2 > 0.0 ? d / g2 : 0.0;
        x = x - step * gx;
        y = y - step * gy;
    }
    return true;
}

// the state in device memory, in place (hm_project_mask with a host mask: the fine-grained path)
__global__ __launch_bounds__(PROJ_NT) void k_project_mask(ProjArgs a)
{
#pragma clang fp contract(off)
    __shared__ double red[PROJ_NT / 64 * 3];
    const int v = blockIdx.x;
    const double x0 = a.X[2 * v], y0 = a.X[2 * v + 1];
    double x, y;
    if (!d_project_vertex(a, x0, y0, red, x, y)) return;
    if (threadIdx.x == 0) {
        a.X[2 * v] = x;
        a.X[2 * v + 1] = y;
        a.X[2 * a.N + 2 * v] += x - x0;
        a.X[2 * a.N + 2 * v + 1] += y - y0;
        atomicAdd(&a.o.count[2], 1);
    }
}

// The same with the state read from and written to page-locked host memory (a frame of the filter: the predicted
// state comes from the host's Newton loop and goes back to it): `io` = [X (4N) | projected X (4N) | vertices moved |
// ticket].  Every workgroup writes its vertex's four entries; the one that finishes last adds the count and, behind a
// system-scope fence, the ticket the host is watching -- no copies, no stream synchronisation (hm_project_mask).
__global__ __launch_bounds__(PROJ_NT) void k_project_mask_host(ProjArgs a, double *io, int *done, double ticket)
{
#pragma clang fp contract(off)
    __shared__ double red[PROJ_NT / 64 * 3];
    const int v = blockIdx.x, N = a.N;
    const double x0 = io[2 * v], y0 = io[2 * v + 1];
    double x, y;
    const bool moved = d_project_vertex(a, x0, y0, red, x, y);
    if (threadIdx.x != 0) return;
    double *out = io + 4 * N;
    out[2 * v] = x;
    out[2 * v + 1] = y;
    out[2 * N + 2 * v] = moved ? io[2 * N + 2 * v] + (x - x0) : io[2 * N + 2 * v];
    out[2 * N + 2 * v + 1] = moved ? io[2 * N + 2 * v + 1] + (y - y0) : io[2 * N + 2 * v + 1];
    if (moved) atomicAdd(&a.o.count[2], 1);
    __threadfence_system();
    if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
        __threadfence();
        io[8 * N] = (double)atomicExch(&a.o.count[2], 0);        // (left clean for another projection onto the same mask)
        *done = 0;
        __threadfence_system();
        io[8 * N + 1] = ticket;
    }
}
