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
// The reference's contour pruning (imgproc.py:205-228: only the largest object and its holes of area >= 40) is
// applied to the mask first, per frame (k_ccl_* below); k_outline and the walk work on the pruned mask.
// The arithmetic is the oracle's (oracle/ekf_ref.py:outline_distance, project_mask), operation by
// operation with contraction off, so the projected state is the same f64 numbers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "host_block.h"

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

// ---- the reference's contour pruning on the device (imgproc.py:198-228 via kalman.py:725) --------------------------
// findObjectThreshold keeps the outer contour of the LARGEST object (cv2.contourArea) and of its holes those of area
// >= 40; smaller objects, smaller holes and everything nested deeper go (oracle/ekf_ref.py:pruned_object has the
// restatement this follows, areas by Pick's theorem).  Per frame, on the mask in device memory:
//   k_ccl_local    a 64 x 16 tile per workgroup, labelled in LDS: runs of a row by ballot, rows joined by a union-find
//                  (atomicMin towards the smaller index, so a component's root is its first pixel in raster order):
//                  object pixels join their W / NW / N / NE neighbours (8-connected), background pixels their W / N
//                  neighbours (4-connected) -- both kinds in the same pass, a pixel belongs to one of the two;
//   k_ccl_border   the same unions across tile borders, on the labels in memory;
//   k_ccl_roots    the first pixels of the tile-local components find their roots;
//   k_ccl_flatten  every pixel points to its root; background components that reach the frame edge are marked
//                  (findContours treats the frame as surrounded by background: they are outside, not holes);
//   k_ccl_stats    nesting comes from the roots: the pixel ABOVE a component's first pixel belongs to the component
//                  that encloses it (an enclosed component cannot have pixels above whatever encloses it).  Every
//                  pixel adds 1 to each contour it lies inside or on (its own object's outer contour, the hole that
//                  object sits in, ...); object pixels 4-adjacent to the outside / to a hole of their own object count
//                  the contours' boundary points;
//   k_ccl_select   the level-0 object with the largest area (ties: first in raster order);
//   k_ccl_write    the pruned mask: inside that object's outer contour and not inside one of its holes of area >= 40.
// All counts are whole numbers, areas are compared doubled (2 A = 2 inside -/+ boundary - 2): same decisions as the oracle.
struct Ccl {
    int *L;          // W*H: labels (pixel indices)
    int *cnt;        // W*H, used at roots: pixels inside or on the component's contour
    int *bnd;        // W*H, used at roots: boundary points of the contour (object: pixels 4-adjacent to the outside;
                     // hole: pixels of the enclosing object 4-adjacent to it)
    uint8_t *edge;   // W*H, used at background roots: 1 = reaches the frame edge
    unsigned long long *best;     // [0]: (2 A << 32) | ~root of the best level-0 object so far
    int W, H;
};

__device__ __forceinline__ int d_ccl_find(int *L, int i)
{
    int r = i;
    for (int p = L[r]; p != r; p = L[r]) r = p;
    for (int p = L[i]; p != r && p != i; p = L[i]) { atomicMin(&L[i], r); i = p; }      // (labels only ever decrease)
    return r;
}
__device__ __forceinline__ void d_ccl_unite(int *L, int a, int b)
{
    for (;;) {
        a = d_ccl_find(L, a);
        b = d_ccl_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&L[b], a);
        if (old == b) return;
        b = old;
    }
}

#define CCL_NT 256
#define CCL_TW 64
#define CCL_TH 16
// the same union-find on a tile's labels in LDS (local pixel indices)
__device__ __forceinline__ int d_ccl_find_lds(int *L, int i)
{
    int r = i;
    for (int p = L[r]; p != r; p = L[r]) r = p;
    return r;
}
__device__ __forceinline__ void d_ccl_unite_lds(int *L, int a, int b)
{
    for (;;) {
        a = d_ccl_find_lds(L, a);
        b = d_ccl_find_lds(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&L[b], a);
        if (old == b) return;
        b = old;
    }
}

// grid (ceil(W / 64), ceil(H / 16)): a workgroup labels a 64 x 16 tile in LDS -- runs of a row by ballot, the rows of
// the tile joined by the union-find above -- and writes, per pixel, the GLOBAL index of its component's first pixel
// inside the tile: flat trees, one hop deep, for the unions across tile borders that follow (k_ccl_border).  (One
// union-find over the whole frame from single pixels took 0.6 ms at 1024^2: chains as long as the object is high.)
__global__ __launch_bounds__(CCL_NT) void k_ccl_local(const uint8_t *__restrict__ ym, Ccl c)
{
    __shared__ int sl[CCL_TH * CCL_TW];
    __shared__ signed char scls[CCL_TH][CCL_TW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = blockIdx.x * CCL_TW, y0 = blockIdx.y * CCL_TH, x = x0 + lane;
    const int W = c.W, H = c.H;
#pragma unroll
    for (int q = 0; q < CCL_TH / 4; q++) {
        const int r = wv * (CCL_TH / 4) + q, y = y0 + r;
        const bool in = x < W && y < H;
        const bool fg = in && ym[(size_t)y * W + x] > 0;
        const unsigned long long bf = __ballot(fg), bi = __ballot(in);
        // lanes that start a run: lane 0, a class different from the lane before, the first lane off the frame
        const unsigned long long prev_f = (bf << 1) | (bf & 1ull), prev_i = (bi << 1) | (bi & 1ull);
        const unsigned long long starts = (bf ^ prev_f) | (bi ^ prev_i) | 1ull;
        const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        sl[r * CCL_TW + lane] = r * CCL_TW + (63 - __clzll(upto));
        scls[r][lane] = in ? (fg ? 1 : 0) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CCL_TH / 4; q++) {
        const int r = wv * (CCL_TH / 4) + q;
        if (r == 0) continue;
        const int cl = scls[r][lane], me = r * CCL_TW + lane, up = me - CCL_TW;
        if (cl < 0) continue;
        const bool n = scls[r - 1][lane] == 1;
        if (cl == 1) {
            // the run above through N; when N is background, the (different) runs of NW and NE
            if (n) d_ccl_unite_lds(sl, me, up);
            else {
                if (lane > 0 && scls[r - 1][lane - 1] == 1) d_ccl_unite_lds(sl, me, up - 1);
                if (lane < 63 && scls[r - 1][lane + 1] == 1) d_ccl_unite_lds(sl, me, up + 1);
            }
        } else if (scls[r - 1][lane] == 0) {
            d_ccl_unite_lds(sl, me, up);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CCL_TH / 4; q++) {
        const int r = wv * (CCL_TH / 4) + q, y = y0 + r;
        if (x >= W || y >= H) continue;
        const int root = d_ccl_find_lds(sl, r * CCL_TW + lane);
        const size_t p = (size_t)y * W + x;
        c.L[p] = (y0 + root / CCL_TW) * W + x0 + (root % CCL_TW);
        c.cnt[p] = root == r * CCL_TW + lane ? 1 : 0;      // marks the first pixel of a tile-local component (k_ccl_roots)
        c.bnd[p] = 0;
        c.edge[p] = 0;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) c.best[0] = 0ull;
}

// the unions across tile borders: grid (ceil(W / 64), ceil(H / 4)), a wave per 64-pixel row segment.  Along a border
// most neighbouring pixels ask for the same union (the same two tile-local components meet along a whole edge): a lane
// whose pair of labels is the pair of the lane before it -- or, down a tile's side, of the row above it -- leaves the
// union to that one (100 000 -> a few thousand unions at 1024^2, all of which would otherwise fight over one root).
__device__ __forceinline__ void d_ccl_unite_once(int *L, int p, int q, bool dup)
{
    if (!dup) d_ccl_unite(L, p, q);
}
__global__ __launch_bounds__(CCL_NT) void k_ccl_border(const uint8_t *__restrict__ ym, Ccl c)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * (CCL_NT / 64) + (threadIdx.x >> 6);
    if (x >= c.W || y >= c.H) return;
    const bool top = (y % CCL_TH) == 0, left = lane == 0, right = lane == 63;
    if (!top && !left && !right) return;
    const int W = c.W, p = y * W + x;
    const uint8_t *row = ym + (size_t)y * W;
    const bool fg = row[x] > 0;
    const int lp = c.L[p];                           // tile-local labels (k_ccl_local): what "the same union" is judged by
    // W: down the left side of a tile the pair (label of p, label of p - 1) usually repeats row after row
    if (left && x > 0 && (row[x - 1] > 0) == fg) {
        const bool dup = !top && c.L[p - W] == lp && c.L[p - W - 1] == c.L[p - 1];
        d_ccl_unite_once(c.L, p, p - 1, dup);
    }
    if (y == 0) return;
    const uint8_t *up = row - W;
    const bool n = up[x] > 0;
    if (fg) {
        if (n) {
            if (top) {
                // along the top row: the lane before asked for the same union when both its labels are the same
                const bool dup = lane > 0 && c.L[p - 1] == lp && c.L[p - W - 1] == c.L[p - W] && row[x - 1] > 0 && up[x - 1] > 0;
                d_ccl_unite_once(c.L, p, p - W, dup);
            }
        } else {
            if ((top || left) && x > 0 && up[x - 1] > 0) d_ccl_unite(c.L, p, p - W - 1);
            if ((top || right) && x + 1 < W && up[x + 1] > 0) d_ccl_unite(c.L, p, p - W + 1);
        }
    } else if (!n && top) {
        const bool dup = lane > 0 && c.L[p - 1] == lp && c.L[p - W - 1] == c.L[p - W] && !(row[x - 1] > 0) && !(up[x - 1] > 0);
        d_ccl_unite_once(c.L, p, p - W, dup);
    }
}

// the first pixels of the tile-local components (a few thousand) walk to their roots and point at them: every other pixel
// is then two hops from its root
__global__ __launch_bounds__(CCL_NT) void k_ccl_roots(Ccl c)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * (CCL_NT / 64) + (threadIdx.x >> 6);
    if (x >= c.W || y >= c.H) return;
    const int p = y * c.W + x;
    if (!c.cnt[p]) return;
    c.cnt[p] = 0;
    (void)d_ccl_find(c.L, p);
}

__global__ __launch_bounds__(CCL_NT) void k_ccl_flatten(const uint8_t *__restrict__ ym, Ccl c)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * (CCL_NT / 64) + (threadIdx.x >> 6);
    if (x >= c.W || y >= c.H) return;
    const int p = y * c.W + x;
    // p -> its tile's first pixel of the component -> the root (k_ccl_roots has shortened that hop; a pixel that lost a
    // race there walks on): without the compressing atomics of d_ccl_find, a million of which took 45 us
    int r = c.L[p];
    for (int q = c.L[r]; q != r; q = c.L[r]) r = q;
    c.L[p] = r;
    if (!(ym[p] > 0) && (x == 0 || y == 0 || x == c.W - 1 || y == c.H - 1)) c.edge[r] = 1;
}

// the component that encloses the component with root r (a pixel index): the one the pixel above r belongs to;
// -1: the frame edge (r in row 0)
__device__ __forceinline__ int d_ccl_parent(const Ccl &c, int r) { return r < c.W ? -1 : c.L[r - c.W]; }

__global__ __launch_bounds__(CCL_NT) void k_ccl_stats(const uint8_t *__restrict__ ym, Ccl c)
{
    __shared__ int s_root, s_n, s_b;
    if (threadIdx.x == 0) { s_root = -1; s_n = 0; s_b = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * (CCL_NT / 64) + (threadIdx.x >> 6);
    const int W = c.W, H = c.H;
    const bool in = x < W && y < H;
    const int p = in ? y * W + x : 0;
    const bool fg = in && ym[p] > 0;
    const int r = in ? c.L[p] : -1;
    // the workgroup's commonest object gathers its counts in LDS (one object covers most of a mask)
    if (fg) atomicCAS(&s_root, -1, r);
    __syncthreads();
    const int hot = s_root;
    if (in) {
        // 1: the contours this pixel lies inside or on
        int cur = r;
        bool curfg = fg;
        for (int depth = 0; depth < 64; depth++) {
            if (curfg) {
                if (cur == hot) atomicAdd(&s_n, 1); else atomicAdd(&c.cnt[cur], 1);
                const int par = d_ccl_parent(c, cur);
                if (par < 0 || c.edge[par]) break;              // level 0: outside is the frame or background that reaches it
                cur = par; curfg = false;
            } else {
                if (c.edge[cur]) break;                          // outside
                atomicAdd(&c.cnt[cur], 1);
                cur = d_ccl_parent(c, cur);                      // (an enclosed background component is never in row 0)
                curfg = true;
            }
        }
        // 2: boundary points
        if (fg) {
            bool outer = x == 0 || y == 0 || x == W - 1 || y == H - 1;
            int holes[4], nh = 0;
            const int nbr[4] = {x > 0 ? p - 1 : -1, x + 1 < W ? p + 1 : -1, y > 0 ? p - W : -1, y + 1 < H ? p + W : -1};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (nbr[k] < 0 || ym[nbr[k]] > 0) continue;
                const int b = c.L[nbr[k]];
                if (c.edge[b]) { outer = true; continue; }
                if (d_ccl_parent(c, b) != r) continue;           // this object sits IN that hole: the hole's contour runs elsewhere
                bool seen = false;
                for (int q = 0; q < nh; q++) seen = seen || holes[q] == b;
                if (!seen) holes[nh++] = b;
            }
            if (outer) { if (r == hot) atomicAdd(&s_b, 1); else atomicAdd(&c.bnd[r], 1); }
            for (int q = 0; q < nh; q++) atomicAdd(&c.bnd[holes[q]], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && hot >= 0) {
        if (s_n) atomicAdd(&c.cnt[hot], s_n);
        if (s_b) atomicAdd(&c.bnd[hot], s_b);
    }
}

// doubled area of the outer contour of the object with root r / of the hole with root r
__device__ __forceinline__ long long d_ccl_area2_object(const Ccl &c, int r) { return 2ll * c.cnt[r] - c.bnd[r] - 2; }
__device__ __forceinline__ long long d_ccl_area2_hole(const Ccl &c, int r) { return 2ll * c.cnt[r] + c.bnd[r] - 2; }

__global__ __launch_bounds__(CCL_NT) void k_ccl_select(const uint8_t *__restrict__ ym, Ccl c)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * (CCL_NT / 64) + (threadIdx.x >> 6);
    if (x >= c.W || y >= c.H) return;
    const int p = y * c.W + x;
    if (!(ym[p] > 0) || c.L[p] != p) return;                     // roots of objects only
    const int par = d_ccl_parent(c, p);
    if (par >= 0 && !c.edge[par]) return;                        // nested in a hole: not level 0
    const long long a2 = d_ccl_area2_object(c, p);
    if (a2 < 0) return;
    atomicMax(c.best, ((unsigned long long)a2 << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p));
}

// pruned mask: 1 inside the kept outer contour and outside every kept hole (doubled areas against 2 x 40)
__global__ __launch_bounds__(CCL_NT) void k_ccl_write(const uint8_t *__restrict__ ym, Ccl c, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * (CCL_NT / 64) + (threadIdx.x >> 6);
    if (x >= c.W || y >= c.H) return;
    const int p = y * c.W + x;
    const unsigned long long key = c.best[0];
    const long long a2 = (long long)(key >> 32);
    const int best = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    uint8_t v = 0;
    if (key != 0ull && a2 >= 80) {
        int cur = c.L[p];
        bool curfg = ym[p] > 0;
        for (int depth = 0; depth < 64; depth++) {
            if (curfg) {
                if (cur == best) { v = 1; break; }
                const int par = d_ccl_parent(c, cur);
                if (par < 0 || c.edge[par]) break;               // another level-0 object
                cur = par; curfg = false;
            } else {
                if (c.edge[cur]) break;                          // outside
                const int own = d_ccl_parent(c, cur);
                if (own == best) { v = d_ccl_area2_hole(c, cur) >= 80 ? 0 : 1; break; }    // a hole of the kept object: kept or filled
                cur = own; curfg = true;
            }
        }
    }
    out[p] = v;
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

// The same for a frame of the filter, without copy operations and without a stream synchronisation (hm_project_mask,
// hm_chain_project).  xin: the predicted state (4N) -- page-locked host memory written before the launch was queued, or
// the device buffer the state prediction's kernel left it in; read with system-scope loads either way.  blk: a result
// block of host_block.h in page-locked host memory, 4N + 1 values: the projected state and the number of vertices that
// were moved (published by the workgroup that finishes last).  dev_out (may be NULL): the projected state in device
// memory for the kernels queued behind this one (the prior mean of the update; dev_out2: a second copy, the update's first
// iterate).
__global__ __launch_bounds__(PROJ_NT) void k_project_mask_host(ProjArgs a, const double *xin, double *blk, double *dev_out, double *dev_out2,
                                                               int *done, double ticket, int delay_us)
{
#pragma clang fp contract(off)
    __shared__ double red[PROJ_NT / 64 * 3];
    const int v = blockIdx.x, N = a.N;
    const double x0 = hb_host_in(xin + 2 * v), y0 = hb_host_in(xin + 2 * v + 1);
    double x, y;
    const bool moved = d_project_vertex(a, x0, y0, red, x, y);
    if (threadIdx.x != 0) return;
    const unsigned long long stamp = hb_stamp((long long)ticket);
    const double vx0 = hb_host_in(xin + 2 * N + 2 * v), vy0 = hb_host_in(xin + 2 * N + 2 * v + 1);
    const double vx = moved ? vx0 + (x - x0) : vx0, vy = moved ? vy0 + (y - y0) : vy0;
    if (dev_out) { dev_out[2 * v] = x; dev_out[2 * v + 1] = y; dev_out[2 * N + 2 * v] = vx; dev_out[2 * N + 2 * v + 1] = vy; }
    if (dev_out2) { dev_out2[2 * v] = x; dev_out2[2 * v + 1] = y; dev_out2[2 * N + 2 * v] = vx; dev_out2[2 * N + 2 * v + 1] = vy; }
    if (delay_us > 0 && v == 0) {                            // test knob "result_delay": vertex 0's pairs come late
        hb_put(blk, 2 * N, vx, stamp);
        hb_flush();
        hb_delay(delay_us);
    }
    hb_put(blk, 2 * v, x, stamp); hb_put(blk, 2 * v + 1, y, stamp);
    hb_put(blk, 2 * N + 2 * v, vx, stamp); hb_put(blk, 2 * N + 2 * v + 1, vy, stamp);
    if (moved) atomicAdd(&a.o.count[2], 1);
    hb_flush();                                              // (system scope: also orders the count before the arrival below)
    if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
        __threadfence();
        hb_put(blk, 4 * N, (double)atomicExch(&a.o.count[2], 0), stamp);        // (left clean for another projection onto the same mask)
        *done = 0;
        hb_flush();
    }
}
