// Small dense f64 algebra of the EKF update on the device (gfx950): blocked Cholesky of the
// 4N x 4N information matrix, triangular solves, SPD inverse.  Replaces the explicit
// numpy.linalg.inv calls of reference kalman.py:753-754, 785-786, 797-799.
//
// Matrices are row-major, n x n with leading dimension n; the factor L overwrites the lower
// triangle (the strict upper triangle is left untouched and never read).
#pragma once
#include <hip/hip_runtime.h>

#define DNB 32          // block size

// ---- potrf, step k, part 1: factor the diagonal block, solve the panel below it ---------------
// One 256-thread workgroup per block row r >= k.  Every workgroup factors the 32x32 diagonal
// block itself in LDS (11k flops, cheaper than a launch boundary; the rank-1 update of each of
// the 32 steps is spread over all threads).  Workgroup r == k stores the factor; the others
// solve X L_kk^T = A_rk for their 32 rows -- one row per thread in registers, L_kk read from
// LDS at wave-uniform addresses (broadcast) -- and store X.
__global__ __launch_bounds__(256) void k_chol_panel(double *__restrict__ A, int n, int k)
{
    __shared__ double D[DNB][DNB + 1];
    const int t = threadIdx.x;
    const int r = k + blockIdx.x;
    const int d0 = k * DNB;
    const int nd = min(DNB, n - d0);              // the last diagonal block may be short: pad with identity
    for (int e = t; e < DNB * DNB; e += 256) {
        int i = e / DNB, j = e % DNB;
        D[i][j] = (i < nd && j < nd && j <= i) ? A[(size_t)(d0 + i) * n + d0 + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    const int ti = t / DNB, tc = t % DNB;         // 8 x 32 thread grid over (row, column)
    for (int j = 0; j < DNB; j++) {
        const double piv = sqrt(D[j][j]);         // every thread reads the same word; one writes it back
        const double rpiv = 1.0 / piv;
        __syncthreads();
        if (t == 0) D[j][j] = piv;
        if (t > j && t < DNB) D[t][j] = D[t][j] * rpiv;
        __syncthreads();
        if (tc > j) {
            const double lcj = D[tc][j];
            for (int i = ti; i < DNB; i += 256 / DNB)
                if (i >= tc) D[i][tc] = D[i][tc] - D[i][j] * lcj;
        }
        __syncthreads();
    }
    if (r == k) {
        for (int e = t; e < DNB * DNB; e += 256) {
            int i = e / DNB, j = e % DNB;
            if (i < nd && j <= i) A[(size_t)(d0 + i) * n + d0 + j] = D[i][j];
        }
        return;
    }
    // rows of the panel: 8 threads per row, thread `part` keeps the entries x[c], c = part mod 8, in
    // four registers; each of the 32 substitution steps is a 4-term partial dot product per thread,
    // a butterfly sum over the 8 threads and one divide
    __shared__ double rD[DNB];                    // reciprocals of the diagonal of L_kk
    if (t < DNB) rD[t] = 1.0 / D[t][t];
    __syncthreads();
    const int r0 = r * DNB;
    const int nr = min(DNB, n - r0);
    const int row = t / 8, part = t % 8;
    double x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int c = 8 * q + part;
        x[q] = (row < nr && c < nd) ? A[(size_t)(r0 + row) * n + d0 + c] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < DNB; j++) {
        double partial = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = 8 * q + part;
            if (c < j) partial = partial + x[q] * D[j][c];
        }
        partial += __shfl_xor(partial, 4, 8);
        partial += __shfl_xor(partial, 2, 8);
        partial += __shfl_xor(partial, 1, 8);
        const double own = __shfl(x[j / 8], j % 8, 8);      // x[j] before the step, from its owner
        const double xj = (own - partial) * rD[j];
        if (part == j % 8) x[j / 8] = xj;
    }
    if (row < nr) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = 8 * q + part;
            if (c < nd) A[(size_t)(r0 + row) * n + d0 + c] = x[q];
        }
    }
}

// ---- potrf, step k, part 2: trailing update A_rc -= L_rk L_ck^T for r >= c > k -------------------
__global__ __launch_bounds__(256) void k_chol_update(double *__restrict__ A, int n, int k)
{
    const int r = k + 1 + blockIdx.y, c = k + 1 + blockIdx.x;
    if (c > r) return;
    __shared__ double R[DNB][DNB + 1];
    __shared__ double C[DNB][DNB + 1];
    const int t = threadIdx.x;
    const int r0 = r * DNB, c0 = c * DNB, d0 = k * DNB;
    const int nr = min(DNB, n - r0), nc = min(DNB, n - c0);
    for (int e = t; e < DNB * DNB; e += 256) {
        int i = e / DNB, j = e % DNB;
        R[i][j] = i < nr ? A[(size_t)(r0 + i) * n + d0 + j] : 0.0;
        C[i][j] = i < nc ? A[(size_t)(c0 + i) * n + d0 + j] : 0.0;
    }
    __syncthreads();
    const int j = t % DNB;                        // column inside the block
    for (int i = t / DNB; i < DNB; i += 256 / DNB) {
        if (i >= nr || j >= nc) continue;
        double s = 0.0;
#pragma unroll 8
        for (int q = 0; q < DNB; q++) s = s + R[i][q] * C[j][q];
        const size_t p = (size_t)(r0 + i) * n + c0 + j;
        A[p] = A[p] - s;
    }
}

// ---- potrs: solve L L^T X = B for a slab of CH right-hand-side columns per workgroup ------------
// B is n x ldb row-major; workgroup b handles columns [b*CH, b*CH + CH) and overwrites them with X.
// IDENT: the right-hand side is the identity (SPD inverse) and B is only written.
// The slab lives in LDS for the whole forward and backward substitution.  Per 32-row block:
// (a) all 256 threads subtract the contribution of the rows already solved (a 32 x done x CH
// product, L streamed from memory), (b) the diagonal block is staged in LDS and each wave solves
// the 32x32 triangle for its columns, one row per lane, passing x_j between lanes by shuffle.
template <int CH, bool IDENT>
__global__ __launch_bounds__(256) void k_chol_solve(const double *__restrict__ L, int n, double *__restrict__ B,
                                                    int ldb, int ncols)
{
    extern __shared__ double Y[];                 // n x CH
    __shared__ double T[DNB][CH + 1];
    __shared__ double Dg[DNB][DNB + 1];
    __shared__ double rDg[DNB];                   // reciprocals of the diagonal of the staged block
    constexpr int PARTS = 256 / DNB;              // 8 partial sums per (row, column slab)
    __shared__ double S[PARTS][DNB][CH + 1];
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int col0 = blockIdx.x * CH;
    const int nb = (n + DNB - 1) / DNB;
    for (int e = t; e < n * CH; e += 256) {
        int i = e / CH, c = e % CH;
        double v = 0.0;
        if (col0 + c < ncols) v = IDENT ? (i == col0 + c ? 1.0 : 0.0) : B[(size_t)i * ldb + col0 + c];
        Y[e] = v;
    }
    __syncthreads();
    // forward: L Y = B
    for (int kb = 0; kb < nb; kb++) {
        const int i0 = kb * DNB, ni = min(DNB, n - i0);
        for (int e = t; e < DNB * DNB; e += 256) {           // stage the diagonal block (identity padded)
            int i = e / DNB, j = e % DNB;
            const double v = (i < ni && j <= i) ? L[(size_t)(i0 + i) * n + i0 + j] : (i == j ? 1.0 : 0.0);
            Dg[i][j] = v;
            if (i == j) rDg[i] = 1.0 / v;
        }
        {
            const int i = t / PARTS, part = t % PARTS;        // row i of the block, every PARTS-th j
            double acc[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) acc[c] = 0.0;
            if (i < ni) {
                const double *Lrow = L + (size_t)(i0 + i) * n;
#pragma unroll 4
                for (int j = part; j < i0; j += PARTS) {
                    const double l = Lrow[j];
#pragma unroll
                    for (int c = 0; c < CH; c++) acc[c] = acc[c] + l * Y[j * CH + c];
                }
            }
#pragma unroll
            for (int c = 0; c < CH; c++) S[part][i][c] = acc[c];
        }
        __syncthreads();
        for (int e = t; e < DNB * CH; e += 256) {             // fixed-order sum of the partials
            int i = e / CH, c = e % CH;
            double v = 0.0;
            for (int q = 0; q < PARTS; q++) v += S[q][i][c];
            T[i][c] = (i < ni ? Y[(i0 + i) * CH + c] : 0.0) - v;
        }
        __syncthreads();
        for (int c = wv; c < CH; c += 4) {                    // triangle: one row per lane, x_j by shuffle
            const int i = lane & (DNB - 1);
            double val = T[i][c];
            for (int j = 0; j < DNB; j++) {
                const double xj = __shfl(val, j, 64) * rDg[j];
                if (i == j) val = xj;
                else if (i > j) val = val - Dg[i][j] * xj;
            }
            if (lane < ni) Y[(i0 + lane) * CH + c] = val;
        }
        __syncthreads();
    }
    // backward: L^T X = Y
    for (int kb = nb - 1; kb >= 0; kb--) {
        const int i0 = kb * DNB, ni = min(DNB, n - i0);
        const int j0 = i0 + ni;
        for (int e = t; e < DNB * DNB; e += 256) {
            int i = e / DNB, j = e % DNB;
            const double v = (i < ni && j <= i) ? L[(size_t)(i0 + i) * n + i0 + j] : (i == j ? 1.0 : 0.0);
            Dg[i][j] = v;
            if (i == j) rDg[i] = 1.0 / v;
        }
        {
            const int i = t % DNB, part = t / DNB;            // lanes run along i: L[j][i0+i] is contiguous in i
            double acc[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) acc[c] = 0.0;
            if (i < ni) {
#pragma unroll 4
                for (int j = j0 + part; j < n; j += PARTS) {
                    const double l = L[(size_t)j * n + i0 + i];
#pragma unroll
                    for (int c = 0; c < CH; c++) acc[c] = acc[c] + l * Y[j * CH + c];
                }
            }
#pragma unroll
            for (int c = 0; c < CH; c++) S[part][i][c] = acc[c];
        }
        __syncthreads();
        for (int e = t; e < DNB * CH; e += 256) {
            int i = e / CH, c = e % CH;
            double v = 0.0;
            for (int q = 0; q < PARTS; q++) v += S[q][i][c];
            T[i][c] = (i < ni ? Y[(i0 + i) * CH + c] : 0.0) - v;
        }
        __syncthreads();
        for (int c = wv; c < CH; c += 4) {
            const int i = lane & (DNB - 1);
            double val = T[i][c];
            for (int j = DNB - 1; j >= 0; j--) {
                const double xj = __shfl(val, j, 64) * rDg[j];
                if (i == j) val = xj;
                else if (i < j) val = val - Dg[j][i] * xj;
            }
            if (lane < ni) Y[(i0 + lane) * CH + c] = val;
        }
        __syncthreads();
    }
    for (int e = t; e < n * CH; e += 256) {
        int i = e / CH, c = e % CH;
        if (col0 + c < ncols) B[(size_t)i * ldb + col0 + c] = Y[e];
    }
}

// ---- assembly of the update system -----------------------------------------------------------------
// A = invW0 + H (elementwise)
__global__ void k_add_mat(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// y = base - M x   (one workgroup per row, fixed-order reduction)
__global__ __launch_bounds__(256) void k_rhs(const double *__restrict__ M, const double *__restrict__ x,
                                             const double *__restrict__ base, double *__restrict__ y, int n)
{
    __shared__ double s[4];
    const int row = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) acc += M[(size_t)row * n + j] * x[j];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) y[row] = base[row] - (((s[0] + s[1]) + s[2]) + s[3]);
}
