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
// One 64-thread workgroup per block row r >= k.  Every workgroup factors the 32x32 diagonal
// block itself (11k flops, cheaper than a launch boundary); workgroup r == k stores it, the
// others solve X L_kk^T = A_rk for their 32 rows and store X.
__global__ __launch_bounds__(64) void k_chol_panel(double *__restrict__ A, int n, int k)
{
    __shared__ double D[DNB][DNB + 1];
    __shared__ double P[DNB][DNB + 1];
    const int lane = threadIdx.x;
    const int r = k + blockIdx.x;
    const int d0 = k * DNB;
    const int nd = min(DNB, n - d0);              // size of the diagonal block (last one may be short)
    for (int e = lane; e < DNB * DNB; e += 64) {
        int i = e / DNB, j = e % DNB;
        D[i][j] = (i < nd && j < nd && j <= i) ? A[(size_t)(d0 + i) * n + d0 + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int j = 0; j < nd; j++) {
        if (lane == 0) D[j][j] = sqrt(D[j][j]);
        __syncthreads();
        const double piv = D[j][j];
        if (lane > j && lane < nd) D[lane][j] = D[lane][j] / piv;
        __syncthreads();
        // rank-1 update of the remaining lower triangle: lane owns row `lane`
        if (lane > j && lane < nd) {
            const double lij = D[lane][j];
            for (int c = j + 1; c <= lane; c++) D[lane][c] = D[lane][c] - lij * D[c][j];
        }
        __syncthreads();
    }
    if (r == k) {
        for (int e = lane; e < DNB * DNB; e += 64) {
            int i = e / DNB, j = e % DNB;
            if (i < nd && j <= i) A[(size_t)(d0 + i) * n + d0 + j] = D[i][j];
        }
        return;
    }
    const int r0 = r * DNB;
    const int nr = min(DNB, n - r0);
    for (int e = lane; e < DNB * DNB; e += 64) {
        int i = e / DNB, j = e % DNB;
        P[i][j] = (i < nr && j < nd) ? A[(size_t)(r0 + i) * n + d0 + j] : 0.0;
    }
    __syncthreads();
    if (lane < nr) {                              // row `lane` of X: forward substitution along the columns
        for (int j = 0; j < nd; j++) {
            double s = P[lane][j];
            for (int c = 0; c < j; c++) s = s - P[lane][c] * D[j][c];
            P[lane][j] = s / D[j][j];
        }
    }
    __syncthreads();
    for (int e = lane; e < DNB * DNB; e += 64) {
        int i = e / DNB, j = e % DNB;
        if (i < nr && j < nd) A[(size_t)(r0 + i) * n + d0 + j] = P[i][j];
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
// The slab lives in LDS for the whole forward and backward substitution.
template <int CH, bool IDENT>
__global__ __launch_bounds__(256) void k_chol_solve(const double *__restrict__ L, int n, double *__restrict__ B,
                                                    int ldb, int ncols)
{
    extern __shared__ double Y[];                 // n x CH
    __shared__ double T[DNB][CH + 1];
    const int t = threadIdx.x;
    const int col0 = blockIdx.x * CH;
    const int nb = (n + DNB - 1) / DNB;
    for (int e = t; e < n * CH; e += 256) {
        int i = e / CH, c = e % CH;
        double v = 0.0;
        if (col0 + c < ncols) v = IDENT ? (i == col0 + c ? 1.0 : 0.0) : B[(size_t)i * ldb + col0 + c];
        Y[e] = v;
    }
    __syncthreads();
    constexpr int PARTS = 256 / DNB;              // 8 partial sums per (row, column slab)
    // forward: L Y = B
    for (int kb = 0; kb < nb; kb++) {
        const int i0 = kb * DNB, ni = min(DNB, n - i0);
        {
            const int i = t / PARTS, part = t % PARTS;    // row i of the block, every PARTS-th j
            double acc[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) acc[c] = 0.0;
            if (i < ni) {
                const double *Lrow = L + (size_t)(i0 + i) * n;
                for (int j = part; j < i0; j += PARTS) {
                    const double l = Lrow[j];
#pragma unroll
                    for (int c = 0; c < CH; c++) acc[c] = acc[c] + l * Y[j * CH + c];
                }
            }
            // reduce the PARTS partials (consecutive lanes) in a fixed order
#pragma unroll
            for (int c = 0; c < CH; c++) {
                double v = acc[c];
                for (int o = PARTS / 2; o > 0; o >>= 1) v += __shfl_down(v, o, PARTS);
                if (part == 0 && i < ni) T[i][c] = Y[(i0 + i) * CH + c] - v;
            }
        }
        __syncthreads();
        if (t < CH) {                             // one thread per column: 32 sequential rows
            for (int i = 0; i < ni; i++) {
                const double *Lrow = L + (size_t)(i0 + i) * n + i0;
                double s = T[i][t];
                for (int j = 0; j < i; j++) s = s - Lrow[j] * T[j][t];
                T[i][t] = s / Lrow[i];
            }
            for (int i = 0; i < ni; i++) Y[(i0 + i) * CH + t] = T[i][t];
        }
        __syncthreads();
    }
    // backward: L^T X = Y
    for (int kb = nb - 1; kb >= 0; kb--) {
        const int i0 = kb * DNB, ni = min(DNB, n - i0);
        const int j0 = i0 + ni;
        {
            const int i = t % DNB, part = t / DNB;        // lanes run along i: L[j][i0+i] is contiguous in i
            double acc[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) acc[c] = 0.0;
            if (i < ni) {
                for (int j = j0 + part; j < n; j += PARTS) {
                    const double l = L[(size_t)j * n + i0 + i];
#pragma unroll
                    for (int c = 0; c < CH; c++) acc[c] = acc[c] + l * Y[j * CH + c];
                }
            }
            __shared__ double S[PARTS][DNB][CH + 1];
#pragma unroll
            for (int c = 0; c < CH; c++) S[part][i][c] = acc[c];
            __syncthreads();
            if (part == 0 && i < ni) {
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    double v = 0.0;
                    for (int q = 0; q < PARTS; q++) v += S[q][i][c];
                    T[i][c] = Y[(i0 + i) * CH + c] - v;
                }
            }
        }
        __syncthreads();
        if (t < CH) {
            for (int i = ni - 1; i >= 0; i--) {
                double s = T[i][t];
                for (int j = i + 1; j < ni; j++) s = s - L[(size_t)(i0 + j) * n + i0 + i] * T[j][t];
                T[i][t] = s / L[(size_t)(i0 + i) * n + i0 + i];
            }
            for (int i = 0; i < ni; i++) Y[(i0 + i) * CH + t] = T[i][t];
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
