// Small dense f64 algebra of the EKF update on the device (gfx950): blocked Cholesky of the
// 4N x 4N information matrix, the solve with it, the SPD inverse.  Replaces the explicit
// numpy.linalg.inv calls of reference kalman.py:753-754, 785-786, 797-799.
//
// Matrices are row-major with leading dimension n.  A factorisation reads a working copy A (which
// it destroys) and writes three things: the factor's blocks below the 32x32 block diagonal to an
// array L of the same shape and the inverses of its 32x32 diagonal blocks to a side array Lt
// (identity padded when n is not a multiple of 32).  A may carry extra rows below row nb*32
// (nb = ceil(n/32)): right-hand sides stored as ROWS.  The factorisation treats them like any other
// block row, which turns them into (L^-1 b)^T -- the forward substitution of a solve comes for
// free; T = L^-1 comes out of the same launches (the elimination applied to an identity as well) and
// x = T^T y (k_tvec) finishes it.  32x32x32 block products run on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64); launches communicate through memory only, in a fixed order, so results
// do not depend on workgroup scheduling.
#pragma once
#include <hip/hip_runtime.h>
#include "ekf_kernels.h"      // d_tile_partial_sums (k_iter_result)
#include "host_block.h"

#define DNB 32          // block size
#define TTT_PF 3        // block products whose operands are in flight (k_ttt)

// 1/sqrt(d) from the hardware estimate (2^-24 relative) plus ONE third-order step, r (1 + e/2 + 3 e^2/8) with
// e = 1 - d r^2: 0.62 ulp at most, 0.20 on average -- the figures of two Newton steps, measured over 4 M arguments --
// in five dependent operations instead of six.  The factorisation has a column-by-column dependency chain; a
// correctly rounded square root and divide (a few hundred cycles each in f64) would sit on it 32 times per block.
__device__ __forceinline__ double d_rsqrt(double d)
{
    const double r = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * r), r, 1.0);
#ifdef HM_RSQRT_ORDER2
    return fma(r * 0.5, e, r);                     // (experiment: one dependent operation less, ~2^-47 instead of 0.62 ulp)
#else
    return fma(r, fma(e, 0.375, 0.5) * e, r);
#endif
}

// ---- potrf, one launch per block column -----------------------------------------------------------
// The factorisation reads a working copy A and writes the factor to a separate array L (same
// shape; blocks below the block diagonal) and the inverses of the factored diagonal blocks to Lt.  Launch k (k_chol_step) does everything that involves block column k and has no later
// dependency:
//   workgroup (r, c), r >= c > k:  X_r = A_rk T_k^T and X_c = A_ck T_k^T with T_k = L_kk^-1 (Lt[k], left
//       by the previous launch; each workgroup forms the two panel blocks it needs itself, from the raw
//       panel in A), then A_rc -= X_r X_c^T -- 32x32x32 products on the f64 matrix cores;
//   workgroups of block column c = k+1 also store X_r as L_rk;
//   workgroup (k+1, k+1) goes on to turn its updated block into the inverse of its factor, Lt[k+1]
//       (chol32_tinv_wave) -- the only part that is sequential across launches.
// Nothing is written that another workgroup of the same launch reads (the panel stays raw in A),
// so the result does not depend on workgroup scheduling.
__device__ __forceinline__ double d_readlane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---- the 32x32 diagonal block: T = chol(B)^-1 in the registers of one wave ---------------------------
// The only part of the factorisation that is sequential across launches, so it is kept short: no
// LDS, no barrier, no substitution.  B (symmetric, both triangles, identity padded) and T (identity on
// entry) are held as 2x2 tiles of 16x16 in the accumulator layout of v_mfma_f64_16x16x4_f64 (element e
// of a lane: row 16R + (lane >> 4) + 4e, column 16C + (lane & 15)) and forward elimination is applied
// to [B | I], four columns at a time -- it ends as [L^T | L^-1].  Strip g (columns j = 4g .. j+3):
//   P = B[j:j+4, j:j+4] is broadcast (v_readlane) and every lane forms T44 = chol(P)^-1, a 4x4 job;
//   U = T44 [B | T][j:j+4, :] -- the pivot rows are accumulator element g & 3 of the tiles in tile row
//       g >> 2, which is exactly the B-operand layout (k = lane >> 4), and T44 padded to 16x4 is the
//       A operand: one MFMA per column tile, whose element 0 is U, again in B-operand layout;
//   rows below: [B | T][r, :] -= X[r, :] U with X[r, :] = B[r, j:j+4] T44^T = U[:, r]^T (symmetry): the A
//       operand of row tile R is the very register that holds U for column tile R;
//   rows j .. j+3 of T are final: U's T half.
// 48 MFMAs and eight 4x4 factorisations instead of 32 column steps with 16 barriers: 3.5 us instead of 6.1.
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4_t d_mfma4(double a, double b, d4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <int G>
__device__ __forceinline__ void chol32_strip(d4_t (&b)[2][2], d4_t (&t)[2][2], int lane)
{
    constexpr int j = 4 * G, RJ = G >> 2, E = G & 3, JJ = j & 15;
    // P[a][c] sits in lane 16 a + JJ + c of accumulator element E of tile (RJ, RJ)
    const double pv = b[RJ][RJ][E];
    const double p00 = d_readlane(pv, JJ);
    const double p10 = d_readlane(pv, 16 + JJ), p11 = d_readlane(pv, 16 + JJ + 1);
    const double p20 = d_readlane(pv, 32 + JJ), p21 = d_readlane(pv, 32 + JJ + 1), p22 = d_readlane(pv, 32 + JJ + 2);
    const double p30 = d_readlane(pv, 48 + JJ), p31 = d_readlane(pv, 48 + JJ + 1), p32 = d_readlane(pv, 48 + JJ + 2),
                 p33 = d_readlane(pv, 48 + JJ + 3);
    // L44 = chol(P) (l = p / sqrt(p) on the diagonal is never needed, only the reciprocals r); fused
    // multiply-adds throughout: this is the chain every launch of the factorisation waits for
    const double r0 = d_rsqrt(p00);
    const double l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
    const double r1 = d_rsqrt(fma(-l10, l10, p11));
    const double l21 = fma(-l20, l10, p21) * r1, l31 = fma(-l30, l10, p31) * r1;
    const double r2 = d_rsqrt(fma(-l21, l21, fma(-l20, l20, p22)));
    const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * r2;
    const double r3 = d_rsqrt(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33))));
    // T44 = L44^-1
    double t10 = -(l10 * r0) * r1;
    double t20 = -fma(l21, t10, l20 * r0) * r2, t21 = -(l21 * r1) * r2;
    double t30 = -fma(l32, t20, fma(l31, t10, l30 * r0)) * r3, t31 = -fma(l32, t21, l31 * r1) * r3, t32 = -(l32 * r2) * r3;
    // (the 4x4 job is the same in every lane and stays so: without this the compiler sinks each entry into
    // the lanes that select it below and the wave walks through the branches one after the other)
    asm volatile("" : "+v"(t10), "+v"(t20), "+v"(t21), "+v"(t30), "+v"(t31), "+v"(t32));
    // A operand: lane holds T44[a][kq], a = lane & 15 (rows 4..15: zero), kq = lane >> 4
    const int a = lane & 15, kq = lane >> 4;
    const int idx = a < 4 ? 4 * a + kq : 16;
    double ta = 0.0;                               // a flat chain of selects (v_cndmask), no control flow
    ta = idx == 0 ? r0 : ta;
    ta = idx == 4 ? t10 : ta; ta = idx == 5 ? r1 : ta;
    ta = idx == 8 ? t20 : ta; ta = idx == 9 ? t21 : ta; ta = idx == 10 ? r2 : ta;
    ta = idx == 12 ? t30 : ta; ta = idx == 13 ? t31 : ta; ta = idx == 14 ? t32 : ta; ta = idx == 15 ? r3 : ta;
    const d4_t z = {0.0, 0.0, 0.0, 0.0};
    // which tiles still matter: rows / columns >= j + 4 of B, columns <= j + 3 of T
    constexpr bool live0 = j + 4 <= 15, live1 = j + 4 <= 31;      // tile row / column 0, 1 of B
    constexpr bool tcol1 = j >= 16;                                // T has entries in column tile 1
    double uB0 = 0.0, uB1 = 0.0, uT0, uT1 = 0.0;
    if (live0) uB0 = d_mfma4(ta, b[RJ][0][E], z)[0];
    if (live1) uB1 = d_mfma4(ta, b[RJ][1][E], z)[0];
    uT0 = d_mfma4(ta, t[RJ][0][E], z)[0];
    if (tcol1) uT1 = d_mfma4(ta, t[RJ][1][E], z)[0];
    if (live0) {
        const double xa = a >= j + 4 ? -uB0 : 0.0;
        b[0][0] = d_mfma4(xa, uB0, b[0][0]);
        b[0][1] = d_mfma4(xa, uB1, b[0][1]);
        t[0][0] = d_mfma4(xa, uT0, t[0][0]);
        if (tcol1) t[0][1] = d_mfma4(xa, uT1, t[0][1]);
    }
    if (live1) {
        const double xa = 16 + a >= j + 4 ? -uB1 : 0.0;
        if (live0) b[1][0] = d_mfma4(xa, uB0, b[1][0]);
        b[1][1] = d_mfma4(xa, uB1, b[1][1]);
        t[1][0] = d_mfma4(xa, uT0, t[1][0]);
        if (tcol1) t[1][1] = d_mfma4(xa, uT1, t[1][1]);
    }
    t[RJ][0][E] = uT0;
    if (tcol1) t[RJ][1][E] = uT1;
}

// ---- the same elimination split over two waves ---------------------------------------------------------------
// Measured (tools/chol32_bench.hip, one block in a loop): 3.29 us in one wave, of which 0.78 are the 22 MFMAs that
// are not on the way to the next pivot and 0.71 the eight 4x4 jobs; the MFMA pipe of ONE SIMD and in-order issue of
// ONE wave carry all of it.  The T half never feeds back into the pivots: wave B keeps the B tiles (pivot, 4x4 job,
// U of the B tiles, their updates; tile (1,0) is never read and is not kept at all), wave T keeps the T tiles and
// gets, per strip, T44 in A-operand layout (`ta`) and the two A operands of the row updates (`xa`) through LDS.
// Hand-off as between the tasks of k_chol_flow: a slot holds a NaN pattern no computation produces until its value
// is stored; the T wave re-loads until all its lanes see values, and puts the pattern back when it has them (wave B
// writes a slot once per block, and blocks are separated by workgroup barriers).  Every MFMA has the operands it has
// in chol32_strip, so T has the same bits: 2.87 us per block.
#define CHOL_X_SENTINEL 0x7FF8DEADBEEF0002ull
struct CholX { double v[8][3][64]; };               // [strip][ta, xa of row tile 0, xa of row tile 1][lane]
__device__ __forceinline__ void d_lds_put(double *slot, double v)
{
    __hip_atomic_store((unsigned long long *)slot, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double d_lds_get(double *slot)
{
    unsigned long long x;
    do {
        x = __hip_atomic_load((unsigned long long *)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } while (!__all(x != CHOL_X_SENTINEL));
    return __longlong_as_double((long long)x);
}
__device__ __forceinline__ void chol32_x_clear(CholX &X, int t, int nt)
{
    for (int e = t; e < 8 * 3 * 64; e += nt) (&X.v[0][0][0])[e] = __longlong_as_double((long long)CHOL_X_SENTINEL);
}

template <int G>
__device__ __forceinline__ void chol32_strip_b(d4_t &b00, d4_t &b01, d4_t &b11, CholX &X, int lane)
{
    constexpr int j = 4 * G, RJ = G >> 2, E = G & 3, JJ = j & 15;
    const double pv = RJ ? b11[E] : b00[E];
    const double p00 = d_readlane(pv, JJ);
    const double p10 = d_readlane(pv, 16 + JJ), p11 = d_readlane(pv, 16 + JJ + 1);
    const double p20 = d_readlane(pv, 32 + JJ), p21 = d_readlane(pv, 32 + JJ + 1), p22 = d_readlane(pv, 32 + JJ + 2);
    const double p30 = d_readlane(pv, 48 + JJ), p31 = d_readlane(pv, 48 + JJ + 1), p32 = d_readlane(pv, 48 + JJ + 2),
                 p33 = d_readlane(pv, 48 + JJ + 3);
    const double r0 = d_rsqrt(p00);
    const double l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
    const double r1 = d_rsqrt(fma(-l10, l10, p11));
    const double l21 = fma(-l20, l10, p21) * r1, l31 = fma(-l30, l10, p31) * r1;
    const double r2 = d_rsqrt(fma(-l21, l21, fma(-l20, l20, p22)));
    const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * r2;
    const double r3 = d_rsqrt(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33))));
    double t10 = -(l10 * r0) * r1;
    double t20 = -fma(l21, t10, l20 * r0) * r2, t21 = -(l21 * r1) * r2;
    double t30 = -fma(l32, t20, fma(l31, t10, l30 * r0)) * r3, t31 = -fma(l32, t21, l31 * r1) * r3, t32 = -(l32 * r2) * r3;
    asm volatile("" : "+v"(t10), "+v"(t20), "+v"(t21), "+v"(t30), "+v"(t31), "+v"(t32));
    const int a = lane & 15, kq = lane >> 4;
    const int idx = a < 4 ? 4 * a + kq : 16;
    double ta = 0.0;
    ta = idx == 0 ? r0 : ta;
    ta = idx == 4 ? t10 : ta; ta = idx == 5 ? r1 : ta;
    ta = idx == 8 ? t20 : ta; ta = idx == 9 ? t21 : ta; ta = idx == 10 ? r2 : ta;
    ta = idx == 12 ? t30 : ta; ta = idx == 13 ? t31 : ta; ta = idx == 14 ? t32 : ta; ta = idx == 15 ? r3 : ta;
    d_lds_put(&X.v[G][0][lane], ta);
    const d4_t z = {0.0, 0.0, 0.0, 0.0};
    constexpr bool live0 = j + 4 <= 15, live1 = j + 4 <= 31;
    double uB0 = 0.0, uB1 = 0.0;
    if (live0) uB0 = d_mfma4(ta, b00[E], z)[0];
    if (live1) uB1 = d_mfma4(ta, (RJ ? b11 : b01)[E], z)[0];
    if (live0) {
        const double xa = a >= j + 4 ? -uB0 : 0.0;
        d_lds_put(&X.v[G][1][lane], xa);
        b00 = d_mfma4(xa, uB0, b00);
        b01 = d_mfma4(xa, uB1, b01);
    }
    if (live1) {
        const double xa = 16 + a >= j + 4 ? -uB1 : 0.0;
        d_lds_put(&X.v[G][2][lane], xa);
        b11 = d_mfma4(xa, uB1, b11);
    }
}

// `rows_done(j, u0, u1)`: rows j .. j+3 of T are final -- lane holds row j + (lane >> 4), columns (lane & 15) and
// 16 + (lane & 15); the caller publishes them while the wave waits for the next strip anyway
template <int G, typename Done>
__device__ __forceinline__ void chol32_strip_t(d4_t (&t)[2][2], CholX &X, int lane, Done &&rows_done)
{
    constexpr int j = 4 * G, RJ = G >> 2, E = G & 3;
    constexpr bool live0 = j + 4 <= 15, live1 = j + 4 <= 31;
    constexpr bool tcol1 = j >= 16;
    const d4_t z = {0.0, 0.0, 0.0, 0.0};
    const double sent = __longlong_as_double((long long)CHOL_X_SENTINEL);
    const double ta = d_lds_get(&X.v[G][0][lane]);
    double uT0, uT1 = 0.0;
    uT0 = d_mfma4(ta, t[RJ][0][E], z)[0];
    if (tcol1) uT1 = d_mfma4(ta, t[RJ][1][E], z)[0];
    if (live0) {
        const double xa = d_lds_get(&X.v[G][1][lane]);
        t[0][0] = d_mfma4(xa, uT0, t[0][0]);
        if (tcol1) t[0][1] = d_mfma4(xa, uT1, t[0][1]);
        X.v[G][1][lane] = sent;
    }
    if (live1) {
        const double xa = d_lds_get(&X.v[G][2][lane]);
        t[1][0] = d_mfma4(xa, uT0, t[1][0]);
        if (tcol1) t[1][1] = d_mfma4(xa, uT1, t[1][1]);
        X.v[G][2][lane] = sent;
    }
    X.v[G][0][lane] = sent;
    t[RJ][0][E] = uT0;
    if (tcol1) t[RJ][1][E] = uT1;
    rows_done(j, uT0, uT1);
}

// wave B of the pair: the B tiles of W (LDS, full symmetric block, identity padded) through the eight strips
__device__ __forceinline__ void chol32_wave_b(double (*W)[DNB + 1], CholX &X, int lane)
{
    d4_t b00, b01, b11;
    const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        b00[e] = W[lr + 4 * e][lc];
        b01[e] = W[lr + 4 * e][16 + lc];
        b11[e] = W[16 + lr + 4 * e][16 + lc];
    }
    chol32_strip_b<0>(b00, b01, b11, X, lane); chol32_strip_b<1>(b00, b01, b11, X, lane);
    chol32_strip_b<2>(b00, b01, b11, X, lane); chol32_strip_b<3>(b00, b01, b11, X, lane);
    chol32_strip_b<4>(b00, b01, b11, X, lane); chol32_strip_b<5>(b00, b01, b11, X, lane);
    chol32_strip_b<6>(b00, b01, b11, X, lane); chol32_strip_b<7>(b00, b01, b11, X, lane);
}

// wave T of the pair: T = chol(B)^-1 in the registers of this wave (accumulator layout, 2 x 2 tiles)
template <typename Done>
__device__ __forceinline__ void chol32_wave_t(d4_t (&t)[2][2], CholX &X, int lane, Done &&rows_done)
{
    const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
    for (int R = 0; R < 2; R++)
#pragma unroll
        for (int C = 0; C < 2; C++)
#pragma unroll
            for (int e = 0; e < 4; e++) t[R][C][e] = (16 * R + lr + 4 * e) == (16 * C + lc) ? 1.0 : 0.0;
    chol32_strip_t<0>(t, X, lane, rows_done); chol32_strip_t<1>(t, X, lane, rows_done); chol32_strip_t<2>(t, X, lane, rows_done);
    chol32_strip_t<3>(t, X, lane, rows_done); chol32_strip_t<4>(t, X, lane, rows_done); chol32_strip_t<5>(t, X, lane, rows_done);
    chol32_strip_t<6>(t, X, lane, rows_done); chol32_strip_t<7>(t, X, lane, rows_done);
}

// One wave: B from LDS (W, full symmetric block, identity padded) -> T = chol(B)^-1 to global memory
// (row-major 32x32 at `out`)
__device__ __forceinline__ void chol32_tinv_wave(double (*W)[DNB + 1], double *__restrict__ out, int lane)
{
    d4_t b[2][2], t[2][2];
    const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
    for (int R = 0; R < 2; R++)
#pragma unroll
        for (int C = 0; C < 2; C++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 16 * R + lr + 4 * e, jc = 16 * C + lc;
                b[R][C][e] = W[i][jc];
                t[R][C][e] = i == jc ? 1.0 : 0.0;
            }
    chol32_strip<0>(b, t, lane); chol32_strip<1>(b, t, lane); chol32_strip<2>(b, t, lane); chol32_strip<3>(b, t, lane);
    chol32_strip<4>(b, t, lane); chol32_strip<5>(b, t, lane); chol32_strip<6>(b, t, lane); chol32_strip<7>(b, t, lane);
#pragma unroll
    for (int R = 0; R < 2; R++)
#pragma unroll
        for (int C = 0; C < 2; C++)
#pragma unroll
            for (int e = 0; e < 4; e++) out[(16 * R + lr + 4 * e) * DNB + 16 * C + lc] = t[R][C][e];
}

// One 16x16 tile of X Y^T for 32x32 blocks X, Y in LDS on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64, eight k-steps): wave wv computes tile (wv >> 1, wv & 1); element e of the
// result sits at row 16 (wv >> 1) + (lane >> 4) + 4 e, column 16 (wv & 1) + (lane & 15).
__device__ __forceinline__ d4_t d_mfma_nt(double (*X)[DNB + 1], double (*Y)[DNB + 1], int wv, int lane)
{
    d4_t c = {0.0, 0.0, 0.0, 0.0};
    const int i = 16 * (wv >> 1) + (lane & 15), j = 16 * (wv & 1) + (lane & 15), kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < DNB / 4; kk++)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(X[i][4 * kk + kq], Y[j][4 * kk + kq], c, 0, 0, 0);
    return c;
}

// the same for X Y
__device__ __forceinline__ d4_t d_mfma_nn(double (*X)[DNB + 1], double (*Y)[DNB + 1], int wv, int lane, d4_t c)
{
    const int i = 16 * (wv >> 1) + (lane & 15), j = 16 * (wv & 1) + (lane & 15), kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < DNB / 4; kk++)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(X[i][4 * kk + kq], Y[4 * kk + kq][j], c, 0, 0, 0);
    return c;
}

// the first diagonal block: the inverse of the factor of A_00 -> Lt[0]
__global__ __launch_bounds__(64) void k_chol_first(const double *__restrict__ A, double *__restrict__ Lt, int n)
{
    __shared__ double W[DNB][DNB + 1];
    const int t = threadIdx.x;
    const int nd = min(DNB, n);
    for (int e = t; e < DNB * DNB; e += 64) {
        const int i = e / DNB, j = e % DNB;
        W[i][j] = (i < nd && j < nd) ? A[(size_t)i * n + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    chol32_tinv_wave(W, Lt, t);
}

// Step k.  Workgroup (r, c), r >= c > k:  X_r = A_rk T^T, X_c = A_ck T^T with T = L_kk^-1 (Lt[k], from
// the previous launch), A_rc -= X_r X_c^T -- three 32x32x32 products on the matrix cores; column
// c = k+1 also stores X_r as L_rk; workgroup (k+1, k+1) goes on to factor its updated block into
// and to invert the factor into Lt[k+1].
//
// The inverse of the whole factor comes out of the same launches: the elimination is applied to an
// identity as well ([A | I] -> [L^T | L^-1], what chol32_tinv_wave does inside a block, here block by
// block).  M holds the right half while it is worked on, Tinv receives its finished rows:
//   workgroups (r, j), k < r < nb, j <= k (grid columns mc ..):   M_rj -= X_r (T_k M_kj)
//   workgroups (j), j <= k (the extra grid row):                   Tinv_kj = T_k M_kj
// with M_kk = I and M_rj = 0 before its first update (step j), so neither array needs clearing.  These
// workgroups are as independent as the others and shorter than the diagonal one: the steps get no
// longer, and the ten launches of a separate inversion are gone.
__global__ __launch_bounds__(256) void k_chol_step(double *__restrict__ A, double *__restrict__ L, double *__restrict__ Lt,
                                                   double *__restrict__ Tinv, double *__restrict__ M,
                                                   int n, int nrows, int nb, int k, int mc, int gx, int gy)
{
    // The grid is one-dimensional and decoded here: linear workgroup ids are dealt to the XCDs round robin,
    // and the three workgroups whose output the next launch's diagonal workgroup reads -- (0,0) itself (T),
    // (0,1) (its panel block) and (1,1) (its diagonal block) -- take the ids 0, 8 and 16, i.e. one XCD: part of
    // what that workgroup waits for then comes out of its own L2 (8.8 instead of 9.3 us per step).
    int lin = blockIdx.x;
    if (gy >= 2 && gx >= 2 && gx * gy > 16) {
        const int n1 = gx, n2 = gx + 1;               // the natural ids of (0,1) and (1,1)
        if (lin == 8) lin = n1;
        else if (lin == 16) lin = n2;
        else {                                        // the others keep their order
            lin -= (lin > 8) + (lin > 16);
            lin += lin >= n1;
            lin += lin >= n2;
        }
    }
    const int bidx = lin % gx, bidy = lin / gx;
    __shared__ double Ts[DNB][DNB + 1];           // T = L_kk^-1
    __shared__ double Br[DNB][DNB + 1];           // A_rk, then X_r, then the updated block k+1
    __shared__ double Bc[DNB][DNB + 1];           // A_ck, then X_c (M_kj, then T_k M_kj in the inverse's workgroups)
    if (bidy == gy - 1 || bidx >= mc) {
        const bool fin = bidy == gy - 1;             // finish row k of the inverse
        const int j = fin ? bidx : bidx - mc;
        const int r = k + 1 + bidy;
        if (j > k || (!fin && r >= nb)) return;
        const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
        const int d0 = k * DNB, j0 = j * DNB, r0 = r * DNB;
        const int nd = min(DNB, n - d0), nr = fin ? 0 : min(DNB, n - r0);
        const int mj = 16 * (wv & 1) + (lane & 15);
        int mi[4];
#pragma unroll
        for (int e = 0; e < 4; e++) mi[e] = 16 * (wv >> 1) + (lane >> 4) + 4 * e;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (!fin && j < k) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (mi[e] < nr) a[e] = M[(size_t)(r0 + mi[e]) * n + j0 + mj];
        }
        for (int e = t; e < DNB * DNB; e += 256) {
            const int i = e / DNB, c = e % DNB;
            Ts[i][c] = Lt[(size_t)k * DNB * DNB + e];
            Bc[i][c] = j == k ? (i == c ? 1.0 : 0.0) : (i < nd ? M[(size_t)(d0 + i) * n + j0 + c] : 0.0);
            if (!fin) Br[i][c] = (i < nr && c < nd) ? A[(size_t)(r0 + i) * n + d0 + c] : 0.0;
        }
        __syncthreads();
        const d4_t z = {0.0, 0.0, 0.0, 0.0};
        const d4_t u = d_mfma_nn(Ts, Bc, wv, lane, z);            // T_k M_kj
        if (fin) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (mi[e] < nd && j0 + mj < n) Tinv[(size_t)(d0 + mi[e]) * n + j0 + mj] = u[e];
            return;
        }
        const d4_t xr = d_mfma_nt(Br, Ts, wv, lane);              // X_r = A_rk T_k^T
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; e++) { Br[mi[e]][mj] = xr[e]; Bc[mi[e]][mj] = u[e]; }
        __syncthreads();
        const d4_t s = d_mfma_nn(Br, Bc, wv, lane, z);
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (mi[e] < nr) M[(size_t)(r0 + mi[e]) * n + j0 + mj] = a[e] - s[e];
        return;
    }
    const int r = k + 1 + bidy;
    const int c = k + 1 + bidx;             // c >= nb: no block to update, the panel row only
    const bool panel_only = c >= nb;
    if (!panel_only && c > r) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int d0 = k * DNB;
    const int nd = min(DNB, n - d0);
    const int r0 = r * DNB, c0 = c * DNB;
    const int nr = min(DNB, nrows - r0);
    const int nc = panel_only ? 0 : min(DNB, n - c0);
    const bool two = !panel_only && c != r;
    // coordinates of this thread's four elements of a product tile
    const int mj = 16 * (wv & 1) + (lane & 15);
    int mi[4];
#pragma unroll
    for (int e = 0; e < 4; e++) mi[e] = 16 * (wv >> 1) + (lane >> 4) + 4 * e;
    double a[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
        a[e] = (!panel_only && mi[e] < nr && mj < nc) ? A[(size_t)(r0 + mi[e]) * n + c0 + mj] : 0.0;
    for (int e = t; e < DNB * DNB; e += 256) {
        const int i = e / DNB, j = e % DNB;
        Ts[i][j] = Lt[(size_t)k * DNB * DNB + e];
        Br[i][j] = (i < nr && j < nd) ? A[(size_t)(r0 + i) * n + d0 + j] : 0.0;
        Bc[i][j] = (two && i < nc && j < nd) ? A[(size_t)(c0 + i) * n + d0 + j] : 0.0;
    }
    __syncthreads();
    const d4_t xr = d_mfma_nt(Br, Ts, wv, lane);
    d4_t xc = xr;
    if (two) xc = d_mfma_nt(Bc, Ts, wv, lane);
    __syncthreads();                              // all reads of A_rk, A_ck done
#pragma unroll
    for (int e = 0; e < 4; e++) {
        Br[mi[e]][mj] = xr[e];
        Bc[mi[e]][mj] = xc[e];
        if (bidx == 0 && mi[e] < nr && mj < nd) L[(size_t)(r0 + mi[e]) * n + d0 + mj] = xr[e];
    }
    if (panel_only) return;
    __syncthreads();
    const d4_t s = d_mfma_nt(Br, Bc, wv, lane);
    const bool diag = r == k + 1 && c == k + 1;
    if (!diag) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (mi[e] < nr && mj < nc) A[(size_t)(r0 + mi[e]) * n + c0 + mj] = a[e] - s[e];
        return;
    }
    // the next diagonal block: finish its update (the whole symmetric block), then wave 0 alone turns it
    // into the inverse of its factor
    __syncthreads();                              // everyone is done reading X_r
#pragma unroll
    for (int e = 0; e < 4; e++) Br[mi[e]][mj] = (mi[e] < nc && mj < nc) ? a[e] - s[e] : (mi[e] == mj ? 1.0 : 0.0);
    __syncthreads();
    if (wv != 0) return;
    chol32_tinv_wave(Br, Lt + (size_t)(k + 1) * DNB * DNB, lane);
}

// ---- x = T^T y: the second half of a solve with A = L L^T once T = L^-1 is at hand ---------------
// y = L^-1 b is the right-hand-side row that went through the factorisation; x = A^-1 b = T^T y.
// One 1024-thread workgroup per TV_COLS columns: thread (g, c) adds T[i][col] y[i] over the rows i = i0 + g,
// + TV_ROWS, ... from the first row i0 of the column's diagonal block on (what lies above the diagonal inside that
// block is stored as zeros; above the block nothing is stored) -- at most ceil(n / TV_ROWS) loads per thread, all in
// flight together: T was written by another launch a moment ago and comes from memory, one round trip instead of the
// four of the 32-column form (26 workgroups, 26 loads per thread); 64-byte pieces of a row per workgroup.  The TV_ROWS
// partial sums per column are added in ascending order.  x goes to xout (not onto y: other workgroups still read it);
// if xn is given, xn = x0 + x is written as well (the update's new iterate).
#define TV_COLS 8
#define TV_ROWS 128
__global__ __launch_bounds__(TV_COLS * TV_ROWS) void k_tvec(const double *__restrict__ T, int n, const double *__restrict__ y,
                                                           double *__restrict__ xout, const double *__restrict__ x0,
                                                           double *__restrict__ xn)
{
    __shared__ double S[TV_ROWS][TV_COLS + 1];
    const int t = threadIdx.x, g = t / TV_COLS, c = t % TV_COLS;
    const int col0 = blockIdx.x * TV_COLS, col = col0 + c;
    const int i0 = (col0 / DNB) * DNB;            // TV_COLS divides DNB: the columns of a workgroup share their diagonal block
    double acc = 0.0;
    if (col < n) {
#pragma unroll 8
        for (int i = i0 + g; i < n; i += TV_ROWS) acc = acc + T[(size_t)i * n + col] * y[i];
    }
    S[g][c] = acc;
    __syncthreads();
    if (t < TV_COLS && col0 + t < n) {
        double v = 0.0;
#pragma unroll 8
        for (int q = 0; q < TV_ROWS; q++) v = v + S[q][t];
        xout[col0 + t] = v;
        if (xn) xn[col0 + t] = x0[col0 + t] + 1.0 * v;
    }
}

// ---- W = T^T T for lower-triangular T (= L^-1): the SPD inverse from the triangular inverse --------
// One workgroup per 32x32 tile (I >= J) of W: W_IJ = sum over block rows k >= I of T_kI^T T_kJ, each
// term one 32x32x32 product on the f64 matrix cores.  The blocks come from memory written by other
// XCDs (~1.7 us away), so three steps' worth of them are kept in flight in registers; LDS is double
// buffered, one barrier per step.  Both mirror images are written (the host wants the full matrix).
__device__ __forceinline__ d4_t d_mfma_tn(double (*X)[DNB + 1], double (*Y)[DNB + 1], int wv, int lane, d4_t c)
{
    const int i = 16 * (wv >> 1) + (lane & 15), j = 16 * (wv & 1) + (lane & 15), kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < DNB / 4; kk++)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(X[4 * kk + kq][i], Y[4 * kk + kq][j], c, 0, 0, 0);
    return c;
}

__global__ __launch_bounds__(256) void k_ttt(const double *__restrict__ Tm, int n, double *__restrict__ W)
{
    const int I = blockIdx.y, J = blockIdx.x;
    if (J > I) return;
    __shared__ double TI[2][DNB][DNB + 1];
    __shared__ double TJ[2][DNB][DNB + 1];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int i0 = I * DNB, j0 = J * DNB;
    const int ni = min(DNB, n - i0), nj = min(DNB, n - j0);
    const int nsteps = (n - i0 + DNB - 1) / DNB;
    double pi[TTT_PF][4], pj[TTT_PF][4];
    // this thread's four entries of the two blocks of step s (zeros past the end; T is lower
    // triangular: entries right of the diagonal are not stored)
    auto fetch = [&](int s, double (&a)[4], double (&b)[4]) {
        const int k0 = i0 + s * DNB;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = t + 256 * q, kk = e / DNB, c = e % DNB;
            const bool row = s < nsteps && k0 + kk < n;
            a[q] = (row && c < ni && i0 + c <= k0 + kk) ? Tm[(size_t)(k0 + kk) * n + i0 + c] : 0.0;
            b[q] = (row && c < nj && j0 + c <= k0 + kk) ? Tm[(size_t)(k0 + kk) * n + j0 + c] : 0.0;
        }
    };
#pragma unroll
    for (int p = 0; p < TTT_PF; p++) fetch(p, pi[p], pj[p]);
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = 0; s0 < nsteps; s0 += TTT_PF) {
#pragma unroll
        for (int p = 0; p < TTT_PF; p++) {
            const int s = s0 + p;
            if (s < nsteps) {
                const int buf = s & 1;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int e = t + 256 * q;
                    TI[buf][e / DNB][e % DNB] = pi[p][q];
                    TJ[buf][e / DNB][e % DNB] = pj[p][q];
                }
                fetch(s + TTT_PF, pi[p], pj[p]);
                __syncthreads();
                acc = d_mfma_tn(TI[buf], TJ[buf], wv, lane, acc);
            }
        }
    }
    const int jj = 16 * (wv & 1) + (lane & 15);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int ii = 16 * (wv >> 1) + (lane >> 4) + 4 * e;
        if (ii < ni && jj < nj) {
            W[(size_t)(i0 + ii) * n + j0 + jj] = acc[e];
            W[(size_t)(j0 + jj) * n + i0 + ii] = acc[e];
        }
    }
}

// ---- assembly of the update system -----------------------------------------------------------------
// The update system of one iteration in one pass over H (one workgroup per row of the augmented
// array): A = invW0 + H, the right-hand side Hz - H (X0 - X) as row `rhs_row` of the block below the
// matrix, zeros in the padding rows.  One more workgroup (the last) does what k_chol_first would do in a
// launch of its own: it forms the first diagonal block itself and leaves the inverse of its factor in Lt[0].
__global__ __launch_bounds__(256) void k_assemble(const double *__restrict__ invW0, const double *__restrict__ H,
                                                  const double *__restrict__ X0, const double *__restrict__ X,
                                                  const double *__restrict__ Hz, double *__restrict__ A, int n, int rhs_row,
                                                  double *__restrict__ Lt)
{
    __shared__ double s[4];
    const int row = blockIdx.x;
    if (row == (int)gridDim.x - 1) {
        __shared__ double W[DNB][DNB + 1];
        const int nd = min(DNB, n);
        for (int e = threadIdx.x; e < DNB * DNB; e += 256) {
            const int i = e / DNB, j = e % DNB;
            W[i][j] = (i < nd && j < nd) ? invW0[(size_t)i * n + j] + H[(size_t)i * n + j] : (i == j ? 1.0 : 0.0);
        }
        __syncthreads();
        if (threadIdx.x < 64) chol32_tinv_wave(W, Lt, threadIdx.x);
        return;
    }
    if (row >= n) {
        if (row != rhs_row)
            for (int j = threadIdx.x; j < n; j += 256) A[(size_t)row * n + j] = 0.0;
        return;
    }
    double acc = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const double h = H[(size_t)row * n + j];
        A[(size_t)row * n + j] = invW0[(size_t)row * n + j] + h;
        acc += h * (X0[j] + -1.0 * X[j]);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) A[(size_t)rhs_row * n + row] = Hz[row] - (((s[0] + s[1]) + s[2]) + s[3]);
}

// ---- end of one iteration of hm_update_run ---------------------------------------------------------------
// res (page-locked host memory, a block of host_block.h: every value a stamped pair of words) = [step (n) | the four
// error sums of the render's per-strip partials (fixed order: d_tile_partial_sums) | overflow flag | the ticket |
// (spare) | factorisation time-out | flow x, flow y error sums against the raw flow].  One workgroup.  The host takes
// the block when every pair carries this launch's stamp; nothing depends on the order the words arrive in (delay_us,
// the test knob "result_delay", has the LAST value published first and everything else that much later).
#define RES_HEAD 10           // doubles behind the step in a result block

__global__ __launch_bounds__(256) void k_iter_result(const double *__restrict__ step, int n, const double *__restrict__ partial,
                                                     int ntiles, const int *__restrict__ overflow,
                                                     const unsigned *__restrict__ flow_ctl, double *__restrict__ res, double ticket,
                                                     int delay_us)
{
    __shared__ double sp[RI_GROUPS * RI_NV];
    __shared__ double sums[RI_NV];
    const int t = threadIdx.x;
    const unsigned long long stamp = hb_stamp((long long)ticket);
    d_tile_partial_sums(partial, ntiles, sp, sums);
    __syncthreads();
    double v = 0.0;
    int slot = -1;
    if (t < 4) { slot = t; v = sums[t]; }
    else if (t == 4) { slot = 4; v = (double)*overflow; }
    else if (t == 5) { slot = 5; v = ticket; }
    else if (t == 6) { slot = 6; v = 0.0; }
    else if (t == 7) { slot = 7; v = (double)flow_ctl[1]; }  // a wait of the persistent factorisation launch timed out
    else if (t == 8) { slot = 8; v = sums[4]; }
    else if (t == 9) { slot = 9; v = sums[5]; }
    if (delay_us > 0) {
        if (t == 9) { hb_put(res, n + 9, v, stamp); hb_flush(); }
        hb_delay(delay_us);
    }
    for (int i = t; i < n; i += 256) hb_put(res, i, step[i], stamp);
    if (slot >= 0) hb_put(res, n + slot, v, stamp);
    hb_flush();
}

// gains of the three measurement channels (kalman.py:828-830): out[0] = W c0, out[1] = W (c1 + c2),
// out[2] = W c3 for the columns c* of Hzc (n x 4); one workgroup per row, fixed-order reduction
__global__ __launch_bounds__(256) void k_gains(const double *__restrict__ W, const double *__restrict__ Hzc, int n,
                                               double *__restrict__ out)
{
    __shared__ double sm[3][4];
    const int row = blockIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const double w = W[(size_t)row * n + j];
        const double *c = Hzc + (size_t)4 * j;
        a0 += w * c[0];
        a1 += w * (c[1] + c[2]);
        a2 += w * c[3];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 64);
        a1 += __shfl_down(a1, o, 64);
        a2 += __shfl_down(a2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = a0;
        sm[1][threadIdx.x >> 6] = a1;
        sm[2][threadIdx.x >> 6] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double *q = sm[threadIdx.x];
        out[(size_t)threadIdx.x * n + row] = ((q[0] + q[1]) + q[2]) + q[3];
    }
}

// what the host wants of the end of an update -- Hz components (n x 4), gains (3 x n) -- as one block of host_block.h
// (7n values) in page-locked host memory
__global__ __launch_bounds__(1024) void k_tail_result(const double *__restrict__ Hzc, const double *__restrict__ gain, int n,
                                                      double *__restrict__ blk, double ticket, int delay_us)
{
    const unsigned long long stamp = hb_stamp((long long)ticket);
    if (delay_us > 0) {
        if (threadIdx.x == 0) { hb_put(blk, 7 * n - 1, gain[3 * n - 1], stamp); hb_flush(); }
        hb_delay(delay_us);
    }
    for (int i = threadIdx.x; i < 4 * n; i += 1024) hb_put(blk, i, Hzc[i], stamp);
    for (int i = threadIdx.x; i < 3 * n; i += 1024) hb_put(blk, 4 * n + i, gain[i], stamp);
    hb_flush();
}

// ---- covariance prediction W' = F W F^T + Weps on the device -------------------------------------------
// F = [[I, a I], [A, I]] with A = s * dfdy, dfdy assembled from one symmetric 2x2 block per spring
// (see predict.cpp): (dfdy M)[rows of vertex v] = - sum over springs (v,u) of B (M[rows v] - M[rows u]).
// Weps = eps * [[I/4, I/2], [I/2, I]] (reference kalman.py:182).  n2 = 2N, n4 = 4N.
struct SpringTopo {
    const int *off;           // N+1: springs of each vertex
    const int *bar;           // spring id
    const int *other;         // the vertex at the other end
    const double *blk;        // I x 3: Bxx, Bxy, Byy
};

// P = F W, one thread per (vertex, column)
__global__ void k_fw_rows(const double *__restrict__ W, double *__restrict__ P, int N, SpringTopo tp, double a, double s)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    const int n2 = 2 * N, n4 = 4 * N;
    if (c >= n4) return;
    const double wx = W[(size_t)(2 * v) * n4 + c], wy = W[(size_t)(2 * v + 1) * n4 + c];
    const double bx = W[(size_t)(n2 + 2 * v) * n4 + c], by = W[(size_t)(n2 + 2 * v + 1) * n4 + c];
    double ax = 0.0, ay = 0.0;
    for (int q = tp.off[v]; q < tp.off[v + 1]; q++) {
        const int u = tp.other[q];
        const double *B = tp.blk + 3 * tp.bar[q];
        const double dx = wx - W[(size_t)(2 * u) * n4 + c], dy = wy - W[(size_t)(2 * u + 1) * n4 + c];
        ax -= B[0] * dx + B[1] * dy;
        ay -= B[1] * dx + B[2] * dy;
    }
    P[(size_t)(2 * v) * n4 + c] = wx + a * bx;
    P[(size_t)(2 * v + 1) * n4 + c] = wy + a * by;
    P[(size_t)(n2 + 2 * v) * n4 + c] = s * ax + bx;
    P[(size_t)(n2 + 2 * v + 1) * n4 + c] = s * ay + by;
}

// W' = P F^T + Weps, one thread per (row, vertex)
__global__ void k_pft_cols(const double *__restrict__ P, double *__restrict__ Wn, int N, SpringTopo tp, double a, double s,
                           double eps)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    const int n2 = 2 * N, n4 = 4 * N;
    if (r >= n4) return;
    const double *Pr = P + (size_t)r * n4;
    const double px = Pr[2 * v], py = Pr[2 * v + 1], qx = Pr[n2 + 2 * v], qy = Pr[n2 + 2 * v + 1];
    double ax = 0.0, ay = 0.0;
    for (int q = tp.off[v]; q < tp.off[v + 1]; q++) {
        const int u = tp.other[q];
        const double *B = tp.blk + 3 * tp.bar[q];
        const double dx = px - Pr[2 * u], dy = py - Pr[2 * u + 1];
        ax -= dx * B[0] + dy * B[1];
        ay -= dx * B[1] + dy * B[2];
    }
    double o0 = px + a * qx, o1 = py + a * qy, o2 = s * ax + qx, o3 = s * ay + qy;
    if (r == 2 * v) { o0 += eps / 4; o2 += eps / 2; }
    if (r == 2 * v + 1) { o1 += eps / 4; o3 += eps / 2; }
    if (r == n2 + 2 * v) { o0 += eps / 2; o2 += eps; }
    if (r == n2 + 2 * v + 1) { o1 += eps / 2; o3 += eps; }
    double *Wr = Wn + (size_t)r * n4;
    Wr[2 * v] = o0; Wr[2 * v + 1] = o1; Wr[n2 + 2 * v] = o2; Wr[n2 + 2 * v + 1] = o3;
}
