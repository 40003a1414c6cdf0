// The blocked Cholesky factorisation + inverse of the factor as ONE persistent launch (gfx950).
//
// k_chol_step (dense_kernels.h) takes one launch per 32-column block step: 26 dependent launches for n = 804,
// each ~8.8 us of kernel (2.0 waiting for operands written by the previous launch, 1.1 products, 3.5 factoring
// the next diagonal block, the rest ramp) plus ~2 us between launches -- the chain of the diagonal blocks is
// what a factorisation costs, and 4-5 us per step of it are launch boundary.  Here the same block operations
// run as tasks of one launch and hand their results to each other through memory:
//
//   S(c)    diagonal block c, all terms but the last:  A_cc - sum_{j<c-1} L_cj L_cj^T  ->  Q_c
//   D(c)    the last term and the factor:  Q_c - L_c,c-1 L_c,c-1^T  ->  T_c = chol(.)^-1       (Lt[c], Tinv_cc)
//   F(r,c)  block below it:    (A_rc - sum_{j<c} L_rj L_cj^T) T_c^T  ->  L_rc                 (r = nb: right-hand sides)
//   I(k,j)  inverse, k > j:    T_k (0 - sum_{i=j..k-1} L_ki Tinv_ij)  ->  Tinv_kj
//
// -- left-looking forms of what the step kernels do right-looking: every sum runs over j (or i) in ascending order
// with the same 32x32x32 MFMA products subtracted one by one, so every block has the bits the step kernels give
// (tests compare the two exactly).
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16, form R2 "the
// data is the flag"): the output arrays are pre-filled with a NaN bit pattern no computation produces; a producer
// stores every element of a block with ONE 8-byte write-through store (agent scope, sc1); a consumer loads the block
// with sc1 loads (they bypass its CU's L1) and takes it when no element is the pattern any more.  No flags, no
// fences, no dependence on which XCD a workgroup runs on.  While a block is not there one lane polls one of its
// words (s_sleep between polls); every wait is bounded and a time-out ends the launch with an error word set.
//
// Scheduling: workgroups draw numbers from one counter in the order they start running.  Number 0 makes its
// workgroup the CHAIN: it runs D(0), D(1), ... one after the other, T_c staying in its LDS for D(c+1) -- the
// diagonal blocks are what a factorisation waits for (4.2 of the ~6 us a column takes are the 32 dependent
// columns of one block, profiles/README.md), and a hand-off between two of them through memory costs 1.3 us.  The
// other numbers are tasks, step by step S(c), F(., c), I(c, .), every one depending only on lower numbers and on
// the chain.  A workgroup therefore only ever waits for what other RUNNING workgroups hold -- no deadlock whatever
// number of workgroups is resident -- and the ~250 workgroups in flight work ~9 steps ahead of the chain: a block's
// sum is complete but for its last term by the time that term's operand appears.  The chain per column: wait for
// Q_c and the block (c, c-1) -- published before its triangular solve by F(c, c-1) -- multiply the latter by
// T_{c-1}^T, subtract its square from Q_c, factor.
#pragma once
#include <hip/hip_runtime.h>
#include "dense_kernels.h"

#define FLOW_NT 256
#ifdef HM_STAMP
// development builds only (tools/stamp_chol.py): clock stamps of the chain's columns, 8 words per column behind the
// 4096 x 8 words k_render_iter uses
#define HM_CHAIN_STAMP(c, k) do { if (g_stamp && (threadIdx.x & 63) == 0) g_stamp[(size_t)(4096 + (c)) * 8 + (k)] = clock64(); } while (0)
#else
#define HM_CHAIN_STAMP(c, k) do { } while (0)
#endif
#define FLOW_SENTINEL 0x7FF8DEADBEEF0001ull       // a quiet NaN with a payload the hardware never generates
#define FLOW_POLL_LIMIT 4000000                   // polls of one wait before it gives up (~seconds)

struct FlowArgs {
    const double *A;          // n x n matrix (+ right-hand-side rows up to nrows), read only
    double *L;                // nrows x n: blocks below the block diagonal
    double *Lt;               // nb x 32 x 32: T_c
    double *Tinv;             // n x n: L^-1
    double *P;                // 3 x nb x 32 x 32: block (c, c-1) before its triangular solve; Q_c; block (c, c-2) likewise (Y_c)
    int n, nrows, nb, nbr;    // nbr: block rows including the right-hand-side rows
    unsigned *ctl;            // [0] next task, [1] set when a wait timed out
    int stall;                // test knob (hm_ctx_tune "chol_flow_stall"): the chain sleeps ~4 us x stall before every diagonal block
};

__device__ __forceinline__ unsigned long long flow_ld_bits(const double *p)
{
    return __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flow_st(double *p, double v)
{
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Pre-fill of everything the launch produces -- the blocks below the block diagonal of L (and its
// right-hand-side rows), the blocks on and below it of Tinv, Lt, P -- and the task counter (the error word is
// cleared by the caller of a sequence of factorisations, so that a time-out is still seen at the end of it).
// One row of the arrays per call (k_assemble does it for the row it assembles anyway) ...
__device__ __forceinline__ void flow_fill_row(const FlowArgs &a, int row, int t, int nt)
{
    const double s = __longlong_as_double((long long)FLOW_SENTINEL);
    if (row < a.nrows) {
        const int upto = row < a.nb * DNB ? (row / DNB) * DNB : a.n;       // columns left of the diagonal block; all of a rhs row
        for (int j = t; j < upto; j += nt) a.L[(size_t)row * a.n + j] = s;
    }
    if (row < a.n) {
        const int upto = min(a.n, (row / DNB + 1) * DNB);
        for (int j = t; j < upto; j += nt) a.Tinv[(size_t)row * a.n + j] = s;
    }
}
// (thread t of nt, over all the workgroups that share the fill: FLOW_FILL_WGS of them -- one workgroup alone took ~10 us
// for the 850 KB of Lt and P at n = 804 and was what k_solve_prep's 17 us hung on)
#define FLOW_FILL_WGS 32
__device__ __forceinline__ void flow_fill_blocks(const FlowArgs &a, int t, int nt)
{
    const double s = __longlong_as_double((long long)FLOW_SENTINEL);
    for (int i = t; i < a.nb * DNB * DNB; i += nt) a.Lt[i] = s;
    for (int i = t; i < 3 * a.nb * DNB * DNB; i += nt) a.P[i] = s;
    if (t == 0) a.ctl[0] = 0;
}
// ... or a launch of its own (the inverse of the prior has no assembly pass): one workgroup per row, one more for Lt / P
__global__ __launch_bounds__(256) void k_flow_fill(FlowArgs a)
{
    if ((int)blockIdx.x >= a.nrows) flow_fill_blocks(a, ((int)blockIdx.x - a.nrows) * 256 + threadIdx.x, 256 * FLOW_FILL_WGS);
    else flow_fill_row(a, blockIdx.x, threadIdx.x, 256);
}

// Wait for one or two blocks other tasks produce (rows < nr and columns < nc of a block are produced; the rest
// reads as 0) and put them into LDS.  First every thread simply loads its elements of both blocks -- one round trip
// when they are there already, the usual case for a task that runs ahead of the diagonal chain; while something is
// missing, one lane watches one word of the missing block (a short s_sleep between polls) before the block is
// loaded again.  false: gave up (error word set by this or another workgroup).
struct FlowBlock {
    const double *src;
    int ld, nr, nc;
    double (*dst)[DNB + 1];
};

// hot: the caller is on the chain of the diagonal blocks and what it waits for is about to appear -- every thread
// keeps re-loading its own elements (one workgroup's worth of traffic) instead of handing the watch to one lane,
// which saves the round trip of loading the block again once the watched word has changed.
__device__ __forceinline__ bool flow_fetch2(const FlowBlock &b0, const FlowBlock &b1, bool two, unsigned *ctl, bool hot = false)
{
    const int t = threadIdx.x;
    __shared__ int s_ok;
    double v0[4], v1[4];
    int patient = 0;                                  // rounds of the patient way (the hot re-loads are not counted)
    for (int tries = 0;; tries++) {
        int ok0 = 1, ok1 = 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = t + FLOW_NT * q, i = e / DNB, j = e % DNB;
            unsigned long long x0 = 0, x1 = 0;
            if (i < b0.nr && j < b0.nc) x0 = flow_ld_bits(b0.src + (size_t)i * b0.ld + j);
            if (two && i < b1.nr && j < b1.nc) x1 = flow_ld_bits(b1.src + (size_t)i * b1.ld + j);
            ok0 &= x0 != FLOW_SENTINEL;
            ok1 &= x1 != FLOW_SENTINEL;
            v0[q] = __longlong_as_double((long long)x0);
            v1[q] = __longlong_as_double((long long)x1);
        }
        const int got0 = __syncthreads_and(ok0), got1 = two ? __syncthreads_and(ok1) : 1;
        if (got0 && got1) break;                      // every element of what was asked for has arrived
        if (hot && tries < 4096) continue;            // ~ a few ms of re-loading at most, then the patient way
        if (t == 0) {                                 // one lane watches one word of the block that is missing
            int ok = 1;
            const FlowBlock &m = got0 ? b1 : b0;
            const double *canary = m.src + (size_t)(m.nr - 1) * m.ld + (m.nc - 1);
            for (int polls = 0; flow_ld_bits(canary) == FLOW_SENTINEL; polls++) {
                __builtin_amdgcn_s_sleep(2);
                if ((polls & 255) == 255 && __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (polls > FLOW_POLL_LIMIT) { __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
            if (++patient > 1000) { __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return false;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e = t + FLOW_NT * q;
        b0.dst[e / DNB][e % DNB] = v0[q];
        if (two) b1.dst[e / DNB][e % DNB] = v1[q];
    }
    return true;
}

__device__ __forceinline__ bool flow_fetch(const double *src, int ld, int nr, int nc, double (*dst)[DNB + 1], unsigned *ctl)
{
    const FlowBlock b = {src, ld, nr, nc, dst};
    return flow_fetch2(b, b, false, ctl);
}

// Two blocks other tasks publish (DNB columns; rows < nr of the first, all of the second; what is not published
// reads as 0; the second only if `two`) into LDS by the 128 threads of two waves (tt = 0 .. 127) that poll for
// them on their own -- no workgroup barrier inside: every wave re-loads its elements of both until none is the
// fill pattern.  false: gave up.
__device__ __forceinline__ bool flow_prefetch2(const double *src0, int nr, double (*dst0)[DNB + 1], const double *src1,
                                               double (*dst1)[DNB + 1], bool two, int tt, unsigned *ctl)
{
    double v0[8], v1[8];
    for (int polls = 0;; polls++) {
        int ok = 1;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int e = tt + 128 * q, i = e / DNB;
            unsigned long long x0 = 0, x1 = 0;
            if (i < nr) x0 = flow_ld_bits(src0 + e);
            if (two) x1 = flow_ld_bits(src1 + e);
            ok &= x0 != FLOW_SENTINEL && x1 != FLOW_SENTINEL;
            v0[q] = __longlong_as_double((long long)x0);
            v1[q] = __longlong_as_double((long long)x1);
        }
        if (__all(ok)) break;
        if ((polls & 255) == 255 && __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (polls > FLOW_POLL_LIMIT) { __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int e = tt + 128 * q;
        dst0[e / DNB][e % DNB] = v0[q];
        if (two) dst1[e / DNB][e % DNB] = v1[q];
    }
    return true;
}

// wave T of the diagonal block (chol32_wave_t): the rows of T are published strip by strip, as soon as they are
// final -- write-through stores to Lt[c] (the next tasks are waiting for it) and to the diagonal block of Tinv, a copy in
// LDS for the chain's next block.  (Stored all at the end they cost the chain ~1 us per column: the stamps of
// tools/stamp_chol.py show 2 600 clocks for 32 store instructions whose lines other workgroups are polling, 650 when
// nobody does; spread over the strips they are issued while this wave waits for wave B.)
__device__ __forceinline__ void flow_chol32_t(CholX &X, double *lt, double *tinv, int ld, int nc, int lane, double (*keep)[DNB + 1])
{
    d4_t t[2][2];
    const int lr = lane >> 4, lc = lane & 15;
    chol32_wave_t(t, X, lane, [&](int j, double u0, double u1) {
        const int i = j + lr;
        flow_st(lt + i * DNB + lc, u0);
        flow_st(lt + i * DNB + 16 + lc, u1);
        keep[i][lc] = u0;
        keep[i][16 + lc] = u1;
        if (i < nc && lc < nc) flow_st(tinv + (size_t)i * ld + lc, u0);
        if (i < nc && 16 + lc < nc) flow_st(tinv + (size_t)i * ld + 16 + lc, u1);
    });
}

__global__ __launch_bounds__(FLOW_NT) void k_chol_flow(FlowArgs a)
{
    __shared__ double Ts[DNB][DNB + 1];
    __shared__ double Br[DNB][DNB + 1];
    __shared__ double Bc[DNB][DNB + 1];
    __shared__ double Pn[DNB][DNB + 1];               // the chain's next operands, fetched while it factors
    __shared__ double Qn[DNB][DNB + 1];
    __shared__ CholX Xs;                              // the chain's hand-off between its B wave and its T wave
    __shared__ unsigned s_task;
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n = a.n, nb = a.nb;
    const unsigned ntasks = (unsigned)nb * (unsigned)(nb + 1);
    // coordinates of this thread's four elements of a 32x32 product tile (accumulator layout of d_mfma_*)
    const int mj = 16 * (wv & 1) + (lane & 15);
    int mi[4];
#pragma unroll
    for (int e = 0; e < 4; e++) mi[e] = 16 * (wv >> 1) + (lane >> 4) + 4 * e;
    const d4_t z = {0.0, 0.0, 0.0, 0.0};
    bool first = true;
    for (;;) {
        __syncthreads();                              // everyone is done with the previous task's LDS and s_task
        if (t == 0) s_task = atomicAdd(a.ctl, 1u);
        __syncthreads();
        if (first && s_task == 0) {
            // ---- the chain: D(0), D(1), ... ------------------------------------------------------------------
            // While waves 0 and 1 factor block c (dense_kernels.h: the B tiles in one, the T tiles in the other), the
            // other two waves fetch what block c + 1 starts from -- the block (c+1, c) before its triangular solve and
            // Q_{c+1}; both appear about now -- so that the chain never waits for a load it could have issued earlier
            // (a load of a block that is already there still takes 1.7 us).
            __builtin_amdgcn_s_setprio(3);            // ahead of whatever shares its SIMDs
            if (t == 0) s_fail = 0;
            chol32_x_clear(Xs, t, FLOW_NT);
            for (int c = 0; c < nb; c++) {
                const int c0 = c * DNB, nc = min(DNB, n - c0);
                for (int q = 0; q < a.stall; q++) __builtin_amdgcn_s_sleep(127);      // tests: everyone who waits for the chain waits long
                if (wv == 0) HM_CHAIN_STAMP(c, 0);
                double acc[4];
                if (c < 2) {                          // Q_c is the block of A itself
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = (mi[e] < nc && mj < nc) ? a.A[(size_t)(c0 + mi[e]) * n + c0 + mj] : 0.0;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = Qn[mi[e]][mj];
                }
                if (c > 0) {
                    const d4_t xr = d_mfma_nt(Pn, Ts, wv, lane);        // block (c, c-1) as F(c, c-1) left it, times T_{c-1}^T
#pragma unroll
                    for (int e = 0; e < 4; e++) Br[mi[e]][mj] = xr[e];
                    __syncthreads();
                    const d4_t s = d_mfma_nt(Br, Br, wv, lane);
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = acc[e] - s[e];
                    __syncthreads();
                }
#pragma unroll
                for (int e = 0; e < 4; e++) Br[mi[e]][mj] = (mi[e] < nc && mj < nc) ? acc[e] : (mi[e] == mj ? 1.0 : 0.0);
                if (wv == 0) HM_CHAIN_STAMP(c, 1);
                __syncthreads();
                if (wv == 0) {
                    HM_CHAIN_STAMP(c, 2);
                    chol32_wave_b(Br, Xs, lane);
                    HM_CHAIN_STAMP(c, 3);
                } else if (wv == 1) {
                    flow_chol32_t(Xs, a.Lt + (size_t)c * DNB * DNB, a.Tinv + (size_t)c0 * n + c0, n, nc, lane, Ts);
                    HM_CHAIN_STAMP(c, 5);
                } else if (c + 1 < nb) {
                    const int nr = min(DNB, n - (c0 + DNB));
                    const double *Pb = a.P + (size_t)(c + 1) * DNB * DNB, *Qb = a.P + (size_t)(nb + c + 1) * DNB * DNB;
                    if (!flow_prefetch2(Pb, nr, Pn, Qb, Qn, c + 1 >= 2, t - 128, a.ctl)) s_fail = 1;
                    if (wv == 2) HM_CHAIN_STAMP(c, 7);
                }
                __syncthreads();
                if (wv == 0) HM_CHAIN_STAMP(c, 6);
                if (s_fail) return;
            }
            __builtin_amdgcn_s_setprio(0);
            first = false;
            continue;
        }
        first = false;
        if (s_task > ntasks) return;
        const unsigned task = s_task - 1;
        const int c = (int)(task / (unsigned)(nb + 1)), idx = (int)(task % (unsigned)(nb + 1));
        const int c0 = c * DNB, nc = min(DNB, n - c0);
        if (idx == 0) {
            // ---- S(c) --------------------------------------------------------------------------------------
            if (c < 2) continue;
            double acc[4];
#pragma unroll
            for (int e = 0; e < 4; e++) acc[e] = (mi[e] < nc && mj < nc) ? a.A[(size_t)(c0 + mi[e]) * n + c0 + mj] : 0.0;
            for (int j = 0; j + 2 < c; j++) {
                if (!flow_fetch(a.L + (size_t)c0 * n + j * DNB, n, nc, DNB, Br, a.ctl)) return;
                __syncthreads();
                const d4_t s = d_mfma_nt(Br, Br, wv, lane);
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = acc[e] - s[e];
                __syncthreads();
            }
            {   // the last term, j = c-2: L_{c,c-2} = Y_c T_{c-2}^T is formed here -- the task that stores it is one
                // hop through memory further from T_{c-2} than this one, and the chain waits for Q_c two columns on
                const FlowBlock by = {a.P + (size_t)(2 * nb + c) * DNB * DNB, DNB, nc, DNB, Br};
                const FlowBlock bt = {a.Lt + (size_t)(c - 2) * DNB * DNB, DNB, DNB, DNB, Ts};
                if (!flow_fetch2(bt, by, true, a.ctl, true)) return;
                __syncthreads();
                const d4_t xr = d_mfma_nt(Br, Ts, wv, lane);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; e++) Br[mi[e]][mj] = xr[e];
                __syncthreads();
                const d4_t s = d_mfma_nt(Br, Br, wv, lane);
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = acc[e] - s[e];
            }
            double *Qb = a.P + (size_t)(nb + c) * DNB * DNB;
#pragma unroll
            for (int e = 0; e < 4; e++) flow_st(Qb + mi[e] * DNB + mj, acc[e]);
        } else if (idx <= nb - c) {
            // ---- F(r, c) -----------------------------------------------------------------------------------
            const int r = c + idx;
            if (r >= a.nbr) continue;                 // no right-hand-side rows in this factorisation
            const int r0 = r * DNB, nr = min(DNB, a.nrows - r0);
            // F(c+1, c) feeds the chain (block (c+1, c) before its triangular solve): its last term, j = c-1, is formed
            // from the two blocks of column c-1 BEFORE their triangular solves and T_{c-1} -- one hop through
            // memory after T_{c-1} instead of two
            const bool feeds = r == c + 1 && r < nb && c >= 1;
            double acc[4];
#pragma unroll
            for (int e = 0; e < 4; e++) acc[e] = (mi[e] < nr && mj < nc) ? a.A[(size_t)(r0 + mi[e]) * n + c0 + mj] : 0.0;
            for (int j = 0; j < (feeds ? c - 1 : c); j++) {
                const FlowBlock br = {a.L + (size_t)r0 * n + j * DNB, n, nr, DNB, Br};
                const FlowBlock bc = {a.L + (size_t)c0 * n + j * DNB, n, nc, DNB, Bc};
                if (!flow_fetch2(br, bc, true, a.ctl)) return;
                __syncthreads();
                const d4_t s = d_mfma_nt(Br, Bc, wv, lane);
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = acc[e] - s[e];
                __syncthreads();
            }
            if (feeds) {
                const FlowBlock by = {a.P + (size_t)(2 * nb + r) * DNB * DNB, DNB, nr, DNB, Br};       // block (r, c-1)
                const FlowBlock bp = {a.P + (size_t)c * DNB * DNB, DNB, nc, DNB, Bc};                  // block (c, c-1)
                const FlowBlock bt = {a.Lt + (size_t)(c - 1) * DNB * DNB, DNB, DNB, DNB, Ts};
                if (!flow_fetch2(by, bp, true, a.ctl)) return;
                if (!flow_fetch2(bt, bt, false, a.ctl, true)) return;
                __syncthreads();
                const d4_t x1 = d_mfma_nt(Br, Ts, wv, lane), x2 = d_mfma_nt(Bc, Ts, wv, lane);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; e++) { Br[mi[e]][mj] = x1[e]; Bc[mi[e]][mj] = x2[e]; }
                __syncthreads();
                const d4_t s = d_mfma_nt(Br, Bc, wv, lane);
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = acc[e] - s[e];
                __syncthreads();
            }
            if (r == c + 2 && r < nb) {               // F(r, r-1) and S(r) form L_{r,r-2} from this themselves
                double *Yb = a.P + (size_t)(2 * nb + r) * DNB * DNB;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (mi[e] < nr) flow_st(Yb + mi[e] * DNB + mj, mj < nc ? acc[e] : 0.0);
            }
            if (r == c + 1 && r < nb) {               // the diagonal task of column r takes it from here
                double *Pb = a.P + (size_t)r * DNB * DNB;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (mi[e] < nr) flow_st(Pb + mi[e] * DNB + mj, mj < nc ? acc[e] : 0.0);
            }
            {
                const FlowBlock bt = {a.Lt + (size_t)c * DNB * DNB, DNB, DNB, DNB, Ts};
                if (!flow_fetch2(bt, bt, false, a.ctl, r <= c + 2)) return;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) Br[mi[e]][mj] = acc[e];
            __syncthreads();
            const d4_t xr = d_mfma_nt(Br, Ts, wv, lane);
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (mi[e] < nr && mj < nc) flow_st(a.L + (size_t)(r0 + mi[e]) * n + c0 + mj, xr[e]);
        } else {
            // ---- I(k, j): block (k, j) of the inverse, k = c ------------------------------------------------------
            const int k = c, j = idx - (nb - c) - 1, k0 = c0, nk = nc, j0 = j * DNB;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = j; i < k; i++) {
                const FlowBlock bl = {a.L + (size_t)k0 * n + i * DNB, n, nk, DNB, Br};
                const FlowBlock bi = {a.Tinv + (size_t)(i * DNB) * n + j0, n, DNB, DNB, Bc};
                if (!flow_fetch2(bl, bi, true, a.ctl)) return;
                __syncthreads();
                const d4_t s = d_mfma_nn(Br, Bc, wv, lane, z);
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = acc[e] - s[e];
                __syncthreads();
            }
            if (!flow_fetch(a.Lt + (size_t)k * DNB * DNB, DNB, DNB, DNB, Ts, a.ctl)) return;
#pragma unroll
            for (int e = 0; e < 4; e++) Bc[mi[e]][mj] = mi[e] < nk ? acc[e] : 0.0;
            __syncthreads();
            const d4_t u = d_mfma_nn(Ts, Bc, wv, lane, z);
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (mi[e] < nk) flow_st(a.Tinv + (size_t)(k0 + mi[e]) * n + j0 + mj, u[e]);
        }
    }
}

// k_assemble (dense_kernels.h) for a factorisation by k_chol_flow: the same pass over H -- A = invW0 + H, the
// right-hand side Hz - H (X0 - X) as row rhs_row below the matrix, zeros in the padding rows -- and, row by row, the
// pre-fill of what the factorisation launch produces; the last workgroup pre-fills Lt and P (the first diagonal
// block is task D(0) of that launch).  Same arithmetic as k_assemble: the systems are bit-identical.
__global__ __launch_bounds__(256) void k_assemble_flow(const double *__restrict__ invW0, const double *__restrict__ H,
                                                       const double *__restrict__ X0, const double *__restrict__ X,
                                                       const double *__restrict__ Hz, double *__restrict__ A, int n, int rhs_row,
                                                       FlowArgs f)
{
    __shared__ double s[4];
    const int row = blockIdx.x;
    if (row >= (int)gridDim.x - FLOW_FILL_WGS) {
        flow_fill_blocks(f, (row - ((int)gridDim.x - FLOW_FILL_WGS)) * 256 + threadIdx.x, 256 * FLOW_FILL_WGS);
        return;
    }
    flow_fill_row(f, row, threadIdx.x, 256);
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

// ---- the update system straight from the measurement's job sums -------------------------------------------------
// k_hth_scatter + k_assemble_flow in one pass (hm_update_step / hm_update_run): H = HTH is block sparse -- row (v, a)
// has entries in the columns of v and of its mesh neighbours only (kalman.py:202-205) -- so a row's workgroup copies
// its row of inv(W0) into A, forms the row's few entries of H from the sums of the vertex job v and of the edge jobs
// around v (the arithmetic of k_hth_scatter: parts added in order, / eps, / d / d), adds them in, and gets the row's
// right-hand side Hz - H (X0 - X) from the same entries; Hz and the Hz components of the row are stored for the
// gains.  The dense H is not written at all (hm_measure still does, with k_hth_scatter).  Row by row the pre-fill of
// what the factorisation launch produces, as k_assemble_flow does.
struct PrepArgs {
    const double *out;        // job sums (njobs x MEAS_VSPLIT_MAX x MEAS_OUT)
    int N, vsplit, esplit;
    double eZ, eJ, eM, d;
    const int *nb_off, *nb_u, *nb_e;          // per vertex: its neighbours (ascending) and the edge job of each
    const double *invW0, *X0, *X;
    double *A, *Hz, *Hzc;
    int n, rhs_row;
    FlowArgs f;
};

__device__ __forceinline__ double d_job_sum(const double *__restrict__ out, int job, int parts, int idx)
{
    const double *src = out + (size_t)job * MEAS_VSPLIT_MAX * MEAS_OUT;
    double v = src[idx];
    for (int q = 1; q < parts; q++) v += src[q * MEAS_OUT + idx];
    return v;
}

#define PREP_MAX_ENTRIES (4 * (EKF_MAX_STAR + 2))
__global__ __launch_bounds__(256) void k_solve_prep(PrepArgs p)
{
    __shared__ double s_term[PREP_MAX_ENTRIES];
    __shared__ double s_hz;
    const int row = blockIdx.x, t = threadIdx.x, n = p.n;
    if (row >= (int)gridDim.x - FLOW_FILL_WGS) {
        flow_fill_blocks(p.f, (row - ((int)gridDim.x - FLOW_FILL_WGS)) * 256 + t, 256 * FLOW_FILL_WGS);
        return;
    }
    flow_fill_row(p.f, row, t, 256);
    if (row >= n) {
        if (row != p.rhs_row)
            for (int j = t; j < n; j += 256) p.A[(size_t)row * n + j] = 0.0;
        return;
    }
    double *Arow = p.A + (size_t)row * n;
    const double *Wrow = p.invW0 + (size_t)row * n;
    {   // the copy of the row, its loads in flight together (a plain loop waits for every load before its store)
        double w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = (t + 256 * q < n) ? Wrow[t + 256 * q] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (t + 256 * q < n) Arow[t + 256 * q] = w[q];
        for (int j = t + 1024; j < n; j += 256) Arow[j] = Wrow[j];
    }
    const int N = p.N, n2 = 2 * N;
    const int v = (row % n2) >> 1, ca = (row & 1) + (row >= n2 ? 2 : 0);      // component of the row: x, y, vx, vy
    const int deg = p.nb_off[v + 1] - p.nb_off[v];
    const int entries = min(PREP_MAX_ENTRIES, 4 * (deg + 1));
    double val = 0.0;
    int col = 0;
    if (t < entries) {
        const int q = t >> 2, cb = t & 3;
        int idx = -1, job, parts, u;
        bool weighted;                              // geometry x geometry sums come weighted by 1 / eps already
        if (q == 0) {                               // the 4x4 block of the vertex itself (symmetric)
            u = v; job = v; parts = p.vsplit;
            const int lo = min(ca, cb), hi = max(ca, cb);
            const int tab[4][4] = {{A_XX, A_XY, A_XVX, A_XVY}, {-1, A_YY, A_YVX, A_YVY}, {-1, -1, A_VXVX, -1}, {-1, -1, -1, A_VYVY}};
            idx = tab[lo][hi];
            weighted = hi < 2;
        } else {
            u = p.nb_u[p.nb_off[v] + q - 1];
            job = N + p.nb_e[p.nb_off[v] + q - 1]; parts = p.esplit;
            const int c1 = v < u ? ca : cb, c2 = v < u ? cb : ca;           // component of the lower / the higher vertex
            const int tab[4][4] = {{B_XX, B_XY, B_XVX, B_XVY}, {B_YX, B_YY, B_YVX, B_YVY}, {B_VXX, B_VXY, B_VXVX, -1},
                                   {B_VYX, B_VYY, -1, B_VYVY}};
            idx = tab[c1][c2];
            weighted = c1 < 2 && c2 < 2;
        }
        col = 2 * u + (cb & 1) + (cb >= 2 ? n2 : 0);
        if (idx >= 0) {
            const double raw = d_job_sum(p.out, job, parts, idx);
            val = (weighted ? raw : raw / p.eJ) / p.d / p.d;               // d_put of k_hth_scatter
        }
        s_term[t] = val * (p.X0[col] + -1.0 * p.X[col]);
    } else if (t == 255) {
        // central differences of jz (kalman.py:499-515), as k_hth_scatter forms them
        double c[4] = {0.0, 0.0, 0.0, 0.0};
        if (ca < 2) {
            const int b0 = ca == 0 ? A_X : A_Y;
            c[0] = d_job_sum(p.out, v, p.vsplit, b0) / p.eZ; c[1] = d_job_sum(p.out, v, p.vsplit, b0 + 1) / p.eJ;
            c[2] = -d_job_sum(p.out, v, p.vsplit, b0 + 2) / p.eJ; c[3] = d_job_sum(p.out, v, p.vsplit, b0 + 3) / p.eM;
        } else if (ca == 2) {
            c[1] = d_job_sum(p.out, v, p.vsplit, A_VX) / p.eJ;
        } else {
            c[2] = -d_job_sum(p.out, v, p.vsplit, A_VY) / p.eJ;
        }
        const double hz = (((c[0] + c[1]) + c[2]) + c[3]) / p.d / 2;
        p.Hz[row] = hz;
        for (int ch = 0; ch < 4; ch++) p.Hzc[(size_t)row * 4 + ch] = c[ch] / p.d / 2;
        s_hz = hz;
    }
    __syncthreads();                                  // the copy of the row is complete, the terms are in LDS
    if (t < entries) Arow[col] = Wrow[col] + val;
    if (t == 0) {
        double acc = 0.0;
        for (int i = 0; i < entries; i++) acc += s_term[i];
        p.A[(size_t)p.rhs_row * n + row] = s_hz - acc;
    }
}
