// Mass-spring state prediction on the device (gfx950): the Newton / implicit-Euler loop of
// IteratedMSKalmanFilter._newton (reference kalman.py:923-960 with _jacobian :865-902, _dgdx :914-921) as ONE
// workgroup that keeps the whole problem in LDS -- the state is 4N doubles, the work a chain of ~50 Newton
// iterations of ~10 conjugate-gradient steps each, every one a few hundred flops per thread between two
// workgroup barriers: latency, not throughput.  Same algebra as the host version (csrc/predict.cpp): the dense
// 4N x 4N system [[I, -dt I], [-A, I]] is block-eliminated to (I - dt A) s1 = g1 + dt g2, the spring operator
// is applied bar by bar (gathered per vertex in ascending bar order: the order the host adds in), conjugate
// gradients to the rounding floor.  Sums over the vector (norms, dot products) are block reductions in a fixed
// order -- not the host's left-to-right order, so the two agree to rounding, not to the bit.
#pragma once
#include "host_block.h"
#include <hip/hip_runtime.h>

#define NEWTON_NT 512

struct NewtonArgs {
    int N, I;
    const int *bars;          // I x 2
    const double *l0;         // I
    const int *voff, *vbar;   // CSR: the bars of every vertex, ascending
    double kappa, M, dt, tol;
    int maxiter, steps;
    double *X;                // 4N, advanced in place
    int *info;                // [0] Newton iterations in total, [1] 0 ok / 1 the inner solve did not converge
};

// sum of v over the workgroup, the same value in every thread (fixed order: wave shuffles, then the eight
// wave sums left to right)
__device__ __forceinline__ double d_wg_sum(double v, double *s_part)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();                                // s_part free again
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NEWTON_NT / 64; w++) t += s_part[w];
    return t;
}

__global__ __launch_bounds__(NEWTON_NT) void k_ms_newton(NewtonArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int N = a.N, I = a.I, n2 = 2 * N, n4 = 4 * N, t = threadIdx.x;
    double *X = sm, *x = X + n4, *xp = x + n4, *xo = xp + n4, *g = xo + n4;                 // 5 x 4N
    double *f = g + n4, *rhs = f + n2, *s1 = rhs + n2, *r = s1 + n2, *p = r + n2, *Sp = p + n2, *tmp = Sp + n2;   // 7 x 2N
    double *Bxx = tmp + n2, *Bxy = Bxx + I, *Byy = Bxy + I, *kk = Byy + I, *ddx = kk + I, *ddy = ddx + I;        // 6 x I
    double *s_part = ddy + I, *l0 = s_part + 8;                                              // 8, I
    // the topology in LDS too: the gathers below are chains of dependent index loads
    int *voff = (int *)(l0 + I), *vbar = voff + (N + 1), *bars = vbar + 2 * I;               // N + 1, 2I, 2I ints
    for (int i = t; i < n4; i += NEWTON_NT) X[i] = a.X[i];
    for (int i = t; i < I; i += NEWTON_NT) l0[i] = a.l0[i];
    for (int i = t; i <= N; i += NEWTON_NT) voff[i] = a.voff[i];
    for (int i = t; i < 2 * I; i += NEWTON_NT) { vbar[i] = a.vbar[i]; bars[i] = a.bars[i]; }
    __syncthreads();
    // out = dfdy * s, per vertex component: - B d on the bar's first vertex, + B d on its second (predict.cpp)
    auto apply = [&](const double *s, double *out) {
        for (int v = t; v < N; v += NEWTON_NT) {              // one thread per vertex: both components share the differences
            double ax = 0.0, ay = 0.0;
            for (int q = voff[v]; q < voff[v + 1]; q++) {
                const int b = vbar[q], va = bars[2 * b], vb = bars[2 * b + 1];
                const double sx = s[2 * va] - s[2 * vb], sy = s[2 * va + 1] - s[2 * vb + 1];
                const double tx = Bxx[b] * sx + Bxy[b] * sy, ty = Bxy[b] * sx + Byy[b] * sy;
                if (va == v) { ax -= tx; ay -= ty; } else { ax += tx; ay += ty; }
            }
            out[2 * v] = ax; out[2 * v + 1] = ay;
        }
        __syncthreads();                                       // out is read by other threads than the ones that wrote it
    };
    const double dt = a.dt, M = a.M, a2 = dt * dt / M;
    int total = 0, bad = 0;
    for (int st = 0; st < a.steps; st++) {
        for (int i = t; i < n4; i += NEWTON_NT) { x[i] = X[i]; xp[i] = X[i]; xo[i] = 0.0; }
        __syncthreads();
        for (int n = 0;; n++) {
            // while n < maxiter and |xo - xp| > tol |xp|  (kalman.py:939)
            double dn = 0.0, pn = 0.0;
            for (int i = t; i < n4; i += NEWTON_NT) { const double d = xo[i] - xp[i]; dn += d * d; pn += xp[i] * xp[i]; }
            dn = d_wg_sum(dn, s_part);
            pn = d_wg_sum(pn, s_part);
            if (!(n < a.maxiter && sqrt(dn) > a.tol * sqrt(pn))) break;
            for (int i = t; i < n4; i += NEWTON_NT) xo[i] = xp[i];
            // forces and Jacobian blocks at the current state, bar by bar
            for (int b = t; b < I; b += NEWTON_NT) {
                const int va = bars[2 * b], vb = bars[2 * b + 1];
                const double dx = X[2 * va] - X[2 * vb], dy = X[2 * va + 1] - X[2 * vb + 1];
                const double l = sqrt(dx * dx + dy * dy);
                const double k = a.kappa * (1.0 - l0[b] / l), c = a.kappa * l0[b] / (l * l * l);
                kk[b] = k; ddx[b] = dx; ddy[b] = dy;
                Bxx[b] = k + c * dx * dx; Bxy[b] = c * dx * dy; Byy[b] = k + c * dy * dy;
            }
            __syncthreads();
            for (int i = t; i < n2; i += NEWTON_NT) {
                const int v = i >> 1, c = i & 1;
                double acc = 0.0;
                for (int q = voff[v]; q < voff[v + 1]; q++) {
                    const int b = vbar[q];
                    const double kd = kk[b] * (c == 0 ? ddx[b] : ddy[b]);
                    acc = bars[2 * b] == v ? acc + kd : acc - kd;
                }
                f[i] = acc;
            }
            __syncthreads();
            // g = xp - x - dt [v; f / M];  (I - dt A) s1 = g1 + dt g2
            double bn = 0.0;
            for (int i = t; i < n2; i += NEWTON_NT) {
                const double g1 = xp[i] - x[i] - dt * X[n2 + i];
                const double g2 = xp[n2 + i] - x[n2 + i] - dt * (f[i] / M);
                g[i] = g1; g[n2 + i] = g2;
                const double rr = g1 + dt * g2;
                rhs[i] = rr; s1[i] = rr;
                bn += rr * rr;
            }
            bn = sqrt(d_wg_sum(bn, s_part));        // (its barriers also publish g, rhs, s1)
            if (bn == 0.0) {
                for (int i = t; i < n2; i += NEWTON_NT) s1[i] = 0.0;
            } else {
                apply(s1, tmp);
                double rs = 0.0;
                for (int i = t; i < n2; i += NEWTON_NT) {
                    const double ri = rhs[i] - (s1[i] - a2 * tmp[i]);
                    r[i] = ri; p[i] = ri; rs += ri * ri;
                }
                rs = d_wg_sum(rs, s_part);
                bool ok = false;
                for (int it = 0; it < 200; it++) {
                    if (sqrt(rs) <= 1e-15 * bn) { ok = true; break; }
                    apply(p, tmp);
                    double pSp = 0.0;
                    for (int i = t; i < n2; i += NEWTON_NT) { const double q = p[i] - a2 * tmp[i]; Sp[i] = q; pSp += p[i] * q; }
                    pSp = d_wg_sum(pSp, s_part);
                    const double alpha = rs / pSp;
                    double rs_new = 0.0;
                    for (int i = t; i < n2; i += NEWTON_NT) {
                        s1[i] += alpha * p[i];
                        const double ri = r[i] - alpha * Sp[i];
                        r[i] = ri; rs_new += ri * ri;
                    }
                    rs_new = d_wg_sum(rs_new, s_part);
                    const double beta = rs_new / rs;
                    for (int i = t; i < n2; i += NEWTON_NT) p[i] = r[i] + beta * p[i];
                    rs = rs_new;
                    __syncthreads();
                }
                if (!ok && !(sqrt(rs) <= 1e-13 * bn)) bad = 1;
            }
            __syncthreads();
            // s2 = g2 + A s1;  xp -= [s1; s2];  X = xp
            apply(s1, tmp);
            for (int i = t; i < n2; i += NEWTON_NT) {
                const double s2 = g[n2 + i] + (dt / M) * tmp[i];
                const double a1 = xp[i] - s1[i], b1 = xp[n2 + i] - s2;
                xp[i] = a1; xp[n2 + i] = b1;
            }
            __syncthreads();
            for (int i = t; i < n4; i += NEWTON_NT) X[i] = xp[i];
            __syncthreads();
            total++;
        }
    }
    for (int i = t; i < n4; i += NEWTON_NT) a.X[i] = X[i];
    if (t == 0) { a.info[0] = total; a.info[1] = bad; }
}


// ---- the same loop, four waves, one vertex per lane ------------------------------------------------------------------
// k_ms_newton above spends its time in workgroup barriers (six per conjugate-gradient step, 512 threads with one vector
// component each: ~2 us per step, 1.2 ms per frame at 201 vertices against 0.3 ms on a host core).  Here a lane owns a
// VERTEX: both components of everything that belongs to it live in registers, its neighbours (padded to DEG slots, in
// ascending order of the bar that joins them: the order the host adds in) and the spring blocks towards them are
// fetched into registers once per Newton iteration, and a step of the inner solve is one gather of the neighbours'
// direction vector from LDS plus two sums over the workgroup (DPP inside a wave, four partial sums through LDS, one
// barrier each).  Per vertex the operations and their order are those of csrc/predict.cpp (a padded slot contributes
// B = 0 times d = 0); only the sums over the whole vector are added in another order, so host and device agree to
// rounding.  N <= NEWTON4_NT, every vertex degree <= DEG (else the caller takes the host version).
#define NEWTON4_NT 256

struct Newton4Args {
    int N, I, deg_stride;
    const int *bars;          // I x 2
    const double *l0;         // I
    const int *nbr, *nbb;     // N x deg_stride: neighbour vertex / bar of slot q (padding: the vertex itself / bar I)
    double kappa, M, dt, tol;
    int maxiter, steps;
    const double *Xin;        // 4N, page-locked host memory written before the launch was queued (system-scope loads)
    double *out;              // a result block of host_block.h in page-locked host memory, 4N + 2 values: the advanced state,
                              // the Newton iterations, 1 = inner solve failed
    double *dev_out;          // 4N + 2 in device memory: the same for kernels queued behind this one (hm_chain_project), or NULL
    double ticket;
    int delay_us;             // test knob "result_delay" (host_block.h)
    int force_bad;            // test knob "newton_fail": report a failed inner solve
};

// wave-wide sums of a and b in lane 63 (row shifts, then the row broadcasts of gfx9)
__device__ __forceinline__ double d_dpp_f64(double v, const int ctrl, const int row_mask)
{
    const long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    // masked-off rows and lanes without a source receive 0.0
    if (ctrl == 0x111) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); }
    else if (ctrl == 0x112) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true); }
    else if (ctrl == 0x114) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true); }
    else if (ctrl == 0x118) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true); }
    else if (ctrl == 0x142) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false); }
    else { lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false); }
    (void)row_mask;
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ void d_wave_sum2(double &a, double &b)
{
    a += d_dpp_f64(a, 0x111, 0xf); b += d_dpp_f64(b, 0x111, 0xf);      // row_shr:1
    a += d_dpp_f64(a, 0x112, 0xf); b += d_dpp_f64(b, 0x112, 0xf);      // row_shr:2
    a += d_dpp_f64(a, 0x114, 0xf); b += d_dpp_f64(b, 0x114, 0xf);      // row_shr:4
    a += d_dpp_f64(a, 0x118, 0xf); b += d_dpp_f64(b, 0x118, 0xf);      // row_shr:8: lane 15 of a row holds the row's sum
    a += d_dpp_f64(a, 0x142, 0xa); b += d_dpp_f64(b, 0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    a += d_dpp_f64(a, 0x143, 0xc); b += d_dpp_f64(b, 0x143, 0xc);      // row_bcast:31 into rows 2 and 3: lane 63 holds all
}

// sums of a and b over the workgroup, the same values in every thread; `part` alternates between two halves so that
// one barrier per call is enough
__device__ __forceinline__ void d_wg_sum2(double &a, double &b, double *part, int &flip)
{
    d_wave_sum2(a, b);
    double *p = part + flip * 2 * (NEWTON4_NT / 64);
    flip ^= 1;
    if ((threadIdx.x & 63) == 63) { p[2 * (threadIdx.x >> 6)] = a; p[2 * (threadIdx.x >> 6) + 1] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < NEWTON4_NT / 64; w++) { sa += p[2 * w]; sb += p[2 * w + 1]; }
    a = sa; b = sb;
}

template <int DEG>
__global__ __launch_bounds__(NEWTON4_NT) void k_ms_newton4(Newton4Args a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int N = a.N, I = a.I, t = threadIdx.x, v = t;
    const bool act = v < N;
    double *Y = sm;                                  // 2N: positions of the current iterate (the springs are evaluated at it)
    double *P = Y + 2 * NEWTON4_NT;                  // 2 x 2N: the vector the neighbours gather from, two buffers
    double *KK = P + 4 * NEWTON4_NT;                 // I + 1 each: spring terms per bar; entry I stays 0 (padding)
    double *BXX = KK + (I + 1), *BXY = BXX + (I + 1), *BYY = BXY + (I + 1);
    double *part = BYY + (I + 1);                    // 2 x 2 x waves
    int *bars = (int *)(part + 4 * (NEWTON4_NT / 64));
    for (int i = t; i < 2 * I; i += NEWTON4_NT) bars[i] = a.bars[i];
    if (t == 0) { KK[I] = 0.0; BXX[I] = 0.0; BXY[I] = 0.0; BYY[I] = 0.0; }
    int nb[DEG], bb[DEG];
#pragma unroll
    for (int q = 0; q < DEG; q++) {
        nb[q] = act ? a.nbr[v * a.deg_stride + q] : 0;
        bb[q] = act ? a.nbb[v * a.deg_stride + q] : I;
    }
    // the state: X = [y; w] (positions, velocities); x = the sub-step's start, xp = the iterate, xo = the one before
    double Xy[2] = {0, 0}, Xw[2] = {0, 0};
    if (act) {
        Xy[0] = hb_host_in(a.Xin + 2 * v); Xy[1] = hb_host_in(a.Xin + 2 * v + 1);
        Xw[0] = hb_host_in(a.Xin + 2 * N + 2 * v); Xw[1] = hb_host_in(a.Xin + 2 * N + 2 * v + 1);
    }
    const double dt = a.dt, M = a.M, a2 = dt * dt / M;
    int total = 0, bad = 0, flip = 0, pb = 0;
    // out = dfdy * s for this vertex, s of the neighbours from P[pb]
    double Bq[DEG][3];
    auto apply = [&](const double sx, const double sy, double &ox, double &oy) {
        const double *S = P + pb * 2 * NEWTON4_NT;
        double ax = 0.0, ay = 0.0;
#pragma unroll
        for (int q = 0; q < DEG; q++) {
            const double dx = sx - S[2 * nb[q]], dy = sy - S[2 * nb[q] + 1];
            const double tx = Bq[q][0] * dx + Bq[q][1] * dy, ty = Bq[q][1] * dx + Bq[q][2] * dy;
            ax -= tx; ay -= ty;
        }
        ox = ax; oy = ay;
    };
    // publish a vector for the neighbours' gathers: into the buffer nobody reads any more, then one barrier
    auto publish = [&](const double sx, const double sy) {
        pb ^= 1;
        if (act) { P[pb * 2 * NEWTON4_NT + 2 * v] = sx; P[pb * 2 * NEWTON4_NT + 2 * v + 1] = sy; }
        __syncthreads();
    };
    // publish a vector AND sum a value over the workgroup behind ONE barrier: the vector goes into the buffer nobody reads
    // any more, the wave's partial sum into the half of `part` nobody reads any more, then everybody sees both
    auto publish_sum = [&](const double sx, const double sy, double val) -> double {
        pb ^= 1;
        if (act) { P[pb * 2 * NEWTON4_NT + 2 * v] = sx; P[pb * 2 * NEWTON4_NT + 2 * v + 1] = sy; }
        double unused2 = 0.0;
        d_wave_sum2(val, unused2);
        double *q = part + flip * 2 * (NEWTON4_NT / 64);
        flip ^= 1;
        if ((threadIdx.x & 63) == 63) q[2 * (threadIdx.x >> 6)] = val;
        __syncthreads();
        double sa = 0.0;
#pragma unroll
        for (int w = 0; w < NEWTON4_NT / 64; w++) sa += q[2 * w];
        return sa;
    };
    for (int st = 0; st < a.steps; st++) {
        const double xy[2] = {Xy[0], Xy[1]}, xw[2] = {Xw[0], Xw[1]};
        double py[2] = {Xy[0], Xy[1]}, pw[2] = {Xw[0], Xw[1]};
        double oy[2] = {0, 0}, ow[2] = {0, 0};
        for (int n = 0;; n++) {
            // while n < maxiter and |xo - xp| > tol |xp|  (kalman.py:939)
            double dn = 0.0, pn = 0.0;
            {
                const double d0 = oy[0] - py[0], d1 = oy[1] - py[1], d2 = ow[0] - pw[0], d3 = ow[1] - pw[1];
                dn = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
                pn = ((py[0] * py[0] + py[1] * py[1]) + pw[0] * pw[0]) + pw[1] * pw[1];
            }
            // (the positions the springs are evaluated at go to LDS behind the same barrier as the two sums)
            if (act) { Y[2 * v] = Xy[0]; Y[2 * v + 1] = Xy[1]; }
            d_wg_sum2(dn, pn, part, flip);
            if (!(n < a.maxiter && sqrt(dn) > a.tol * sqrt(pn))) break;
            oy[0] = py[0]; oy[1] = py[1]; ow[0] = pw[0]; ow[1] = pw[1];
            // the springs at the current state X, bar by bar
            for (int b = t; b < I; b += NEWTON4_NT) {
                const int va = bars[2 * b], vb = bars[2 * b + 1];
                const double dx = Y[2 * va] - Y[2 * vb], dy = Y[2 * va + 1] - Y[2 * vb + 1];
                const double l = sqrt(dx * dx + dy * dy);
                const double l0 = a.l0[b];
                const double k = a.kappa * (1.0 - l0 / l), c = a.kappa * l0 / (l * l * l);
                KK[b] = k; BXX[b] = k + c * dx * dx; BXY[b] = c * dx * dy; BYY[b] = k + c * dy * dy;
            }
            __syncthreads();
            double f0 = 0.0, f1 = 0.0;
#pragma unroll
            for (int q = 0; q < DEG; q++) {
                const double k = KK[bb[q]];
                Bq[q][0] = BXX[bb[q]]; Bq[q][1] = BXY[bb[q]]; Bq[q][2] = BYY[bb[q]];
                // (host: f[a] += k d, f[b] -= k d with d = y_a - y_b: from either end  + k (y_v - y_other))
                f0 += k * (Xy[0] - Y[2 * nb[q]]);
                f1 += k * (Xy[1] - Y[2 * nb[q] + 1]);
            }
            // g = xp - x - dt [w; f / M];  (I - dt A) s1 = g1 + dt g2
            const double g1[2] = {py[0] - xy[0] - dt * Xw[0], py[1] - xy[1] - dt * Xw[1]};
            const double g2[2] = {pw[0] - xw[0] - dt * (f0 / M), pw[1] - xw[1] - dt * (f1 / M)};
            const double rhs[2] = {g1[0] + dt * g2[0], g1[1] + dt * g2[1]};
            // |rhs|, and rhs -- the first term of the series below -- published behind the same barrier
            double bn = sqrt(publish_sum(rhs[0], rhs[1], rhs[0] * rhs[0] + rhs[1] * rhs[1]));
            double s1[2] = {0.0, 0.0};
            if (bn != 0.0) {
                // S = I - a2 dfdy with |a2 dfdy| of a few percent: the series s = sum (a2 dfdy)^k rhs gains a digit and a
                // half per term, and a term costs one gather of the neighbours' vector and ONE barrier (the next term is
                // published and its norm summed behind the same one) -- the host's conjugate gradients take nine steps
                // instead of twelve, but each with two sums and two divisions in its dependent chain (measured: a Newton
                // iteration in 7.7 us with conjugate gradients, 6.7 with the series -- with two barriers per term and with
                // one alike: the gather's dependent chain and the lane sums, not the barriers, are what a term costs).  Same
                // stopping rule: the remainder below the rounding of the state the correction is subtracted from.
                const double stop = fmax(1e-15 * bn, 1e-16 * sqrt(pn));
                double r[2] = {rhs[0], rhs[1]};              // the term being added (in P[pb] already)
                bool ok = false;
                for (int it = 0; it < 200; it++) {
                    s1[0] += r[0]; s1[1] += r[1];
                    double tx, ty;
                    apply(r[0], r[1], tx, ty);
                    r[0] = a2 * tx; r[1] = a2 * ty;
                    const double rs = publish_sum(r[0], r[1], r[0] * r[0] + r[1] * r[1]);
                    if (sqrt(rs) <= stop) { ok = true; break; }
                }
                if (!ok) bad = 1;
            }
            // s2 = g2 + A s1;  xp -= [s1; s2];  X = xp
            publish(s1[0], s1[1]);
            double tx, ty;
            apply(s1[0], s1[1], tx, ty);
            py[0] -= s1[0]; py[1] -= s1[1];
            pw[0] -= g2[0] + (dt / M) * tx; pw[1] -= g2[1] + (dt / M) * ty;
            Xy[0] = py[0]; Xy[1] = py[1]; Xw[0] = pw[0]; Xw[1] = pw[1];
            total++;
        }
    }
    const unsigned long long stamp = hb_stamp((long long)a.ticket);
    bad |= a.force_bad;
    if (a.dev_out) {
        if (act) {
            a.dev_out[2 * v] = Xy[0]; a.dev_out[2 * v + 1] = Xy[1];
            a.dev_out[2 * N + 2 * v] = Xw[0]; a.dev_out[2 * N + 2 * v + 1] = Xw[1];
        }
        if (t == 0) { a.dev_out[4 * N] = (double)total; a.dev_out[4 * N + 1] = (double)bad; }
    }
    if (a.delay_us > 0) {
        if (t == 0) { hb_put(a.out, 4 * N + 1, (double)bad, stamp); hb_flush(); }
        hb_delay(a.delay_us);
    }
    if (act) {
        hb_put(a.out, 2 * v, Xy[0], stamp); hb_put(a.out, 2 * v + 1, Xy[1], stamp);
        hb_put(a.out, 2 * N + 2 * v, Xw[0], stamp); hb_put(a.out, 2 * N + 2 * v + 1, Xw[1], stamp);
    }
    if (t == 0) { hb_put(a.out, 4 * N, (double)total, stamp); hb_put(a.out, 4 * N + 1, (double)bad, stamp); }
    hb_flush();
}
