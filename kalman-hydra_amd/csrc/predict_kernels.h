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
