// Mass-spring prediction of the mesh state on the host (native, no GPU work):
// the Newton / implicit-Euler loop of IteratedMSKalmanFilter._newton
// (reference kalman.py:923-960 with _jacobian :865-902 and _dgdx :914-921).
//
// The reference inverts the dense 4N x 4N matrix G = [[I, -dt I], [-A, I]], A = dt/M dfdy,
// in every Newton iteration.  Eliminating the first block row leaves the 2N x 2N system
// (I - dt A) s1 = g1 + dt g2, s2 = g2 + A s1.  dfdy is a sum of 2x2 blocks per spring
// (bar i between vertices a, b with d = y_a - y_b, l = |d|:  B_i = k_i I + c_i d d^T,
// k_i = kappa (1 - l0_i / l), c_i = kappa l0_i / l^3;  -B_i on (a,a), (b,b), +B_i on (a,b), (b,a)),
// so the operator is applied bar by bar and never assembled; I - dt A is symmetric with
// eigenvalues within a few percent of 1 and conjugate gradients reach the rounding floor in
// about ten products.
#include "hm_common.h"
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {
struct Springs {
    int N, I;
    const int32_t *bars;
    std::vector<double> Bxx, Bxy, Byy;       // per-bar block of dfdy

    // out = dfdy * s   (s, out: 2N)
    void apply(const double *s, double *out) const
    {
        for (int i = 0; i < 2 * N; i++) out[i] = 0.0;
        for (int i = 0; i < I; i++) {
            const int a = bars[2 * i], b = bars[2 * i + 1];
            const double sx = s[2 * a] - s[2 * b], sy = s[2 * a + 1] - s[2 * b + 1];
            const double tx = Bxx[i] * sx + Bxy[i] * sy, ty = Bxy[i] * sx + Byy[i] * sy;
            out[2 * a] -= tx; out[2 * a + 1] -= ty;
            out[2 * b] += tx; out[2 * b + 1] += ty;
        }
    }
};

double nrm2(const std::vector<double> &a)
{
    double s = 0.0;
    for (double v : a) s += v * v;
    return std::sqrt(s);
}
}  // namespace

// X: 4N doubles [y; v], advanced in place over one frame (ceil(1/dt) sub-steps).
// Returns HM_OK, or HM_ERR_STATE if the inner solve does not converge (never observed).
extern "C" int hm_ms_newton(int N, int I, const int32_t *bars, const double *l0, double kappa, double M, double dt,
                            int maxiter, double tol, double *X, int *newton_iterations)
{
    HM_ARG(N >= 1 && I >= 0 && bars && l0 && X, "hm_ms_newton: bad argument");
    HM_ARG(dt > 0 && M > 0 && maxiter >= 1 && tol > 0, "hm_ms_newton: bad parameter");
    for (int i = 0; i < 2 * I; i++) HM_ARG(bars[i] >= 0 && bars[i] < N, "hm_ms_newton: bar refers to vertex %d", bars[i]);
    const int n2 = 2 * N, n4 = 4 * N;
    Springs sp;
    sp.N = N; sp.I = I; sp.bars = bars;
    sp.Bxx.resize(I); sp.Bxy.resize(I); sp.Byy.resize(I);
    std::vector<double> x(n4), xp(n4), xo(n4), g(n4), f(n2), rhs(n2), s1(n2), s2(n2), r(n2), p(n2), Sp(n2), tmp(n2);
    const int steps = (int)std::ceil(1.0 / dt);
    int total = 0;
    for (int st = 0; st < steps; st++) {
        for (int i = 0; i < n4; i++) { x[i] = X[i]; xp[i] = X[i]; xo[i] = 0.0; }
        int n = 0;
        for (;;) {
            // while n < maxiter and |xo - xp| > tol |xp|
            double dn = 0.0, pn = 0.0;
            for (int i = 0; i < n4; i++) { dn += (xo[i] - xp[i]) * (xo[i] - xp[i]); pn += xp[i] * xp[i]; }
            if (!(n < maxiter && std::sqrt(dn) > tol * std::sqrt(pn))) break;
            xo = xp;
            // forces and Jacobian blocks at the current state X (= xp after the first iteration)
            for (int i = 0; i < n2; i++) f[i] = 0.0;
            for (int i = 0; i < I; i++) {
                const int a = bars[2 * i], b = bars[2 * i + 1];
                const double dx = X[2 * a] - X[2 * b], dy = X[2 * a + 1] - X[2 * b + 1];
                const double l = std::sqrt(dx * dx + dy * dy);
                const double k = kappa * (1.0 - l0[i] / l), c = kappa * l0[i] / (l * l * l);
                f[2 * a] += k * dx; f[2 * a + 1] += k * dy;
                f[2 * b] -= k * dx; f[2 * b + 1] -= k * dy;
                sp.Bxx[i] = k + c * dx * dx; sp.Bxy[i] = c * dx * dy; sp.Byy[i] = k + c * dy * dy;
            }
            // g = xp - x - dt [v; f / M]
            for (int i = 0; i < n2; i++) {
                g[i] = xp[i] - x[i] - dt * X[n2 + i];
                g[n2 + i] = xp[n2 + i] - x[n2 + i] - dt * (f[i] / M);
            }
            // (I - dt A) s1 = g1 + dt g2,  A = dt/M dfdy:   S s = s - (dt^2/M) dfdy s
            const double a2 = dt * dt / M;
            for (int i = 0; i < n2; i++) rhs[i] = g[i] + dt * g[n2 + i];
            const double bn = nrm2(rhs);
            if (bn == 0.0) {
                for (int i = 0; i < n2; i++) s1[i] = 0.0;
            } else {
                s1 = rhs;
                sp.apply(s1.data(), tmp.data());
                double rs = 0.0;
                for (int i = 0; i < n2; i++) { r[i] = rhs[i] - (s1[i] - a2 * tmp[i]); p[i] = r[i]; rs += r[i] * r[i]; }
                // Stop where the correction is exact to the rounding of the state it is subtracted from: S is within a
                // few percent of I, so the error of s1 is the residual, and a residual of 1e-16 |x| changes no digit
                // of xp that the reference's dense solve would get right.  Relative to the right-hand side alone
                // (1e-15 |rhs|) the second Newton iteration of a sub-step -- whose right-hand side is ~1e-7 of the
                // first's -- took as many products as the first: 40 % of all of them.
                const double floor_abs = 1e-16 * std::sqrt(pn);
                bool ok = false;
                for (int it = 0; it < 200; it++) {
                    if (std::sqrt(rs) <= std::max(1e-15 * bn, floor_abs)) { ok = true; break; }
                    sp.apply(p.data(), tmp.data());
                    double pSp = 0.0;
                    for (int i = 0; i < n2; i++) { Sp[i] = p[i] - a2 * tmp[i]; pSp += p[i] * Sp[i]; }
                    const double alpha = rs / pSp;
                    double rs_new = 0.0;
                    for (int i = 0; i < n2; i++) { s1[i] += alpha * p[i]; r[i] -= alpha * Sp[i]; rs_new += r[i] * r[i]; }
                    const double beta = rs_new / rs;
                    for (int i = 0; i < n2; i++) p[i] = r[i] + beta * p[i];
                    rs = rs_new;
                }
                if (!ok && !(std::sqrt(rs) <= std::max(1e-13 * bn, floor_abs))) {
                    hm_set_error("hm_ms_newton: the inner solve did not converge (residual %g of %g)", std::sqrt(rs), bn);
                    return HM_ERR_STATE;
                }
            }
            // s2 = g2 + A s1
            sp.apply(s1.data(), tmp.data());
            for (int i = 0; i < n2; i++) s2[i] = g[n2 + i] + (dt / M) * tmp[i];
            for (int i = 0; i < n2; i++) { xp[i] -= s1[i]; xp[n2 + i] -= s2[i]; }
            for (int i = 0; i < n4; i++) X[i] = xp[i];
            n++;
            total++;
        }
    }
    if (newton_iterations) *newton_iterations = total;
    return HM_OK;
}


// ---- the same prediction started ahead of time --------------------------------------------------------------
// The state a frame ends with is the state the next frame's prediction starts from, and it is known on the host
// the moment the update's last iteration has reported: hm_ms_newton for the NEXT frame can run on a host thread
// while the device finishes the covariance of this one and the caller does its bookkeeping between frames.  One
// persistent thread per worker object (posting a job costs a few microseconds; creating a thread per frame ~60).
// (the device version of the loop: hm_newton_dev_start / _finish, csrc/ekf.hip; 1 = this job is for the host)
namespace {
struct NewtonWorker {
    hm_ctx_t ctx = nullptr;          // hm_ms_worker_attach: jobs go to this handle's device when they fit its kernel
    bool on_device = false;          // the job in flight is a launch, not a post to the thread
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool quit = false, posted = false, busy = false;
    // the job
    int N = 0, I = 0, maxiter = 0, its = 0, rc = HM_OK;
    double kappa = 0, M = 0, dt = 0, tol = 0;
    std::vector<int32_t> bars;
    std::vector<double> l0, X;
    char err[512] = "";

    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return quit || posted; });
            if (quit) return;
            posted = false;
            lk.unlock();
            const int r = hm_ms_newton(N, I, bars.data(), l0.data(), kappa, M, dt, maxiter, tol, X.data(), &its);
            if (r != HM_OK) snprintf(err, sizeof err, "%s", hm_last_error());
            lk.lock();
            rc = r;
            busy = false;
            cv.notify_all();
        }
    }
};
}  // namespace

extern "C" int hm_ms_worker_create(void **out)
{
    HM_ARG(out != nullptr, "hm_ms_worker_create: out is NULL");
    NewtonWorker *w = new NewtonWorker();
    w->th = std::thread([w] { w->run(); });
    *out = w;
    return HM_OK;
}

// Jobs started on this worker run as one launch on the device of the filter handle `ctx` (hm_ctx_t; NULL: back to the
// host thread) whenever the mesh fits the kernel (k_ms_newton4: at most 256 vertices, 12 springs per vertex); the
// handle must outlive the worker's jobs.
extern "C" int hm_ms_worker_attach(void *worker, hm_ctx_t ctx)
{
    NewtonWorker *w = (NewtonWorker *)worker;
    HM_ARG(w != nullptr, "hm_ms_worker_attach: NULL worker");
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return !w->busy; });
    w->ctx = ctx;
    return HM_OK;
}

extern "C" int hm_ms_worker_destroy(void *worker)
{
    NewtonWorker *w = (NewtonWorker *)worker;
    if (!w) return HM_OK;
    {
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return !w->busy; });
        w->quit = true;
        w->cv.notify_all();
    }
    w->th.join();
    delete w;
    return HM_OK;
}

extern "C" int hm_ms_newton_start(void *worker, int N, int I, const int32_t *bars, const double *l0, double kappa, double M,
                                  double dt, int maxiter, double tol, const double *X)
{
    NewtonWorker *w = (NewtonWorker *)worker;
    HM_ARG(w && N >= 1 && I >= 0 && bars && l0 && X, "hm_ms_newton_start: bad argument");
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return !w->busy; });       // a job whose result nobody fetched is superseded
    w->N = N; w->I = I; w->kappa = kappa; w->M = M; w->dt = dt; w->maxiter = maxiter; w->tol = tol;
    w->bars.assign(bars, bars + 2 * (size_t)I);
    w->l0.assign(l0, l0 + I);
    w->X.assign(X, X + 4 * (size_t)N);
    w->on_device = false;
    if (w->ctx && I >= 1) {
        const int rc = hm_newton_dev_start(w->ctx, N, I, bars, l0, kappa, M, dt, maxiter, tol, X);
        if (rc == HM_OK) { w->on_device = true; w->rc = HM_OK; return HM_OK; }
        if (rc != 1) return rc;
    }
    w->busy = true;
    w->posted = true;
    w->cv.notify_all();
    return HM_OK;
}

// waits for the job; X (4N) receives the advanced state.  HM_ERR_STATE: nothing was started.
extern "C" int hm_ms_newton_finish(void *worker, double *X, int *newton_iterations)
{
    NewtonWorker *w = (NewtonWorker *)worker;
    HM_ARG(w && X, "hm_ms_newton_finish: bad argument");
    std::unique_lock<std::mutex> lk(w->m);
    if (w->X.empty()) { hm_set_error("hm_ms_newton_finish: no job was started"); return HM_ERR_STATE; }
    if (w->on_device) {
        w->on_device = false;
        const int rc = hm_newton_dev_finish(w->ctx, X, newton_iterations);
        if (rc != 1) return rc;
        // the kernel's inner solve gave up: once more, here
        memcpy(X, w->X.data(), w->X.size() * sizeof(double));
        return hm_ms_newton(w->N, w->I, w->bars.data(), w->l0.data(), w->kappa, w->M, w->dt, w->maxiter, w->tol, X, newton_iterations);
    }
    w->cv.wait(lk, [&] { return !w->busy && !w->posted; });
    if (w->rc != HM_OK) { hm_set_error("%s", w->err); return w->rc; }
    memcpy(X, w->X.data(), w->X.size() * sizeof(double));
    if (newton_iterations) *newton_iterations = w->its;
    return HM_OK;
}
