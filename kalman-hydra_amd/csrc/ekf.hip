// EKF measurement model on MI355X: host orchestration and C-ABI (include/hydra_mi.h).
// Replaces reference renderer.py (OpenGL rasteriser), cuda.py and cuda_multi.py
// (reduction kernels + PBO plumbing).
#include "hm_common.h"
#include "ekf_kernels.h"
#include "dense_kernels.h"
#include "chol_flow_kernels.h"
#include "project_kernels.h"
#include "predict_kernels.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <thread>
#include <utility>
#include <vector>

// A persistent helper thread of a handle: it queues launches the calling thread does not have to wait for (the tail of
// hm_update_run: ~20 launches, ~80 us of host time at a frame boundary, where the caller's way to the NEXT frame's first
// launches is what the device ends up waiting for).  One job at a time; every entry point joins it first (ctx_join).
struct Helper {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool posted = false, busy = false, quit = false;
    int rc = HM_OK;
    char err[512] = "";
    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return quit || posted; });
            if (quit) return;
            posted = false;
            std::function<int()> f = std::move(job);
            lk.unlock();
            const int r = f();
            if (r != HM_OK) snprintf(err, sizeof err, "%s", hm_last_error());
            lk.lock();
            rc = r;
            busy = false;
            cv.notify_all();
        }
    }
    void post(std::function<int()> f)
    {
        std::unique_lock<std::mutex> lk(m);
        if (!th.joinable()) th = std::thread([this] { run(); });
        job = std::move(f);
        posted = busy = true;
        cv.notify_all();
    }
    // waits for the job in flight; its return code (reported once)
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
        const int r = rc;
        rc = HM_OK;
        return r;
    }
    void stop()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return !busy; });
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct hm_ctx {
    int device, W, H, N, T, E, njobs;
    Helper helper;
    bool helper_used = false;
    int tail_async = 1;              // hm_ctx_tune "tail_async": the tail of hm_update_run queued by the helper thread (same results)
    double eps_Z, eps_J, eps_M;
    hipStream_t stream;
    // mesh
    int *d_tri, *d_star_off, *d_star_tri, *d_edges;
    int *d_nb_off = nullptr, *d_nb_u = nullptr, *d_nb_e = nullptr;   // per vertex: neighbours (ascending) and their edge jobs (k_solve_prep)
    float *d_uv;
    uint8_t *d_tex;
    std::vector<int> edges;          // host copy, E*2
    // observation
    uint8_t *d_yim, *d_ym;
    float *d_yfx, *d_yfy, *d_yfxm, *d_yfym;
    bool obs_owned;                  // false when set_observation_dev aliases caller memory
    const uint8_t *o_yim, *o_ym;     // pointers in use (owned or caller's)
    const float *o_yfx, *o_yfy;
    bool have_tex, have_obs, have_ref;
    // renders
    Targets ref, P, Q;
    TriSetup *d_setup, *d_cfgs;
    int4 *d_ubox;
    int *d_tlist = nullptr, *d_tcount = nullptr;   // ... and the list / count of the tiles with a non-zero word
    unsigned *d_tmask = nullptr;     // per vertex and tile of its star region: the star triangles that can reach the tile (k_measure_vertex)
    double *d_X, *d_out, *d_partial;
    uint8_t *d_im8, *d_m8;
    std::vector<double> X0;          // state of the reference render
    // dense update on the device (n4 = 4N)
    double *d_HTH;                   // dense HTH of the last measurement: zero outside the J pattern (cleared once;
                                     // every pattern entry is rewritten by every measurement), used for nothing else
    double *d_H, *d_Hz, *d_Hzc, *d_invW0, *d_Af[2], *d_T[2], *d_step, *d_Wtmp, *d_X0, *d_Xn, *d_Wprior, *d_gain, *d_Awork, *d_Lt[2];
    std::vector<double> upd_X0;      // prior mean given to hm_update_begin
    int upd_last, upd_prev;          // which d_Af holds the factor of the last / previous step (-1: none)
    double *d_Wres;                  // the covariance resident on the device (the result of the last
                                     // hm_cov_predict / hm_update_cov / hm_update_run): d_Wtmp, d_H or
                                     // d_Wprior, or NULL when that buffer has been reused since
    double *pin;                     // page-locked, device-mapped: hm_update_run's result blocks (host_block.h) -- per iteration
                                     // [step (n4) | RES_HEAD values], then the tail block [Hzc (n4 x 4) | gains (3 x n4)], every value
                                     // a pair of words -- and two words of scratch (pin_scratch)
    size_t pin_n;
    int *pin_scratch;                // ... where hm_measure / hm_update_step have a flag copied to
    std::vector<double> resv, tailv; // the host's copies of the two blocks, taken when whole (hb_wait)
    int result_delay = 0;            // test knob: the result kernels publish a block's last word first, the rest this many us later
    // the tail block of the last hm_update_run (Hz components, gains): taken by hm_update_tail, or by hm_update_run itself
    // when its caller wants them at once
    bool tail_pending = false;
    long long tail_ticket = 0;
    hipStream_t tail_stream = nullptr;
    const uint8_t *armed_mask = nullptr;     // hm_update_arm_mask: the next hm_update_run queues this mask's outline when its state is final
    std::vector<int> sp_h_off, sp_h_bar, sp_h_other;   // host staging of the spring topology
    std::vector<double> sp_h_blk;
    std::vector<int32_t> sp_bars_cached;   // the springs whose topology is on the device (d_sp_off / _bar / _other)
    std::vector<int> tri;            // host copy of the triangles (orientation test of hm_update_run)
    DPool pool;                      // parked difference images (see ekf_kernels.h)
    int *d_area;
    int *d_sp_off, *d_sp_bar, *d_sp_other;
    double *d_sp_blk;
    size_t sp_cap;                   // capacity (springs) of the d_sp_* arrays
    bool upd_open;
    bool prefactored;                // d_invW0 is the inverse of the resident covariance d_Wprior (hm_update_prefactor)
    std::vector<double> h_partial;
    int red_blocks;
    double *d_tpart = nullptr;       // per-tile partial sums of Renderer.error from k_render_iter (tiles x 4)
    std::vector<double> h_tpart;
    int ntiles = 0;
    int render_rows = RI_H;          // strip height of k_render_iter (16 or 8)
    // Renderer.error of the state hm_update_run kept, against the raw flow, from the render of its last iteration
    // (hm_update_last_error): valid until the observation or anything else on the handle changes
    bool last_err_valid = false;
    double last_err[4] = {0, 0, 0, 0};
    std::vector<double> last_err_X;
    long long run_ticket;            // sequence number of hm_update_run's per-iteration result blocks
    int vsplit, esplit;              // workgroups per vertex / per edge job of the measurement (hm_ctx_tune)
    // hm_update_arm_newton: what the next hm_update_run starts when its state is final
    bool pn_armed = false;
    void *pn_worker = nullptr;
    std::vector<int32_t> pn_bars;
    std::vector<double> pn_l0;
    double pn_par[4] = {0, 0, 0, 0};
    int pn_maxiter = 0;
    // hm_update_arm_cov: the covariance half of the next frame's prediction queued by hm_update_run itself (pq = "pre-queued")
    bool pq_armed = false;           // the next hm_update_run is asked to queue it
    double pq_eps_F = 0.0;
    bool pq_valid = false;           // queued and not yet taken: d_Wprior / d_invW0 hold the prediction made from ...
    std::vector<double> pq_X, pq_l0; // ... this state, these springs and parameters (kappa, a, s, eps_F)
    std::vector<int32_t> pq_bars;
    double pq_par[4] = {0, 0, 0, 0};
    // hm_newton_dev_start / _finish: the state prediction's Newton loop as a four-wave kernel on a stream of its own
    hipStream_t stream3 = nullptr;
    int *d_n4nbr = nullptr, *d_n4nbb = nullptr, *d_n4bars = nullptr;
    double *d_n4l0 = nullptr;
    size_t n4cap = 0;                // bars the device arrays hold
    int n4deg = 0;                   // padded degree of the table on the device (8 or 12), 0: this mesh does not fit the kernel
    std::vector<int32_t> n4_bars;    // the springs the table was built for
    std::vector<double> n4_l0;
    double *pin_n4 = nullptr;        // page-locked: [X in (4N) | result block of 4N + 2 values (host_block.h): X out, iterations, failed]
    double *d_n4X = nullptr;         // the same 4N + 2 values in device memory, for hm_chain_project
    std::vector<double> n4v;         // the host's copy of that block
    long long n4_ticket = 0;
    bool n4_pending = false;
    int newton_fail = 0;             // test knob: the next device predictions report a failed inner solve
    hipEvent_t ev_n4 = nullptr, ev_pm = nullptr;     // the prediction's kernel / the chained projection have run
    // hm_chain_project: projectmask of the prediction in flight queued behind it; the projected state (d_X0) is the
    // prior mean of the next hm_update_run, which also collects what the two kernels report
    bool chain_pending = false;
    std::vector<double> chain_pred, chain_proj;     // ... the predicted / the projected state of the last chained update
    int chain_its = 0, chain_moved = 0;
    double *pin_blk = nullptr;       // page-locked staging of the spring blocks of that prediction
    size_t pin_blk_cap = 0;
    std::thread worker;              // hm_update_prefactor queues its launches from here while the caller predicts the state
    bool worker_active;
    int worker_rc;
    char worker_err[512];
    int chol_flow, flow_wgs;         // the factorisation as one persistent launch (chol_flow_kernels.h) / its workgroups
    int flow_stall = 0;              // test knob: FlowArgs.stall
    int speculate = 1;               // hm_update_run queues the next iteration's measurement before it knows that there is one
    double *d_flowP;                 // 3 x nb x 32 x 32 scratch of that launch
    unsigned *d_flowctl;             // its task counter and time-out word
    hipStream_t stream2;             // hm_ms_predict: the state prediction runs beside the covariance half of the update
    // The tail of hm_update_run -- covariance of the kept state, gains, their result block, and the covariance half of the
    // NEXT frame's prediction (queue_predict_ahead) -- runs on a stream of its own: beside the measurement hm_update_run
    // queued for an iteration that did not happen, and beside the next frame's reference render and measurement, which
    // need none of its buffers.  tail_on_stream4: it may still be running; `stream` waits for ev_tail (on the device)
    // before anything there touches the dense buffers (ctx_join; hm_update_run itself only before its first solve).
    hipStream_t stream4 = nullptr;
    hipEvent_t ev_tail = nullptr;
    bool tail_on_stream4 = false;
    int tail_split = 1;              // hm_ctx_tune "tail_split": 0 keeps the tail on `stream` (same results either way)
    int *d_nbars, *d_nvoff, *d_nvbar, *d_ninfo;     // its spring topology (bars, CSR of the bars of every vertex), result words
    double *d_nl0, *d_nX;
    size_t ncap;                     // bars the buffers hold
    int *d_ids[3];                   // hm_jz_multi / hm_j_multi: id images of the reference and the two perturbed renders,
    int *d_labels;                   // the label palette (T), per-label boxes and sums; allocated on first use
    int4 *d_lbox;
    double *d_lout;
    int lcap;                        // labels the last two hold
    std::vector<double> h_lout;
    int2 *d_outline;                 // hm_project_mask: outline pixels (W*H), counters, uploaded mask; allocated on first use
    int *d_outline_cnt;
    uint8_t *d_pm_mask;
    uint8_t *d_pm_flag = nullptr;    // border-pixel flags of that mask (W*H)
    uint8_t *d_pm_pruned = nullptr;  // the mask after the reference's contour pruning (k_ccl_*): what the outline and the walk use
    Ccl ccl = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};     // its working arrays
    double *d_pm_X = nullptr;        // hm_project_mask's copy of the state (second stream)
    double *pin_pm = nullptr;        // page-locked: [X in (4N) | result block of 4N + 1 values: projected X, vertices moved] (k_project_mask_host)
    std::vector<double> pmv;         // the host's copy of that block
    int *d_pm_done = nullptr;        // workgroups of k_project_mask_host that have finished
    long long pm_ticket = 0;
    bool outline_ready = false;      // the outline of the resident mask (o_ym) has been queued on the second stream
    const uint8_t *prepared_mask = nullptr;   // hm_prepare_mask: the outline in the buffers is that of this mask (device memory)
    hipEvent_t ev_outline = nullptr; // ... recorded behind every outline queued on the second stream
    hipEvent_t ev_m0 = nullptr;      // the first measurement of an update has run (hm_update_arm_mask)
};

static int alloc_targets(Targets &t, size_t n)
{
    HM_HIP(hm_malloc((void **)&t.acc, n * sizeof(int)));
    HM_HIP(hm_malloc((void **)&t.fx, n * sizeof(float)));
    HM_HIP(hm_malloc((void **)&t.fy, n * sizeof(float)));
    HM_HIP(hm_malloc((void **)&t.cnt, n * sizeof(int)));
    return HM_OK;
}
static void free_targets(Targets &t)
{
    if (t.acc) (void)hipFree(t.acc);
    if (t.fx) (void)hipFree(t.fx);
    if (t.fy) (void)hipFree(t.fy);
    if (t.cnt) (void)hipFree(t.cnt);
}

// Every entry point waits for the launches hm_update_prefactor is still queueing (one handle = one
// stream = one thread at a time, as far as the device can tell) and reports their failure, if any.
static int ctx_join(hm_ctx *h, bool lazy = false)
{
    if (!h) return HM_OK;
    if (h->helper_used) {                        // launches the helper thread is still queueing (the tail of the last update)
        const int rc = h->helper.wait();
        if (rc != HM_OK) { hm_set_error("%s", h->helper.err); return rc; }
    }
    if (!lazy && h->tail_on_stream4) {
        h->tail_on_stream4 = false;
        if (hipSetDevice(h->device) != hipSuccess || hipStreamWaitEvent(h->stream, h->ev_tail, 0) != hipSuccess) {
            hm_set_error("the handle's stream cannot wait for the tail of the last update");
            return HM_ERR_HIP;
        }
    }
    if (h->worker_active) {
        h->worker.join();
        h->worker_active = false;
        if (h->worker_rc != HM_OK) {
            const int rc = h->worker_rc;
            h->worker_rc = HM_OK;
            h->prefactored = false;
            hm_set_error("%s", h->worker_err);
            return rc;
        }
    }
    return HM_OK;
}
#define HM_JOIN(h) do { int _j = ctx_join(h); if (_j) return _j; } while (0)
// entry points of the frame loop that touch none of the tail's buffers (or wait for it themselves, where they do)
#define HM_JOIN_LAZY(h) do { int _j = ctx_join(h, true); if (_j) return _j; } while (0)

// Wait for the stream: poll it for a while (a few microseconds of latency) before falling back on
// hipStreamSynchronize, whose wake-up costs ~20 us -- three of those per frame on the compute() path.
// (A short pure spin, then the core is offered to other threads between polls: with several trackers
// per GPU plus their flow and prefactor helper threads on a CPU-quota'd host the pollers must not take
// the cores away from the threads that queue the launches.)
static hipError_t stream_wait(hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    const auto t_yield = t0 + std::chrono::microseconds(50), t_give_up = t0 + std::chrono::milliseconds(20);
    bool polite = false;
    for (int spin = 0;; spin++) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (polite) std::this_thread::yield();
        else __builtin_ia32_pause();
        if ((spin & 15) == 15) {
            const auto now = std::chrono::steady_clock::now();
            if (now > t_give_up) return hipStreamSynchronize(s);
            polite = now > t_yield;
        }
    }
}

// Take values [first, first + n) of a result block in page-locked host memory (host_block.h) into `out` once every one of
// their pairs carries the launch's stamp.  The last pair is looked at first (one pair of loads per poll while nothing is
// there yet); a block of which some words are still missing is simply looked at again -- no order of arrival is assumed.
// A couple of microseconds against ~20 for waking up from a stream synchronisation, which remains as the fallback: after
// 20 ms the stream is waited for (everything queued behind the block's kernel included) and a block that is not whole
// then is an error.
static int hb_wait(hipStream_t st, const double *blk, size_t first, size_t n, unsigned long long stamp, double *out, const char *who)
{
    const volatile unsigned long long *canary = (const volatile unsigned long long *)blk + 2 * (first + n - 1);
    const auto t_start = std::chrono::steady_clock::now();
    const auto t_yield = t_start + std::chrono::microseconds(700), t_give_up = t_start + std::chrono::milliseconds(20);
    bool polite = false;
    for (int spin = 0;; spin++) {
        if ((canary[0] ^ canary[1]) == stamp && hb_take(blk, first, n, stamp, out)) return HM_OK;
        if (polite) std::this_thread::yield();
        else __builtin_ia32_pause();
        if ((spin & 255) == 255) {
            const auto now = std::chrono::steady_clock::now();
            if (now > t_give_up) break;
            polite = now > t_yield;
        }
    }
    HM_HIP(hipStreamSynchronize(st));
    if (hb_take(blk, first, n, stamp, out)) return HM_OK;
    hm_set_error("%s: a result block of the device is not whole although its stream has completed (stamp %016llx)", who, stamp);
    return HM_ERR_HIP;
}

// ---- the pool of parked difference images (DPool, ekf_kernels.h) -------------------------------------------------
static void pool_release(hm_ctx *h)
{
    void *ptrs[] = {h->pool.live, h->pool.xi, h->pool.yi, h->pool.xfx, h->pool.xfy, h->pool.yfx, h->pool.yfy, h->pool.vxfx, h->pool.vyfy};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    h->pool.live = nullptr; h->pool.xi = h->pool.yi = nullptr;
    h->pool.xfx = h->pool.xfy = h->pool.yfx = h->pool.yfy = h->pool.vxfx = h->pool.vyfy = nullptr;
    h->pool.cap = 0;
}
static int pool_alloc(hm_ctx *h, long long cap)
{
    pool_release(h);
    const size_t pc = (size_t)cap;
    HM_HIP(hm_malloc((void **)&h->pool.live, (pc / 64 + 1) * sizeof(int)));
    short2 **sp[] = {&h->pool.xi, &h->pool.yi};
    for (short2 **q : sp) HM_HIP(hm_malloc((void **)q, pc * sizeof(short2)));
    float **fp[] = {&h->pool.xfx, &h->pool.xfy, &h->pool.yfx, &h->pool.yfy, &h->pool.vxfx, &h->pool.vyfy};
    for (float **q : fp) HM_HIP(hm_malloc((void **)q, pc * sizeof(float)));
    h->pool.cap = cap;
    return HM_OK;
}
// A measurement reported that its star regions do not fit (the reference has no such limit: its renders are whole
// frames): make room for them -- the region areas are on the device -- plus a quarter.  The stream must be idle.
static int pool_grow(hm_ctx *h, const char *who)
{
    std::vector<int> area(h->N);
    HM_HIP(hipMemcpyAsync(area.data(), h->d_area, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    long long total = 0;
    for (int a : area) total += a;
    if (total <= h->pool.cap) { hm_set_error("%s: the difference-image pool reported an overflow it does not have", who); return HM_ERR_STATE; }
    const long long want = total + total / 4;
    if (want > ((long long)1 << 32)) {              // 128 GB of planes: not a mesh this path is meant for
        hm_set_error("%s: the star regions need %lld pixels of difference images, more than the pool may hold", who, total);
        return HM_ERR_STATE;
    }
    if (getenv("HYDRA_MI_TRACE")) fprintf(stderr, "[hydra_mi] %s: difference-image pool %lld -> %lld pixels\n", who, h->pool.cap, want);
    int rc = pool_alloc(h, want);
    if (rc) { hm_set_error("%s: cannot grow the difference-image pool to %lld pixels: %s", who, want, hm_last_error()); return rc; }
    return HM_OK;
}

static int ctx_free(hm_ctx *h)
{
    if (!h) return HM_OK;
    (void)ctx_join(h);
    if (h->helper_used) h->helper.stop();
    (void)hipSetDevice(h->device);
    // every stream of the handle first: a state prediction started ahead is always pending after the last frame (it writes
    // into pin_n4), and so may be launches of an update that ended in an error
    { hipStream_t q[] = {h->stream, h->stream2, h->stream3, h->stream4}; for (hipStream_t x : q) if (x) (void)hipStreamSynchronize(x); }
    void *ptrs[] = {h->d_tri, h->d_star_off, h->d_star_tri, h->d_edges, h->d_uv, h->d_tex, h->d_yim, h->d_ym, h->d_yfx,
                    h->d_yfy, h->d_yfxm, h->d_yfym, h->d_setup, h->d_cfgs, h->d_ubox, h->d_X, h->d_out, h->d_partial, h->d_im8, h->d_m8,
                    h->d_HTH, h->d_H, h->d_Hz, h->d_Hzc, h->d_invW0, h->d_Af[0], h->d_Af[1], h->d_T[0], h->d_T[1], h->d_step, h->d_Wprior, h->d_gain, h->d_Awork, h->d_Lt[0], h->d_Lt[1],
                    h->d_Wtmp, h->d_X0, h->d_Xn, h->d_sp_off, h->d_sp_bar, h->d_sp_other, h->d_sp_blk,
                    h->pool.hdr, h->pool.overflow, h->d_area,
                    h->d_outline, h->d_outline_cnt, h->d_pm_mask, h->d_pm_flag, h->d_pm_X, h->d_pm_pruned, h->ccl.L, h->ccl.cnt, h->ccl.bnd, h->ccl.edge, h->ccl.best, h->d_ids[0], h->d_ids[1], h->d_ids[2], h->d_labels, h->d_lbox,
                    h->d_lout, h->d_tpart, h->d_tmask, h->d_tlist, h->d_tcount, h->d_nb_off, h->d_nb_u, h->d_nb_e, h->d_flowP, h->d_flowctl, h->d_nbars, h->d_nvoff, h->d_nvbar, h->d_ninfo, h->d_nl0, h->d_nX};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    pool_release(h);
    free_targets(h->ref);
    free_targets(h->P);
    free_targets(h->Q);
    { void *q[] = {h->d_n4nbr, h->d_n4nbb, h->d_n4bars, h->d_n4l0, h->d_n4X, h->d_pm_done}; for (void *x : q) if (x) (void)hipFree(x); }
    // (the streams were drained at the top: nothing writes into the page-locked blocks any more)
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->pin_pm) (void)hipHostFree(h->pin_pm);
    if (h->pin_blk) (void)hipHostFree(h->pin_blk);
    if (h->pin_n4) (void)hipHostFree(h->pin_n4);
    if (h->ev_n4) (void)hipEventDestroy(h->ev_n4);
    if (h->ev_pm) (void)hipEventDestroy(h->ev_pm);
    if (h->ev_outline) (void)hipEventDestroy(h->ev_outline);
    if (h->ev_m0) (void)hipEventDestroy(h->ev_m0);
    if (h->ev_tail) (void)hipEventDestroy(h->ev_tail);
    if (h->stream4) (void)hipStreamDestroy(h->stream4);
    if (h->stream3) (void)hipStreamDestroy(h->stream3);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return HM_OK;
}

template <typename T>
static int upload(T **dst, const T *src, size_t n)
{
    HM_HIP(hm_malloc((void **)dst, (n ? n : 1) * sizeof(T)));
    if (n) HM_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return HM_OK;
}

extern "C" int hm_ctx_create(int device, int W, int H, int N, int T, const int32_t *tri, const float *uv, double eps_Z,
                             double eps_J, double eps_M, hm_ctx_t *out)
{
    HM_ARG(out != nullptr, "hm_ctx_create: out is NULL");
    *out = nullptr;
    HM_ARG(W >= 1 && H >= 1 && W <= 4096 && H <= 4096, "hm_ctx_create: frame size %dx%d outside 1..4096", W, H);
    HM_ARG(N >= 3 && T >= 1 && T <= EKF_MAX_TRI, "hm_ctx_create: need N >= 3 vertices and 1..%d triangles (N=%d, T=%d)",
           EKF_MAX_TRI, N, T);
    HM_ARG(tri && uv, "hm_ctx_create: NULL mesh arrays");
    HM_ARG(eps_Z > 0 && eps_J > 0 && eps_M > 0, "hm_ctx_create: eps_Z, eps_J, eps_M must be positive");
    std::vector<std::vector<int>> star(N);
    std::set<std::pair<int, int>> eset;
    for (int t = 0; t < T; t++) {
        int v[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
        for (int k = 0; k < 3; k++) {
            HM_ARG(v[k] >= 0 && v[k] < N, "hm_ctx_create: triangle %d refers to vertex %d (N=%d)", t, v[k], N);
            star[v[k]].push_back(t);
            int a = v[k], b = v[(k + 1) % 3];
            if (a != b) eset.insert(std::make_pair(std::min(a, b), std::max(a, b)));
        }
    }
    std::vector<int> off(N + 1, 0), flat;
    for (int v = 0; v < N; v++) {
        // a vertex listed twice in one (degenerate) triangle would be listed twice here
        std::sort(star[v].begin(), star[v].end());
        star[v].erase(std::unique(star[v].begin(), star[v].end()), star[v].end());
        HM_ARG((int)star[v].size() <= EKF_MAX_STAR, "hm_ctx_create: vertex %d has %d triangles, limit %d", v,
               (int)star[v].size(), EKF_MAX_STAR);
        off[v + 1] = off[v] + (int)star[v].size();
        flat.insert(flat.end(), star[v].begin(), star[v].end());
    }
    HM_HIP(hipSetDevice(device));
    hm_ctx *h = new hm_ctx();
    h->device = device; h->W = W; h->H = H; h->N = N; h->T = T;
    h->eps_Z = eps_Z; h->eps_J = eps_J; h->eps_M = eps_M;
    h->obs_owned = true; h->have_tex = h->have_obs = h->have_ref = false;
    h->o_yim = h->o_ym = nullptr; h->o_yfx = h->o_yfy = nullptr;
    h->d_yim = h->d_ym = nullptr; h->d_yfx = h->d_yfy = h->d_yfxm = h->d_yfym = nullptr;
    h->ref = Targets{nullptr, nullptr, nullptr, nullptr}; h->P = h->ref; h->Q = h->ref;
    h->d_setup = nullptr; h->d_cfgs = nullptr; h->d_ubox = nullptr; h->d_X = h->d_out = h->d_partial = nullptr; h->d_im8 = h->d_m8 = nullptr;
    h->d_HTH = nullptr;
    h->d_H = h->d_Hz = h->d_Hzc = h->d_invW0 = h->d_Af[0] = h->d_Af[1] = h->d_Wtmp = nullptr;
    h->d_Wprior = h->d_gain = h->d_Awork = h->d_Lt[0] = h->d_Lt[1] = nullptr; h->d_Wres = nullptr; h->pin = nullptr; h->pin_n = 0;
    h->tri.assign(tri, tri + (size_t)3 * T);
    h->d_outline = nullptr; h->d_outline_cnt = nullptr; h->d_pm_mask = nullptr;
    // (192 workgroups of the factorisation launch: 304.9 -> 306.9 us per iteration with the filter alone, but 260.5 -> 264.2 and
    // 318.6 -> 324.1 frames/s in the two benches -- the workgroups that poll for blocks take issue slots from the flow's
    // kernels on every compute unit they sit on; 160 and 128 starve the chain: profiles/r04_ab_tunes.txt)
    h->chol_flow = 1; h->flow_wgs = 192; h->d_flowP = nullptr; h->d_flowctl = nullptr;
    h->stream2 = nullptr; h->d_nbars = h->d_nvoff = h->d_nvbar = h->d_ninfo = nullptr; h->d_nl0 = h->d_nX = nullptr; h->ncap = 0;
    h->d_ids[0] = h->d_ids[1] = h->d_ids[2] = nullptr; h->d_labels = nullptr; h->d_lbox = nullptr; h->d_lout = nullptr; h->lcap = 0;
    h->worker_active = false; h->worker_rc = HM_OK; h->worker_err[0] = 0;
    h->d_T[0] = h->d_T[1] = h->d_step = nullptr; h->d_X0 = h->d_Xn = nullptr;
    memset(&h->pool, 0, sizeof h->pool); h->d_area = nullptr;
    h->d_sp_off = h->d_sp_bar = h->d_sp_other = nullptr; h->d_sp_blk = nullptr; h->sp_cap = 0;
    h->upd_last = h->upd_prev = -1; h->upd_open = false; h->prefactored = false;
    for (const auto &e : eset) { h->edges.push_back(e.first); h->edges.push_back(e.second); }
    h->E = (int)eset.size();
    h->njobs = N + h->E;
    h->red_blocks = 512;
    h->run_ticket = 0;
    h->vsplit = 5;
    // two workgroups per edge job as long as they are ONE round of the chip (122 VGPRs: four workgroups per CU, 1 024 slots),
    // else one: at 201 vertices 556 edges x 2 = 1 112 workgroups had 88 of them wait for a slot and the launch take a second
    // round (308.6 -> 305.1 us per iteration with one; the sums of an edge are then added in another order: rounding-level
    // differences); small meshes keep the parallelism of two
    h->esplit = 2 * h->E > 1024 ? 1 : 2;
    const size_t n = (size_t)W * H;
    int rc = HM_OK;
    auto step = [&](int r) { if (rc == HM_OK) rc = r; };
    step(upload(&h->d_tri, tri, (size_t)3 * T));
    step(upload(&h->d_uv, uv, (size_t)2 * N));
    step(upload(&h->d_star_off, off.data(), off.size()));
    step(upload(&h->d_star_tri, flat.data(), flat.size()));
    step(upload(&h->d_edges, h->edges.data(), h->edges.size()));
    {   // neighbours of every vertex with the edge job that holds their block of HTH
        std::vector<std::vector<std::pair<int, int>>> nb(N);
        for (int e = 0; e < h->E; e++) {
            const int a = h->edges[2 * e], b = h->edges[2 * e + 1];
            nb[a].push_back(std::make_pair(b, e));
            nb[b].push_back(std::make_pair(a, e));
        }
        std::vector<int> noff(N + 1, 0), nu, ne;
        for (int v = 0; v < N; v++) {
            std::sort(nb[v].begin(), nb[v].end());
            noff[v + 1] = noff[v] + (int)nb[v].size();
            for (const auto &q : nb[v]) { nu.push_back(q.first); ne.push_back(q.second); }
            if (rc == HM_OK && 4 * ((int)nb[v].size() + 1) > PREP_MAX_ENTRIES) {
                hm_set_error("hm_ctx_create: vertex %d has %d neighbours, limit %d", v, (int)nb[v].size(), PREP_MAX_ENTRIES / 4 - 1);
                rc = HM_ERR_ARG;
            }
        }
        step(upload(&h->d_nb_off, noff.data(), noff.size()));
        step(upload(&h->d_nb_u, nu.data(), nu.size()));
        step(upload(&h->d_nb_e, ne.data(), ne.size()));
    }
    if (rc == HM_OK) {
        // the filter's launches are a dependent chain of short kernels beside the flow's long ones on another stream:
        // its queue gets the highest dispatch priority the device offers (HYDRA_MI_EKF_PRIORITY=0 turns that off)
        int least = 0, greatest = 0;
        const char *pe = getenv("HYDRA_MI_EKF_PRIORITY");
        hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess && greatest != least && !(pe && atoi(pe) == 0))
            e = hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, greatest);
        else if (e == hipSuccess)
            e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_tex, n);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_yim, n);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_ym, n);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_yfx, n * sizeof(float));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_yfy, n * sizeof(float));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_yfxm, n * sizeof(float));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_yfym, n * sizeof(float));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_setup, (size_t)T * sizeof(TriSetup));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_cfgs, (size_t)N * MEAS_NCFG * (EKF_MAX_STAR + 1) * sizeof(TriSetup));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_ubox, (size_t)N * UBOX_STRIDE * sizeof(int4));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_tmask, (size_t)N * TMASK_STRIDE * sizeof(unsigned));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_tlist, (size_t)N * TMASK_STRIDE * sizeof(int));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_tcount, (size_t)N * sizeof(int));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_X, (size_t)4 * N * sizeof(double));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_out, (size_t)h->njobs * MEAS_VSPLIT_MAX * MEAS_OUT * sizeof(double));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_partial, (size_t)h->red_blocks * 4 * sizeof(double));
        h->ntiles = hm_cdiv(W, RI_W) * hm_cdiv(H, 8);           // most strips of k_render_iter (render_rows 8)
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_tpart, (size_t)h->ntiles * RI_NV * sizeof(double));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_im8, n);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_m8, n);
        const size_t n4 = (size_t)4 * N, nn = n4 * n4 * sizeof(double);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_H, nn);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_HTH, nn);
        if (e == hipSuccess) e = hipMemsetAsync(h->d_HTH, 0, nn, h->stream);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_invW0, nn);
        const size_t nn_aug = (size_t)(hm_cdiv((int)n4, DNB) * DNB + DNB) * n4 * sizeof(double);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Af[0], nn_aug);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Af[1], nn_aug);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Awork, nn_aug);
        const size_t ld_bytes = (size_t)hm_cdiv((int)n4, DNB) * DNB * DNB * sizeof(double);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_T[0], nn);       // L^-1 of the factor in the same slot
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_T[1], nn);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_step, n4 * sizeof(double));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Lt[0], ld_bytes);  // inverses of the factored diagonal blocks
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Lt[1], ld_bytes);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Wtmp, nn);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_flowP, 3 * ld_bytes);      // P, Q and Y blocks of k_chol_flow
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_flowctl, 4 * sizeof(unsigned));
        if (e == hipSuccess) e = hipMemsetAsync(h->d_flowctl, 0, 4 * sizeof(unsigned), h->stream);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Hz, n4 * sizeof(double));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Hzc, n4 * 4 * sizeof(double));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Wprior, nn);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_gain, n4 * 3 * sizeof(double));
        h->pin_n = 2 * (n4 + RES_HEAD + n4 * 4 + n4 * 3) + 2;
        if (e == hipSuccess) e = hipHostMalloc((void **)&h->pin, h->pin_n * sizeof(double), hipHostMallocCoherent);
        if (e == hipSuccess) memset(h->pin, 0, h->pin_n * sizeof(double));
        h->pin_scratch = (int *)(h->pin + h->pin_n - 2);
        h->resv.assign(n4 + RES_HEAD, 0.0);
        h->tailv.assign(n4 * 7, 0.0);
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_X0, n4 * sizeof(double));
        if (e == hipSuccess) e = hm_malloc((void **)&h->pool.hdr, (size_t)4 * N * sizeof(int));
        if (e == hipSuccess) e = hm_malloc((void **)&h->pool.overflow, sizeof(int));
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_area, (size_t)N * sizeof(int));
        h->pool.area = h->d_area;
        // pool of parked difference images (32 B per pixel): the star regions overlap about six times, so a mesh that
        // covers the whole frame needs about six frames' worth of pixels plus the padding of every region to whole 8x8
        // tiles (the bench's disk, a third of the frame: 1.6); it starts at 8 frames' worth = 268 MB at 1024^2 and grows
        // when a measurement reports that its regions do not fit (pool_grow)
        if (e == hipSuccess && pool_alloc(h, (long long)8 * W * H) != HM_OK) e = hipErrorOutOfMemory;
        if (e == hipSuccess) e = hm_malloc((void **)&h->d_Xn, n4 * sizeof(double));
        if (e != hipSuccess) {
            hm_set_error("hm_ctx_create: device allocation failed: %s", hipGetErrorString(e));
            rc = HM_ERR_HIP;
        }
    }
    step(alloc_targets(h->ref, n));
    step(alloc_targets(h->P, n));
    step(alloc_targets(h->Q, n));
    if (rc != HM_OK) {
        ctx_free(h);
        return rc;
    }
    h->h_partial.resize((size_t)h->red_blocks * 4);
    h->h_tpart.resize((size_t)h->ntiles * RI_NV);
    *out = h;
    return HM_OK;
}

extern "C" int hm_ctx_destroy(hm_ctx_t h) { return ctx_free(h); }

#ifdef HM_STAMP
// development builds only: where k_render_iter writes its per-workgroup time stamps (device pointer, 8 words per workgroup)
extern "C" int hm_debug_stamps(void *dev_ptr)
{
    long long *p = (long long *)dev_ptr;
    HM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &p, sizeof p));
    return HM_OK;
}
#endif

extern "C" int hm_ctx_tune(hm_ctx_t h, const char *key, int value)
{
    HM_ARG(h != nullptr && key != nullptr, "hm_ctx_tune: NULL argument");
    HM_JOIN(h);
    if (!strcmp(key, "measure_split")) {
        HM_ARG(value >= 1 && value <= MEAS_VSPLIT_MAX, "hm_ctx_tune: measure_split must be in 1..%d", MEAS_VSPLIT_MAX);
        h->vsplit = value;
    } else if (!strcmp(key, "chol_flow")) {
        HM_ARG(value == 0 || value == 1, "hm_ctx_tune: chol_flow must be 0 (one launch per block step) or 1 (one persistent launch)");
        h->chol_flow = value;
    } else if (!strcmp(key, "chol_flow_wgs")) {
        // at least two: the first workgroup becomes the chain of the diagonal blocks and waits for blocks that only
        // the task workgroups produce
        HM_ARG(value >= 2 && value <= 2048, "hm_ctx_tune: chol_flow_wgs must be in 2..2048 (the chain workgroup and at least one task workgroup)");
        h->flow_wgs = value;
    } else if (!strcmp(key, "render_rows")) {
        HM_ARG(value == 8 || value == 16, "hm_ctx_tune: render_rows must be 8 or 16");
        h->render_rows = value;
    } else if (!strcmp(key, "chol_flow_stall")) {      // tests only: results must not depend on it
        HM_ARG(value >= 0 && value <= 100000, "hm_ctx_tune: chol_flow_stall must be in 0..100000");
        h->flow_stall = value;
    } else if (!strcmp(key, "result_delay")) {         // tests only: results must not depend on it (host_block.h)
        HM_ARG(value >= 0 && value <= 5000, "hm_ctx_tune: result_delay must be in 0..5000 (microseconds)");
        h->result_delay = value;
    } else if (!strcmp(key, "tail_async")) {           // same results either way
        HM_ARG(value == 0 || value == 1, "hm_ctx_tune: tail_async must be 0 or 1");
        h->tail_async = value;
    } else if (!strcmp(key, "tail_split")) {           // same results either way
        HM_ARG(value == 0 || value == 1, "hm_ctx_tune: tail_split must be 0 or 1");
        h->tail_split = value;
    } else if (!strcmp(key, "newton_fail")) {          // tests only: the device predictions report a failed inner solve
        HM_ARG(value == 0 || value == 1, "hm_ctx_tune: newton_fail must be 0 or 1");
        h->newton_fail = value;
    } else if (!strcmp(key, "speculate")) {            // same results either way
        HM_ARG(value == 0 || value == 1, "hm_ctx_tune: speculate must be 0 or 1");
        h->speculate = value;
    } else if (!strcmp(key, "edge_split")) {
        HM_ARG(value >= 1 && value <= MEAS_VSPLIT_MAX, "hm_ctx_tune: edge_split must be in 1..%d", MEAS_VSPLIT_MAX);
        h->esplit = value;
    } else {
        hm_set_error("hm_ctx_tune: unknown key '%s'", key);
        return HM_ERR_ARG;
    }
    return HM_OK;
}
extern "C" void *hm_ctx_stream(hm_ctx_t h)
{
    (void)ctx_join(h);
    return h ? (void *)h->stream : nullptr;
}
extern "C" int hm_ctx_sync(hm_ctx_t h)
{
    HM_ARG(h != nullptr, "hm_ctx_sync: NULL handle");
    HM_JOIN(h);
    HM_HIP(hipSetDevice(h->device));
    HM_HIP(hipStreamSynchronize(h->stream));
    return HM_OK;
}

extern "C" int hm_set_texture(hm_ctx_t h, const uint8_t *tex)
{
    HM_ARG(h && tex, "hm_set_texture: NULL argument");
    HM_JOIN(h);
    HM_HIP(hipSetDevice(h->device));
    HM_HIP(hipMemcpyAsync(h->d_tex, tex, (size_t)h->W * h->H, hipMemcpyHostToDevice, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    h->last_err_valid = false;
    h->have_tex = true;
    return HM_OK;
}

// The second stream (projectmask, hm_ms_predict): short launches beside the first stream's, same dispatch priority.
static int ensure_stream2(hm_ctx *h)
{
    if (h->stream2) return HM_OK;
    int least = 0, greatest = 0;
    HM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (greatest != least) { HM_HIP(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, greatest)); }
    else HM_HIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    return HM_OK;
}

// hm_project_mask's buffers and the outline of a mask in device memory, queued on the second stream
static int project_buffers(hm_ctx *h)
{
    if (h->d_outline) return HM_OK;
    const size_t n = (size_t)h->W * h->H, n4 = (size_t)4 * h->N;
    HM_HIP(hm_malloc((void **)&h->d_outline, n * sizeof(int2)));
    HM_HIP(hm_malloc((void **)&h->d_outline_cnt, 4 * sizeof(int)));
    HM_HIP(hm_malloc((void **)&h->d_pm_flag, n));
    HM_HIP(hm_malloc((void **)&h->d_pm_pruned, n));
    HM_HIP(hm_malloc((void **)&h->ccl.L, n * sizeof(int)));
    HM_HIP(hm_malloc((void **)&h->ccl.cnt, n * sizeof(int)));
    HM_HIP(hm_malloc((void **)&h->ccl.bnd, n * sizeof(int)));
    HM_HIP(hm_malloc((void **)&h->ccl.edge, n));
    HM_HIP(hm_malloc((void **)&h->ccl.best, sizeof(unsigned long long)));
    h->ccl.W = h->W; h->ccl.H = h->H;
    HM_HIP(hm_malloc((void **)&h->d_pm_X, n4 * sizeof(double)));
    HM_HIP(hm_malloc((void **)&h->d_pm_done, sizeof(int)));
    // zeroed on the stream the kernel that counts in it runs on: a hipMemset on the null stream is not ordered with a
    // non-blocking stream and may land in the middle of that kernel -- no workgroup is the last one then, the count of
    // moved vertices never comes and the projection of a context's first frame was silently dropped (seen once in ~10 runs)
    HM_HIP(hipMemsetAsync(h->d_pm_done, 0, sizeof(int), h->stream2));
    HM_HIP(hipHostMalloc((void **)&h->pin_pm, (n4 + 2 * (n4 + 1)) * sizeof(double), hipHostMallocCoherent));
    memset(h->pin_pm, 0, (n4 + 2 * (n4 + 1)) * sizeof(double));
    h->pmv.assign(n4 + 1, 0.0);
    HM_HIP(hipEventCreateWithFlags(&h->ev_pm, hipEventDisableTiming));
    HM_HIP(hipEventCreateWithFlags(&h->ev_outline, hipEventDisableTiming));
    return HM_OK;
}
// the reference's contour pruning of the mask (imgproc.py:198-228: the largest object, its holes of area >= 40) into
// d_pm_pruned, then the border pixels of what is left
static int queue_outline(hm_ctx *h, const uint8_t *mask)
{
    HM_HIP(hipMemsetAsync(h->d_outline_cnt, 0, 4 * sizeof(int), h->stream2));
    const dim3 cg(hm_cdiv(h->W, 64), hm_cdiv(h->H, CCL_NT / 64)), cb(CCL_NT);
    hipLaunchKernelGGL(k_ccl_local, dim3(hm_cdiv(h->W, CCL_TW), hm_cdiv(h->H, CCL_TH)), cb, 0, h->stream2, mask, h->ccl);
    hipLaunchKernelGGL(k_ccl_border, cg, cb, 0, h->stream2, mask, h->ccl);
    hipLaunchKernelGGL(k_ccl_roots, cg, cb, 0, h->stream2, h->ccl);
    hipLaunchKernelGGL(k_ccl_flatten, cg, cb, 0, h->stream2, mask, h->ccl);
    hipLaunchKernelGGL(k_ccl_stats, cg, cb, 0, h->stream2, mask, h->ccl);
    hipLaunchKernelGGL(k_ccl_select, cg, cb, 0, h->stream2, mask, h->ccl);
    hipLaunchKernelGGL(k_ccl_write, cg, cb, 0, h->stream2, mask, h->ccl, h->d_pm_pruned);
    Outline o = {h->d_outline, h->d_outline_cnt, h->W * h->H, h->d_pm_flag};
    hipLaunchKernelGGL(k_outline, dim3(hm_cdiv(h->W, 64), hm_cdiv(h->H, OUTLINE_NT / 64)), dim3(OUTLINE_NT), 0, h->stream2,
                       (const uint8_t *)h->d_pm_pruned, h->W, h->H, o);
    HM_HIP(hipGetLastError());
    HM_HIP(hipEventRecord(h->ev_outline, h->stream2));
    h->prepared_mask = nullptr;
    return HM_OK;
}

static int finish_observation(hm_ctx *h)
{
    h->last_err_valid = false;
    const int n = h->W * h->H;
    hipLaunchKernelGGL(k_mask_flow, dim3(hm_cdiv(n, 256)), dim3(256), 0, h->stream, h->o_ym, h->o_yfx, h->o_yfy,
                       h->d_yfxm, h->d_yfym, n);
    HM_HIP(hipGetLastError());
    h->have_obs = true;
    return HM_OK;
}

extern "C" int hm_set_observation(hm_ctx_t h, const uint8_t *y_im, const float *y_fx, const float *y_fy,
                                  const uint8_t *y_m)
{
    HM_ARG(h && y_im && y_fx && y_fy && y_m, "hm_set_observation: NULL argument");
    HM_JOIN(h);
    HM_HIP(hipSetDevice(h->device));
    const size_t n = (size_t)h->W * h->H;
    HM_HIP(hipMemcpyAsync(h->d_yim, y_im, n, hipMemcpyHostToDevice, h->stream));
    HM_HIP(hipMemcpyAsync(h->d_ym, y_m, n, hipMemcpyHostToDevice, h->stream));
    HM_HIP(hipMemcpyAsync(h->d_yfx, y_fx, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HM_HIP(hipMemcpyAsync(h->d_yfy, y_fy, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->o_yim = h->d_yim; h->o_ym = h->d_ym; h->o_yfx = h->d_yfx; h->o_yfy = h->d_yfy;
    int rc = finish_observation(h);
    if (rc) return rc;
    HM_HIP(hipStreamSynchronize(h->stream));
    h->outline_ready = false;
    if (h->d_outline) {              // a tracker that projects onto the mask: its outline, ahead of hm_project_mask
        rc = queue_outline(h, h->o_ym);
        if (rc) return rc;
        h->outline_ready = true;
    }
    return HM_OK;
}

extern "C" int hm_set_observation_dev(hm_ctx_t h, const uint8_t *d_y_im, const float *d_y_fx, const float *d_y_fy,
                                      const uint8_t *d_y_m)
{
    HM_ARG(h && d_y_im && d_y_fx && d_y_fy && d_y_m, "hm_set_observation_dev: NULL argument");
    HM_JOIN_LAZY(h);
    HM_HIP(hipSetDevice(h->device));
    h->o_yim = d_y_im; h->o_ym = d_y_m; h->o_yfx = d_y_fx; h->o_yfy = d_y_fy;
    int rc = finish_observation(h);
    if (rc) return rc;
    h->outline_ready = false;
    if (h->d_outline && h->prepared_mask == d_y_m) {      // hm_prepare_mask queued the outline of exactly this mask ahead
        h->outline_ready = true;
    } else if (h->d_outline) {       // as in hm_set_observation: the caller's mask is complete in device memory by contract
        rc = queue_outline(h, h->o_ym);
        if (rc) return rc;
        h->outline_ready = true;
    }
    h->prepared_mask = nullptr;
    return HM_OK;
}

// render state X (host, 4N doubles) into target t on the handle's stream
// render the device-resident state dX into target t
static int render_dev(hm_ctx *h, const double *dX, Targets t)
{
    Mesh m = {h->W, h->H, h->N, h->T, h->d_tri, h->d_uv, h->d_tex};
    hipLaunchKernelGGL(k_setup_all, dim3(hm_cdiv(h->T, 64)), dim3(64), 0, h->stream, m, dX, h->d_setup);
    hipLaunchKernelGGL((k_render<0>), dim3(hm_cdiv(h->W, EKF_TILE), hm_cdiv(h->H, EKF_TILE)), dim3(EKF_TILE, EKF_TILE), 0,
                       h->stream, m, dX, h->d_setup, t, (const int *)nullptr, (int *)nullptr);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

static int render_into(hm_ctx *h, const double *X, Targets t)
{
    HM_HIP(hipMemcpyAsync(h->d_X, X, (size_t)4 * h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return render_dev(h, h->d_X, t);
}

static void measure_args(hm_ctx *h, const double *dX, double deltaX, int masked, MeasureArgs &a);
static int render_strips(const hm_ctx *h) { return hm_cdiv(h->W, RI_W) * hm_cdiv(h->H, h->render_rows); }

// The render of the device-resident state dX into target t as ONE launch (k_render_iter): triangle setups made per
// tile, optionally the per-tile partial sums of Renderer.error against the observation (d_tpart) and, riding along
// as extra workgroups, the star regions of the measurement at dX (they read nothing but the state).
static int render_iter(hm_ctx *h, const double *dX, Targets t, bool with_err, int masked, bool regions, double deltaX)
{
    IterRenderArgs r;
    r.m = Mesh{h->W, h->H, h->N, h->T, h->d_tri, h->d_uv, h->d_tex};
    r.X = dX;
    r.out = t;
    r.o = Obs{h->o_yim, masked ? h->d_yfxm : h->o_yfx, masked ? h->d_yfym : h->o_yfy, h->o_ym};
    r.raw_fx = h->o_yfx; r.raw_fy = h->o_yfy;
    r.partial = h->d_tpart;
    r.tiles_x = hm_cdiv(h->W, RI_W); r.tiles_y = hm_cdiv(h->H, h->render_rows);
    const int nstrips = r.tiles_x * r.tiles_y;
    r.with_err = with_err ? 1 : 0;
    r.n_regions = regions ? h->N : 0;
    MeasureArgs a;
    measure_args(h, dX, deltaX, masked, a);
    if (h->render_rows == 8) hipLaunchKernelGGL((k_render_iter<8>), dim3(r.n_regions + nstrips), dim3(256), 0, h->stream, r, a, h->d_area);
    else hipLaunchKernelGGL((k_render_iter<16>), dim3(r.n_regions + nstrips), dim3(256), 0, h->stream, r, a, h->d_area);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

// the RI_NV sums of Renderer.error from the per-strip partials, added in the order of d_tile_partial_sums
static void hm_tile_partial_sums(const double *p, int ntiles, double s[RI_NV])
{
    static thread_local double g[RI_GROUPS][RI_NV];
    for (int t = 0; t < RI_GROUPS; t++) {
        double a[RI_NV];
        for (int k = 0; k < RI_NV; k++) a[k] = 0.0;
        for (int i = t; i < ntiles; i += RI_GROUPS)
            for (int k = 0; k < RI_NV; k++) a[k] += p[RI_NV * (size_t)i + k];
        for (int k = 0; k < RI_NV; k++) g[t][k] = a[k];
    }
    for (int k = 0; k < RI_NV; k++) {
        double v = 0.0;
        for (int t = 0; t < RI_GROUPS; t++) v += g[t][k];
        s[k] = v;
    }
}

#define NEED_TEX(h, who) \
    if (!(h)->have_tex) { hm_set_error(who ": hm_set_texture has not been called"); return HM_ERR_STATE; }
#define NEED_OBS(h, who) \
    if (!(h)->have_obs) { hm_set_error(who ": hm_set_observation has not been called"); return HM_ERR_STATE; }
#define NEED_REF(h, who) \
    if (!(h)->have_ref) { hm_set_error(who ": hm_initjacobian has not been called"); return HM_ERR_STATE; }

extern "C" int hm_render(hm_ctx_t h, const double *X, uint8_t *im, float *fx, float *fy, uint8_t *mk)
{
    HM_ARG(h && X, "hm_render: NULL argument");
    HM_JOIN(h);
    NEED_TEX(h, "hm_render");
    HM_HIP(hipSetDevice(h->device));
    int rc = render_into(h, X, h->P);
    if (rc) return rc;
    const int n = h->W * h->H;
    hipLaunchKernelGGL(k_resolve, dim3(hm_cdiv(n, 256)), dim3(256), 0, h->stream, h->P, h->d_im8, h->d_m8, n);
    if (im) HM_HIP(hipMemcpyAsync(im, h->d_im8, n, hipMemcpyDeviceToHost, h->stream));
    if (mk) HM_HIP(hipMemcpyAsync(mk, h->d_m8, n, hipMemcpyDeviceToHost, h->stream));
    if (fx) HM_HIP(hipMemcpyAsync(fx, h->P.fx, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (fy) HM_HIP(hipMemcpyAsync(fy, h->P.fy, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    return HM_OK;
}

static Obs obs_of(hm_ctx *h, int masked)
{
    Obs o = {h->o_yim, masked ? h->d_yfxm : h->o_yfx, masked ? h->d_yfym : h->o_yfy, h->o_ym};
    return o;
}

extern "C" int hm_initjacobian(hm_ctx_t h, const double *X, int masked)
{
    HM_ARG(h && X, "hm_initjacobian: NULL argument");
    HM_JOIN(h);
    (void)masked;
    NEED_TEX(h, "hm_initjacobian");
    NEED_OBS(h, "hm_initjacobian");
    HM_HIP(hipSetDevice(h->device));
    int rc = render_into(h, X, h->ref);
    if (rc) return rc;
    HM_HIP(hipStreamSynchronize(h->stream));
    h->X0.assign(X, X + 4 * h->N);
    h->have_ref = true;
    return HM_OK;
}

// sum the per-workgroup partials in index order
static int collect4(hm_ctx *h, double s[4])
{
    HM_HIP(hipMemcpyAsync(h->h_partial.data(), h->d_partial, (size_t)h->red_blocks * 4 * sizeof(double),
                          hipMemcpyDeviceToHost, h->stream));
    HM_HIP(stream_wait(h->stream));
    for (int k = 0; k < 4; k++) s[k] = 0.0;
    for (int b = 0; b < h->red_blocks; b++)
        for (int k = 0; k < 4; k++) s[k] += h->h_partial[(size_t)4 * b + k];
    return HM_OK;
}

extern "C" int hm_jz(hm_ctx_t h, const double *Xp, int masked, double *jz, double jzc[4])
{
    HM_ARG(h && Xp, "hm_jz: NULL argument");
    HM_JOIN(h);
    NEED_REF(h, "hm_jz");
    HM_HIP(hipSetDevice(h->device));
    int rc = render_into(h, Xp, h->P);
    if (rc) return rc;
    hipLaunchKernelGGL(k_jz, dim3(h->red_blocks), dim3(RED_NT), 0, h->stream, h->ref, h->P, obs_of(h, masked),
                       h->W * h->H, h->d_partial);
    double s[4];
    rc = collect4(h, s);
    if (rc) return rc;
    double c[4] = {s[0] / h->eps_Z, s[1] / h->eps_J, -s[2] / h->eps_J, s[3] / h->eps_M};
    if (jzc) for (int k = 0; k < 4; k++) jzc[k] = c[k];
    if (jz) *jz = ((c[0] + c[1]) + c[2]) + c[3];
    return HM_OK;
}

extern "C" int hm_j(hm_ctx_t h, const double *X, double deltaX, int i, int j, double *out)
{
    HM_ARG(h && X && out, "hm_j: NULL argument");
    HM_JOIN(h);
    HM_ARG(i >= 0 && i < 4 * h->N && j >= 0 && j < 4 * h->N, "hm_j: state index outside 0..%d", 4 * h->N - 1);
    NEED_REF(h, "hm_j");
    HM_HIP(hipSetDevice(h->device));
    std::vector<double> Xp(X, X + 4 * h->N), Xq(X, X + 4 * h->N);
    Xp[i] += deltaX;
    Xq[j] += deltaX;
    int rc = render_into(h, Xp.data(), h->P);
    if (rc) return rc;
    HM_HIP(hipStreamSynchronize(h->stream));     // d_X is reused by the second render
    rc = render_into(h, Xq.data(), h->Q);
    if (rc) return rc;
    hipLaunchKernelGGL(k_j, dim3(h->red_blocks), dim3(RED_NT), 0, h->stream, h->ref, h->P, h->Q, h->W * h->H,
                       h->d_partial);
    double s[4];
    rc = collect4(h, s);
    if (rc) return rc;
    *out = ((s[0] / h->eps_Z + s[1] / h->eps_J) + s[2] / h->eps_J) + s[3] / h->eps_M;
    return HM_OK;
}

// ---- the reference's multi-perturbation operators (cuda_multi.py:81-248, 721-845, 979-1129) ---------------------
// label palette, id images and per-label boxes of the renders involved; states: host copies of the 2 or 3 states
static int multi_setup(hm_ctx *h, const int32_t *labels, int n_labels, const double *const states[], int n_states)
{
    const size_t n = (size_t)h->W * h->H;
    if (!h->d_ids[0]) {
        for (int k = 0; k < 3; k++) HM_HIP(hm_malloc((void **)&h->d_ids[k], n * sizeof(int)));
        HM_HIP(hm_malloc((void **)&h->d_labels, (size_t)h->T * sizeof(int)));
    }
    if (n_labels > h->lcap) {
        if (h->d_lbox) (void)hipFree(h->d_lbox);
        if (h->d_lout) (void)hipFree(h->d_lout);
        h->d_lbox = nullptr; h->d_lout = nullptr; h->lcap = 0;
        HM_HIP(hm_malloc((void **)&h->d_lbox, (size_t)n_labels * sizeof(int4)));
        HM_HIP(hm_malloc((void **)&h->d_lout, (size_t)n_labels * 5 * sizeof(double)));
        h->lcap = n_labels;
    }
    std::vector<int4> box(n_labels, make_int4(1, 0, 1, 0));
    for (int t = 0; t < h->T; t++) {
        const int lab = labels[t];
        if (lab < 0) continue;
        HM_ARG(lab < n_labels && lab < 65535, "multi-perturbation label %d of triangle %d outside 0..%d", lab, t, n_labels - 1);
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (int sidx = 0; sidx < n_states; sidx++)
            for (int k = 0; k < 3; k++) {
                const int v = h->tri[3 * t + k];
                const double x = states[sidx][2 * v], y = states[sidx][2 * v + 1];
                x0 = std::min(x0, x); x1 = std::max(x1, x); y0 = std::min(y0, y); y1 = std::max(y1, y);
            }
        if (!(x0 <= x1 && y0 <= y1)) continue;                              // non-finite state: nothing is drawn
        // pixel (c, r) is drawn iff its centre (c + .5, r + .5) is inside: one pixel of margin each way
        const int c0 = (int)std::max(0.0, std::floor(std::min(x0, (double)h->W)) - 1.0);
        const int c1 = (int)std::min((double)h->W - 1.0, std::ceil(std::max(x1, -1.0)) + 1.0);
        const int r0 = (int)std::max(0.0, std::floor(std::min(y0, (double)h->H)) - 1.0);
        const int r1 = (int)std::min((double)h->H - 1.0, std::ceil(std::max(y1, -1.0)) + 1.0);
        int4 &b = box[lab];
        if (b.x > b.y) b = make_int4(c0, c1, r0, r1);
        else b = make_int4(std::min(b.x, c0), std::max(b.y, c1), std::min(b.z, r0), std::max(b.w, r1));
    }
    HM_HIP(hipMemcpyAsync(h->d_labels, labels, (size_t)h->T * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HM_HIP(hipMemcpyAsync(h->d_lbox, box.data(), (size_t)n_labels * sizeof(int4), hipMemcpyHostToDevice, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));                                   // box is a local
    return HM_OK;
}

// render state X (host) into target t with its id image (MODE 1), or the id image only (MODE 2)
static int render_ids(hm_ctx *h, const double *X, Targets t, int *ids, bool ids_only)
{
    HM_HIP(hipMemcpyAsync(h->d_X, X, (size_t)4 * h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    Mesh m = {h->W, h->H, h->N, h->T, h->d_tri, h->d_uv, h->d_tex};
    hipLaunchKernelGGL(k_setup_all, dim3(hm_cdiv(h->T, 64)), dim3(64), 0, h->stream, m, h->d_X, h->d_setup);
    const dim3 grid(hm_cdiv(h->W, EKF_TILE), hm_cdiv(h->H, EKF_TILE)), block(EKF_TILE, EKF_TILE);
    if (ids_only) hipLaunchKernelGGL((k_render<2>), grid, block, 0, h->stream, m, h->d_X, h->d_setup, t, (const int *)h->d_labels, ids);
    else hipLaunchKernelGGL((k_render<1>), grid, block, 0, h->stream, m, h->d_X, h->d_setup, t, (const int *)h->d_labels, ids);
    HM_HIP(hipGetLastError());
    HM_HIP(hipStreamSynchronize(h->stream));                                   // d_X is reused by the next render
    return HM_OK;
}

extern "C" int hm_jz_multi(hm_ctx_t h, const double *Xp, int masked, const int32_t *labels, int n_labels, double *hz,
                           double *hzc)
{
    HM_ARG(h && Xp && labels && hz && n_labels >= 1, "hm_jz_multi: bad argument");
    HM_JOIN(h);
    NEED_REF(h, "hm_jz_multi");
    HM_HIP(hipSetDevice(h->device));
    const double *states[2] = {h->X0.data(), Xp};
    int rc = multi_setup(h, labels, n_labels, states, 2);
    if (rc) return rc;
    rc = render_ids(h, h->X0.data(), h->ref, h->d_ids[0], true);               // the reference render's ids (its targets stand)
    if (rc == HM_OK) rc = render_ids(h, Xp, h->P, h->d_ids[1], false);
    if (rc) return rc;
    MultiArgs a = {h->ref, h->P, h->P, h->d_ids[0], h->d_ids[1], h->d_ids[1], obs_of(h, masked), h->W, h->d_lbox, h->d_lout};
    hipLaunchKernelGGL(k_jz_multi, dim3(n_labels), dim3(RED_NT), 0, h->stream, a);
    HM_HIP(hipGetLastError());
    h->h_lout.resize((size_t)n_labels * 5);
    HM_HIP(hipMemcpyAsync(h->h_lout.data(), h->d_lout, (size_t)n_labels * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    for (int l = 0; l < n_labels; l++) {
        const double *s = h->h_lout.data() + 4 * l;
        const double c[4] = {s[0] / h->eps_Z, s[1] / h->eps_J, -s[2] / h->eps_J, s[3] / h->eps_M};
        hz[l] = ((c[0] + c[1]) + c[2]) + c[3];
        if (hzc) for (int k = 0; k < 4; k++) hzc[4 * l + k] = c[k];
    }
    return HM_OK;
}

extern "C" int hm_j_multi(hm_ctx_t h, const double *X, double deltaX, int n_pairs, const int32_t *ee, const int32_t *labels,
                          int n_labels, double *hsum, double *nz, double *hcomp)
{
    HM_ARG(h && X && ee && labels && hsum && nz && n_pairs >= 1 && n_labels >= 1, "hm_j_multi: bad argument");
    HM_JOIN(h);
    NEED_REF(h, "hm_j_multi");
    HM_HIP(hipSetDevice(h->device));
    const int n4 = 4 * h->N;
    std::vector<double> Xp(X, X + n4), Xq(X, X + n4);
    for (int k = 0; k < n_pairs; k++) {
        HM_ARG(ee[2 * k] >= 0 && ee[2 * k] < n4 && ee[2 * k + 1] >= 0 && ee[2 * k + 1] < n4, "hm_j_multi: state index outside 0..%d", n4 - 1);
        Xp[ee[2 * k]] += deltaX;                                            // all first indices at once (cuda_multi.py:987-990)
        Xq[ee[2 * k + 1]] += deltaX;                                        // all second indices at once (:1004-1007)
    }
    const double *states[3] = {h->X0.data(), Xp.data(), Xq.data()};
    int rc = multi_setup(h, labels, n_labels, states, 3);
    if (rc) return rc;
    rc = render_ids(h, h->X0.data(), h->ref, h->d_ids[0], true);
    if (rc == HM_OK) rc = render_ids(h, Xp.data(), h->P, h->d_ids[1], false);
    if (rc == HM_OK) rc = render_ids(h, Xq.data(), h->Q, h->d_ids[2], false);
    if (rc) return rc;
    MultiArgs a = {h->ref, h->P, h->Q, h->d_ids[0], h->d_ids[1], h->d_ids[2], obs_of(h, 0), h->W, h->d_lbox, h->d_lout};
    hipLaunchKernelGGL(k_j_multi, dim3(n_labels), dim3(RED_NT), 0, h->stream, a);
    HM_HIP(hipGetLastError());
    h->h_lout.resize((size_t)n_labels * 5);
    HM_HIP(hipMemcpyAsync(h->h_lout.data(), h->d_lout, (size_t)n_labels * 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    for (int l = 0; l < n_labels; l++) {
        const double *s = h->h_lout.data() + 5 * l;
        const double c[4] = {s[0] / h->eps_Z, s[1] / h->eps_J, s[2] / h->eps_J, s[3] / h->eps_M};
        hsum[l] = ((c[0] + c[1]) + c[2]) + c[3];
        nz[l] = s[4];
        if (hcomp) for (int k = 0; k < 4; k++) hcomp[4 * l + k] = c[k];
    }
    return HM_OK;
}

extern "C" int hm_error(hm_ctx_t h, const double *X, int masked, double err[4], float *fx, float *fy)
{
    HM_ARG(h && X && err, "hm_error: NULL argument");
    HM_JOIN(h);
    NEED_TEX(h, "hm_error");
    NEED_OBS(h, "hm_error");
    HM_HIP(hipSetDevice(h->device));
    // the launch and the order of additions of the update loop (k_render_iter, strip partials in the fixed two-level
    // order): the error of a state is the same number whether an iteration of hm_update_run reports it or this call
    HM_HIP(hipMemcpyAsync(h->d_X, X, (size_t)4 * h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = render_iter(h, h->d_X, h->P, true, masked, false, 2.0);
    if (rc) return rc;
    const int n = h->W * h->H;
    if (fx) HM_HIP(hipMemcpyAsync(fx, h->P.fx, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (fy) HM_HIP(hipMemcpyAsync(fy, h->P.fy, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipMemcpyAsync(h->h_tpart.data(), h->d_tpart, (size_t)render_strips(h) * RI_NV * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(stream_wait(h->stream));
    double s6[RI_NV];
    hm_tile_partial_sums(h->h_tpart.data(), render_strips(h), s6);
    for (int k = 0; k < 4; k++) err[k] = s6[k];
    return HM_OK;
}

// render X as the reference, run the fused perturb-and-reduce kernel, unpack into d_H / d_Hz / d_Hzc
// the measurement at the device-resident state dX, whose render is (ref_ready) or is to be put in h->ref
// scatter: the job sums go to the dense HTH / Hz / Hzc (hm_measure); the update loop forms its system from the job
// sums directly (k_solve_prep) and leaves that launch out
static int measure_dev(hm_ctx *h, const double *dX, bool ref_ready, double deltaX, int masked, bool regions_ready = false,
                       bool scatter = true)
{
    if (!ref_ready) {                             // the reference render, and the star regions with it when they are wanted too
        int rc = render_iter(h, dX, h->ref, false, masked, !regions_ready, deltaX);
        if (rc) return rc;
        regions_ready = true;
    }
    MeasureArgs a;
    measure_args(h, dX, deltaX, masked, a);
    if (!regions_ready) hipLaunchKernelGGL(k_star_regions, dim3(h->N), dim3(REGION_NT), 0, h->stream, a, h->d_area);
    hipLaunchKernelGGL(k_measure_vertex, dim3(h->N, h->vsplit), dim3(MEAS_NT), 0, h->stream, a,
                       (const TriSetup *)h->d_cfgs, (const int4 *)h->d_ubox, (const unsigned *)h->d_tmask);
    if (h->E > 0) hipLaunchKernelGGL(k_measure_edge, dim3(h->E, h->esplit), dim3(MEAS_NT), 0, h->stream, a);
    if (scatter) {
        ScatterArgs s = {h->d_out, h->d_edges, h->N, h->E, h->vsplit, h->esplit, h->eps_Z, h->eps_J, h->eps_M, deltaX, h->d_HTH, h->d_Hz, h->d_Hzc};
        hipLaunchKernelGGL(k_hth_scatter, dim3(hm_cdiv(h->njobs, 4)), dim3(256), 0, h->stream, s);
    }
    HM_HIP(hipGetLastError());
    return HM_OK;
}

static void measure_args(hm_ctx *h, const double *dX, double deltaX, int masked, MeasureArgs &a)
{
    a.m = Mesh{h->W, h->H, h->N, h->T, h->d_tri, h->d_uv, h->d_tex};
    a.topo = StarTopo{h->d_star_off, h->d_star_tri, h->d_edges, h->E};
    a.ref = h->ref;
    a.obs = obs_of(h, masked);
    a.X = dX;
    a.delta = deltaX;
    a.out = h->d_out;
    a.pool = h->pool;
    a.vsplit = h->vsplit;
    a.esplit = h->esplit;
    a.iZ = 1.0 / h->eps_Z; a.iJ = 1.0 / h->eps_J; a.iM = 1.0 / h->eps_M;
    a.cfgs = h->d_cfgs;
    a.ubox = h->d_ubox;
    a.tmask = h->d_tmask;
    a.tlist = h->d_tlist;
    a.tcount = h->d_tcount;
    a.tmask_stride = TMASK_STRIDE;
}

static int measure_on_device(hm_ctx *h, const double *X, double deltaX, int masked, bool scatter = true)
{
    HM_HIP(hipMemcpyAsync(h->d_X, X, (size_t)4 * h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->X0.assign(X, X + 4 * h->N);
    h->have_ref = true;
    return measure_dev(h, h->d_X, false, deltaX, masked, false, scatter);
}

extern "C" int hm_measure(hm_ctx_t h, const double *X, double deltaX, int masked, double *Hz, double *Hzc, double *HTH)
{
    HM_ARG(h && X && Hz && HTH, "hm_measure: NULL argument");
    HM_JOIN(h);
    HM_ARG(deltaX > 0, "hm_measure: deltaX must be positive");
    NEED_TEX(h, "hm_measure");
    NEED_OBS(h, "hm_measure");
    HM_HIP(hipSetDevice(h->device));
    const size_t n4 = (size_t)4 * h->N;
    int *ovf = h->pin_scratch;
    for (int attempt = 0;; attempt++) {              // (once more after the pool has been grown)
        int rc = measure_on_device(h, X, deltaX, masked);
        if (rc) return rc;
        // the flag lands in the handle's pinned block, not in this frame: an error return below must not leave a
        // copy in flight towards a dead stack slot
        *ovf = 0;
        HM_HIP(hipMemcpyAsync(ovf, h->pool.overflow, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HM_HIP(hipStreamSynchronize(h->stream));
        if (!*ovf) break;
        if (attempt) { hm_set_error("hm_measure: the star regions do not fit the difference-image pool"); return HM_ERR_STATE; }
        rc = pool_grow(h, "hm_measure");
        if (rc) return rc;
    }
    HM_HIP(hipMemcpyAsync(Hz, h->d_Hz, n4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (Hzc) HM_HIP(hipMemcpyAsync(Hzc, h->d_Hzc, n4 * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipMemcpyAsync(HTH, h->d_HTH, n4 * n4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    return HM_OK;
}

// ---- the dense part of the update on the device ----------------------------------------------------
// Layout of a factor buffer d_Af[s]: n4 matrix rows, padding up to a multiple of 32 rows, then one
// 32-row block whose first row carries the right-hand side (see dense_kernels.h).
static int aug_rows(int n) { return hm_cdiv(n, DNB) * DNB + DNB; }

// Cholesky of the n x n matrix in the working copy A (destroyed) into L / Lt, and the inverse of the
// factor into T (M: n x n scratch) -- both from the same launches; with_rhs: the right-hand-side rows
// below the matrix go through the elimination too (dense_kernels.h)
// first_done: Lt[0] is there already (k_assemble)
static void chol_factor(hm_ctx *h, double *A, double *L, double *Lt, double *T, double *M, int n, bool with_rhs,
                        bool first_done = false, hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    const int nb = hm_cdiv(n, DNB);
    const int nrows = with_rhs ? aug_rows(n) : n;
    const int nbr = hm_cdiv(nrows, DNB);
    if (h->chol_flow) {
        // one persistent launch: the block operations below as tasks that hand their results over through memory
        // (chol_flow_kernels.h); the same bits as the launch-per-step form.  first_done: the caller's assembly pass
        // (k_assemble_flow) has pre-filled the outputs
        FlowArgs a = {A, L, Lt, T, h->d_flowP, n, nrows, nb, nbr, h->d_flowctl, h->flow_stall};
        if (!first_done) hipLaunchKernelGGL(k_flow_fill, dim3(nrows + FLOW_FILL_WGS), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_chol_flow, dim3(h->flow_wgs), dim3(FLOW_NT), 0, st, a);
        return;
    }
    if (!first_done) hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(64), 0, st, A, Lt, n);
    for (int k = 0; k < nb; k++) {
        const int mr = nbr - k - 1, mc = std::max(nb - k - 1, 1);
        // grid: mr block rows below the diagonal (update of A in columns < mc, of the inverse in the k+1
        // columns after them) and one more row that finishes row k of the inverse
        hipLaunchKernelGGL(k_chol_step, dim3((mc + k + 1) * (mr + 1)), dim3(256), 0, st, A, L, Lt, T, M, n, nrows, nb, k, mc, mc + k + 1, mr + 1);
    }
}

// SPD inverse from the inverse T = L^-1 of the factor: inv = T^T T
static void chol_inverse(hm_ctx *h, int n, const double *T, double *out, hipStream_t st = nullptr)
{
    const int nb = hm_cdiv(n, DNB);
    hipLaunchKernelGGL(k_ttt, dim3(nb, nb), dim3(256), 0, st ? st : h->stream, T, n, out);
}

// the time-out word of the persistent factorisation launches since the last hm_update_begin (stream must be idle)
static int flow_status(hm_ctx *h, const char *who)
{
    if (!h->chol_flow) return HM_OK;
    unsigned ctl[2] = {0, 0};
    HM_HIP(hipMemcpyAsync(ctl, h->d_flowctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream));      // (not the null stream: it
    HM_HIP(hipStreamSynchronize(h->stream));                                                         // waits for blocking streams)
    if (ctl[1]) { hm_set_error("%s: the factorisation launch gave up waiting for a block (chol_flow time-out)", who); return HM_ERR_HIP; }
    return HM_OK;
}

// one iteration's worth of launches of the update: system assembly, factorisation, solve.
// d_X holds the iterate the measurement was taken at; returns the step (d_step).
static double *solve_step(hm_ctx *h, int slot, double deltaX)
{
    const int n4 = 4 * h->N;
    double *A = h->d_Awork;
    // A = invW0 + H; the right-hand side Hz - H (X0 - X) as the first row of the block below the
    // matrix; the rows in between and the rest of that block zero
    const int rhs_index = hm_cdiv(n4, DNB) * DNB;
    double *rhs_row = A + (size_t)rhs_index * n4;
    {   // one pass from the job sums of the measurement: A, the right-hand side, Hz / Hzc, and the pre-fill of what
        // the persistent factorisation launch produces (harmless for the launch-per-step form, which overwrites it)
        const int nb = hm_cdiv(n4, DNB), nrows = aug_rows(n4);
        PrepArgs p;
        p.out = h->d_out; p.N = h->N; p.vsplit = h->vsplit; p.esplit = h->esplit;
        p.eZ = h->eps_Z; p.eJ = h->eps_J; p.eM = h->eps_M; p.d = deltaX;
        p.nb_off = h->d_nb_off; p.nb_u = h->d_nb_u; p.nb_e = h->d_nb_e;
        p.invW0 = h->d_invW0; p.X0 = h->d_X0; p.X = h->d_X;
        p.A = A; p.Hz = h->d_Hz; p.Hzc = h->d_Hzc; p.n = n4; p.rhs_row = rhs_index;
        p.f = FlowArgs{A, h->d_Af[slot], h->d_Lt[slot], h->d_T[slot], h->d_flowP, n4, nrows, nb, hm_cdiv(nrows, DNB), h->d_flowctl, 0};
        hipLaunchKernelGGL(k_solve_prep, dim3(nrows + FLOW_FILL_WGS), dim3(256), 0, h->stream, p);
    }
    if (h->d_Wres == h->d_Wtmp) h->d_Wres = nullptr;          // a covariance predicted since hm_update_begin is lost
    chol_factor(h, A, h->d_Af[slot], h->d_Lt[slot], h->d_T[slot], h->d_Wtmp, n4, true, h->chol_flow != 0);
    // y = L^-1 b came out of the factorisation as the extra row and T = L^-1 with it (d_Wtmp was the
    // scratch: it is free between hm_update_begin and the next hm_cov_predict); x = T^T y.  T stays in the
    // slot for hm_update_cov (inv = T^T T).
    const double *yrow = h->d_Af[slot] + (rhs_row - A);
    hipLaunchKernelGGL(k_tvec, dim3(hm_cdiv(n4, TV_COLS)), dim3(TV_COLS * TV_ROWS), 0, h->stream, h->d_T[slot], n4, yrow, h->d_step, h->d_X0,
                       h->d_Xn);                              // also d_Xn = X0 + step
    return h->d_step;
}

// the covariance half of hm_update_begin: the prior into d_Wprior, its factor, inv(W) into d_invW0
static int prior_inverse(hm_ctx *h, const double *W_prior, hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    const int n4 = 4 * h->N;
    // the prior stays in d_Wprior: it is the covariance to keep when no iterate is accepted
    const size_t nnb = (size_t)n4 * n4 * sizeof(double);
    if (W_prior)
        HM_HIP(hipMemcpyAsync(h->d_Wprior, W_prior, nnb, hipMemcpyHostToDevice, st));
    else if (h->d_Wres != h->d_Wprior)
        HM_HIP(hipMemcpyAsync(h->d_Wprior, h->d_Wres, nnb, hipMemcpyDeviceToDevice, st));
    // the launch-per-step factorisation destroys its input: it gets a copy; the persistent launch only reads it
    if (!h->chol_flow) HM_HIP(hipMemcpyAsync(h->d_Awork, h->d_Wprior, nnb, hipMemcpyDeviceToDevice, st));
    h->d_Wres = h->d_Wprior;                     // d_Wtmp is scratch from here on
    HM_HIP(hipMemsetAsync(h->d_flowctl, 0, 4 * sizeof(unsigned), st));      // a new sequence of factorisations
    chol_factor(h, h->chol_flow ? h->d_Wprior : h->d_Awork, h->d_Af[0], h->d_Lt[0], h->d_Wtmp, h->d_invW0, n4, false, false, st);    // T in d_Wtmp
    chol_inverse(h, n4, h->d_Wtmp, h->d_invW0, st);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

// Start the covariance half of the next hm_update_begin / hm_update_run(h, NULL, ...) now: the
// factorisation and inversion of the covariance resident on the device are queued and the call
// returns.  They need the predicted covariance only, not the predicted state, so a caller whose
// state prediction runs on the host (hm_ms_newton) overlaps the two.
extern "C" int hm_update_prefactor(hm_ctx_t h)
{
    HM_ARG(h != nullptr, "hm_update_prefactor: NULL handle");
    HM_JOIN(h);
    if (!h->d_Wres) { hm_set_error("hm_update_prefactor: no covariance resident on the device"); return HM_ERR_STATE; }
    HM_HIP(hipSetDevice(h->device));
    h->pq_valid = false;
    // (launched one by one: replaying this series as a hipGraph saved 0.08 ms of host time per frame, but
    // graph replays proved unreliable next to allocations by the caller -- see hm_brox_tune "graph")
    // With the factorisation as one persistent launch this is a copy, a fill and three launches: queued right here
    // (a helper thread started per call had them reach the device ~0.2 ms later -- thread start-up and the first
    // runtime calls of a new thread -- and the first measurement of the frame waited for them: kernel trace of the
    // bench, tools/frame_gap_timeline.py).
    if (h->chol_flow) {
        const int rc = prior_inverse(h, nullptr);
        if (rc == HM_OK) {
            h->prefactored = true;
            h->upd_open = false;                  // the factor slots are being reused
        }
        return rc;
    }
    // The launch-per-step form: queueing its ~30 launches takes the host about as long as the state prediction the
    // caller does next (hm_ms_newton): a helper thread does it, and whatever is called next on this handle joins it.
    h->worker_active = true;
    h->worker = std::thread([h]() {
        int rc = hipSetDevice(h->device) == hipSuccess ? HM_OK : HM_ERR_HIP;
        if (rc == HM_OK) rc = prior_inverse(h, nullptr);
        if (rc == HM_OK) {
            h->prefactored = true;
            h->upd_open = false;                  // the factor slots are being reused
        } else {
            snprintf(h->worker_err, sizeof h->worker_err, "hm_update_prefactor: %s", hm_last_error());
        }
        h->worker_rc = rc;
    });
    return HM_OK;
}

static int update_begin(hm_ctx *h, const double *W_prior, const double *X0);
extern "C" int hm_update_begin(hm_ctx_t h, const double *W_prior, const double *X0)
{
    HM_ARG(h && X0, "hm_update_begin: NULL argument");
    HM_JOIN(h);
    h->chain_pending = false;
    return update_begin(h, W_prior, X0);
}
// X0 == NULL: the prior mean is being left in d_X0 by hm_chain_project's kernel on the second stream
static int update_begin(hm_ctx *h, const double *W_prior, const double *X0)
{
    if (!W_prior && !h->d_Wres) {
        hm_set_error("hm_update_begin: no prior given and none resident on the device");
        return HM_ERR_STATE;
    }
    HM_HIP(hipSetDevice(h->device));
    const int n4 = 4 * h->N;
    h->pq_valid = false;                         // (a prediction queued ahead that nobody took: d_Wres is still the posterior)
    const bool taken = h->prefactored && !W_prior && h->d_Wres == h->d_Wprior;
    if (!taken) {
        int rc = ctx_join(h);                    // (the tail of the last update is rewriting what prior_inverse rewrites)
        if (rc) return rc;
    }
    if (!taken) {
        int rc = prior_inverse(h, W_prior);
        if (rc) return rc;
    }
    h->prefactored = false;
    if (X0) {
        h->upd_X0.assign(X0, X0 + n4);
        HM_HIP(hipMemcpyAsync(h->d_X0, h->upd_X0.data(), (size_t)n4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
        HM_HIP(hipStreamWaitEvent(h->stream, h->ev_pm, 0));
    }
    h->upd_last = h->upd_prev = -1;
    h->upd_open = true;
    return HM_OK;
}

extern "C" int hm_update_step(hm_ctx_t h, const double *X, double deltaX, int masked, double *step, double *Hzc,
                              double err[4])
{
    HM_ARG(h && X && step, "hm_update_step: NULL argument");
    HM_JOIN(h);
    HM_ARG(deltaX > 0, "hm_update_step: deltaX must be positive");
    if (!h->upd_open) { hm_set_error("hm_update_step: hm_update_begin has not been called"); return HM_ERR_STATE; }
    NEED_TEX(h, "hm_update_step");
    NEED_OBS(h, "hm_update_step");
    HM_HIP(hipSetDevice(h->device));
    const int n4 = 4 * h->N;
    int rc = HM_OK;
    const int slot = h->upd_last == 0 ? 1 : 0;
    for (int attempt = 0;; attempt++) {              // (once more after the pool of difference images has been grown)
    rc = measure_on_device(h, X, deltaX, masked, false);   // leaves X in d_X; the job sums go straight into the system
    if (rc) return rc;
    double *rhs_row = solve_step(h, slot, deltaX);
    HM_HIP(hipGetLastError());
    HM_HIP(hipMemcpyAsync(step, rhs_row, (size_t)n4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (Hzc) HM_HIP(hipMemcpyAsync(Hzc, h->d_Hzc, (size_t)n4 * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    int *ovf = h->pin_scratch;                               // pinned, lives with the handle (see hm_measure)
    *ovf = 0;
    HM_HIP(hipMemcpyAsync(ovf, h->pool.overflow, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (err) {
        // Renderer.error of the new iterate X0 + step, without another host round trip: the launch and the order of
        // additions hm_update_run uses
        rc = render_iter(h, h->d_Xn, h->P, true, masked, false, deltaX);
        if (rc == HM_OK) {
            hipError_t e = hipMemcpyAsync(h->h_tpart.data(), h->d_tpart, (size_t)render_strips(h) * RI_NV * sizeof(double), hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = stream_wait(h->stream);
            if (e != hipSuccess) { hm_set_error("hm_update_step: %s", hipGetErrorString(e)); rc = HM_ERR_HIP; }
            else { double s6[RI_NV]; hm_tile_partial_sums(h->h_tpart.data(), render_strips(h), s6); for (int k = 0; k < 4; k++) err[k] = s6[k]; }
        }
        if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }   // nothing of this call stays in flight
    } else {
        HM_HIP(hipStreamSynchronize(h->stream));
    }
    if (!*ovf) break;
    if (attempt) { hm_set_error("hm_update_step: the star regions do not fit the difference-image pool"); return HM_ERR_STATE; }
    rc = pool_grow(h, "hm_update_step");
    if (rc) return rc;
    }
    rc = flow_status(h, "hm_update_step");
    if (rc) return rc;
    h->upd_prev = h->upd_last;
    h->upd_last = slot;
    return HM_OK;
}

static int update_cov_on(hm_ctx *h, int which, double *W_out, hipStream_t st);
extern "C" int hm_update_cov(hm_ctx_t h, int which, double *W_out)
{
    HM_ARG(h != nullptr, "hm_update_cov: NULL handle");
    HM_JOIN(h);
    return update_cov_on(h, which, W_out, h->stream);
}
static int update_cov_on(hm_ctx *h, int which, double *W_out, hipStream_t st)
{
    HM_ARG(which >= -1 && which <= 1, "hm_update_cov: which must be 0 (last step), 1 (the step before) or -1 (the prior)");
    if (!h->upd_open) { hm_set_error("hm_update_cov: hm_update_begin has not been called"); return HM_ERR_STATE; }
    HM_HIP(hipSetDevice(h->device));
    const int n4 = 4 * h->N;
    if (which < 0) {
        h->d_Wres = h->d_Wprior;
    } else {
        const int slot = which == 0 ? h->upd_last : h->upd_prev;
        if (slot < 0) { hm_set_error("hm_update_cov: no such step"); return HM_ERR_STATE; }
        {   // T of this slot is there since its solve: inv = T^T T
            const int nb = hm_cdiv(n4, DNB);
            hipLaunchKernelGGL(k_ttt, dim3(nb, nb), dim3(256), 0, st, h->d_T[slot], n4, h->d_H);
        }
        HM_HIP(hipGetLastError());
        h->d_Wres = h->d_H;
    }
    if (W_out) {
        HM_HIP(hipMemcpyAsync(W_out, h->d_Wres, (size_t)n4 * n4 * sizeof(double), hipMemcpyDeviceToHost, st));
        HM_HIP(stream_wait(st));
    }
    return HM_OK;
}

// KalmanFilter.projectmask (kalman.py:724-742): vertices more than 1 px outside the object are
// walked back onto its outline, their displacement is added to their velocity.  y_m: a W*H host
// mask (object where > 0), or NULL for the mask of the observation in place.  X (4N) is updated in
// place; *moved (may be NULL) receives the number of vertices that were outside.
extern "C" int hm_project_mask(hm_ctx_t h, const uint8_t *y_m, double *X, int *moved)
{
    HM_ARG(h && X, "hm_project_mask: NULL argument");
    HM_JOIN_LAZY(h);
    if (!y_m) { NEED_OBS(h, "hm_project_mask"); }
    HM_HIP(hipSetDevice(h->device));
    // On the handle's second stream, with buffers of its own, and without joining the helper thread: the projection
    // needs the mask and the predicted state only, and in a frame of the filter it comes while the covariance half of
    // the update (hm_update_prefactor: ~0.25 ms of launches) is still being queued and run on the first stream --
    // behind those it would have its caller wait for them.
    int rc = ensure_stream2(h);
    if (rc) return rc;
    rc = project_buffers(h);
    if (rc) return rc;
    hipStream_t s = h->stream2;
    const size_t n = (size_t)h->W * h->H;
    const size_t n4 = (size_t)4 * h->N, xb = n4 * sizeof(double);
    Outline o = {h->d_outline, h->d_outline_cnt, (int)n, h->d_pm_flag};
    if (!y_m) {
        // The mask of the resident observation: its outline was queued when the observation was set (or is now, the
        // first time); the state goes through page-locked memory both ways, the result as a block the host takes when
        // it is whole (host_block.h) -- one launch and no copy operations on the way (0.15 -> ~0.03 ms per frame of the streaming pipeline).
        if (!h->outline_ready) {
            rc = queue_outline(h, h->o_ym);
            if (rc) return rc;
            h->outline_ready = true;
        }
        memcpy(h->pin_pm, X, xb);
        ProjArgs a = {h->d_pm_pruned, h->W, h->H, h->N, o, nullptr};
        hipLaunchKernelGGL(k_project_mask_host, dim3(h->N), dim3(PROJ_NT), 0, s, a, (const double *)h->pin_pm, h->pin_pm + n4, (double *)nullptr,
                           (double *)nullptr, h->d_pm_done, (double)(++h->pm_ticket), h->result_delay);
        HM_HIP(hipGetLastError());
        rc = hb_wait(s, h->pin_pm + n4, 0, n4 + 1, hb_stamp(h->pm_ticket), h->pmv.data(), "hm_project_mask");
        if (rc) return rc;
        const int nmoved = (int)h->pmv[n4];
        if (nmoved) memcpy(X, h->pmv.data(), xb);
        if (moved) *moved = nmoved;
        return HM_OK;
    }
    h->outline_ready = false;                    // the buffers are about to hold the outline of the caller's mask
    if (!h->d_pm_mask) HM_HIP(hm_malloc((void **)&h->d_pm_mask, n));
    HM_HIP(hipMemcpyAsync(h->d_pm_mask, y_m, n, hipMemcpyHostToDevice, s));
    const uint8_t *mask = h->d_pm_mask;
    rc = queue_outline(h, mask);
    if (rc) return rc;
    HM_HIP(hipMemcpyAsync(h->d_pm_X, X, xb, hipMemcpyHostToDevice, s));
    ProjArgs a = {h->d_pm_pruned, h->W, h->H, h->N, o, h->d_pm_X};
    hipLaunchKernelGGL(k_project_mask, dim3(h->N), dim3(PROJ_NT), 0, s, a);
    HM_HIP(hipGetLastError());
    int cnt[4] = {0, 0, 0, 0};
    HM_HIP(hipMemcpyAsync(cnt, h->d_outline_cnt, sizeof(cnt), hipMemcpyDeviceToHost, s));
    HM_HIP(hipMemcpyAsync(X, h->d_pm_X, xb, hipMemcpyDeviceToHost, s));
    HM_HIP(stream_wait(s));
    if (moved) *moved = cnt[2];
    return HM_OK;
}

// The reference's contour pruning of a mask (imgproc.py:198-228: the largest object and its holes of area >= 40) as
// hm_project_mask applies it -- fine-grained operator for the parity tests.  y_m: W*H host mask (object where > 0);
// out: W*H, 1 where the pruned object is.
extern "C" int hm_prune_mask(hm_ctx_t h, const uint8_t *y_m, uint8_t *out)
{
    HM_ARG(h && y_m && out, "hm_prune_mask: NULL argument");
    HM_JOIN_LAZY(h);
    HM_HIP(hipSetDevice(h->device));
    int rc = ensure_stream2(h);
    if (rc) return rc;
    rc = project_buffers(h);
    if (rc) return rc;
    const size_t n = (size_t)h->W * h->H;
    h->outline_ready = false;                    // the buffers are about to hold the outline of the caller's mask
    if (!h->d_pm_mask) HM_HIP(hm_malloc((void **)&h->d_pm_mask, n));
    HM_HIP(hipMemcpyAsync(h->d_pm_mask, y_m, n, hipMemcpyHostToDevice, h->stream2));
    rc = queue_outline(h, h->d_pm_mask);
    if (rc) return rc;
    HM_HIP(hipMemcpyAsync(out, h->d_pm_pruned, n, hipMemcpyDeviceToHost, h->stream2));
    HM_HIP(hipStreamSynchronize(h->stream2));
    return HM_OK;
}

// The covariance resident on the device (the result of the last hm_cov_predict, hm_update_cov or
// hm_update_run) copied to the host.
extern "C" int hm_cov_fetch(hm_ctx_t h, double *W_out)
{
    HM_ARG(h && W_out, "hm_cov_fetch: NULL argument");
    HM_JOIN(h);
    if (!h->d_Wres) { hm_set_error("hm_cov_fetch: no covariance resident on the device"); return HM_ERR_STATE; }
    HM_HIP(hipSetDevice(h->device));
    const size_t n4 = (size_t)4 * h->N;
    HM_HIP(hipMemcpyAsync(W_out, h->d_Wres, n4 * n4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    return HM_OK;
}

// IteratedKalmanFilter.update (kalman.py:774-831) in one call: hm_update_begin, up to max_iter
// hm_update_step's with the reference's acceptance logic between them, hm_update_cov of the state
// that is kept.  Per iteration the host sees one small result block (step, the four error sums)
// that the last kernel writes into pinned memory; the iterate itself never leaves the device, and
// the render that gave an iterate's error is the reference render of the next measurement.
extern "C" int hm_update_arm_newton(hm_ctx_t h, void *worker, int n_bars, const int32_t *bars, const double *l0, double kappa,
                                    double M, double dt, int maxiter, double tol)
{
    HM_ARG(h && worker && n_bars >= 0 && bars && l0, "hm_update_arm_newton: bad argument");
    HM_JOIN_LAZY(h);                               // (the tail of the last update still reads what this call replaces)
    h->pn_worker = worker;
    h->pn_bars.assign(bars, bars + 2 * (size_t)n_bars);
    h->pn_l0.assign(l0, l0 + n_bars);
    h->pn_par[0] = kappa; h->pn_par[1] = M; h->pn_par[2] = dt; h->pn_par[3] = tol;
    h->pn_maxiter = maxiter;
    h->pn_armed = true;
    return HM_OK;
}

// With hm_update_arm_newton armed as well: the next hm_update_run also queues the covariance half of the next frame's
// prediction (queue_predict_ahead above; eps_F as for hm_cov_predict) behind the covariance of the state it keeps.
// hm_update_cov is not available after such a run (the factor slots are reused); hm_cov_fetch still returns the
// posterior until hm_predict_take.
extern "C" int hm_update_arm_cov(hm_ctx_t h, double eps_F)
{
    HM_ARG(h != nullptr, "hm_update_arm_cov: NULL handle");
    h->pq_armed = true;
    h->pq_eps_F = eps_F;
    return HM_OK;
}

static int queue_predict_ahead(hm_ctx *h, const double *X, int n_bars, const int32_t *bars, const double *l0, double kappa,
                               double M, double dt, double eps_F, hipStream_t st);
static int prepare_mask(hm_ctx *h, const uint8_t *d_y_m);

// one-shot: when the next hm_update_run on h has its final state it also queues hm_prepare_mask(h, d_y_m) -- the contour
// pruning and outline of the NEXT frame's mask run beside the state prediction and the update's tail instead of at the
// start of the next frame, where the projection would wait for them
extern "C" int hm_update_arm_mask(hm_ctx_t h, const uint8_t *d_y_m)
{
    HM_ARG(h != nullptr, "hm_update_arm_mask: NULL handle");
    h->armed_mask = d_y_m;
    return HM_OK;
}

// Hz components (4N x 4) and gains (3 x 4N) of the last hm_update_run that was called with Hzc = gains = NULL (such a
// call does not wait for the kernels that form them: the caller gets on with the next frame); either may be NULL.
// Available until the next hm_update_run on h; zeros for an update without iterations.
extern "C" int hm_update_tail(hm_ctx_t h, double *Hzc, double *gains)
{
    HM_ARG(h != nullptr, "hm_update_tail: NULL handle");
    HM_JOIN_LAZY(h);                               // (the helper thread may still be queueing the kernels that form them)
    const size_t n4 = (size_t)4 * h->N;
    if (h->tail_pending) {
        HM_HIP(hipSetDevice(h->device));
        const int rc = hb_wait(h->tail_stream, h->pin + 2 * (n4 + RES_HEAD), 0, n4 * 7, hb_stamp(h->tail_ticket), h->tailv.data(), "hm_update_tail");
        if (rc) return rc;
        h->tail_pending = false;
    }
    if (Hzc) memcpy(Hzc, h->tailv.data(), n4 * 4 * sizeof(double));
    if (gains) memcpy(gains, h->tailv.data() + n4 * 4, n4 * 3 * sizeof(double));
    return HM_OK;
}

extern "C" int hm_update_run(hm_ctx_t h, const double *W_prior, double *X, double deltaX, int masked, int max_iter,
                             double reltol, int info[4], double *errs, double *Hzc, double *gains, double *W_out)
{
    HM_ARG(h && X && info, "hm_update_run: NULL argument");
    HM_JOIN_LAZY(h);                               // (the tail of the last update: waited for before the first solve below)
    HM_ARG(deltaX > 0 && max_iter >= 0, "hm_update_run: deltaX must be positive, max_iter >= 0");
    // hm_chain_project: the prior mean is being left in d_X0 by the kernels of the state path; X is output only
    const bool chained = h->chain_pending;
    h->chain_pending = false;
    // what hm_update_arm_newton armed is for THIS call only: taken out of the handle before anything can fail, so that an
    // error return never leaves a worker pointer behind for a later call to start a job on
    h->last_err_valid = false;
    const bool pn_go = h->pn_armed;
    void *const pn_worker = h->pn_worker;
    h->pn_armed = false;
    h->pn_worker = nullptr;
    const bool pq_go = h->pq_armed && pn_go;
    h->pq_armed = false;
    h->pq_valid = false;
    const uint8_t *next_mask = h->armed_mask;
    h->armed_mask = nullptr;
    h->tail_pending = false;                       // (the block of the last update is about to be overwritten)
    NEED_TEX(h, "hm_update_run");
    NEED_OBS(h, "hm_update_run");
    const bool dbg = getenv("HYDRA_MI_TRACE") != nullptr;
    const auto dbg_a = std::chrono::steady_clock::now();
    auto dbg_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_a).count(); };
    double dbg_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc = update_begin(h, W_prior, chained ? nullptr : X);
    if (rc) return rc;
    dbg_stage[0] = dbg_ms();
    const int N = h->N, n4 = 4 * N;
    std::vector<double> X0, Xcur, Xold;
    if (!chained) { X0.assign(X, X + n4); Xcur = X0; Xold = X0; }
    // chained: what the prediction's and the projection's kernels report (host_block.h), taken while the first iteration
    // runs.  2: the prediction's inner solve gave up (never observed) -- the caller predicts on the host and calls again.
    auto chain_collect = [&]() -> int {
        int r = hb_wait(h->stream3, h->pin_n4 + n4, 0, (size_t)n4 + 2, hb_stamp(h->n4_ticket), h->n4v.data(), "hm_update_run (state prediction)");
        h->n4_pending = false;
        if (r) return r;
        r = hb_wait(h->stream3, h->pin_pm + n4, 0, (size_t)n4 + 1, hb_stamp(h->pm_ticket), h->pmv.data(), "hm_update_run (projectmask)");
        if (r) return r;
        h->chain_pred.assign(h->n4v.begin(), h->n4v.begin() + n4);
        h->chain_proj.assign(h->pmv.begin(), h->pmv.begin() + n4);
        h->chain_its = (int)h->n4v[n4];
        h->chain_moved = (int)h->pmv[n4];
        if (h->n4v[n4 + 1] != 0.0) return 2;
        X0 = h->chain_proj; Xcur = X0; Xold = X0;
        h->upd_X0 = X0;
        h->X0 = X0;
        return HM_OK;
    };
    bool collected = !chained;
    // (chained: the projection's kernel has written the first iterate, d_X, next to the prior mean)
    if (!chained) HM_HIP(hipMemcpyAsync(h->d_X, h->d_X0, (size_t)n4 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    double *const pin_res = h->pin, *const pin_tail = h->pin + 2 * ((size_t)n4 + RES_HEAD);
    const double *const res = h->resv.data();        // this call's copy of an iteration's block, taken when whole
    int niter = 0, accepted = 0;
    bool reverted = false, conv = false, ref_ready = false, regions_ahead = false, grown = false;
    double eold = 0.0;
    // spec: the measurement at the new iterate has been queued already -- behind k_iter_result, before the host has seen
    // that iteration's result, on the assumption that the loop goes on (it does 7 times out of 8): the device goes from
    // one iteration into the next without the ~14 us it takes the host to notice the result block, decide and launch.  ref / P
    // and d_X / d_Xn are swapped when it is queued; an iteration that turns out to be the last swaps them back (the
    // wasted measurement only wrote job sums and parked differences nobody reads, and the launches behind it on this
    // stream -- the covariance of the kept state -- are not what the next frame waits for).
    bool spec = false;
    auto unspec = [&]() {
        if (spec) { std::swap(h->ref, h->P); std::swap(h->d_X, h->d_Xn); spec = false; }
    };
    for (int it = 0; it < max_iter; it++) {
        if (!spec) {
            rc = measure_dev(h, h->d_X, ref_ready, deltaX, masked, regions_ahead, false);
            if (rc) return rc;
        }
        spec = false;
        if (next_mask && it == 0) {
            // The next frame's mask (hm_update_arm_mask): its pruning and outline (~0.2 ms of kernels on the second
            // stream) start when this frame's first measurement has run -- beside the first factorisation, which leaves
            // the chip idle -- not beside the first render and measurement, and not in the window between two frames,
            // where they slowed the state prediction's one workgroup down (0.33 instead of 0.27 ms)
            // (the event now; the launches when this iteration's own launches are queued and the host has nothing to do
            // but wait: eight launches here kept the first system of the frame waiting for the host, ~17 us per iteration
            // on average in the kernel trace)
            if (!h->ev_m0) HM_HIP(hipEventCreateWithFlags(&h->ev_m0, hipEventDisableTiming));
            HM_HIP(hipEventRecord(h->ev_m0, h->stream));
        }
        if (collected) h->X0 = Xcur;               // the state of the reference render (hm_jz / hm_j)
        h->have_ref = true;
        const int slot = h->upd_last == 0 ? 1 : 0;
        rc = ctx_join(h);                          // the tail of the last update may still be writing inv(W) and the factor slots
        if (rc) return rc;
        double *rhs_row = solve_step(h, slot, deltaX);
        // the new iterate: its render, the partial sums of Renderer.error (kalman.py:813) and, as extra workgroups of
        // the same launch, the star regions of the next measurement (they need the new iterate only; wasted when the
        // loop ends here)
        regions_ahead = it + 1 < max_iter;
        rc = render_iter(h, h->d_Xn, h->P, true, masked, regions_ahead, deltaX);
        if (rc) return rc;
        hipLaunchKernelGGL(k_iter_result, dim3(1), dim3(256), 0, h->stream, rhs_row, n4, h->d_tpart,
                           render_strips(h), h->pool.overflow, (const unsigned *)h->d_flowctl, pin_res, (double)(++h->run_ticket),
                           h->result_delay);
        HM_HIP(hipGetLastError());
        if (h->speculate && regions_ahead) {
            std::swap(h->ref, h->P);
            std::swap(h->d_X, h->d_Xn);
            spec = true;
            rc = measure_dev(h, h->d_X, true, deltaX, masked, true, false);
            if (rc) { unspec(); return rc; }
        }
        const auto dbg_t0 = std::chrono::steady_clock::now();
        if (next_mask && it == 0) {
            rc = ensure_stream2(h);
            if (rc == HM_OK) { HM_HIP(hipStreamWaitEvent(h->stream2, h->ev_m0, 0)); rc = prepare_mask(h, next_mask); }
            if (rc) { unspec(); return rc; }
            next_mask = nullptr;
        }
        if (!collected) {
            rc = chain_collect();
            if (rc) { unspec(); (void)hipStreamSynchronize(h->stream); return rc; }
            collected = true;
        }
        rc = hb_wait(h->stream, pin_res, 0, (size_t)n4 + RES_HEAD, hb_stamp(h->run_ticket), h->resv.data(), "hm_update_run");
        if (rc) { unspec(); return rc; }
        if (getenv("HYDRA_MI_TRACE")) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count();
            if (ms > 1.5) fprintf(stderr, "[hydra_mi] hm_update_run: iteration %d waited %.2f ms for its result\n", it, ms);
        }
        if (res[n4 + 4] != 0.0) {
            unspec();
            // the star regions of this measurement did not fit the pool of difference images: grow it and take the
            // iteration again (nothing of it has been kept; the regions the failed pass computed for its -- meaningless
            // -- next iterate are computed anew)
            if (grown) { hm_set_error("hm_update_run: the star regions do not fit the difference-image pool"); return HM_ERR_STATE; }
            if (regions_ahead) {
                // d_area no longer describes the measurement that overflowed: this iteration's render launch has put the
                // regions of its (meaningless) new iterate there.  Those of the iterate that was measured, again:
                MeasureArgs a;
                measure_args(h, h->d_X, deltaX, masked, a);
                hipLaunchKernelGGL(k_star_regions, dim3(h->N), dim3(REGION_NT), 0, h->stream, a, h->d_area);
                HM_HIP(hipGetLastError());
            }
            HM_HIP(hipStreamSynchronize(h->stream));
            rc = pool_grow(h, "hm_update_run");
            if (rc) return rc;
            grown = true;
            regions_ahead = false;
            HM_HIP(hipMemsetAsync(h->d_flowctl, 0, 4 * sizeof(unsigned), h->stream));      // (a NaN system may have timed out)
            it--;
            continue;
        }
        grown = false;
        h->upd_prev = h->upd_last;
        h->upd_last = slot;
        niter++;
        if (res[n4 + 7] != 0.0) {
            unspec();
            hm_set_error("hm_update_run: the factorisation launch gave up waiting for a block (chol_flow time-out)");
            return HM_ERR_HIP;
        }
        bool finite = true;
        for (int i = 0; i < n4; i++) finite = finite && std::isfinite(res[i]);
        if (!finite) {
            unspec();
            hm_set_error("hm_update_run: the update system inv(W) + HTH is not positive definite "
                         "(non-finite state, covariance or observation?)");
            return HM_ERR_NUMERIC;
        }
        for (int i = 0; i < n4; i++) Xcur[i] = X0[i] + res[i];
        if (errs) for (int k = 0; k < 4; k++) errs[4 * it + k] = res[n4 + k];
        // a triangle that flipped: back to the last good state (kalman.py:806-811)
        bool flipped = false;
        for (int t = 0; t < h->T && !flipped; t++) {
            const int a = h->tri[3 * t], b = h->tri[3 * t + 1], c = h->tri[3 * t + 2];
            const double ax = Xcur[2 * b] - Xcur[2 * a], ay = Xcur[2 * b + 1] - Xcur[2 * a + 1];
            const double bx = Xcur[2 * c] - Xcur[2 * a], by = Xcur[2 * c + 1] - Xcur[2 * a + 1];
            flipped = ax * by - ay * bx < 0.0;
        }
        if (flipped) {
            unspec();
            Xcur = Xold;
            reverted = true;
            break;
        }
        // the error sums of the image and mask terms are whole numbers (the reference truncates
        // them to int before combining, kalman.py:813-816)
        const double e_im = std::trunc(res[n4]), e_m = std::trunc(res[n4 + 3]);
        const double enew = std::sqrt(e_im * e_im + res[n4 + 1] * res[n4 + 1] + res[n4 + 2] * res[n4 + 2] + e_m * e_m);
        accepted++;
        if (std::fabs(enew - eold) / enew < reltol) { unspec(); conv = true; break; }
        eold = enew;
        Xold = Xcur;
        // the new iterate becomes the point of the next measurement; its render is already there
        if (!spec) {
            std::swap(h->ref, h->P);
            std::swap(h->d_X, h->d_Xn);
        }
        ref_ready = true;
        h->X0 = Xcur;
    }
    unspec();                                      // (max_iter reached: regions_ahead was false, nothing was queued)
    if (!collected) {                              // (max_iter = 0)
        rc = chain_collect();
        if (rc) return rc;
    }
    dbg_stage[1] = dbg_ms();
    // the state is final: a caller that armed it gets the next frame's state prediction started now, on its worker
    // thread, beside the covariance launches below (hm_update_arm_newton)
    if (pn_go) {
        rc = hm_ms_newton_start(pn_worker, N, (int)h->pn_l0.size(), h->pn_bars.data(), h->pn_l0.data(), h->pn_par[0],
                                h->pn_par[1], h->pn_par[2], h->pn_maxiter, h->pn_par[3], Xcur.data());
        if (rc) return rc;
    }
    dbg_stage[2] = dbg_ms();
    // covariance of the state that is kept (kalman.py:806-811, 826)
    int which = -1;                                // -1: the prior
    if (reverted) which = accepted == 0 ? -1 : 1;
    else which = niter == 0 ? -1 : 0;
    // The tail runs on a stream of its own (stream4) when the caller does not want the covariance on the host: everything
    // it reads is complete -- the host has seen the result block of the last iteration, whose kernel is behind all of
    // them on `stream` -- and what is still queued on `stream` (the measurement of an iteration that does not happen)
    // and what the next frame queues there first (reference render, measurement) touch none of its buffers; `stream` waits
    // for its end (ev_tail, on the device) before the next solve (ctx_join).
    hipStream_t ts = h->stream;
    rc = ctx_join(h);                              // (a tail nobody has joined since: an update without iterations)
    if (rc) return rc;
    if (!W_out && h->tail_split) {
        if (!h->stream4) {
            int least = 0, greatest = 0;
            HM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            if (greatest != least) { HM_HIP(hipStreamCreateWithPriority(&h->stream4, hipStreamNonBlocking, greatest)); }
            else HM_HIP(hipStreamCreateWithFlags(&h->stream4, hipStreamNonBlocking));
            HM_HIP(hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming));
        }
        ts = h->stream4;
    }
    const long long tail_ticket = niter > 0 ? ++h->run_ticket : h->run_ticket;
    const double eps_F = h->pq_eps_F;
    const bool ahead = pq_go && h->chol_flow && which >= 0 && niter > 0 && !W_out;
    // the launches of the tail; Xk: the state that is kept
    auto tail = [h, next_mask, which, ts, niter, n4, pin_tail, tail_ticket, W_out, ahead, eps_F](const std::vector<double> &Xk) -> int {
        HM_HIP(hipSetDevice(h->device));
        int r = HM_OK;
        if (next_mask) {                           // the next frame's outline, beside the prediction (hm_update_arm_mask)
            r = prepare_mask(h, next_mask);
            if (r) return r;
        }
        r = update_cov_on(h, which, nullptr, ts);
        if (r) return r;
        if (niter > 0) {
            // gains and Hz components go to the host as a result block of their own (host_block.h), as the iterations'
            // results do: two blit launches and the wake-up from a stream synchronisation less per frame
            hipLaunchKernelGGL(k_gains, dim3(n4), dim3(256), 0, ts, h->d_Wres, h->d_Hzc, n4, h->d_gain);
            hipLaunchKernelGGL(k_tail_result, dim3(1), dim3(1024), 0, ts, h->d_Hzc, h->d_gain, n4, pin_tail, (double)tail_ticket,
                               h->result_delay);
        }
        if (W_out) HM_HIP(hipMemcpyAsync(W_out, h->d_Wres, (size_t)n4 * n4 * sizeof(double), hipMemcpyDeviceToHost, ts));
        HM_HIP(hipGetLastError());
        if (ahead) {
            // the covariance half of the next frame's prediction, behind the launches above (hm_update_arm_cov)
            r = queue_predict_ahead(h, Xk.data(), (int)h->pn_l0.size(), h->pn_bars.data(), h->pn_l0.data(), h->pn_par[0],
                                    h->pn_par[1], h->pn_par[2], eps_F, ts);
            if (r) return r;
        }
        if (ts != h->stream) {
            HM_HIP(hipEventRecord(h->ev_tail, ts));
            h->tail_on_stream4 = true;
        }
        return HM_OK;
    };
    dbg_stage[3] = dbg_ms();
    if (niter > 0 && !W_out && !Hzc && !gains) {
        // nobody wants the gains now: they are taken when asked for (hm_update_tail) and the caller gets on with its next
        // frame -- the tail's ~20 launches are queued by the handle's helper thread (every entry point joins it first)
        h->tail_pending = true;
        h->tail_ticket = tail_ticket;
        h->tail_stream = ts;
        if (h->tail_async) {
            h->helper_used = true;
            h->helper.post([tail, Xk = Xcur]() { return tail(Xk); });
        } else {
            rc = tail(Xcur);
            if (rc) return rc;
        }
    } else if (niter > 0 && !W_out) {
        rc = tail(Xcur);
        if (rc) return rc;
        rc = hb_wait(ts, pin_tail, 0, (size_t)n4 * 7, hb_stamp(tail_ticket), h->tailv.data(), "hm_update_run");
        if (rc) return rc;
    } else {
        rc = tail(Xcur);
        if (rc) return rc;
        HM_HIP(stream_wait(ts));
        if (niter > 0 && !hb_take(pin_tail, 0, (size_t)n4 * 7, hb_stamp(tail_ticket), h->tailv.data())) {
            hm_set_error("hm_update_run: the gains did not arrive although the stream has completed");
            return HM_ERR_HIP;
        }
    }
    dbg_stage[4] = dbg_ms();
    if (dbg && dbg_stage[4] > 5.0)
        fprintf(stderr, "[hydra_mi] hm_update_run %d iterations: begin %.2f loop-end %.2f newton-started %.2f tail-queued %.2f tail-done %.2f ms\n",
                niter, dbg_stage[0], dbg_stage[1], dbg_stage[2], dbg_stage[3], dbg_stage[4]);
    if (niter > 0) {
        if (Hzc) memcpy(Hzc, h->tailv.data(), (size_t)n4 * 4 * sizeof(double));
        if (gains) memcpy(gains, h->tailv.data() + (size_t)n4 * 4, (size_t)n4 * 3 * sizeof(double));
    } else {
        if (Hzc) memset(Hzc, 0, (size_t)n4 * 4 * sizeof(double));
        if (gains) memset(gains, 0, (size_t)n4 * 3 * sizeof(double));
        std::fill(h->tailv.begin(), h->tailv.end(), 0.0);
    }
    memcpy(X, Xcur.data(), (size_t)n4 * sizeof(double));
    info[0] = niter; info[1] = accepted; info[2] = reverted ? 1 : 0; info[3] = conv ? 1 : 0;
    if (niter > 0 && !reverted) {
        // the state that is kept is the last iterate: its render produced Renderer.error against the raw flow as well
        h->last_err[0] = res[n4]; h->last_err[1] = res[n4 + 8]; h->last_err[2] = res[n4 + 9]; h->last_err[3] = res[n4 + 3];
        h->last_err_X = Xcur;
        h->last_err_valid = true;
    }
    return HM_OK;
}

// Renderer.error (renderer.py:485-501) of the state the last hm_update_run kept, against the observation as given
// (the raw flow), when that state is its last iterate: the sums came out of that iterate's render (k_render_iter).
// Returns HM_OK and fills err, or 1 when they are not at hand (another state, a reverted update, a new observation):
// the caller then asks hm_error.
extern "C" int hm_update_last_error(hm_ctx_t h, const double *X, double err[4])
{
    HM_ARG(h && X && err, "hm_update_last_error: NULL argument");
    if (!h->last_err_valid || h->last_err_X.size() != (size_t)4 * h->N) return 1;
    if (memcmp(X, h->last_err_X.data(), (size_t)4 * h->N * sizeof(double)) != 0) return 1;
    for (int k = 0; k < 4; k++) err[k] = h->last_err[k];
    return HM_OK;
}

// Covariance prediction W' = F W F^T + Weps (kalman.py:717, 863) on the device.
//   W_in   : the covariance to propagate (host), or NULL to use the one hm_update_cov returned last,
//            which is still on the device;
//   bars / blocks : the springs and, per spring, the symmetric 2x2 block (Bxx, Bxy, Byy) of the force
//            Jacobian at the state before the step; n_bars = 0 gives the constant-velocity F (A = 0);
//   a, s   : F = [[I, a I], [s dfdy, I]];  eps_F : Weps = eps_F [[I/4, I/2], [I/2, I]].
// The result is copied to W_out and stays on the device as the prior of the next hm_update_begin(NULL).
// ahead: called by hm_update_run for the NEXT frame, behind launches that are still running -- nothing here may wait for
// the stream: the spring blocks go through page-locked memory (`blocks` is h->pin_blk), the result goes straight to
// d_Wprior (where prior_inverse wants it) and the resident covariance (d_Wres, the posterior the caller may still
// fetch) is left alone.
static int cov_predict_core(hm_ctx *h, const double *W_in, int n_bars, const int32_t *bars, const double *blocks,
                            double a, double s, double eps_F, double *W_out, bool ahead, hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    const int N = h->N, n4 = 4 * N;
    const size_t nn = (size_t)n4 * n4 * sizeof(double);
    std::vector<int> &off = h->sp_h_off, &bar = h->sp_h_bar, &other = h->sp_h_other;   // live until the copies ran
    if (!ahead) HM_HIP(stream_wait(st));      // ... of the previous call
    // the springs' topology rarely changes between frames: its device copy is kept and only the per-spring blocks go up
    const bool same_topo = h->d_sp_off && h->sp_bars_cached.size() == 2 * (size_t)n_bars &&
                           (n_bars == 0 || memcmp(h->sp_bars_cached.data(), bars, 2 * (size_t)n_bars * sizeof(int32_t)) == 0);
    if (!same_topo) {
        off.assign(N + 1, 0);
        for (int i = 0; i < n_bars; i++) {
            HM_ARG(bars[2 * i] >= 0 && bars[2 * i] < N && bars[2 * i + 1] >= 0 && bars[2 * i + 1] < N,
                   "hm_cov_predict: spring %d refers to a vertex outside 0..%d", i, N - 1);
            off[bars[2 * i] + 1]++;
            off[bars[2 * i + 1] + 1]++;
        }
        for (int v = 0; v < N; v++) off[v + 1] += off[v];
        bar.resize(2 * (size_t)n_bars); other.resize(2 * (size_t)n_bars);
        {
            std::vector<int> fill(off.begin(), off.end() - 1);
            for (int i = 0; i < n_bars; i++) {
                const int p = bars[2 * i], q = bars[2 * i + 1];
                bar[fill[p]] = i; other[fill[p]++] = q;
                bar[fill[q]] = i; other[fill[q]++] = p;
            }
        }
        if (!h->d_sp_off) HM_HIP(hm_malloc((void **)&h->d_sp_off, (size_t)(N + 1) * sizeof(int)));
        if ((size_t)n_bars > h->sp_cap) {
            if (h->d_sp_bar) (void)hipFree(h->d_sp_bar);
            if (h->d_sp_other) (void)hipFree(h->d_sp_other);
            if (h->d_sp_blk) (void)hipFree(h->d_sp_blk);
            h->d_sp_bar = h->d_sp_other = nullptr; h->d_sp_blk = nullptr;
            HM_HIP(hm_malloc((void **)&h->d_sp_bar, 2 * (size_t)n_bars * sizeof(int)));
            HM_HIP(hm_malloc((void **)&h->d_sp_other, 2 * (size_t)n_bars * sizeof(int)));
            HM_HIP(hm_malloc((void **)&h->d_sp_blk, 3 * (size_t)n_bars * sizeof(double)));
            h->sp_cap = n_bars;
        }
        HM_HIP(hipMemcpyAsync(h->d_sp_off, off.data(), (size_t)(N + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        if (n_bars > 0) {
            HM_HIP(hipMemcpyAsync(h->d_sp_bar, bar.data(), bar.size() * sizeof(int), hipMemcpyHostToDevice, st));
            HM_HIP(hipMemcpyAsync(h->d_sp_other, other.data(), other.size() * sizeof(int), hipMemcpyHostToDevice, st));
        }
        h->sp_bars_cached.assign(bars, bars + 2 * (size_t)n_bars);
    }
    if (n_bars > 0) {
        const double *from = blocks;
        if (!ahead) {
            h->sp_h_blk.assign(blocks, blocks + 3 * (size_t)n_bars);
            from = h->sp_h_blk.data();
        }
        HM_HIP(hipMemcpyAsync(h->d_sp_blk, from, 3 * (size_t)n_bars * sizeof(double), hipMemcpyHostToDevice, st));
    }
    const double *src = h->d_Wres;
    if (W_in) {
        HM_HIP(hipMemcpyAsync(h->d_H, W_in, nn, hipMemcpyHostToDevice, st));
        src = h->d_H;
    } else if (src == h->d_Wtmp) {               // the output buffer: move the input out of the way
        HM_HIP(hipMemcpyAsync(h->d_H, h->d_Wtmp, nn, hipMemcpyDeviceToDevice, st));
        src = h->d_H;
    }
    SpringTopo tp = {h->d_sp_off, h->d_sp_bar, h->d_sp_other, h->d_sp_blk};
    double *P = h->d_Awork;                      // scratch
    double *dst = ahead ? h->d_Wprior : h->d_Wtmp;
    hipLaunchKernelGGL(k_fw_rows, dim3(hm_cdiv(n4, 256), N), dim3(256), 0, st, src, P, N, tp, a, s);
    hipLaunchKernelGGL(k_pft_cols, dim3(hm_cdiv(n4, 256), N), dim3(256), 0, st, P, dst, N, tp, a, s, eps_F);
    HM_HIP(hipGetLastError());
    if (ahead) return HM_OK;
    if (W_out) HM_HIP(hipMemcpyAsync(W_out, h->d_Wtmp, nn, hipMemcpyDeviceToHost, st));
    if (W_out) HM_HIP(stream_wait(st));
    h->d_Wres = h->d_Wtmp;
    h->prefactored = false;
    return HM_OK;
}

extern "C" int hm_cov_predict(hm_ctx_t h, const double *W_in, int n_bars, const int32_t *bars, const double *blocks,
                              double a, double s, double eps_F, double *W_out)
{
    HM_ARG(h && n_bars >= 0 && (n_bars == 0 || (bars && blocks)), "hm_cov_predict: bad argument");
    HM_JOIN(h);
    if (!W_in && !h->d_Wres) {
        hm_set_error("hm_cov_predict: no covariance given and none resident on the device");
        return HM_ERR_STATE;
    }
    HM_HIP(hipSetDevice(h->device));
    h->pq_valid = false;                         // a prediction queued ahead (hm_update_arm_cov) is not what this caller wants
    return cov_predict_core(h, W_in, n_bars, bars, blocks, a, s, eps_F, W_out, false);
}

// The per-spring blocks (Bxx, Bxy, Byy) of the force Jacobian at the vertices of X (kalman.py:892-901): bar i between
// a and b, d = y_a - y_b, l = |d|:  B = k I + c d d^T,  k = kappa (1 - l0/l),  c = kappa l0 / l^3.
static void spring_blocks(int n_bars, const int32_t *bars, const double *l0, double kappa, const double *X, double *blk)
{
    for (int i = 0; i < n_bars; i++) {
        const int a = bars[2 * i], b = bars[2 * i + 1];
        const double dx = X[2 * a] - X[2 * b], dy = X[2 * a + 1] - X[2 * b + 1];
        const double l = std::sqrt(dx * dx + dy * dy);
        const double k = kappa * (1.0 - l0[i] / l), c = kappa * l0[i] / (l * l * l);
        blk[3 * i] = k + c * dx * dx; blk[3 * i + 1] = c * dx * dy; blk[3 * i + 2] = k + c * dy * dy;
    }
}

// hm_update_run, armed by hm_update_arm_newton + hm_update_arm_cov, when its state is final and the covariance of that
// state (d_Wres) is queued: the covariance half of the NEXT frame's prediction -- W' = F W F^T + Weps with F at this
// state (hm_cov_predict) and the factorisation / inverse of W' the next update starts with (hm_update_prefactor) --
// goes onto the stream right behind it, ~0.3 ms before the caller could ask for it (it first has to get back to
// IteratedMSKalmanFilter.predict).  Nothing is made current: hm_predict_take does that when the caller's inputs turn
// out to be the ones used here; otherwise the caller's own hm_cov_predict starts from the posterior, which is intact.
static int queue_predict_ahead(hm_ctx *h, const double *X, int n_bars, const int32_t *bars, const double *l0, double kappa,
                               double M, double dt, double eps_F, hipStream_t st)
{
    for (int i = 0; i < 2 * n_bars; i++)
        if (bars[i] < 0 || bars[i] >= h->N) return HM_OK;                 // the caller's own hm_cov_predict reports it
    if ((size_t)n_bars > h->pin_blk_cap) {
        if (h->pin_blk) (void)hipHostFree(h->pin_blk);
        h->pin_blk = nullptr; h->pin_blk_cap = 0;
        HM_HIP(hipHostMalloc((void **)&h->pin_blk, 3 * (size_t)n_bars * sizeof(double), hipHostMallocDefault));
        h->pin_blk_cap = n_bars;
    }
    spring_blocks(n_bars, bars, l0, kappa, X, h->pin_blk);
    double *const post = h->d_Wres;
    int rc = cov_predict_core(h, nullptr, n_bars, bars, h->pin_blk, dt, dt / M, eps_F, nullptr, true, st);
    if (rc) return rc;
    h->d_Wres = h->d_Wprior;                     // (prior_inverse: the prior is where it belongs already)
    rc = prior_inverse(h, nullptr, st);
    h->d_Wres = post;
    if (rc) return rc;
    h->upd_open = false;                         // the factor slots and d_Wprior belong to the next update now
    h->pq_X.assign(X, X + 4 * (size_t)h->N);
    h->pq_bars.assign(bars, bars + 2 * (size_t)n_bars);
    h->pq_l0.assign(l0, l0 + n_bars);
    h->pq_par[0] = kappa; h->pq_par[1] = dt; h->pq_par[2] = dt / M; h->pq_par[3] = eps_F;
    h->pq_valid = true;
    return HM_OK;
}

// Makes the prediction hm_update_run queued ahead current -- as if hm_cov_predict(h, NULL, n_bars, bars, blocks at X, a, s,
// eps_F, NULL) and hm_update_prefactor(h) had just been called -- when it was made from exactly these inputs (compared
// bit for bit) and the posterior it started from is still the resident covariance.  Returns HM_OK when taken, 1 when
// there is nothing to take (the caller then makes those two calls itself).
extern "C" int hm_predict_take(hm_ctx_t h, const double *X, int n_bars, const int32_t *bars, const double *l0, double kappa,
                               double a, double s, double eps_F)
{
    HM_ARG(h && X && n_bars >= 0 && (n_bars == 0 || (bars && l0)), "hm_predict_take: bad argument");
    HM_JOIN_LAZY(h);
    if (!h->pq_valid) return 1;
    h->pq_valid = false;
    const size_t n4 = (size_t)4 * h->N;
    const bool same = h->pq_X.size() == n4 && memcmp(h->pq_X.data(), X, n4 * sizeof(double)) == 0 &&
                      h->pq_l0.size() == (size_t)n_bars &&
                      (n_bars == 0 || (memcmp(h->pq_bars.data(), bars, 2 * (size_t)n_bars * sizeof(int32_t)) == 0 &&
                                       memcmp(h->pq_l0.data(), l0, (size_t)n_bars * sizeof(double)) == 0)) &&
                      h->pq_par[0] == kappa && h->pq_par[1] == a && h->pq_par[2] == s && h->pq_par[3] == eps_F;
    if (!same || !h->d_Wres) return 1;
    h->d_Wres = h->d_Wprior;
    h->prefactored = true;
    h->upd_open = false;
    return HM_OK;
}


// IteratedMSKalmanFilter.predict (kalman.py:850-863) in one call: F from the spring Jacobian at the state before the
// step (:856, _dfdx :904-912), the state advanced by _newton (:923-960), W <- F W F^T + Weps (:863) -- and, since
// it only needs the predicted covariance, the covariance half of the update that follows (factorisation and inverse,
// what hm_update_prefactor queues).  The Newton iterations run as one workgroup on a second stream
// (csrc/predict_kernels.h) while this thread queues the ~30 launches of the covariance half on the handle's stream;
// meshes too large for that kernel's LDS footprint take the host version (hm_ms_newton).
extern "C" int hm_ms_predict(hm_ctx_t h, int n_bars, const int32_t *bars, const double *l0, double kappa, double M, double dt,
                             int maxiter, double tol, double eps_F, double *X, int *newton_iterations, int prefactor)
{
    HM_ARG(h && bars && l0 && X && n_bars >= 1, "hm_ms_predict: bad argument");
    HM_ARG(dt > 0 && M > 0 && maxiter >= 1 && tol > 0, "hm_ms_predict: bad parameter");
    HM_JOIN(h);
    if (!h->d_Wres) { hm_set_error("hm_ms_predict: no covariance resident on the device"); return HM_ERR_STATE; }
    HM_HIP(hipSetDevice(h->device));
    const int N = h->N, n4 = 4 * N;
    for (int i = 0; i < 2 * n_bars; i++) HM_ARG(bars[i] >= 0 && bars[i] < N, "hm_ms_predict: bar refers to vertex %d", bars[i]);
    // the spring blocks of dfdy at the state before the step (kalman.py:892-901)
    std::vector<double> blk(3 * (size_t)n_bars);
    spring_blocks(n_bars, bars, l0, kappa, X, blk.data());
    const size_t lds = ((size_t)34 * N + (size_t)7 * n_bars + 8) * sizeof(double) + ((size_t)N + 1 + 4 * (size_t)n_bars) * sizeof(int);
    const bool on_device = lds <= 160 * 1024;
    if (on_device) {
        { const int rc2 = ensure_stream2(h); if (rc2) return rc2; }
        if (!h->d_nX) {
            HM_HIP(hm_malloc((void **)&h->d_nX, (size_t)n4 * sizeof(double)));
            HM_HIP(hm_malloc((void **)&h->d_nvoff, (size_t)(N + 1) * sizeof(int)));
            HM_HIP(hm_malloc((void **)&h->d_ninfo, 2 * sizeof(int)));
            HM_HIP(hipFuncSetAttribute((const void *)k_ms_newton, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if ((size_t)n_bars > h->ncap) {
            if (h->d_nbars) (void)hipFree(h->d_nbars);
            if (h->d_nvbar) (void)hipFree(h->d_nvbar);
            if (h->d_nl0) (void)hipFree(h->d_nl0);
            h->d_nbars = h->d_nvbar = nullptr; h->d_nl0 = nullptr; h->ncap = 0;
            HM_HIP(hm_malloc((void **)&h->d_nbars, 2 * (size_t)n_bars * sizeof(int)));
            HM_HIP(hm_malloc((void **)&h->d_nvbar, 2 * (size_t)n_bars * sizeof(int)));
            HM_HIP(hm_malloc((void **)&h->d_nl0, (size_t)n_bars * sizeof(double)));
            h->ncap = n_bars;
        }
        std::vector<int> off(N + 1, 0), vbar(2 * (size_t)n_bars);
        for (int i = 0; i < n_bars; i++) { off[bars[2 * i] + 1]++; off[bars[2 * i + 1] + 1]++; }
        for (int v = 0; v < N; v++) off[v + 1] += off[v];
        {
            std::vector<int> fill(off.begin(), off.end() - 1);
            for (int i = 0; i < n_bars; i++) { vbar[fill[bars[2 * i]]++] = i; vbar[fill[bars[2 * i + 1]]++] = i; }   // ascending per vertex
        }
        HM_HIP(hipMemcpyAsync(h->d_nbars, bars, 2 * (size_t)n_bars * sizeof(int), hipMemcpyHostToDevice, h->stream2));
        HM_HIP(hipMemcpyAsync(h->d_nl0, l0, (size_t)n_bars * sizeof(double), hipMemcpyHostToDevice, h->stream2));
        HM_HIP(hipMemcpyAsync(h->d_nvoff, off.data(), (size_t)(N + 1) * sizeof(int), hipMemcpyHostToDevice, h->stream2));
        HM_HIP(hipMemcpyAsync(h->d_nvbar, vbar.data(), vbar.size() * sizeof(int), hipMemcpyHostToDevice, h->stream2));
        HM_HIP(hipMemcpyAsync(h->d_nX, X, (size_t)n4 * sizeof(double), hipMemcpyHostToDevice, h->stream2));
        HM_HIP(hipStreamSynchronize(h->stream2));            // off / vbar are locals (pageable copies have returned anyway)
        NewtonArgs a = {N, n_bars, h->d_nbars, h->d_nl0, h->d_nvoff, h->d_nvbar, kappa, M, dt, tol, maxiter,
                        (int)std::ceil(1.0 / dt), h->d_nX, h->d_ninfo};
        hipLaunchKernelGGL(k_ms_newton, dim3(1), dim3(NEWTON_NT), lds, h->stream2, a);
        HM_HIP(hipGetLastError());
    }
    // covariance: W <- F W F^T + Weps on the handle's stream, then (prefactor) its factor and inverse
    int rc = hm_cov_predict(h, nullptr, n_bars, bars, blk.data(), dt, dt / M, eps_F, nullptr);
    if (rc == HM_OK && prefactor) {
        rc = prior_inverse(h, nullptr);
        if (rc == HM_OK) { h->prefactored = true; h->upd_open = false; }
    }
    int info[2] = {0, 0};
    if (on_device) {
        hipError_t e = hipMemcpyAsync(X, h->d_nX, (size_t)n4 * sizeof(double), hipMemcpyDeviceToHost, h->stream2);
        if (e == hipSuccess) e = hipMemcpyAsync(info, h->d_ninfo, sizeof info, hipMemcpyDeviceToHost, h->stream2);
        if (e == hipSuccess) e = stream_wait(h->stream2);
        if (e != hipSuccess) { hm_set_error("hm_ms_predict: %s", hipGetErrorString(e)); return HM_ERR_HIP; }
        if (rc) return rc;
        if (info[1]) { hm_set_error("hm_ms_predict: the inner solve did not converge"); return HM_ERR_STATE; }
        if (newton_iterations) *newton_iterations = info[0];
        return HM_OK;
    }
    if (rc) return rc;
    return hm_ms_newton(N, n_bars, bars, l0, kappa, M, dt, maxiter, tol, X, newton_iterations);
}


// ---- the state prediction's Newton loop on the device, started ahead (csrc/predict_kernels.h: k_ms_newton4) -----------------
// hm_newton_dev_start queues ONE launch on a stream of its own: the state goes in and comes out through page-locked
// memory, the host watches a ticket (hm_newton_dev_finish).  Called by the worker object of csrc/predict.cpp when a
// handle is attached to it (hm_ms_worker_attach): from hm_update_run the moment its state is final, i.e. the kernel
// runs beside the covariance launches of the frame that ends and the caller's way back to predict().  Returns 1 when
// this mesh does not fit the kernel (more than 256 vertices, a vertex with more than 12 springs): the caller takes the
// host loop.
extern "C" int hm_newton_dev_start(hm_ctx_t h, int N, int n_bars, const int32_t *bars, const double *l0, double kappa, double M,
                                   double dt, int maxiter, double tol, const double *X)
{
    HM_ARG(h && bars && l0 && X && n_bars >= 1, "hm_newton_dev_start: bad argument");
    if (N != h->N || N > NEWTON4_NT) return 1;
    HM_HIP(hipSetDevice(h->device));
    if (h->n4_pending) {                          // a prediction nobody fetched: let it finish (its result block is dropped)
        HM_HIP(hipStreamSynchronize(h->stream3));
        h->n4_pending = false;
    }
    if (!h->stream3) {
        int least = 0, greatest = 0;
        HM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (greatest != least) { HM_HIP(hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, greatest)); }
        else HM_HIP(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
        // page-locked and coherent (hipHostMalloc without flags already is: the flag states the intent, it was not the
        // cause of round 3's wrong tracks -- host_block.h has that story): [X in (4N) | a result block of 4N + 2 values]
        const size_t words = (size_t)4 * N + 2 * ((size_t)4 * N + 2);
        HM_HIP(hipHostMalloc((void **)&h->pin_n4, words * sizeof(double), hipHostMallocCoherent));
        memset(h->pin_n4, 0, words * sizeof(double));
        HM_HIP(hm_malloc((void **)&h->d_n4X, ((size_t)4 * N + 2) * sizeof(double)));
        h->n4v.assign((size_t)4 * N + 2, 0.0);
        HM_HIP(hipEventCreateWithFlags(&h->ev_n4, hipEventDisableTiming));
    }
    const bool same = h->n4_bars.size() == 2 * (size_t)n_bars && memcmp(h->n4_bars.data(), bars, 2 * (size_t)n_bars * sizeof(int32_t)) == 0;
    if (!same) {
        for (int i = 0; i < 2 * n_bars; i++) HM_ARG(bars[i] >= 0 && bars[i] < N, "hm_newton_dev_start: bar refers to vertex %d", bars[i]);
        std::vector<int> deg(N, 0);
        for (int i = 0; i < 2 * n_bars; i++) deg[bars[i]]++;
        const int maxdeg = *std::max_element(deg.begin(), deg.end());
        h->n4_bars.assign(bars, bars + 2 * (size_t)n_bars);
        h->n4_l0.clear();
        h->n4deg = maxdeg <= 8 ? 8 : maxdeg <= 12 ? 12 : 0;
        if (h->n4deg) {
            const int D = h->n4deg;
            std::vector<int> nbr((size_t)N * D), nbb((size_t)N * D, n_bars), fill(N, 0);
            for (int v = 0; v < N; v++)
                for (int q = 0; q < D; q++) nbr[(size_t)v * D + q] = v;             // padding: the vertex itself, bar I (zero terms)
            for (int i = 0; i < n_bars; i++) {                                        // ascending bar order per vertex
                const int a = bars[2 * i], b = bars[2 * i + 1];
                nbr[(size_t)a * D + fill[a]] = b; nbb[(size_t)a * D + fill[a]++] = i;
                nbr[(size_t)b * D + fill[b]] = a; nbb[(size_t)b * D + fill[b]++] = i;
            }
            if ((size_t)n_bars > h->n4cap || !h->d_n4nbr) {
                void *q[] = {h->d_n4nbr, h->d_n4nbb, h->d_n4bars, h->d_n4l0};
                for (void *x : q) if (x) (void)hipFree(x);
                h->d_n4nbr = h->d_n4nbb = h->d_n4bars = nullptr; h->d_n4l0 = nullptr; h->n4cap = 0;
                HM_HIP(hm_malloc((void **)&h->d_n4nbr, (size_t)N * 12 * sizeof(int)));
                HM_HIP(hm_malloc((void **)&h->d_n4nbb, (size_t)N * 12 * sizeof(int)));
                HM_HIP(hm_malloc((void **)&h->d_n4bars, 2 * (size_t)n_bars * sizeof(int)));
                HM_HIP(hm_malloc((void **)&h->d_n4l0, (size_t)n_bars * sizeof(double)));
                h->n4cap = n_bars;
            }
            // on the kernel's own stream and waited for (nbr / nbb are locals): ordered before the launch by the stream
            // itself rather than by hipMemcpy's return (that was not the cause of round 3's wrong tracks either -- a track
            // uploads these tables once, the wrong predictions came at frame 13 -- but it is the form that needs no argument)
            HM_HIP(hipMemcpyAsync(h->d_n4nbr, nbr.data(), nbr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream3));
            HM_HIP(hipMemcpyAsync(h->d_n4nbb, nbb.data(), nbb.size() * sizeof(int), hipMemcpyHostToDevice, h->stream3));
            HM_HIP(hipMemcpyAsync(h->d_n4bars, h->n4_bars.data(), 2 * (size_t)n_bars * sizeof(int), hipMemcpyHostToDevice, h->stream3));
            HM_HIP(hipStreamSynchronize(h->stream3));            // (nbr / nbb are locals)
        }
    }
    if (!h->n4deg) return 1;
    if (h->n4_l0.size() != (size_t)n_bars || memcmp(h->n4_l0.data(), l0, (size_t)n_bars * sizeof(double)) != 0) {
        h->n4_l0.assign(l0, l0 + n_bars);
        HM_HIP(hipMemcpyAsync(h->d_n4l0, h->n4_l0.data(), (size_t)n_bars * sizeof(double), hipMemcpyHostToDevice, h->stream3));
        HM_HIP(hipStreamSynchronize(h->stream3));
    }
    const size_t n4 = (size_t)4 * N;
    memcpy(h->pin_n4, X, n4 * sizeof(double));
    Newton4Args a;
    a.N = N; a.I = n_bars; a.deg_stride = h->n4deg;
    a.bars = h->d_n4bars; a.l0 = h->d_n4l0; a.nbr = h->d_n4nbr; a.nbb = h->d_n4nbb;
    a.kappa = kappa; a.M = M; a.dt = dt; a.tol = tol; a.maxiter = maxiter; a.steps = (int)std::ceil(1.0 / dt);
    a.Xin = h->pin_n4; a.out = h->pin_n4 + n4; a.dev_out = h->d_n4X; a.ticket = (double)(++h->n4_ticket);
    a.delay_us = h->result_delay; a.force_bad = h->newton_fail;
    const size_t lds = ((size_t)6 * NEWTON4_NT + 4 * ((size_t)n_bars + 1) + 4 * (NEWTON4_NT / 64)) * sizeof(double) + 2 * (size_t)n_bars * sizeof(int);
    if (lds > 64 * 1024) return 1;
    if (h->n4deg == 8) hipLaunchKernelGGL((k_ms_newton4<8>), dim3(1), dim3(NEWTON4_NT), lds, h->stream3, a);
    else hipLaunchKernelGGL((k_ms_newton4<12>), dim3(1), dim3(NEWTON4_NT), lds, h->stream3, a);
    HM_HIP(hipGetLastError());
    HM_HIP(hipEventRecord(h->ev_n4, h->stream3));
    h->n4_pending = true;
    h->chain_pending = false;
    return HM_OK;
}

// Waits for the launch of hm_newton_dev_start; X (4N) receives the advanced state.  Returns 1 when the kernel's inner solve
// did not converge (never observed): the caller repeats the prediction on the host.
extern "C" int hm_newton_dev_finish(hm_ctx_t h, double *X, int *newton_iterations)
{
    HM_ARG(h && X, "hm_newton_dev_finish: bad argument");
    if (!h->n4_pending) { hm_set_error("hm_newton_dev_finish: no prediction was started"); return HM_ERR_STATE; }
    HM_HIP(hipSetDevice(h->device));
    const size_t n4 = (size_t)4 * h->N;
    // the kernel's result block, taken when it is whole (host_block.h); the kernel is usually long done when this is
    // called (it was started at the end of the previous update)
    const int rc = hb_wait(h->stream3, h->pin_n4 + n4, 0, n4 + 2, hb_stamp(h->n4_ticket), h->n4v.data(), "hm_newton_dev_finish");
    h->n4_pending = false;
    h->chain_pending = false;
    if (rc) return rc;
    if (h->n4v[n4 + 1] != 0.0) return 1;
    memcpy(X, h->n4v.data(), n4 * sizeof(double));
    if (newton_iterations) *newton_iterations = (int)h->n4v[n4];
    return HM_OK;
}


// ---- the state path between two frames without host round trips ------------------------------------------------------
// compute() (kalman.py:676-700) is predict -> projectmask -> update.  With the state prediction started on the device
// by the update that precedes it (hm_newton_dev_start) every step's input is in device memory before the host needs
// to know it: hm_chain_project queues projectmask (kalman.py:724-742, mask of the observation in place) of the
// prediction in flight behind its kernel -- k_project_mask_host reads the kernel's device copy of the predicted state
// and leaves the projected state in d_X0, where the update keeps its prior mean -- and marks the handle: the next
// hm_update_run starts from d_X0 as it is (its X argument is output only), waits on the device for the projection
// instead of uploading a state, and collects the two kernels' result blocks (predicted state, Newton iterations,
// projected state, vertices moved: hm_chain_states) while its first iteration runs.  The numbers are those of the
// three separate calls.  Returns 1 when there is nothing to chain (no prediction in flight): the caller takes the
// three calls.
extern "C" int hm_chain_project(hm_ctx_t h)
{
    HM_ARG(h != nullptr, "hm_chain_project: NULL handle");
    HM_JOIN_LAZY(h);
    NEED_OBS(h, "hm_chain_project");
    if (!h->n4_pending || !h->d_n4X) return 1;
    HM_HIP(hipSetDevice(h->device));
    int rc = ensure_stream2(h);
    if (rc) return rc;
    rc = project_buffers(h);
    if (rc) return rc;
    if (!h->outline_ready) {
        rc = queue_outline(h, h->o_ym);
        if (rc) return rc;
        h->outline_ready = true;
    }
    const size_t n4 = (size_t)4 * h->N;
    Outline o = {h->d_outline, h->d_outline_cnt, h->W * h->H, h->d_pm_flag};
    ProjArgs a = {h->d_pm_pruned, h->W, h->H, h->N, o, nullptr};
    // on the prediction's own stream, right behind its kernel (no hop between streams there); the outline of the mask was
    // queued on the second stream when the observation was set, or a frame ahead (hm_prepare_mask)
    HM_HIP(hipStreamWaitEvent(h->stream3, h->ev_outline, 0));
    hipLaunchKernelGGL(k_project_mask_host, dim3(h->N), dim3(PROJ_NT), 0, h->stream3, a, (const double *)h->d_n4X, h->pin_pm + n4, h->d_X0,
                       h->d_X, h->d_pm_done, (double)(++h->pm_ticket), h->result_delay);
    HM_HIP(hipGetLastError());
    HM_HIP(hipEventRecord(h->ev_pm, h->stream3));
    h->chain_pending = true;
    return HM_OK;
}

// The pruning + outline of a mask in device memory queued AHEAD of the hm_set_observation_dev that will name it (a
// streaming caller has the next frame's mask in device memory a frame early): ~0.2 ms of kernels on the second stream
// that would otherwise start when the next frame does and keep its projection waiting.  Queued behind whatever the
// second stream still has to do with the current outline.  The mask must not change until that hm_set_observation_dev;
// any other observation, or a projection onto a host mask, simply discards the preparation.
extern "C" int hm_prepare_mask(hm_ctx_t h, const uint8_t *d_y_m)
{
    HM_ARG(h && d_y_m, "hm_prepare_mask: NULL argument");
    HM_JOIN_LAZY(h);
    return prepare_mask(h, d_y_m);
}
static int prepare_mask(hm_ctx *h, const uint8_t *d_y_m)
{
    HM_HIP(hipSetDevice(h->device));
    int rc = ensure_stream2(h);
    if (rc) return rc;
    rc = project_buffers(h);
    if (rc) return rc;
    // the projection of the CURRENT frame may still be reading the outline buffers on the prediction's stream
    if (h->ev_pm && h->pm_ticket > 0) HM_HIP(hipStreamWaitEvent(h->stream2, h->ev_pm, 0));
    rc = queue_outline(h, d_y_m);
    if (rc) return rc;
    h->outline_ready = false;                    // (the buffers no longer hold the outline of the observation in place)
    h->prepared_mask = d_y_m;
    return HM_OK;
}

// What the last chained hm_update_run started from: the predicted state, the projected state (its prior mean), the
// Newton iterations of the prediction, the vertices projectmask moved.  Any pointer may be NULL.
extern "C" int hm_chain_states(hm_ctx_t h, double *predicted, double *projected, int *newton_iterations, int *moved)
{
    HM_ARG(h != nullptr, "hm_chain_states: NULL handle");
    const size_t n4 = (size_t)4 * h->N;
    if (h->chain_pred.size() != n4 || h->chain_proj.size() != n4) { hm_set_error("hm_chain_states: no chained update has run"); return HM_ERR_STATE; }
    if (predicted) memcpy(predicted, h->chain_pred.data(), n4 * sizeof(double));
    if (projected) memcpy(projected, h->chain_proj.data(), n4 * sizeof(double));
    if (newton_iterations) *newton_iterations = h->chain_its;
    if (moved) *moved = h->chain_moved;
    return HM_OK;
}
