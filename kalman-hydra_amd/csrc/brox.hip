// Brox optical flow on MI355X: host orchestration and C-ABI (include/hydra_mi.h).
// Replaces cv::cuda::BroxOpticalFlow as called by processflow_gpu
// (reference src/optical_flow_ext.cpp:294-331).
#define HM_ALLOC_CLASS 2          // hm_malloc (HYDRA_MI_POISON): a flow handle's buffers
#include "hm_common.h"
#include <hip/hip_ext.h>
#include "brox_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

// ---- error plumbing (shared by the whole library) --------------------------------
static thread_local char g_err[512] = "";
void hm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *hm_last_error(void) { return g_err; }
extern "C" const char *hm_version(void) { return "hydra_mi 0.1 (gfx950)"; }
extern "C" int hm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        hm_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return HM_ERR_HIP;
    }
    return n;
}

// ---- device buffers of a caller that keeps its frames / flow planes in HBM --------------------------
extern "C" int hm_dev_alloc(int device, uint64_t bytes, void **out)
{
    HM_ARG(out != nullptr && bytes > 0, "hm_dev_alloc: bad argument");
    *out = nullptr;
    HM_HIP(hipSetDevice(device));
    HM_HIP(hm_malloc(out, (size_t)bytes, 4));
    return HM_OK;
}
extern "C" int hm_dev_free(int device, void *ptr)
{
    if (!ptr) return HM_OK;
    HM_HIP(hipSetDevice(device));
    HM_HIP(hipFree(ptr));
    return HM_OK;
}
// The synchronous copies go over a non-blocking stream of their own, one per device: a copy on the null stream would
// wait for every blocking stream of the device -- a flow handle's CU-masked stream (hipExtStreamCreateWithCUMask has no
// non-blocking form) with a whole launch series queued on it.
static hipStream_t xfer_stream(int device)
{
    static std::mutex mu;
    static std::vector<hipStream_t> streams;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64) return nullptr;
    if ((int)streams.size() <= device) streams.resize(device + 1, nullptr);
    if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) streams[device] = nullptr;
    return streams[device];
}
extern "C" int hm_dev_upload(int device, void *dst, const void *src, uint64_t bytes)
{
    HM_ARG(dst && src, "hm_dev_upload: NULL pointer");
    HM_HIP(hipSetDevice(device));
    hipStream_t s = xfer_stream(device);
    HM_ARG(s != nullptr, "hm_dev_upload: no copy stream on device %d", device);
    HM_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, s));
    HM_HIP(hipStreamSynchronize(s));
    return HM_OK;
}
extern "C" int hm_dev_download(int device, void *dst, const void *src, uint64_t bytes)
{
    HM_ARG(dst && src, "hm_dev_download: NULL pointer");
    HM_HIP(hipSetDevice(device));
    hipStream_t s = xfer_stream(device);
    HM_ARG(s != nullptr, "hm_dev_download: no copy stream on device %d", device);
    HM_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    return HM_OK;
}

// ---- streaming uploads: a copy stream and pinned staging memory of the caller --------------------------
// (frame k+B and its mask go to the device while the flow series k .. k+B-1 runs; pipeline.py)
extern "C" int hm_copy_stream_create(int device, void **out)
{
    HM_ARG(out != nullptr, "hm_copy_stream_create: out is NULL");
    *out = nullptr;
    HM_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;
    HM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void *)s;
    return HM_OK;
}
extern "C" int hm_copy_stream_destroy(int device, void *stream)
{
    if (!stream) return HM_OK;
    HM_HIP(hipSetDevice(device));
    HM_HIP(hipStreamSynchronize((hipStream_t)stream));
    HM_HIP(hipStreamDestroy((hipStream_t)stream));
    return HM_OK;
}
extern "C" int hm_copy_stream_sync(int device, void *stream)
{
    HM_ARG(stream != nullptr, "hm_copy_stream_sync: NULL stream");
    HM_HIP(hipSetDevice(device));
    HM_HIP(hipStreamSynchronize((hipStream_t)stream));
    return HM_OK;
}
extern "C" int hm_host_alloc(uint64_t bytes, void **out)
{
    HM_ARG(out != nullptr && bytes > 0, "hm_host_alloc: bad argument");
    *out = nullptr;
    HM_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return HM_OK;
}
extern "C" int hm_host_free(void *ptr)
{
    if (!ptr) return HM_OK;
    HM_HIP(hipHostFree(ptr));
    return HM_OK;
}
extern "C" int hm_dev_upload_async(int device, void *dst, const void *src, uint64_t bytes, void *stream)
{
    HM_ARG(dst && src && stream, "hm_dev_upload_async: NULL pointer");
    HM_HIP(hipSetDevice(device));
    HM_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return HM_OK;
}

// ---- geometry -----------------------------------------------------------------------
static Geo make_geo(int w, int h)
{
    Geo g;
    g.w = w; g.h = h;
    g.pitch = (w + 15) / 16 * 16;
    g.plane = (g.pitch * h + 63) / 64 * 64;
    return g;
}

// level k is ceil(W s^k) x ceil(H s^k) (s accumulated in float); levels are added while
// the previous one exceeds 15 px on both sides and fewer than `outer` exist
static int make_levels(int W, int H, float scale, int outer, std::vector<Geo> &out)
{
    out.clear();
    out.push_back(make_geo(W, H));
    float sc = 1.0f;
    while (out.back().w > 15 && out.back().h > 15 && (int)out.size() < outer && out.size() < 128) {
        sc = sc * scale;
        int w = (int)ceilf((float)W * sc), h = (int)ceilf((float)H * sc);
        if (w < 1) w = 1;
        if (h < 1) h = 1;
        out.push_back(make_geo(w, h));
    }
    return (int)out.size();
}

// sigma = 0.6 sqrt(1/s^2 - 1), R = ceil(3 sigma) >= 1, taps normalised in double
static Taps make_taps(float scale)
{
    Taps t;
    double sigma = 0.6 * sqrt(1.0 / ((double)scale * (double)scale) - 1.0);
    int R = (int)ceil(3.0 * sigma);
    if (R < 1) R = 1;
    if (R > (BROX_MAX_TAPS - 1) / 2) R = (BROX_MAX_TAPS - 1) / 2;
    double tmp[BROX_MAX_TAPS], sum = 0.0;
    for (int i = -R; i <= R; i++) {
        tmp[i + R] = exp(-(double)(i * i) / (2.0 * sigma * sigma));
        sum += tmp[i + R];
    }
    memset(t.g, 0, sizeof t.g);
    for (int i = 0; i <= 2 * R; i++) t.g[i] = (float)(tmp[i] / sum);
    t.R = R;
    return t;
}

static dim3 grid2d(const Geo &g, int n) { return dim3(hm_cdiv(g.w, 64), hm_cdiv(g.h, 4), n); }
static const dim3 kBlock2d(64, 4, 1);

// ---- SOR launch ------------------------------------------------------------------------
#define SOR_TW 64
#define SOR_TH 64

struct SorPlan {
    int K;                      // iterations per launch
    int threads;                // 256 (8 rows per thread) or 512 (4 rows per thread)
    int tw;                     // tile width: SOR_TW, or 2 * SOR_TW (1024 threads, 4 rows per thread: the halo's share of a
                                // tile drops from 2.12 to 1.72 of its interior at K = 5)
    int tiles_x, tiles_y, step_x, step_y, halo_x, halo_y;
};

// iterations fused per launch: everything when the level fits one tile (no halo
// needed); otherwise the largest divisor of `solver` up to 5 (halo 2K = 10 px of a
// 64-px tile is where redundant work starts to outweigh the saved traffic)
static SorPlan sor_plan(const Geo &g, int solver, int fuse, int threads, int n = 1, long long slots = 0, int wide = 0)
{
    SorPlan p;
    p.threads = threads;
    // the wide tile: levels of at least `wide` pixels per side (0: never)
    p.tw = (wide > 0 && g.w >= wide && g.h >= wide) ? 2 * SOR_TW : SOR_TW;
    if (p.tw != SOR_TW) p.threads = threads = 1024;
    const int TWp = p.tw;
    bool fitx = g.w <= TWp, fity = g.h <= SOR_TH;
    int K;
    if (fuse > 0) K = fuse;
    else if (fitx && fity) K = solver;
    else {
        K = 1;
        for (int d = 1; d <= 5 && d <= solver; d++)
            if (solver % d == 0) K = d;
        // A level whose tiles do not fill the chip even with the halo of all `solver` iterations (a small interior:
        // many more tiles) is bound by the latency of one launch after the other, not by work: one launch instead
        // of solver / K (a dependent launch costs ~8 us on the device whatever it does).
        const int step = TWp - 4 * solver;
        if (step >= 16 && solver <= 15) {
            const long long tiles = (long long)(fitx ? 1 : hm_cdiv(g.w, step)) * (fity ? 1 : hm_cdiv(g.h, SOR_TH - 4 * solver)) * n;
            if (tiles * threads <= slots * 1024) K = solver;
        }
    }
    p.K = K;
    p.halo_x = fitx ? 0 : 2 * K;
    p.halo_y = fity ? 0 : 2 * K;
    p.step_x = TWp - 2 * p.halo_x;
    p.step_y = SOR_TH - 2 * p.halo_y;
    p.tiles_x = fitx ? 1 : hm_cdiv(g.w, p.step_x);
    p.tiles_y = fity ? 1 : hm_cdiv(g.h, p.step_y);
    return p;
}

// e0 / e1 (profiling only): events that receive the start and stop time of the kernel itself
// (hipExtLaunchKernelGGL) -- events recorded around the launch would add the command processor's
// dispatch latency to every launch.  Without events the plain launch is used.
static void sor_launch(const SorPlan &p, SorArgs a, int n, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y;
    a.step_x = p.step_x; a.step_y = p.step_y;
    a.halo_x = p.halo_x; a.halo_y = p.halo_y;
    dim3 grid(p.tiles_x * p.tiles_y, 1, n);
    if (p.tw == 2 * SOR_TW) {
        if (e0 && e1) hipExtLaunchKernelGGL((k_sor<2 * SOR_TW, SOR_TH, 1024>), grid, dim3(1024), 0, s, e0, e1, 0, a, p.K);
        else hipLaunchKernelGGL((k_sor<2 * SOR_TW, SOR_TH, 1024>), grid, dim3(1024), 0, s, a, p.K);
        return;
    }
    if (e0 && e1) {
        if (p.threads == 1024)
            hipExtLaunchKernelGGL((k_sor<SOR_TW, SOR_TH, 1024>), grid, dim3(1024), 0, s, e0, e1, 0, a, p.K);
        else if (p.threads == 512)
            hipExtLaunchKernelGGL((k_sor<SOR_TW, SOR_TH, 512>), grid, dim3(512), 0, s, e0, e1, 0, a, p.K);
        else
            hipExtLaunchKernelGGL((k_sor<SOR_TW, SOR_TH, 256>), grid, dim3(256), 0, s, e0, e1, 0, a, p.K);
        return;
    }
    if (p.threads == 1024)
        hipLaunchKernelGGL((k_sor<SOR_TW, SOR_TH, 1024>), grid, dim3(1024), 0, s, a, p.K);
    else if (p.threads == 512)
        hipLaunchKernelGGL((k_sor<SOR_TW, SOR_TH, 512>), grid, dim3(512), 0, s, a, p.K);
    else
        hipLaunchKernelGGL((k_sor<SOR_TW, SOR_TH, 256>), grid, dim3(256), 0, s, a, p.K);
}

// ---- handle ------------------------------------------------------------------------------
struct hm_brox {
    int device, W, H, B;
    float alpha, gamma, scale, omega;
    int inner, outer, solver, fuse, sor_threads;
    int sor_dry;                 // development knob: SOR launches load and store but do not iterate (wrong results)
    int cus;                     // compute units of the device
    int sor_deep;                // levels with few tiles take all solver iterations in one launch (sor_plan)
    int sor_wide;                // levels of at least this many pixels per side use the 128 x 64 tile (0: none)
    int coarse_stagger;          // test knob: the pairs of a k_coarse launch start one after the other
    int coarse_max;              // levels up to this many px per side run inside k_coarse: 0 (none), 32 or 64
    std::vector<Geo> geo;
    Taps taps;
    hipStream_t stream;                 // the stream launches go to: `whole` or `masked`
    hipStream_t whole = nullptr, masked = nullptr;
    // lanes = 2 (experiment, off): a call of four or more pairs on the CU-masked stream runs as two halves side by side,
    // the second on a twin of that stream.  The launches of a series alternate between kernels bound by HBM (k_prepare,
    // the fill of a SOR tile) and by the CU (the sweeps); two halves at different places of that sequence fill each
    // other's gaps -- two handles with 4 pairs each: 14.1 instead of 15.5 ms per 8 x 1024^2 pairs, the two lanes of one
    // call, which end together: 15.0.  Beside the filter it costs more than it gives (the filter's workgroups find
    // fewer free places between two queues of flow launches): 229 against 261 frames/s at 20 frames, 241 against 323
    // at 64.  Same bits: a pair's flow does not depend on what it shares a launch with.
    int lanes = 1;
    hipStream_t masked2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float *arena;
    size_t arena_floats;
    std::vector<float *> pyr0, pyr1;
    float *tmpA, *tmpB;
    float *Ix0, *Iy0, *I1x, *I1y, *I1xx, *I1xy, *I1yy;
    float *Iz, *Ix, *Iy, *Ixz, *Iyz, *Ixx, *Ixy, *Iyy;
    float *nu, *nv, *a12, *idu, *idv, *sx, *sy;
    float *u, *v, *u2, *v2, *du[2], *dv[2];
    float *zero;                 // a plane of zeros, only ever read
    uint8_t *d_f0, *d_f1;        // staging of host frames
    float *d_ox, *d_oy;          // staging of the host-bound result (tight W*H per pair)
    bool warp_window;                     // k_warp stages its taps as an LDS window (hm_brox_tune)
    // profiling
    bool prof;
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t ev_used;
    double prof_ms, prof_pxit;
    long long prof_launches;
    double prof_px;
    std::vector<double> ev_pxit, ev_px;
    std::vector<int> ev_level;   // pyramid level of every recorded launch
    std::vector<double> lev_ms, lev_pxit, lev_px;       // per-level totals (hm_brox_profile_levels)
    std::vector<long long> lev_launches;
};

static int brox_free(hm_brox *h)
{
    if (!h) return HM_OK;
    hipSetDevice(h->device);
    for (hipEvent_t e : h->ev) hipEventDestroy(e);
    if (h->arena) hipFree(h->arena);
    if (h->d_f0) hipFree(h->d_f0);
    if (h->d_f1) hipFree(h->d_f1);
    if (h->d_ox) hipFree(h->d_ox);
    if (h->d_oy) hipFree(h->d_oy);
    if (h->whole) hipStreamDestroy(h->whole);
    if (h->masked) hipStreamDestroy(h->masked);
    if (h->masked2) hipStreamDestroy(h->masked2);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    delete h;
    return HM_OK;
}

extern "C" int hm_brox_create(int device, int W, int H, int max_batch, float alpha, float gamma,
                              float scale, int inner, int outer, int solver, hm_brox_t *out)
{
    HM_ARG(out != nullptr, "hm_brox_create: out is NULL");
    *out = nullptr;
    HM_ARG(W >= 1 && H >= 1 && W <= 16384 && H <= 16384, "hm_brox_create: bad size %dx%d", W, H);
    HM_ARG(max_batch >= 1 && max_batch <= 4096, "hm_brox_create: bad max_batch %d", max_batch);
    HM_ARG(scale > 0.0f && scale < 1.0f, "hm_brox_create: scale_factor must be in (0,1), got %g", scale);
    HM_ARG(inner >= 1 && outer >= 1 && solver >= 1, "hm_brox_create: iteration counts must be >= 1");
    HM_ARG(alpha > 0.0f && gamma >= 0.0f, "hm_brox_create: alpha must be > 0 and gamma >= 0");
    HM_HIP(hipSetDevice(device));
    hm_brox *h = new hm_brox();
    h->device = device; h->W = W; h->H = H; h->B = max_batch;
    h->alpha = alpha; h->gamma = gamma; h->scale = scale; h->omega = 1.99f;
    h->inner = inner; h->outer = outer; h->solver = solver; h->fuse = 0; h->sor_threads = 0; h->sor_dry = 0;
    h->coarse_max = 32; h->sor_deep = 1; h->cus = 0; h->coarse_stagger = 0; h->sor_wide = 0;
    if (hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) h->cus = 0;
    h->arena = nullptr; h->d_f0 = h->d_f1 = nullptr; h->d_ox = h->d_oy = nullptr; h->stream = nullptr;
    h->prof = false; h->ev_used = 0; h->prof_ms = 0; h->prof_pxit = 0; h->prof_px = 0; h->prof_launches = 0;
    h->warp_window = false;
    make_levels(W, H, scale, outer, h->geo);
    h->taps = make_taps(scale);

    const size_t B = (size_t)max_batch;
    const size_t plane0 = (size_t)h->geo[0].plane;
    size_t pyr_floats = 0;
    for (const Geo &g : h->geo) pyr_floats += (size_t)g.plane * B;
    const int nfields = 2 + 7 + 8 + 7 + 4 + 4 + 1;
    h->arena_floats = 2 * pyr_floats + (size_t)nfields * plane0 * B;
    hipError_t e = hm_malloc((void **)&h->arena, h->arena_floats * sizeof(float));
    if (e != hipSuccess) {
        hm_set_error("hm_brox_create: hipMalloc of %zu MiB failed: %s", h->arena_floats * 4 >> 20, hipGetErrorString(e));
        brox_free(h);
        return HM_ERR_HIP;
    }
    // padding columns are read (never used) by float2 loads: keep them finite.  The fill
    // goes on the handle's own stream: that stream is non-blocking, so a fill issued on the
    // null stream could still be running when the first calc starts.
    // (Tried and taken back: this stream as a CU-masked one with every CU in its mask, which has a hardware queue of its
    // own instead of a place among the runtime's four -- the benches ran as before, the native flow tool, which has no
    // other stream in its process, hung in its first series.)
    e = hipStreamCreateWithFlags(&h->whole, hipStreamNonBlocking);
    h->stream = h->whole;
    if (e == hipSuccess) e = hipMemsetAsync(h->arena, 0, h->arena_floats * sizeof(float), h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hm_malloc((void **)&h->d_f0, B * W * H);
    if (e == hipSuccess) e = hm_malloc((void **)&h->d_f1, B * W * H);
    if (e == hipSuccess) e = hm_malloc((void **)&h->d_ox, B * W * H * sizeof(float));
    if (e == hipSuccess) e = hm_malloc((void **)&h->d_oy, B * W * H * sizeof(float));
    if (e != hipSuccess) {
        hm_set_error("hm_brox_create: device setup failed: %s", hipGetErrorString(e));
        brox_free(h);
        return HM_ERR_HIP;
    }
    float *p = h->arena;
    auto take = [&](size_t n) { float *r = p; p += n; return r; };
    for (const Geo &g : h->geo) h->pyr0.push_back(take((size_t)g.plane * B));
    for (const Geo &g : h->geo) h->pyr1.push_back(take((size_t)g.plane * B));
    float **fields[] = {&h->tmpA, &h->tmpB, &h->Ix0, &h->Iy0, &h->I1x, &h->I1y, &h->I1xx, &h->I1xy, &h->I1yy,
                        &h->Iz, &h->Ix, &h->Iy, &h->Ixz, &h->Iyz, &h->Ixx, &h->Ixy, &h->Iyy,
                        &h->nu, &h->nv, &h->a12, &h->idu, &h->idv, &h->sx, &h->sy,
                        &h->u, &h->v, &h->u2, &h->v2, &h->du[0], &h->du[1], &h->dv[0], &h->dv[1], &h->zero};
    static_assert(sizeof(fields) / sizeof(fields[0]) == 33, "field count");
    for (float **f : fields) *f = take(plane0 * B);
    *out = h;
    return HM_OK;
}

extern "C" int hm_brox_destroy(hm_brox_t h) { return brox_free(h); }

extern "C" int hm_brox_levels(hm_brox_t h, int *ws, int *hs, int cap)
{
    HM_ARG(h != nullptr, "hm_brox_levels: NULL handle");
    int n = (int)h->geo.size();
    for (int i = 0; i < n && i < cap; i++) {
        if (ws) ws[i] = h->geo[i].w;
        if (hs) hs[i] = h->geo[i].h;
    }
    return n;
}

extern "C" int hm_brox_set_omega(hm_brox_t h, float omega)
{
    HM_ARG(h != nullptr, "hm_brox_set_omega: NULL handle");
    HM_ARG(omega > 0.0f && omega < 2.0f, "hm_brox_set_omega: omega must be in (0,2), got %g", omega);
    h->omega = omega;
    return HM_OK;
}

extern "C" int hm_brox_tune(hm_brox_t h, const char *key, int value)
{
    HM_ARG(h != nullptr && key != nullptr, "hm_brox_tune: NULL argument");
    if (!strcmp(key, "sor_fuse")) {
        HM_ARG(value == 0 || (value >= 1 && value <= 10 && h->solver % value == 0),
               "hm_brox_tune: sor_fuse=%d must be 0 or a divisor of solver_iterations (%d) not above 10", value, h->solver);
        h->fuse = value;
    } else if (!strcmp(key, "sor_threads")) {
        HM_ARG(value == 0 || value == 256 || value == 512 || value == 1024, "hm_brox_tune: sor_threads must be 0 (choose per call), 256, 512 or 1024");
        h->sor_threads = value;
    } else if (!strcmp(key, "sor_dry")) {            // timing experiments only (tools/): the flow is wrong with 1
        h->sor_dry = value != 0;
    } else if (!strcmp(key, "cu_reserve")) {
        // The handle's stream leaves `value` compute units alone (a CU mask on its queue): launches of other streams
        // -- the filter's chain of short dependent kernels -- find room at once while a flow series fills the rest.
        HM_ARG(value >= 0 && value < h->cus, "hm_brox_tune: cu_reserve must be in 0..%d", h->cus - 1);
        HM_HIP(hipSetDevice(h->device));
        HM_HIP(hipStreamSynchronize(h->stream));
        if (h->masked) { HM_HIP(hipStreamDestroy(h->masked)); h->masked = nullptr; }
        if (h->masked2) { HM_HIP(hipStreamDestroy(h->masked2)); h->masked2 = nullptr; }
        if (value > 0) {
            const int words = (h->cus + 31) / 32;
            std::vector<uint32_t> mask(words, 0u);
            for (int i = 0; i < h->cus - value; i++) mask[i / 32] |= 1u << (i % 32);
            HM_HIP(hipExtStreamCreateWithCUMask(&h->masked, (uint32_t)words, mask.data()));
            if (h->lanes >= 2) HM_HIP(hipExtStreamCreateWithCUMask(&h->masked2, (uint32_t)words, mask.data()));
        }
        h->stream = h->masked ? h->masked : h->whole;
    } else if (!strcmp(key, "whole_chip")) {
        // 1: the next calls use all compute units whatever cu_reserve says (a series that runs with nothing beside it:
        // the first of a phase of the streaming pipeline); 0: back to the masked stream.  Calls of one handle do not
        // overlap, and the stream that is left is drained first.
        HM_ARG(value == 0 || value == 1, "hm_brox_tune: whole_chip must be 0 or 1");
        hipStream_t to = (value || !h->masked) ? h->whole : h->masked;
        if (to != h->stream) {
            HM_HIP(hipSetDevice(h->device));
            HM_HIP(hipStreamSynchronize(h->stream));
            h->stream = to;
        }
    } else if (!strcmp(key, "lanes")) {              // same bits either way
        // (before "cu_reserve": the second lane's stream gets the mask the first one has.  It exists only in a handle
        // that asked for it, and only as a CU-masked stream, which has a hardware queue of its own: a plain stream takes
        // one of the FOUR queues the runtime spreads all plain streams of a priority over (GPU_MAX_HW_QUEUES), and this
        // process has four already -- the null stream, the copy stream of hm_dev_upload, the frame ring's, the handle's.
        // A fifth, even idle, made the ring's uploads share a queue: 173 instead of 261 frames/s)
        HM_ARG(value == 1 || value == 2, "hm_brox_tune: lanes must be 1 or 2");
        HM_ARG(value == 1 || h->masked == nullptr, "hm_brox_tune: lanes must be set before cu_reserve");
        HM_HIP(hipSetDevice(h->device));
        if (value == 2 && !h->ev_fork) {
            HM_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HM_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        h->lanes = value;
    } else if (!strcmp(key, "sor_wide")) {           // same bits either way
        HM_ARG(value >= 0 && value <= 8192, "hm_brox_tune: sor_wide must be 0 (never) or the smallest level side that uses the 128 x 64 tile");
        h->sor_wide = value;
    } else if (!strcmp(key, "sor_deep")) {
        HM_ARG(value >= 0 && value <= 8, "hm_brox_tune: sor_deep must be 0 .. 8");
        h->sor_deep = value;
    } else if (!strcmp(key, "coarse_stagger")) {     // tests only: results must not depend on it
        h->coarse_stagger = value != 0;
    } else if (!strcmp(key, "coarse_max")) {
        HM_ARG(value == 0 || value == 32 || value == 64, "hm_brox_tune: coarse_max must be 0, 32 or 64");
        h->coarse_max = value;
    } else if (!strcmp(key, "warp_window")) {
        HM_ARG(value == 0 || value == 1, "hm_brox_tune: warp_window must be 0 or 1");
        h->warp_window = value != 0;
    } else {
        hm_set_error("hm_brox_tune: unknown key '%s'", key);
        return HM_ERR_ARG;
    }
    return HM_OK;
}

extern "C" void *hm_brox_stream(hm_brox_t h) { return h ? (void *)h->stream : nullptr; }

extern "C" int hm_brox_sync(hm_brox_t h)
{
    HM_ARG(h != nullptr, "hm_brox_sync: NULL handle");
    HM_HIP(hipSetDevice(h->device));
    HM_HIP(hipStreamSynchronize(h->stream));
    return HM_OK;
}

extern "C" int hm_brox_profile(hm_brox_t h, int enable)
{
    HM_ARG(h != nullptr, "hm_brox_profile: NULL handle");
    HM_HIP(hipSetDevice(h->device));
    h->prof = enable != 0;
    if (enable) {                                  // switching off keeps what was recorded for hm_brox_profile_read
        h->ev_used = 0; h->prof_ms = 0; h->prof_pxit = 0; h->prof_px = 0; h->prof_launches = 0;
        h->ev_pxit.clear(); h->ev_px.clear(); h->ev_level.clear();
        const size_t L = h->geo.size();
        h->lev_ms.assign(L, 0.0); h->lev_pxit.assign(L, 0.0); h->lev_px.assign(L, 0.0); h->lev_launches.assign(L, 0);
    }
    return HM_OK;
}

// fold the recorded event pairs into the running totals (stream must be idle)
static int prof_collect(hm_brox *h)
{
    for (size_t i = 0; i < h->ev_used; i += 2) {
        float ms = 0.0f;
        HM_HIP(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        h->prof_ms += ms;
        h->prof_pxit += h->ev_pxit[i / 2];
        h->prof_px += h->ev_px[i / 2];
        h->prof_launches++;
        const size_t lv = (size_t)h->ev_level[i / 2];
        if (lv < h->lev_ms.size()) {
            h->lev_ms[lv] += ms; h->lev_pxit[lv] += h->ev_pxit[i / 2]; h->lev_px[lv] += h->ev_px[i / 2]; h->lev_launches[lv]++;
        }
    }
    h->ev_used = 0;
    h->ev_pxit.clear(); h->ev_px.clear(); h->ev_level.clear();
    return HM_OK;
}

// the same totals per pyramid level (0 = the full frame) since profiling was switched on; returns the number of levels
extern "C" int hm_brox_profile_levels(hm_brox_t h, int cap, double *ms, long long *launches, double *pxit, double *px)
{
    HM_ARG(h != nullptr, "hm_brox_profile_levels: NULL handle");
    HM_HIP(hipSetDevice(h->device));
    HM_HIP(hipStreamSynchronize(h->stream));
    int rc = prof_collect(h);
    if (rc) return rc;
    const int L = (int)h->lev_ms.size();
    for (int k = 0; k < L && k < cap; k++) {
        if (ms) ms[k] = h->lev_ms[k];
        if (launches) launches[k] = h->lev_launches[k];
        if (pxit) pxit[k] = h->lev_pxit[k];
        if (px) px[k] = h->lev_px[k];
    }
    return L;
}

extern "C" int hm_brox_profile_read(hm_brox_t h, double *ms, long long *launches, double *pxit, double *px)
{
    HM_ARG(h != nullptr, "hm_brox_profile_read: NULL handle");
    HM_HIP(hipSetDevice(h->device));
    HM_HIP(hipStreamSynchronize(h->stream));
    int rc = prof_collect(h);
    if (rc) return rc;
    if (ms) *ms = h->prof_ms;
    if (launches) *launches = h->prof_launches;
    if (pxit) *pxit = h->prof_pxit;
    if (px) *px = h->prof_px;
    h->prof_ms = 0; h->prof_pxit = 0; h->prof_px = 0; h->prof_launches = 0;
    return HM_OK;
}

// ---- the pipeline ---------------------------------------------------------------------------
// Pairs z0 .. z0 + n - 1 of a call, queued on stream s.  Every working plane holds max_batch pairs, pair z of a level
// at z x (that level's plane): a lane that starts at pair z0 works in the same buffers from z0 x (plane of level 0) on
// -- room for its n pairs at every level, and clear of the lane in front of it, which has at most z0 pairs.
static int brox_run(hm_brox *h, int z0, int n, hipStream_t s, const uint8_t *d_f0, const uint8_t *d_f1, float *d_ox, float *d_oy)
{
    const size_t fo = (size_t)z0 * h->geo[0].plane;
    auto P0 = [&](int k) { return h->pyr0[k] + (size_t)z0 * h->geo[k].plane; };
    auto P1 = [&](int k) { return h->pyr1[k] + (size_t)z0 * h->geo[k].plane; };
    float *const tmpA = h->tmpA + fo, *const tmpB = h->tmpB + fo;
    float *const Ix0 = h->Ix0 + fo, *const Iy0 = h->Iy0 + fo, *const I1x = h->I1x + fo, *const I1y = h->I1y + fo;
    float *const I1xx = h->I1xx + fo, *const I1xy = h->I1xy + fo, *const I1yy = h->I1yy + fo;
    float *const Iz = h->Iz + fo, *const Ix = h->Ix + fo, *const Iy = h->Iy + fo, *const Ixz = h->Ixz + fo, *const Iyz = h->Iyz + fo;
    float *const Ixx = h->Ixx + fo, *const Ixy = h->Ixy + fo, *const Iyy = h->Iyy + fo;
    float *const c_nu = h->nu + fo, *const c_nv = h->nv + fo, *const c_a12 = h->a12 + fo, *const c_idu = h->idu + fo;
    float *const c_idv = h->idv + fo, *const c_sx = h->sx + fo, *const c_sy = h->sy + fo;
    float *const b_u = h->u + fo, *const b_v = h->v + fo, *const b_u2 = h->u2 + fo, *const b_v2 = h->v2 + fo;
    float *const b_du[2] = {h->du[0] + fo, h->du[1] + fo}, *const b_dv[2] = {h->dv[0] + fo, h->dv[1] + fo};
    const float *const zero = h->zero + fo;
    d_f0 += (size_t)z0 * h->W * h->H; d_f1 += (size_t)z0 * h->W * h->H;
    d_ox += (size_t)z0 * h->W * h->H; d_oy += (size_t)z0 * h->W * h->H;
    const int L = (int)h->geo.size();
    const Geo &g0 = h->geo[0];
    hipLaunchKernelGGL(k_u8_to_f32, grid2d(g0, 2 * n), kBlock2d, 0, s, d_f0, d_f1, h->W, h->W * h->H, P0(0), P1(0), g0, n);
    // Small levels are launch-latency bound: one fused launch per level (k_pyr_down, k_deriv_all recompute
    // their taps instead of storing intermediate images -- ~100 / ~56 cached reads per pixel).  Large levels
    // are bandwidth bound and keep the separate streaming kernels, which read every pixel once per pass
    // (measured at 8 x 1024^2: fused everywhere 17.8 ms per series against 16.9).
    const long long fuse_below = 131072;                // pixels of a launch (all pairs)
    for (int k = 1; k < L; k++) {
        const Geo &gs = h->geo[k - 1], &gd = h->geo[k];
        if ((long long)gd.w * gd.h * n <= fuse_below) {
            hipLaunchKernelGGL(k_pyr_down, grid2d(gd, 2 * n), kBlock2d, 0, s, P0(k - 1), P1(k - 1), gs, P0(k),
                               P1(k), gd, h->taps, n);
            continue;
        }
        // both frames per launch; the second one's intermediates borrow two planes the pyramid does not use yet
        hipLaunchKernelGGL((k_blur<false>), grid2d(gs, 2 * n), kBlock2d, 0, s, P0(k - 1), tmpA, P1(k - 1), Ix0, gs,
                           h->taps, n);
        hipLaunchKernelGGL((k_blur<true>), grid2d(gs, 2 * n), kBlock2d, 0, s, tmpA, tmpB, Ix0, Iy0, gs, h->taps, n);
        hipLaunchKernelGGL(k_resample, grid2d(gd, 2 * n), kBlock2d, 0, s, tmpB, P0(k), Iy0, P1(k), gs, gd, 1.0f, n);
    }
    // u = v = 0 at the coarsest level and du = dv = 0 at the start of every level: a plane of zeros that is
    // only ever read (no memsets); u / v of a level are written by the level above it, never in place
    const float *u = zero, *v = zero;
    float *un = b_u, *vn = b_v, *uo = b_u2, *vo = b_v2;      // next level's u, v; the pair after that
    // The coarse end of the pyramid -- the levels of at most coarse_max x coarse_max px, each a single SOR tile --
    // in one launch per tile size (k_coarse<32> for levels up to 32 x 32 px, then k_coarse<64>) and per COARSE_MAX
    // levels (more only when the scale factor is close to 1).
    int kc = L;
    while (kc > 0 && h->geo[kc - 1].w <= h->coarse_max && h->geo[kc - 1].h <= h->coarse_max) kc--;
    auto tile_of = [](const Geo &g) { return g.w <= 32 && g.h <= 32 ? 32 : 64; };
    for (int hi = L - 1; hi >= kc;) {
        const int T = tile_of(h->geo[hi]);
        int lo = hi;
        while (lo - 1 >= kc && hi - (lo - 1) + 1 <= COARSE_MAX && tile_of(h->geo[lo - 1]) == T) lo--;
        CoarseArgs ca;
        ca.nlev = hi - lo + 1;
        for (int k = hi; k >= lo; k--) {
            ca.g[hi - k] = h->geo[k];
            ca.I0[hi - k] = P0(k);
            ca.I1[hi - k] = P1(k);
        }
        ca.u_in = u; ca.v_in = v;
        ca.scratch_plane = h->geo[lo].plane;            // the finest level of the launch
        ca.stagger = h->coarse_stagger;
        ca.Ix0 = Ix0; ca.Iy0 = Iy0; ca.I1x = I1x; ca.I1y = I1y; ca.I1xx = I1xx; ca.I1xy = I1xy; ca.I1yy = I1yy;
        if (lo > 0) {
            ca.gout = h->geo[lo - 1];
            ca.u_out = un; ca.v_out = vn;
        } else {
            ca.gout.w = ca.gout.h = ca.gout.pitch = ca.gout.plane = 0;
            ca.u_out = d_ox; ca.v_out = d_oy;
        }
        ca.inner = h->inner; ca.solver = h->solver;
        ca.alpha = h->alpha; ca.gamma = h->gamma; ca.om = h->omega; ca.om1 = 1.0f - h->omega;
        if (T == 32) hipLaunchKernelGGL((k_coarse<32>), dim3(n), dim3(256), 0, s, ca);
        else hipLaunchKernelGGL((k_coarse<64>), dim3(n), dim3(1024), 0, s, ca);
        if (lo > 0) {
            u = un; v = vn;
            float *t = un; un = uo; uo = t;
            t = vn; vn = vo; vo = t;
        }
        hi = lo - 1;
    }
    for (int k = kc - 1; k >= 0; k--) {
        const Geo &g = h->geo[k];
        const dim3 gr = grid2d(g, n);
        if ((long long)g.w * g.h * n <= fuse_below) {
            DerivOut dout = {Ix0, Iy0, I1x, I1y, I1xx, I1xy, I1yy};
            hipLaunchKernelGGL(k_deriv_all, gr, kBlock2d, 0, s, P0(k), P1(k), dout, g);
        } else {
            const dim3 gr2 = grid2d(g, 2 * n);           // two images per launch
            hipLaunchKernelGGL(k_deriv, gr2, kBlock2d, 0, s, P0(k), Ix0, Iy0, P1(k), I1x, I1y, g, n);
            hipLaunchKernelGGL(k_deriv, gr2, kBlock2d, 0, s, I1x, I1xx, I1xy, I1y, (float *)nullptr, I1yy, g, n);
        }
        WarpIn wi = {P0(k), Ix0, Iy0, P1(k), I1x, I1y, I1xx, I1xy, I1yy, u, v};
        WarpOut wo = {Iz, Ix, Iy, Ixz, Iyz, Ixx, Ixy, Iyy};
        if (h->warp_window) hipLaunchKernelGGL((k_warp<true>), gr, kBlock2d, 0, s, wi, wo, g);
        else hipLaunchKernelGGL((k_warp<false>), gr, kBlock2d, 0, s, wi, wo, g);
        const float *du = zero, *dv = zero;
        int nxt = 0;
        // workgroup size: with few pairs a launch is bound by the latency of one workgroup (its ten half-sweeps in
        // sequence) -- 1024 threads, two rows per thread, finish a tile soonest; a launch that fills the chip several
        // times over is bound by how many tiles are resident -- 512 threads, two workgroups per CU (measured, one /
        // eight 1024^2 pairs: 4.31 / 12.3 ms of SOR per series with 1024 threads, 4.71 / 10.4 with 512, 6.65 / 13.7 with 256)
        const int threads = h->sor_threads ? h->sor_threads : (n <= 2 ? 1024 : 512);
        SorPlan plan = sor_plan(g, h->solver, h->fuse, threads, n, (long long)h->sor_deep * h->cus, n > 2 ? h->sor_wide : 0);
        const int launches_per_inner = h->solver / plan.K;
        Coef co = {c_nu, c_nv, c_a12, c_idu, c_idv, c_sx, c_sy};
        for (int it = 0; it < h->inner; it++) {
            PrepIn pi = {u, v, du, dv, Iz, Ix, Iy, Ixz, Iyz, Ixx, Ixy, Iyy};
            hipLaunchKernelGGL(k_prepare, dim3(hm_cdiv(g.w, PREP_BX), hm_cdiv(g.h, PREP_BY), n), dim3(PREP_BX, PREP_BY), 0, s,
                               pi, co, g, h->alpha, h->gamma);
            for (int pass = 0; pass < launches_per_inner; pass++) {
                SorArgs a;
                a.du_in = du; a.dv_in = dv;
                a.du_out = b_du[nxt]; a.dv_out = b_dv[nxt];
                a.nu = c_nu; a.nv = c_nv; a.a12 = c_a12; a.idu = c_idu; a.idv = c_idv;
                a.sx = c_sx; a.sy = c_sy;
                a.g = g;
                a.om = h->omega; a.om1 = 1.0f - h->omega;
                bool rec = h->prof;
                if (rec) {
                    if (h->ev_used + 2 > h->ev.size()) {
                        hipEvent_t e0, e1;
                        HM_HIP(hipEventCreate(&e0));
                        HM_HIP(hipEventCreate(&e1));
                        h->ev.push_back(e0);
                        h->ev.push_back(e1);
                    }
                }
                SorPlan run = plan;
                if (h->sor_dry) run.K = 0;
                if (rec) sor_launch(run, a, n, s, h->ev[h->ev_used], h->ev[h->ev_used + 1]);
                else sor_launch(run, a, n, s);
                if (rec) {
                    h->ev_used += 2;
                    h->ev_pxit.push_back((double)g.w * g.h * n * plan.K);
                    h->ev_px.push_back((double)g.w * g.h * n);
                    h->ev_level.push_back(k);
                }
                du = b_du[nxt]; dv = b_dv[nxt];
                nxt ^= 1;
            }
        }
        if (k > 0) {
            const Geo &gf = h->geo[k - 1];
            hipLaunchKernelGGL(k_add_prolong, grid2d(gf, n), kBlock2d, 0, s, u, v, du, dv, g, un, vn, gf,
                               (float)gf.w / (float)g.w, (float)gf.h / (float)g.h);
            u = un; v = vn;
            float *t = un; un = uo; uo = t;
            t = vn; vn = vo; vo = t;
        } else {
            // level-0 planes are pitched; the caller's arrays are tight
            hipLaunchKernelGGL(k_add_out, gr, kBlock2d, 0, s, u, v, du, dv, g, d_ox, d_oy);
        }
    }
    HM_HIP(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_brox_calc_dev(hm_brox_t h, int n, const uint8_t *d_f0, const uint8_t *d_f1, float *d_ox, float *d_oy)
{
    HM_ARG(h != nullptr, "hm_brox_calc_dev: NULL handle");
    HM_ARG(n >= 1 && n <= h->B, "hm_brox_calc_dev: n=%d outside 1..max_batch=%d", n, h->B);
    HM_ARG(d_f0 && d_f1 && d_ox && d_oy, "hm_brox_calc_dev: NULL pointer");
    HM_HIP(hipSetDevice(h->device));
    // (the second lane exists beside the CU-masked stream only: a masked stream has a hardware queue of its own, a plain
    // one takes a place in the runtime's pool of four -- see the note at "lanes" in hm_brox_tune)
    hipStream_t s2 = h->stream == h->masked ? h->masked2 : nullptr;
    if (h->lanes < 2 || n < 4 || h->prof || !s2) return brox_run(h, 0, n, h->stream, d_f0, d_f1, d_ox, d_oy);
    // two lanes: the second starts behind everything queued on the handle's stream so far, and the handle's stream
    // ends behind the second -- to the caller the call is still one piece of work on hm_brox_stream()
    const int n0 = (n + 1) / 2;
    HM_HIP(hipEventRecord(h->ev_fork, h->stream));
    HM_HIP(hipStreamWaitEvent(s2, h->ev_fork, 0));
    int rc = brox_run(h, 0, n0, h->stream, d_f0, d_f1, d_ox, d_oy);
    if (rc == HM_OK) rc = brox_run(h, n0, n - n0, s2, d_f0, d_f1, d_ox, d_oy);
    HM_HIP(hipEventRecord(h->ev_join, s2));
    HM_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return rc;
}

extern "C" int hm_brox_calc_batch(hm_brox_t h, int n, const uint8_t *f0, const uint8_t *f1, float *ox, float *oy)
{
    HM_ARG(h != nullptr, "hm_brox_calc_batch: NULL handle");
    HM_ARG(n >= 1 && n <= h->B, "hm_brox_calc_batch: n=%d outside 1..max_batch=%d", n, h->B);
    HM_ARG(f0 && f1 && ox && oy, "hm_brox_calc_batch: NULL pointer");
    HM_HIP(hipSetDevice(h->device));
    const size_t px = (size_t)h->W * h->H * n;
    HM_HIP(hipMemcpyAsync(h->d_f0, f0, px, hipMemcpyHostToDevice, h->stream));
    HM_HIP(hipMemcpyAsync(h->d_f1, f1, px, hipMemcpyHostToDevice, h->stream));
    int rc = hm_brox_calc_dev(h, n, h->d_f0, h->d_f1, h->d_ox, h->d_oy);
    if (rc) return rc;
    HM_HIP(hipMemcpyAsync(ox, h->d_ox, px * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipMemcpyAsync(oy, h->d_oy, px * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HM_HIP(hipStreamSynchronize(h->stream));
    return HM_OK;
}

extern "C" int hm_brox_calc(hm_brox_t h, const uint8_t *f0, const uint8_t *f1, float *ox, float *oy)
{
    return hm_brox_calc_batch(h, 1, f0, f1, ox, oy);
}

// ---- single operators on host arrays (parity tests) ---------------------------------------------
namespace {
struct Scratch {      // pitched device planes with tight host I/O
    std::vector<float *> bufs;
    ~Scratch() { for (float *b : bufs) hipFree(b); }
    float *plane(const Geo &g)
    {
        float *p = nullptr;
        if (hm_malloc((void **)&p, (size_t)g.plane * sizeof(float)) != hipSuccess) return nullptr;
        hipMemset(p, 0, (size_t)g.plane * sizeof(float));
        hipDeviceSynchronize();          // the fill is on the null stream, the operators run on a non-blocking one
        bufs.push_back(p);
        return p;
    }
    float *up(const Geo &g, const float *host)
    {
        float *p = plane(g);
        if (p && hipMemcpy2D(p, g.pitch * sizeof(float), host, g.w * sizeof(float), g.w * sizeof(float), g.h,
                             hipMemcpyHostToDevice) != hipSuccess)
            return nullptr;
        return p;
    }
    static bool down(const Geo &g, const float *dev, float *host)
    {
        return hipMemcpy2D(host, g.w * sizeof(float), dev, g.pitch * sizeof(float), g.w * sizeof(float), g.h,
                           hipMemcpyDeviceToHost) == hipSuccess;
    }
};
}  // namespace

#define OP_CHECK(p)                                                              \
    do {                                                                         \
        if (!(p)) {                                                              \
            hm_set_error("hm_op: device allocation or copy failed: %s",          \
                         hipGetErrorString(hipGetLastError()));                  \
            return HM_ERR_HIP;                                                   \
        }                                                                        \
    } while (0)

extern "C" int hm_op_blur(const float *src, int w, int h, float scale, float *dst)
{
    HM_ARG(src && dst && w >= 1 && h >= 1 && scale > 0.0f && scale < 1.0f, "hm_op_blur: bad argument");
    Geo g = make_geo(w, h);
    Taps t = make_taps(scale);
    Scratch sc;
    float *a = sc.up(g, src), *b = sc.plane(g), *c = sc.plane(g);
    OP_CHECK(a && b && c);
    hipLaunchKernelGGL((k_blur<false>), grid2d(g, 1), kBlock2d, 0, 0, a, b, (const float *)nullptr, (float *)nullptr, g, t, 1);
    hipLaunchKernelGGL((k_blur<true>), grid2d(g, 1), kBlock2d, 0, 0, b, c, (const float *)nullptr, (float *)nullptr, g, t, 1);
    HM_HIP(hipDeviceSynchronize());
    OP_CHECK(Scratch::down(g, c, dst));
    return HM_OK;
}

extern "C" int hm_op_resample(const float *src, int ws, int hs, float *dst, int wd, int hd, float mul)
{
    HM_ARG(src && dst && ws >= 1 && hs >= 1 && wd >= 1 && hd >= 1, "hm_op_resample: bad argument");
    Geo gs = make_geo(ws, hs), gd = make_geo(wd, hd);
    Scratch sc;
    float *a = sc.up(gs, src), *b = sc.plane(gd);
    OP_CHECK(a && b);
    hipLaunchKernelGGL(k_resample, grid2d(gd, 1), kBlock2d, 0, 0, a, b, (const float *)nullptr, (float *)nullptr, gs, gd, mul, 1);
    HM_HIP(hipDeviceSynchronize());
    OP_CHECK(Scratch::down(gd, b, dst));
    return HM_OK;
}

// the fused launches calc uses: one pyramid level (blur rows, blur columns, resample) ...
extern "C" int hm_op_pyr_down(const float *src, int ws, int hs, float scale, float *dst, int wd, int hd)
{
    HM_ARG(src && dst && ws >= 1 && hs >= 1 && wd >= 1 && hd >= 1 && scale > 0.0f && scale < 1.0f, "hm_op_pyr_down: bad argument");
    Geo gs = make_geo(ws, hs), gd = make_geo(wd, hd);
    Taps t = make_taps(scale);
    Scratch sc;
    float *a = sc.up(gs, src), *b = sc.plane(gd), *c = sc.plane(gd);
    OP_CHECK(a && b && c);
    hipLaunchKernelGGL(k_pyr_down, grid2d(gd, 2), kBlock2d, 0, 0, a, a, gs, b, c, gd, t, 1);     // both "frames" = src
    HM_HIP(hipDeviceSynchronize());
    OP_CHECK(Scratch::down(gd, c, dst));
    return HM_OK;
}

// ... all derivative images of a level (out: Ix0, Iy0, I1x, I1y, I1xx, I1xy, I1yy) ...
extern "C" int hm_op_deriv_all(const float *I0, const float *I1, int w, int h, float *const out[7])
{
    HM_ARG(I0 && I1 && out && w >= 1 && h >= 1, "hm_op_deriv_all: bad argument");
    Geo g = make_geo(w, h);
    Scratch sc;
    float *a = sc.up(g, I0), *b = sc.up(g, I1), *o[7];
    OP_CHECK(a && b);
    for (int i = 0; i < 7; i++) { o[i] = sc.plane(g); OP_CHECK(o[i]); }
    DerivOut d = {o[0], o[1], o[2], o[3], o[4], o[5], o[6]};
    hipLaunchKernelGGL(k_deriv_all, grid2d(g, 1), kBlock2d, 0, 0, a, b, d, g);
    HM_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 7; i++) OP_CHECK(Scratch::down(g, o[i], out[i]));
    return HM_OK;
}

// ... and u + du, v + dv prolonged to the next finer level (wd x hd; wd = ws, hd = hs: the level-0 form,
// the sums written to tight planes)
extern "C" int hm_op_add_prolong(const float *u, const float *v, const float *du, const float *dv, int ws, int hs,
                                 float *u2, float *v2, int wd, int hd)
{
    HM_ARG(u && v && du && dv && u2 && v2 && ws >= 1 && hs >= 1 && wd >= 1 && hd >= 1, "hm_op_add_prolong: bad argument");
    Geo gs = make_geo(ws, hs), gd = make_geo(wd, hd);
    Scratch sc;
    float *a = sc.up(gs, u), *b = sc.up(gs, v), *c = sc.up(gs, du), *d = sc.up(gs, dv);
    OP_CHECK(a && b && c && d);
    if (wd == ws && hd == hs) {
        float *ox = nullptr, *oy = nullptr;
        OP_CHECK(hm_malloc((void **)&ox, (size_t)ws * hs * sizeof(float)) == hipSuccess);
        sc.bufs.push_back(ox);
        OP_CHECK(hm_malloc((void **)&oy, (size_t)ws * hs * sizeof(float)) == hipSuccess);
        sc.bufs.push_back(oy);
        hipLaunchKernelGGL(k_add_out, grid2d(gs, 1), kBlock2d, 0, 0, a, b, c, d, gs, ox, oy);
        HM_HIP(hipDeviceSynchronize());
        OP_CHECK(hipMemcpy(u2, ox, (size_t)ws * hs * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
        OP_CHECK(hipMemcpy(v2, oy, (size_t)ws * hs * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
        return HM_OK;
    }
    float *e = sc.plane(gd), *f = sc.plane(gd);
    OP_CHECK(e && f);
    hipLaunchKernelGGL(k_add_prolong, grid2d(gd, 1), kBlock2d, 0, 0, a, b, c, d, gs, e, f, gd, (float)wd / (float)ws,
                       (float)hd / (float)hs);
    HM_HIP(hipDeviceSynchronize());
    OP_CHECK(Scratch::down(gd, e, u2) && Scratch::down(gd, f, v2));
    return HM_OK;
}

extern "C" int hm_op_deriv(const float *src, int w, int h, float *dx, float *dy)
{
    HM_ARG(src && dx && dy && w >= 1 && h >= 1, "hm_op_deriv: bad argument");
    Geo g = make_geo(w, h);
    Scratch sc;
    float *a = sc.up(g, src), *b = sc.plane(g), *c = sc.plane(g);
    OP_CHECK(a && b && c);
    hipLaunchKernelGGL(k_deriv, grid2d(g, 1), kBlock2d, 0, 0, a, b, c, (const float *)nullptr, (float *)nullptr, (float *)nullptr, g, 1);
    HM_HIP(hipDeviceSynchronize());
    OP_CHECK(Scratch::down(g, b, dx) && Scratch::down(g, c, dy));
    return HM_OK;
}

extern "C" int hm_op_warp(const float *const in[11], int w, int h, float *const out[8], int window)
{
    HM_ARG(in && out && w >= 1 && h >= 1, "hm_op_warp: bad argument");
    Geo g = make_geo(w, h);
    Scratch sc;
    const float *d[11];
    float *o[8];
    for (int i = 0; i < 11; i++) { d[i] = sc.up(g, in[i]); OP_CHECK(d[i]); }
    for (int i = 0; i < 8; i++) { o[i] = sc.plane(g); OP_CHECK(o[i]); }
    WarpIn wi = {d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10]};
    WarpOut wo = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
    if (window) hipLaunchKernelGGL((k_warp<true>), grid2d(g, 1), kBlock2d, 0, 0, wi, wo, g);
    else hipLaunchKernelGGL((k_warp<false>), grid2d(g, 1), kBlock2d, 0, 0, wi, wo, g);
    HM_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 8; i++) OP_CHECK(Scratch::down(g, o[i], out[i]));
    return HM_OK;
}

extern "C" int hm_op_prepare(const float *const in[12], int w, int h, float alpha, float gamma, float *const out[7])
{
    HM_ARG(in && out && w >= 1 && h >= 1, "hm_op_prepare: bad argument");
    Geo g = make_geo(w, h);
    Scratch sc;
    const float *d[12];
    float *o[7];
    for (int i = 0; i < 12; i++) { d[i] = sc.up(g, in[i]); OP_CHECK(d[i]); }
    for (int i = 0; i < 7; i++) { o[i] = sc.plane(g); OP_CHECK(o[i]); }
    PrepIn pi = {d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11]};
    Coef co = {o[0], o[1], o[2], o[3], o[4], o[5], o[6]};
    hipLaunchKernelGGL(k_prepare, dim3(hm_cdiv(w, PREP_BX), hm_cdiv(h, PREP_BY), 1), dim3(PREP_BX, PREP_BY), 0, 0,
                       pi, co, g, alpha, gamma);
    HM_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 7; i++) OP_CHECK(Scratch::down(g, o[i], out[i]));
    return HM_OK;
}

extern "C" int hm_op_sor(float *du, float *dv, const float *const coef[7], int w, int h, int iterations, int fuse,
                         float omega)
{
    HM_ARG(du && dv && coef && w >= 1 && h >= 1 && iterations >= 1, "hm_op_sor: bad argument");
    HM_ARG(fuse % 100 == 0 || (fuse % 100 >= 1 && fuse % 100 <= 10 && iterations % (fuse % 100) == 0),
           "hm_op_sor: fuse=%d must be 0 or a divisor of iterations=%d not above 10", fuse % 100, iterations);
    Geo g = make_geo(w, h);
    Scratch sc;
    const float *c[7];
    for (int i = 0; i < 7; i++) { c[i] = sc.up(g, coef[i]); OP_CHECK(c[i]); }
    float *b[4] = {sc.up(g, du), sc.up(g, dv), sc.plane(g), sc.plane(g)};
    OP_CHECK(b[0] && b[1] && b[2] && b[3]);
    SorPlan plan = sor_plan(g, iterations, fuse % 100, fuse >= 200 ? 1024 : (fuse >= 100 ? 512 : 256));
    int cur = 0;
    for (int done = 0; done < iterations; done += plan.K) {
        SorArgs a;
        a.du_in = b[cur ? 2 : 0]; a.dv_in = b[cur ? 3 : 1];
        a.du_out = b[cur ? 0 : 2]; a.dv_out = b[cur ? 1 : 3];
        a.nu = c[0]; a.nv = c[1]; a.a12 = c[2]; a.idu = c[3]; a.idv = c[4]; a.sx = c[5]; a.sy = c[6];
        a.g = g;
        a.om = omega; a.om1 = 1.0f - omega;
        sor_launch(plan, a, 1, 0);
        cur ^= 1;
    }
    HM_HIP(hipDeviceSynchronize());
    OP_CHECK(Scratch::down(g, b[cur ? 2 : 0], du) && Scratch::down(g, b[cur ? 3 : 1], dv));
    return HM_OK;
}
