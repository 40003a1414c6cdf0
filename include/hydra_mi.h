/*
 * hydra_mi.h -- C-ABI of libhydra_mi.so, the MI355X (gfx950) implementation of
 * HydraGL's per-frame hot loop: Brox optical flow followed by the EKF
 * measurement update of a textured triangle mesh.
 *
 * Every entry point is what the reference's binding for this path would call
 * through ctypes; the reference interface each one replaces is cited as
 * reference file:line.  Conventions:
 *   - plain pointers and sizes only; all host arrays are C-contiguous and owned
 *     by the caller, the library copies in/out; device buffers are owned by the
 *     handle and freed by *_destroy;
 *   - every call returns 0 on success and a negative code on error; the text of
 *     the last error of the calling thread is hm_last_error();
 *   - one handle <-> one HIP stream <-> one host thread; the host-pointer calls
 *     are synchronous at return (the reference synchronises after every launch,
 *     cuda_multi.py:793,804,1083); the *_dev calls enqueue on the handle's
 *     stream and return, hm_*_sync waits.
 */
#ifndef HYDRA_MI_H
#define HYDRA_MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HM_OK 0
#define HM_ERR_ARG (-1)     /* bad argument (reference: Python exception / assert) */
#define HM_ERR_HIP (-2)     /* HIP runtime failure, including "no GPU" */
#define HM_ERR_STATE (-3)   /* call sequence error, e.g. jz before initjacobian (cuda_multi.py:611) */
#define HM_ERR_NUMERIC (-4) /* the update system is not positive definite (non-finite input) */

const char *hm_last_error(void);
const char *hm_version(void);
/* number of visible HIP devices, or a negative error code */
int hm_device_count(void);

/* Device buffers for callers that keep frames / flow planes resident in HBM between calls (the
 * *_dev entry points below; the in-process flow -> EKF pipeline).  Counterpart of the reference's
 * cv::cuda::GpuMat uploads (src/optical_flow_ext.cpp:296-299) and gpuarray.to_gpu
 * (cuda_multi.py:760-790).  upload / download are synchronous. */
int hm_dev_alloc(int device, uint64_t bytes, void **out);
int hm_dev_free(int device, void *ptr);
int hm_dev_upload(int device, void *dst, const void *src, uint64_t bytes);
int hm_dev_download(int device, void *dst, const void *src, uint64_t bytes);
/* Streaming uploads (the frame loop of run_kalmanfilter.py:78-89 reads one frame per iteration): a copy
 * stream of the caller, page-locked host staging memory, and an asynchronous host -> device copy on that
 * stream; hm_copy_stream_sync waits for the copies queued so far. */
int hm_copy_stream_create(int device, void **stream_out);
int hm_copy_stream_destroy(int device, void *stream);
int hm_copy_stream_sync(int device, void *stream);
int hm_host_alloc(uint64_t bytes, void **out);
int hm_host_free(void *ptr);
int hm_dev_upload_async(int device, void *dst, const void *src, uint64_t bytes, void *stream);

/* ------------------------------------------------------------------------
 * Brox optical flow.  Replaces cv::cuda::BroxOpticalFlow as used by
 * processflow_gpu (src/optical_flow_ext.cpp:294-331).
 * ---------------------------------------------------------------------- */
typedef struct hm_brox *hm_brox_t;

/* cuda::BroxOpticalFlow::create(alpha, gamma, scale_factor, inner, outer, solver)
 * (src/optical_flow_ext.cpp:310; parameter meaning :300-308; defaults :453-488).
 * max_batch = number of frame pairs one calc_batch call may carry. */
int hm_brox_create(int device, int width, int height, int max_batch,
                   float alpha, float gamma, float scale_factor,
                   int inner_iterations, int outer_iterations, int solver_iterations,
                   hm_brox_t *out);
int hm_brox_destroy(hm_brox_t h);

/* brox->calc(frame0f, frame1f, flow) + split + download
 * (src/optical_flow_ext.cpp:314-328): two 8-bit gray frames in, flowx / flowy
 * (row-major f32, width*height each) out.  Host pointers. */
int hm_brox_calc(hm_brox_t h, const uint8_t *frame0, const uint8_t *frame1,
                 float *flowx, float *flowy);
/* n independent pairs, pair i at offset i*width*height in each array
 * (the loop of process(), src/optical_flow_ext.cpp:361-412, turned into a batch). */
int hm_brox_calc_batch(hm_brox_t h, int n, const uint8_t *frame0, const uint8_t *frame1,
                       float *flowx, float *flowy);
/* same, all four pointers in device memory; asynchronous on the handle's stream */
int hm_brox_calc_dev(hm_brox_t h, int n, const uint8_t *d_frame0, const uint8_t *d_frame1,
                     float *d_flowx, float *d_flowy);
int hm_brox_sync(hm_brox_t h);
/* the hipStream_t of the handle (so a caller can order its own work against it) */
void *hm_brox_stream(hm_brox_t h);
/* pyramid geometry: returns the number of levels, fills widths/heights (cap entries) */
int hm_brox_levels(hm_brox_t h, int *widths, int *heights, int cap);
/* SOR relaxation factor (default 1.99) */
int hm_brox_set_omega(hm_brox_t h, float omega);
/* launch tuning, never changes results: "sor_fuse" = red-black iterations fused
 * per SOR launch (0 = choose per level, else a divisor of solver_iterations, at most 10),
 * "sor_deep" = 1/0 (default 1): with sor_fuse 0, a level whose tiles would not fill the device even with the halo of
 * all solver_iterations takes them in one launch, "coarse_max" = 0, 32 (default) or 64: pyramid levels of at most
 * that many pixels per side run in one launch per pair (k_coarse) instead of one launch per operator,
 * "cu_reserve" = n (default 0): the handle gets a second stream with a compute-unit mask that leaves n CUs to
 * other streams and uses it from now on (call it before hm_brox_stream's value is kept anywhere; the pipeline sets 32),
 * "whole_chip" = 1/0: with a cu_reserve in force, the next calls go to the unmasked stream / back to the masked one
 * (the stream that is left is drained first; the pipeline gives the first series of a phase the whole chip),
 * "coarse_stagger" = 1/0: test knob, delays some workgroups of k_coarse between phases,
 * "sor_threads" = 256, 512 or 1024 threads per SOR workgroup (0, the default: 1024 for calls of one or two pairs,
 * 512 for larger ones), "warp_window" =
 * 1/0 the warp kernel stages its taps as an LDS window or reads them directly (default 0: measured
 * faster, see brox_kernels.h).  (Round 1 also had "graph": replaying a call's launch series as a captured
 * hipGraph.  Replays went wrong when the caller allocated device memory between calls, the cause was not
 * found, and with the series queued from a helper thread it bought nothing: removed.) */
int hm_brox_tune(hm_brox_t h, const char *key, int value);

/* HIP-event timing of the SOR launches of subsequent calc calls.
 * read: total milliseconds, launches, pixel-iterations (sum over launches of
 * pixels * red-black iterations) and pixels (sum over launches of pixels: every launch is one pass
 * over memory) since profiling was switched on or last read (switching it off stops the recording
 * and keeps the totals for `read`).  Any output pointer may be NULL. */
int hm_brox_profile(hm_brox_t h, int enable);
int hm_brox_profile_read(hm_brox_t h, double *sor_ms, long long *sor_launches,
                         double *sor_pixel_iterations, double *sor_pixels);
/* the same four totals per pyramid level (index 0 = the full frame) since profiling was switched on (not reset
 * by hm_brox_profile_read); fills at most cap entries of each array (any may be NULL), returns the number of levels */
int hm_brox_profile_levels(hm_brox_t h, int cap, double *sor_ms, long long *sor_launches,
                           double *sor_pixel_iterations, double *sor_pixels);

/* Single operators on host arrays, for parity tests against the oracle.
 * Each allocates scratch, runs the same kernel calc uses, and copies back. */
int hm_op_blur(const float *src, int w, int h, float scale_factor, float *dst);
int hm_op_resample(const float *src, int ws, int hs, float *dst, int wd, int hd, float mul);
int hm_op_deriv(const float *src, int w, int h, float *dx, float *dy);
/* the fused launches calc is made of (each gives the bits of the separate operators above applied in
 * turn): one pyramid level = blur + resample; all derivative images of a level (out: Ix0, Iy0 of I0;
 * I1x, I1y, I1xx, I1xy, I1yy of I1); u + du, v + dv prolonged to the next finer level with the flow
 * rescaled (wd = ws and hd = hs: the level-0 form, the sums themselves) */
int hm_op_pyr_down(const float *src, int ws, int hs, float scale_factor, float *dst, int wd, int hd);
int hm_op_deriv_all(const float *I0, const float *I1, int w, int h, float *const out[7]);
int hm_op_add_prolong(const float *u, const float *v, const float *du, const float *dv, int ws, int hs,
                      float *u2, float *v2, int wd, int hd);
/* in: I0,Ix0,Iy0,I1,I1x,I1y,I1xx,I1xy,I1yy,u,v   out: Iz,Ix,Iy,Ixz,Iyz,Ixx,Ixy,Iyy;
 * window: 1 = the LDS-window variant of the kernel (hm_brox_tune "warp_window"), 0 = direct reads */
int hm_op_warp(const float *const in[11], int w, int h, float *const out[8], int window);
/* in: u,v,du,dv + the 8 warped fields   out: nu,nv,a12,idu,idv,sx,sy */
int hm_op_prepare(const float *const in[12], int w, int h, float alpha, float gamma,
                  float *const out[7]);
/* du,dv updated in place; coef = nu,nv,a12,idu,idv,sx,sy; fuse = sor_fuse (+100 selects
 * 512-thread, +200 1024-thread workgroups) */
int hm_op_sor(float *du, float *dv, const float *const coef[7], int w, int h,
              int iterations, int fuse, float omega);

/* ------------------------------------------------------------------------
 * EKF measurement model.  Replaces Renderer (renderer.py:197-737, the OpenGL
 * rasteriser) and CUDAGL / CUDAGL_multi (cuda.py, cuda_multi.py: the reduction
 * kernels and their PBO plumbing).
 * ---------------------------------------------------------------------- */
typedef struct hm_ctx *hm_ctx_t;

/* Renderer.__init__ + loadMesh (renderer.py:199-295, 560-654) and the
 * CUDAGL_multi constructor (cuda_multi.py:24-71): mesh topology, texture
 * coordinates (= initial vertex positions in pixels, renderer.py:579) and the
 * measurement-noise scales that the reference bakes into its kernels
 * (cuda_multi.py:420).  tri: T*3 int32 vertex ids; uv: N*2 float pixels. */
int hm_ctx_create(int device, int width, int height, int n_vertices, int n_triangles,
                  const int32_t *tri, const float *uv,
                  double eps_Z, double eps_J, double eps_M, hm_ctx_t *out);
int hm_ctx_destroy(hm_ctx_t h);
/* gloo.Texture2D(im1) bound as init_texture (renderer.py:222-223,232): W*H u8 */
int hm_set_texture(hm_ctx_t h, const uint8_t *tex);
/* the observed frame of compute()/initjacobian (kalman.py:676-700,
 * renderer.py:674-679): y_im u8, y_flow x/y f32, y_m u8 in {0,1} */
int hm_set_observation(hm_ctx_t h, const uint8_t *y_im, const float *y_fx,
                       const float *y_fy, const uint8_t *y_m);
/* device-resident variant (flow straight from hm_brox_calc_dev) */
int hm_set_observation_dev(hm_ctx_t h, const uint8_t *d_y_im, const float *d_y_fx,
                           const float *d_y_fy, const uint8_t *d_y_m);

/* Renderer.render() of state X (renderer.py:310-325 after update_vertex_buffer
 * :503-524).  X = 4N doubles [x0,y0,..,vx0,vy0,..] (kalman.py:178).  Any output
 * pointer may be NULL.  im u8, fx/fy f32, m u8 (255 where covered), all W*H. */
int hm_render(hm_ctx_t h, const double *X, uint8_t *im, float *fx, float *fy, uint8_t *m);

/* In the calls below `masked` selects which observed flow the residuals use:
 * 0 = the flow as given to hm_set_observation, 1 = that flow multiplied by y_m
 * (what compute() hands to update(), kalman.py:679-687; error() gets the raw
 * flow, :700). */

/* initjacobian (cuda.py:940-950): render X and keep it as the reference render */
int hm_initjacobian(hm_ctx_t h, const double *X, int masked);
/* jz (cuda.py:972-980): render Xp, return the sum and its 4 components
 * (image, flow x, flow y, mask) against the reference render */
int hm_jz(hm_ctx_t h, const double *Xp, int masked, double *jz, double jzc[4]);
/* j (cuda.py:982-1010): renders X + dX e_i and X + dX e_j and reduces the products
 * of their differences to the reference render */
int hm_j(hm_ctx_t h, const double *X, double deltaX, int i, int j, double *out);
/* The reference's multi-perturbation operators: several non-interacting perturbations in one render, the
 * sums separated by the label of the triangle a pixel belongs to.  labels: one per triangle (the palette
 * update_vertex_buffer selects, renderer.py:553-556, built at :610-614; -1 = none); a pixel takes the label
 * shown by the reference render where that covers it, else by the perturbed render(s) in turn
 * (cuda_multi.py:132-143, 215-235).  hm_initjacobian must have been called.
 *   jz_multi (cuda_multi.py:81-157, 721-845): Xp = the state with all its perturbations applied;
 *       hz[n_labels], hzc[n_labels x 4] (may be NULL): the sums of hm_jz per label;
 *   j_multi (cuda_multi.py:159-248, 979-1129): ee = n_pairs x 2 state indices; one render of X with every
 *       ee[k][0] raised by deltaX, one with every ee[k][1]; hsum, nz (pixels counted), hcomp[n_labels x 4]
 *       (may be NULL) per label.
 * The fused hm_measure does not need them (it evaluates every single perturbation inside its own star);
 * they are here so that KFState._jacobian_multi / _hessian_sparse_multi (kalman.py:452-489, 539-581) run
 * as in the reference. */
int hm_jz_multi(hm_ctx_t h, const double *Xp, int masked, const int32_t *labels, int n_labels,
                double *hz, double *hzc);
int hm_j_multi(hm_ctx_t h, const double *X, double deltaX, int n_pairs, const int32_t *ee,
               const int32_t *labels, int n_labels, double *hsum, double *nz, double *hcomp);
/* KalmanFilter.projectmask (kalman.py:724-742) on the device: every vertex whose signed distance fd to the object's
 * outline exceeds 1 px takes 10 steps p -= d g/|g|^2 (g: forward differences of 0.1 px; d and the set of vertices
 * are those before the first step) and its displacement is added to its velocity.  fd is the reference's
 * (findObjectThreshold(y_m, 0.5)[2], imgproc.py:175-248): the mask's contours pruned as there (:205-228: the largest
 * object and its holes of cv2.contourArea >= 40; areas from pixel counts by Pick's theorem), then the distance to the
 * polygon through the centres of the border pixels of what is left (-cv2.pointPolygonTest), negative inside.
 * y_m: W*H host mask (object where > 0), or NULL = the mask of the observation in place.  X (4N) is updated in
 * place; *moved (may be NULL) = number of vertices that were outside. */
int hm_project_mask(hm_ctx_t h, const uint8_t *y_m, double *X, int *moved);
/* the pruning step alone, for parity tests: out (W*H) = 1 where the pruned object is */
int hm_prune_mask(hm_ctx_t h, const uint8_t *y_m, uint8_t *out);
/* Renderer.error (renderer.py:485-501): SSE per channel of render(X) against the
 * observation, with the 8-bit wrap-around the reference's uint8 arithmetic has
 * for the image and mask terms.  err = e_im, e_fx, e_fy, e_m; fx/fy (may be NULL)
 * receive the rendered flow planes */
int hm_error(hm_ctx_t h, const double *X, int masked, double err[4], float *fx, float *fy);

/* KFState.update (kalman.py:437-449) in one call, single-perturbation
 * semantics (_jacobian :491-518, _hessian_sparse :583-606; deltaX = 2 there):
 * renders X once, then evaluates every +-deltaX perturbation only inside the
 * bounding box of the triangles it moves, one workgroup per vertex / per mesh
 * edge.  Hz[4N], Hzc[4N*4] (may be NULL), HTH[4N*4N] (dense, symmetric, zero
 * outside the J pattern kalman.py:202-205; _hessian :521-536 has the same
 * value because non-adjacent vertices have disjoint supports). */
int hm_measure(hm_ctx_t h, const double *X, double deltaX, int masked,
               double *Hz, double *Hzc, double *HTH);
/* The dense part of the update (kalman.py:753-755, 785-799) on the device, in information form:
 *   begin : factor the prior covariance W (4N x 4N, symmetric positive definite) and keep
 *           inv(W) and the prior mean X0 on the device;
 *   step  : hm_measure at X, then  step = (inv(W) + HTH)^-1 (Hz - HTH (X0 - X))  by a blocked
 *           Cholesky factorisation (the reference forms inv(inv(W) + HTH) explicitly and
 *           multiplies); the new iterate is X0 + step.  Hzc (4N x 4, may be NULL) as hm_measure;
 *           err (may be NULL) receives hm_error of the new iterate X0 + step (kalman.py:813);
 *   cov   : (inv(W) + HTH)^-1 of the last step (which = 0) or of the one before (which = 1,
 *           what the reference keeps as W_old for its mesh-inversion rollback, kalman.py:806-811);
 *           which = -1: the prior itself (no iterate was accepted).
 * A non-positive-definite system shows up as NaNs in `step`. */
int hm_update_begin(hm_ctx_t h, const double *W_prior, const double *X0);   /* W_prior NULL: the covariance
                                                                              resident on the device */
/* queue the covariance half of the next hm_update_begin / hm_update_run(h, NULL, ...) -- factoring and
 * inverting the covariance resident on the device -- and return; it does not need the predicted state,
 * so it can run while the host predicts the state (hm_ms_newton) */
int hm_update_prefactor(hm_ctx_t h);
int hm_update_step(hm_ctx_t h, const double *X, double deltaX, int masked, double *step, double *Hzc,
                   double err[4]);
int hm_update_cov(hm_ctx_t h, int which, double *W_out);                     /* W_out NULL: stays on the device */
/* The whole of IteratedKalmanFilter.update (kalman.py:774-831) in one call: begin, up to max_iter
 * steps with the reference's acceptance logic between them, cov of the state that is kept.
 *   X       in: the predicted state (prior mean); out: the state kept;
 *   info    iterations run, iterations accepted, reverted (a triangle flipped, :806-811), converged
 *           (|e_new - e_old| / e_new < reltol, :817-819);
 *   errs    max_iter x 4 (may be NULL): hm_error of every new iterate;
 *   Hzc     4N x 4 (may be NULL) of the last measurement; gains 3 x 4N (may be NULL): W Hzc[:,0],
 *           W (Hzc[:,1] + Hzc[:,2]), W Hzc[:,3] with the covariance kept (:828-830);
 *   W_out   4N x 4N, or NULL to leave the covariance on the device (hm_cov_fetch / hm_cov_predict).
 * Between iterations the state stays on the device and the render that gave an iterate's error is
 * the reference render of the next measurement; the numbers are those of the step-by-step calls.
 * HM_ERR_NUMERIC: the system inv(W) + HTH was not positive definite. */
int hm_update_run(hm_ctx_t h, const double *W_prior, double *X, double deltaX, int masked, int max_iter,
                  double reltol, int info[4], double *errs, double *Hzc, double *gains, double *W_out);
/* the covariance resident on the device (result of the last hm_cov_predict, hm_update_cov or
 * hm_update_run) copied to W_out (4N x 4N) */
int hm_cov_fetch(hm_ctx_t h, double *W_out);
/* IteratedMSKalmanFilter._newton (kalman.py:923-960): the mass-spring state prediction, ceil(1/dt)
 * implicit-Euler sub-steps each solved by Newton's method.  Host code, no GPU involved.
 * bars: I*2 vertex ids (distmesh.bars), l0: rest lengths; X: 4N doubles, advanced in place. */
int hm_ms_newton(int n_vertices, int n_bars, const int32_t *bars, const double *l0, double kappa, double M,
                 double dt, int maxiter, double tol, double *X, int *newton_iterations);
/* The same, started ahead of time on a host thread: the state a frame ends with is the one the next frame's
 * prediction starts from (kalman.py:850-863 run at the top of the next compute()).  create: one persistent
 * thread; start: copies its arguments and returns (a job still running is waited for, its result dropped); finish:
 * waits, X (4N) receives the advanced state of the job started last.  One job at a time per worker. */
int hm_ms_worker_create(void **worker);
int hm_ms_worker_destroy(void *worker);
int hm_ms_newton_start(void *worker, int n_vertices, int n_bars, const int32_t *bars, const double *l0, double kappa,
                       double M, double dt, int maxiter, double tol, const double *X);
int hm_ms_newton_finish(void *worker, double *X, int *newton_iterations);
/* Jobs started on `worker` run as ONE launch on the device of the filter handle `ctx` (NULL: back to the worker's host
 * thread) when the mesh fits the kernel (k_ms_newton4, csrc/predict_kernels.h: at most 256 vertices and 12 springs per
 * vertex; four waves, a vertex per lane) -- the state goes in and out through page-locked memory, hm_ms_newton_finish
 * takes the kernel's result block when it is whole (csrc/host_block.h); larger meshes keep the host loop.  Host and device agree to rounding (sums over the vector are
 * added in another order).  The handle must outlive the worker's jobs. */
int hm_ms_worker_attach(void *worker, hm_ctx_t ctx);
/* what the attached worker calls; 1 = not for the device (mesh too large / inner solve gave up) */
int hm_newton_dev_start(hm_ctx_t h, int N, int n_bars, const int32_t *bars, const double *l0, double kappa, double M,
                        double dt, int maxiter, double tol, const double *X);
int hm_newton_dev_finish(hm_ctx_t h, double *X, int *newton_iterations);
/* one-shot: the next hm_update_run on h calls hm_ms_newton_start(worker, ..., X) with the state it ends with as soon
 * as that state is known -- before the covariance of the kept iterate is formed and fetched */
int hm_update_arm_newton(hm_ctx_t h, void *worker, int n_bars, const int32_t *bars, const double *l0, double kappa,
                         double M, double dt, int maxiter, double tol);
/* one-shot, with hm_update_arm_newton armed as well: the next hm_update_run also queues, right behind the covariance of
 * the state it keeps, the covariance half of the NEXT frame's prediction (reference kalman.py:856-863: F at that state,
 * W' = F W F^T + Weps -- hm_cov_predict with the spring blocks at that state) and the factorisation / inverse of W' the
 * next update starts with (hm_update_prefactor): both depend on the finished update only, and the device has them
 * done while the caller is still on its way back to predict().  hm_cov_fetch keeps returning the posterior;
 * hm_update_cov is not available after such a run. */
int hm_update_arm_cov(hm_ctx_t h, double eps_F);
/* Makes what hm_update_run queued ahead (hm_update_arm_cov) current, as if hm_cov_predict(h, NULL, n_bars, bars,
 * <blocks at X>, a, s, eps_F, NULL) and hm_update_prefactor(h) had just been called -- if it was made from exactly
 * these inputs (X: the 4N state before the step; compared bit for bit).  Returns 0 when taken, 1 when there is nothing
 * to take: the caller then makes the two calls itself (the posterior is still the resident covariance). */
int hm_predict_take(hm_ctx_t h, const double *X, int n_bars, const int32_t *bars, const double *l0, double kappa,
                    double a, double s, double eps_F);
/* Covariance prediction W' = F W F^T + Weps (kalman.py:717 and :863) on the device, with
 * F = [[I, a I], [s dfdy, I]], dfdy given as one symmetric 2x2 block (Bxx, Bxy, Byy) per spring
 * (kalman.py:865-902; n_bars = 0: the constant-velocity model), Weps = eps_F [[I/4, I/2], [I/2, I]]
 * (:182).  W_in NULL: propagate the covariance resident on the device.  The result is copied to
 * W_out (NULL: not copied) and kept on the device for hm_update_begin / hm_update_run(h, NULL, ...). */
int hm_cov_predict(hm_ctx_t h, const double *W_in, int n_bars, const int32_t *bars, const double *blocks,
                   double a, double s, double eps_F, double *W_out);
/* IteratedMSKalmanFilter.predict (kalman.py:850-863) in one call for the covariance resident on the device:
 * the spring Jacobian at X before the step gives F (:856, 904-912); X (4N, in place) is advanced by _newton
 * (:923-960) -- on the device, one workgroup on a second stream, for meshes whose problem fits its LDS (about 350
 * vertices), else by hm_ms_newton; W <- F W F^T + Weps (hm_cov_predict); prefactor != 0: also what
 * hm_update_prefactor does, queued by the calling thread while the state prediction runs. */
int hm_ms_predict(hm_ctx_t h, int n_bars, const int32_t *bars, const double *l0, double kappa, double M, double dt,
                  int maxiter, double tol, double eps_F, double *X, int *newton_iterations, int prefactor);
/* Renderer.error (renderer.py:485-501) of the state the last hm_update_run kept, against the observation as it was
 * given (raw flow), WITHOUT another render: when that state is the update's last iterate, the iterate's own render
 * produced these sums (KalmanFilter.compute ends with exactly this call, kalman.py:700).  X: the state asked about.
 * Returns 0 and fills err, or 1 when the sums are not at hand (another state, a reverted update, a new observation or
 * texture since): call hm_error then. */
int hm_update_last_error(hm_ctx_t h, const double *X, double err[4]);
/* KalmanFilter.compute's predict -> projectmask -> update (kalman.py:676-700) without a host round trip between the
 * three, for a state prediction started on the device (hm_ms_worker_attach + hm_update_arm_newton / hm_ms_newton_start)
 * and the new frame's observation in place: queues projectmask (kalman.py:724-742, mask of the resident observation) of
 * the prediction in flight behind its kernel; the projected state stays in device memory as the prior mean of the NEXT
 * hm_update_run on h, whose X argument is then output only (it must still point at 4N doubles).  Same numbers as
 * hm_ms_newton_finish + hm_project_mask(h, NULL, ...) + hm_update_run.  Returns 0 when queued, 1 when there is nothing
 * to chain (no device prediction in flight): the caller makes the three calls.  A chained hm_update_run returns 2 when
 * the prediction's inner solve gave up (never observed): the caller predicts on the host and calls it again. */
int hm_chain_project(hm_ctx_t h);
/* The contour pruning and outline of a mask in DEVICE memory queued ahead of the hm_set_observation_dev that will name it
 * as d_y_m (a streaming caller holds the next frame's mask a frame early): that call then finds the outline done instead
 * of queueing ~0.2 ms of kernels the projection waits for.  The mask must not change until then; any other observation
 * or a projection onto a host mask discards the preparation. */
int hm_prepare_mask(hm_ctx_t h, const uint8_t *d_y_m);
/* one-shot: the next hm_update_run on h calls hm_prepare_mask(h, d_y_m) the moment its state is final, so that the next
 * frame's outline is computed beside the state prediction and the update's tail */
int hm_update_arm_mask(hm_ctx_t h, const uint8_t *d_y_m);
/* Hz components (4N x 4) and gains (3 x 4N; kalman.py:826-828) of the last hm_update_run that was called with
 * Hzc = gains = NULL: such a call does not wait for the kernels that form them.  Either may be NULL.  Available until the
 * next hm_update_run on h. */
int hm_update_tail(hm_ctx_t h, double *Hzc, double *gains);
/* what the last chained hm_update_run started from: the predicted state, the projected state (its prior mean), the Newton
 * iterations of the prediction, the number of vertices projectmask moved; any pointer may be NULL */
int hm_chain_states(hm_ctx_t h, double *predicted, double *projected, int *newton_iterations, int *moved);
/* tuning knobs: "measure_split" = workgroups per vertex job of the measurement (1..16, default 5),
 * "edge_split" = workgroups per mesh-edge job (1..16, default 2); the sums change in their last
 * bits with them (another summation order); "chol_flow" = 1/0 the blocked Cholesky factorisations of the update as one
 * persistent launch whose block tasks hand their results over through memory, or one launch per 32-column block step
 * (same bits either way), "chol_flow_wgs" = workgroups of that launch (2..2048, default 256: the first becomes the chain of
 * the diagonal blocks, the others run the tasks it waits for), "chol_flow_stall" = n (test knob, default 0): that chain sleeps
 * ~4 us x n before every diagonal block, so that every wait for it takes the patient path; "result_delay" = n microseconds
 * (test knob, default 0): the kernels that hand result blocks to the host (csrc/host_block.h) publish a block's last word
 * first and the rest n us later -- same results; "tail_split" = 1/0 the tail of hm_update_run (covariance of the kept
 * state, gains, the next frame's covariance prediction) on a stream of its own or on the handle's (same results);
 * "newton_fail" = 1 (test knob): device state predictions report a failed inner solve */
int hm_ctx_tune(hm_ctx_t h, const char *key, int value);
int hm_ctx_sync(hm_ctx_t h);
void *hm_ctx_stream(hm_ctx_t h);

#ifdef __cplusplus
}
#endif
#endif /* HYDRA_MI_H */
