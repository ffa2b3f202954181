/*
 * slowflow_amd.h -- C-ABI of the MI355X-native (gfx950 HIP) implementation of slowflow's variational
 * optical-flow refinement hot path.  Plain pointers and sizes only; every entry point names the
 * reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions (identical to the reference, epic_flow_extended/image.h:17-43):
 *   - planes are fp32, row-major, `stride` floats per row (stride >= width); a colour image is 3 planes
 *     of height*stride floats each; only the `width` valid columns of a row are read or written,
 *   - all pointers below are HOST pointers unless a function says "device-resident"; the library owns
 *     every device buffer (through the context),
 *   - functions return 0 (SFA_OK) or a negative sfa_status; they never exit() (the reference does:
 *     image.c:19-30, solver.c:75-78); sfa_last_error() gives the message,
 *   - a context is bound to one GPU and one HIP stream; it is thread-compatible, not thread-safe: use one
 *     context per host thread (the reference runs one Variational_MT per OpenMP thread, slow_flow.cpp:706).
 * There is NO CPU fallback: without a HIP device every compute entry point fails with SFA_ERR_NO_DEVICE.
 */
#ifndef SLOWFLOW_AMD_H
#define SLOWFLOW_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFA_VERSION 1
#define SFA_MAX_REF 8          /* slow_flow_S - 1 <= 8: windows of up to 17 frames (4 until round 5) */

typedef enum {
    SFA_OK = 0,
    SFA_ERR_ARG = -1,          /* bad argument (null pointer, size, unsupported S) */
    SFA_ERR_HIP = -2,          /* a HIP runtime call failed */
    SFA_ERR_NO_DEVICE = -3,    /* no usable GPU */
    SFA_ERR_REF_FRAME = -4,    /* "Frame compared to reference frame is the reference frame itself" (variational_aux_mt.cpp:419) */
    SFA_ERR_TIMEOUT = -5,      /* a bounded in-kernel wait gave up (solver pipeline) */
    SFA_ERR_UNSUPPORTED = -6   /* a path that needs the absent third-party hook (occlusion graph cut) */
} sfa_status;

/* layout-compatible with image_t / color_image_t (epic_flow_extended/image.h:17-33) */
typedef struct sfa_image { int width, height, stride; float *data; } sfa_image;
typedef struct sfa_color_image { int width, height, stride; float *c1, *c2, *c3; } sfa_color_image;

/* penalty ids as Variational_AUX_MT::select_robust_function (variational_aux_mt.cpp:909-925):
 * 0 quadratic, 2 lorentzian, 3 truncated modified L1, 4 geman-mcclure, anything else modified L1 */
typedef struct sfa_penalty { int id; float eps; float trunc; } sfa_penalty;

/* the cfg keys Variational_MT reads (variational_mt.cpp:173-192, 533-568), already parsed */
typedef struct sfa_params {
    int   S;                    /* slow_flow_S */
    int   one_direction;        /* slow_flow_method == "forward" */
    int   smoothing;            /* slow_flow_smoothing */
    int   dataterm_norm;        /* slow_flow_dataterm */
    int   niter_alter, niter_outer, niter_inner, niter_solver;
    float thres_outer, thres_inner;
    float sor_omega;
    float alpha, gamma, delta;
    sfa_penalty robust_color, robust_grad, robust_reg;
    float rho[SFA_MAX_REF], omega[SFA_MAX_REF];      /* slow_flow_rho_<a>, slow_flow_omega_<a> */
    int   hbit;                 /* 16bit */
    float norm_avg[3], norm_std[3];                  /* slow_flow_img_norm_{avg,std}_{1,2,3} */
    int   occlusion_reasoning;  /* slow_flow_occlusion_reasoning */
    int   layers;               /* slow_flow_layers */
    float p_scale;              /* slow_flow_p_scale */
    float presmooth_sigma;      /* > 0 when cfg `sigma` > 0: value of slow_flow_sigma */
    /* discrete occlusion step between alternations (optimizeOcc, variational_aux_mt.cpp:758-887) */
    float occlusion_penalty;    /* slow_flow_occlusion_penalty ("1.0"): cost of the label "occluded in the future" */
    float occlusion_alpha;      /* slow_flow_occlusion_alpha ("0.5"): Potts weight between 4-neighbours */
    int   niter_graphc;         /* slow_flow_niter_graphc ("10"): expansion iterations; a two-label cut is exact after one */
    /* ADDITIVE key slow_flow_sor_order: 0 = "lexicographic" (default; the reference's raster order, results identical to sor_coupled), 1 = "red_black":
     * a DIFFERENT ALGORITHM (two-colour sweeps of the same point update) that does not reproduce the reference -- after 30 sweeps its increment is 1e-2..1e-1
     * away -- kept as a labelled low-latency mode for a single frame pair; every result produced with it is to be labelled as such */
    int   sor_order;
} sfa_params;

/* variational_params_t (epic_flow_extended/variational.h:16-25), same layout */
typedef struct sfa_params_2frame {
    float alpha, gamma, delta, sigma;
    int niter_outer, niter_inner, niter_solver;
    float sor_omega;
} sfa_params_2frame;

typedef struct sfa_ctx sfa_ctx;

/* ---- context ------------------------------------------------------------------------------------ */
int  sfa_device_count(void);
int  sfa_ctx_create(int device, sfa_ctx **out);
void sfa_ctx_destroy(sfa_ctx *ctx);
const char *sfa_last_error(const sfa_ctx *ctx);     /* ctx may be NULL: last error of the calling thread */
int  sfa_ctx_sync(sfa_ctx *ctx);
void sfa_params_default(sfa_params *p);             /* driver defaults, slow_flow.cpp:64-128 */

/* ---- the path itself ---------------------------------------------------------------------------- */

/* Replaces Variational_MT::variational (variational_mt.cpp:526-784): coarse-to-fine refinement of (wx,wy),
 * in place.  frames[f] = first plane of colour frame f (3 consecutive planes), f = 0 .. 2*(S-1), reference
 * frame in the middle; chw = channel weights (setChannelWeights, :521) or NULL for all ones;
 * occlusions_out (h*stride floats) or NULL; change[2] = returned Point2f (mean |du|, mean |dv| of the last
 * outer iteration of level 0). */
int sfa_variational(sfa_ctx *ctx, const sfa_params *p, float *wx, float *wy, int w, int h, int stride,
                    const float *const *frames, int n_frames, const float *const chw[3],
                    float *occlusions_out, float change[2]);

/* The reference's ORIGINAL two-frame refinement, `variational(wx, wy, im1, im2, params)` (epic_flow_extended/variational.c:101-143
 * with variational_aux.c): one level, fixed modified-L1 penalties, weights halved; wx, wy refined in place.  im1, im2: first
 * plane of 3.  Pinned end to end, bit for bit, against the compiled reference.  p == NULL: variational_params_default. */
int sfa_variational_2frame(sfa_ctx *ctx, float *wx, float *wy, int w, int h, int stride, const float *im1, const float *im2, const sfa_params_2frame *p);
void sfa_params_2frame_default(sfa_params_2frame *p);          /* variational.c:86-98 */
/* The reference's own symbol and signature (variational.h:34), for relinking callers such as adaptiveFR / EpicFlow's refinement
 * step: runs on device 0 with a process-wide context; aborts with a message on error like the reference does. */
void variational(sfa_image *wx, sfa_image *wy, const sfa_color_image *im1, const sfa_color_image *im2, sfa_params_2frame *params);

/* Replaces Variational_MT::compute_one_level (variational_mt.cpp:169-493): one pyramid level. */
int sfa_compute_one_level(sfa_ctx *ctx, const sfa_params *p, float *wx, float *wy, int w, int h, int stride,
                          const float *const *frames, int n_frames, const float *const chw[3],
                          float *occlusions_out, float change[2]);

/* Replaces normalize() (variational_mt.cpp:17-85): in-place (I-avg)/std over F colour frames; avg/std are the
 * doubles the reference publishes as slow_flow_img_norm_* params. */
int sfa_normalize(sfa_ctx *ctx, float *const *frames, int n_frames, int w, int h, int stride,
                  double avg[3], double std_dev[3]);

/* Replaces sor_coupled (solver.h:11, solver.c:63-399) on host planes: K lexicographic SOR sweeps, results
 * numerically identical to the reference's raster order (hyperplane-pipelined on the GPU).  a11/a12/a22 are
 * overwritten with the inverted 2x2 blocks exactly as the reference does. */
int sfa_sor_coupled(sfa_ctx *ctx, sfa_image *du, sfa_image *dv, sfa_image *a11, sfa_image *a12, sfa_image *a22,
                    sfa_image *b1, sfa_image *b2, sfa_image *dpsis_horiz, sfa_image *dpsis_vert,
                    int iterations, float omega);
/* LABELLED MODE, not a replacement of anything in the reference: K red-black sweeps of the same per-point update on host planes (a11/a12/a22 are overwritten
 * with the inverted blocks).  Bit-identical to the test checker's red-black restatement, NOT to sor_coupled. */
int sfa_sor_red_black(sfa_ctx *ctx, sfa_image *du, sfa_image *dv, sfa_image *a11, sfa_image *a12, sfa_image *a22,
                      sfa_image *b1, sfa_image *b2, sfa_image *dpsis_horiz, sfa_image *dpsis_vert, int iterations, float omega);
/* the reference's own symbol and signature (solver.h:11); uses a process-wide default context on device 0 and
 * aborts with a message if no GPU is usable (the reference's error style, solver.c:75-78) */
void sor_coupled(sfa_image *du, sfa_image *dv, sfa_image *a11, sfa_image *a12, sfa_image *a22, sfa_image *b1,
                 sfa_image *b2, sfa_image *dpsis_horiz, sfa_image *dpsis_vert, const int iterations, const float omega);

/* ---- stage entry points (host planes; used by the parity tests and by partial integrations) ------ */

/* optimizeOcc, first half (variational_aux_mt.cpp:783-866): the data costs of the labels "occluded in the past" (d0) and
 * "occluded in the future" (d1).  masks[s]: raw warp mask of slot s; succ1/succ2[s], ref1/ref2[s]: the colour image pairs
 * (first plane of 3) whose difference is Iz of the slot's successive-frames / reference-frame derivative stack. */
int sfa_occlusion_costs(sfa_ctx *ctx, const sfa_params *p, float *d0, float *d1, const float *const *masks, const float *const *succ1,
                        const float *const *succ2, const float *const *ref1, const float *const *ref2, int w, int h, int stride);
/* optimizeOcc, second half (:868-880): occ[p] = 2*l_p - 1 for the labelling that minimises sum_p D_{l_p}(p) + alpha * #{4-neighbour
 * pairs with different labels} -- what GCO's two-label expansion computes; exact s-t minimum cut on the GPU. */
int sfa_grid_cut(sfa_ctx *ctx, float *occ, const float *d0, const float *d1, int w, int h, int stride, float alpha);

/* Variational_AUX_MT::image_warp (variational_aux_mt.cpp:722-756); mask may be NULL */
int sfa_image_warp(sfa_ctx *ctx, float *dst3, float *mask, const float *src3, const float *wx, const float *wy,
                   int w, int h, int stride, int factor);
/* one derivative stack of get_derivatives (variational_mt.cpp:113-133): out = Ix,Iy,Iz,Ixx,Ixy,Iyy,Ixz,Iyz,
 * each a colour image (3*h*stride floats), from I1 (im1p) and I2 (im2p) */
int sfa_derivative_stack(sfa_ctx *ctx, float *out8x3, const float *I1, const float *I2, int w, int h, int stride);
/* convolve_horiz / convolve_vert with the path's derivative filters (image.c:400-526); order 1 or 2 */
int sfa_convolve(sfa_ctx *ctx, float *dst, const float *src, int w, int h, int stride, int order, int horizontal);
/* Variational_AUX_MT::compute_dpsis_weight, first output (variational_aux_mt.cpp:673-719) */
int sfa_dpsis_weight(sfa_ctx *ctx, float *dst, const float *im3, int w, int h, int stride, float coef,
                     const float avg[3], const float std_dev[3], int hbit);
/* Variational_AUX_MT::compute_smoothness (variational_aux_mt.cpp:18-127) */
int sfa_smoothness(sfa_ctx *ctx, int method, float *dst_horiz, float *dst_vert, const float *uu, const float *vv,
                   const float *dpsis, int w, int h, int stride, float alpha, const sfa_penalty *reg);
/* Variational_AUX_MT::sub_laplacian (variational_aux_mt.cpp:130-161): dst += div(w grad src) */
int sfa_sub_laplacian(sfa_ctx *ctx, float *dst, const float *src, const float *wh, const float *wv, int w, int h, int stride);
/* Variational_AUX_MT::add_data_and_match / add_data_and_match_ref (variational_aux_mt.cpp:166-403, 408-634):
 * accumulate one data term into a11,a12,a22,b1,b2.  D8x3 as sfa_derivative_stack's output. */
int sfa_add_data_and_match(sfa_ctx *ctx, float *a11, float *a12, float *a22, float *b1, float *b2, const float *mask,
                           const float *du, const float *dv, const float *D8x3, const float *const chw[3],
                           int w, int h, int stride, float delta_over3, float gamma_over3, float s, int ref_term,
                           int dt_norm, const sfa_penalty *color, const sfa_penalty *grad);
/* pyramid arithmetic (cv::GaussianBlur / cv::resize as used at variational_mt.cpp:607,611,672,711) */
int sfa_gaussian_blur(sfa_ctx *ctx, float *dst, const float *src, int w, int h, int stride, float sigma);
int sfa_resize_linear(sfa_ctx *ctx, float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride);
/* optional presmoothing of level 0 (cfg `sigma` > 0, variational_mt.cpp:590-597): gaussian_filter (image.c:310-348) through
 * convolve_horiz / convolve_vert (image.c:529-644; orders 1 and 2 take the 3 / 5-tap routines) */
int sfa_gaussian_presmooth(sfa_ctx *ctx, float *dst, const float *src, int w, int h, int stride, float sigma);
/* cv::resize(src, dst, Size(0,0), fx, fy, INTER_LINEAR) as the driver's input rescaling uses it (slow_flow.cpp:552): the caller
 * passes dw = cvRound(sw*fx), dh = cvRound(sh*fy); source coordinate = (dst + 0.5) / fx - 0.5 */
int sfa_resize_linear_fx(sfa_ctx *ctx, float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride, double fx, double fy);
int sfa_pyramid_sizes(int w, int h, int layers, float p_scale, int *ws, int *hs);

/* ---- a sequence's frames resident in HBM (the multi-pair driver) ----------------------------------------------------------------------
 * The reference keeps the whole sequence in host memory (slow_flow.cpp:447-592), normalises it in place (:673) and hands windows of it to the solver
 * (:721-724).  Here the frames cross PCIe once: uploaded, normalised on the GPU (same arithmetic as sfa_normalize, which is built on this), and copied
 * device-to-device into the jobs that need them. */
typedef struct sfa_sequence sfa_sequence;
int  sfa_sequence_create(sfa_ctx *ctx, int w, int h, int n_frames, sfa_sequence **out);
void sfa_sequence_destroy(sfa_sequence *seq);
int  sfa_sequence_upload(sfa_sequence *seq, int f, const float *frame3, int stride);      /* asynchronous on the context's stream */
int  sfa_sequence_download(sfa_sequence *seq, int f, float *frame3, int stride);
/* normalize() (variational_mt.cpp:17-85) over the frames [f0, f0 + n) in place on the GPU; avg / std as sfa_normalize */
int  sfa_sequence_normalize(sfa_sequence *seq, int f0, int n, double avg[3], double std_dev[3]);
/* The same in its three parts, for a sequence whose frames are spread over several GPUs (each GPU holds the frames its windows read) and must still be
 * normalised with the statistics of ALL loaded frames (slow_flow.cpp:673): (1) the six fp64 sums of the raw frames [f0, f0 + n), sums[6 f + 2 k] = sum of
 * channel k, sums[6 f + 2 k + 1] = sum of its squares -- a deterministic kernel, the same bits on whichever GPU holds a frame; (2) avg / std from per-frame sums
 * in frame order (host arithmetic, variational_mt.cpp:41-52); (3) I <- (I - avg) / std on resident frames. */
int  sfa_sequence_frame_sums(sfa_sequence *seq, int f0, int n, double *sums /* n x 6 */);
int  sfa_normalize_statistics(const double *sums /* n_frames x 6 */, int n_frames, int w, int h, double avg[3], double std_dev[3]);
int  sfa_sequence_apply_normalization(sfa_sequence *seq, int f0, int n, const double avg[3], const double std_dev[3]);

/* ---- device-resident batches (measurement and the multi-pair driver) ------------------------------
 * A job = `batch` independent frame windows of identical size solved in lockstep by the same launches
 * (forward + backward of a jet, several jets ...), all inputs resident in HBM. */
typedef struct sfa_job sfa_job;
int  sfa_job_create(sfa_ctx *ctx, const sfa_params *p, int w, int h, int batch, sfa_job **out);
void sfa_job_destroy(sfa_job *job);
/* frames / initial flow of batch element b (host -> HBM); chw NULL = ones */
int  sfa_job_upload(sfa_job *job, int b, const float *const *frames, int n_frames, const float *wx, const float *wy,
                    int stride, const float *const chw[3]);
/* the same with the frames of the window taken from a resident sequence on the same GPU: frame f of the window = sequence frame frame_index[f] */
int  sfa_job_upload_resident(sfa_job *job, int b, const sfa_sequence *seq, const int *frame_index, int n_frames, const float *wx, const float *wy,
                             int stride, const float *const chw[3]);
int  sfa_job_reset_flow(sfa_job *job);               /* re-arm every element with the uploaded initial flow */
int  sfa_job_run(sfa_job *job);                      /* the whole coarse-to-fine path on the ctx stream; may be called again on the same uploads
                                                        (same result: presmoothing, cfg sigma > 0, is applied once per upload) */
int  sfa_job_download(sfa_job *job, int b, float *wx, float *wy, int stride, float change[2]);
/* Variational_MT::getOcclusions() of window b after the run: -1 occluded in the past / forward terms only, +1 in the future, 0 none */
int  sfa_job_download_occlusions(sfa_job *job, int b, float *occ, int stride);
/* Per-alternation labels (key slow_flow_occlusions_output: the reference writes <prefix><alter>.png after the discrete step of every alternation
 * alter >= 1, variational_mt.cpp:275-285; every pyramid level overwrites the file, the finest level's survives).  Enable before the run; after it
 * download the finest level's labels of alternation 1 <= alter < niter_alter of window b. */
int  sfa_job_keep_alternation_occlusions(sfa_job *job, int on);
int  sfa_job_download_alternation_occlusions(sfa_job *job, int b, int alter, float *occ, int stride);
double sfa_job_mpix_iters(const sfa_job *job);       /* sum over the job's SOR solves of w*h*K / 1e6, per run */
double sfa_job_device_bytes(const sfa_job *job);     /* device memory the job holds right now (arena + solver workspaces shaped so far) */

/* SOR-only resident batch: `batch` independent systems of one size */
typedef struct sfa_sor_batch sfa_sor_batch;
int  sfa_sor_batch_create(sfa_ctx *ctx, int w, int h, int batch, sfa_sor_batch **out);
void sfa_sor_batch_destroy(sfa_sor_batch *sb);
int  sfa_sor_batch_upload(sfa_sor_batch *sb, int b, const float *du, const float *dv, const float *a11, const float *a12,
                          const float *a22, const float *b1, const float *b2, const float *sh, const float *sv, int stride);
int  sfa_sor_batch_run(sfa_sor_batch *sb, int iterations, float omega);   /* prepare + solve + finish, async */
int  sfa_sor_batch_download(sfa_sor_batch *sb, int b, float *du, float *dv, int stride);

/* ---- test hook: the bound of the solver's in-kernel waits --------------------------------------------------------------------------
 * Every wait of one workgroup for another inside the SOR kernels is bounded (2^22 polls); a wait that gives up poisons the launch, every other wait
 * of the launch gives up at its next look at the error word, the kernel drains, and the entry point returns SFA_ERR_TIMEOUT.  `spins` > 0 shortens
 * that bound for this context (1: any wait that is not satisfied at once gives up) so that the path can be exercised; 0 restores the default.
 * A context that has returned SFA_ERR_TIMEOUT stays usable: progress words, tickets and the error word are reset by the next launch. */
int  sfa_ctx_set_wait_bound(sfa_ctx *ctx, unsigned spins);

/* ---- test / tooling hooks: the library's cross-check and what-if paths ------------------------------------------------------------
 * sfa_debug_set: one switch ("SFA_UNFUSED", "SFA_SOR_CHAIN", ... -- the names tools/README.md lists); value NULL = back to the default.  The library reads
 * these names from the ENVIRONMENT only when SFA_DEBUG=1 is set at the first sfa_ctx_create of the process: a drop-in caller's environment cannot otherwise
 * select other kernels.  Process-wide; not to be called while a refinement runs.
 * sfa_ctx_set_verbose: the reference's "inner it / outer it ... avg change" lines (variational_mt.cpp:404-405, 431-432) on stdout for this context; costs a
 * host round trip per iteration. */
int  sfa_debug_set(const char *name, const char *value);
int  sfa_ctx_set_verbose(sfa_ctx *ctx, int on);

/* ---- test hook: the division of the normalised data terms --------------------------------------------------------------------
 * The cfg-default instance of the fused assembly kernel forms the quotients r^2 / n and t / n of variational_aux_mt.cpp:240-250, 333-347, 479-490, 556-572
 * with the hardware's correctly-rounded chain and ONE refined reciprocal per denominator, behind range guards (kernels.hip: recip_of / div_by / num_ok).
 * Per element: q_chain = that chain without any guard, q_exact = the IEEE division, admitted = 1 where the guards let the chain be used.  The parity test
 * asserts q_chain == q_exact bit for bit wherever admitted == 1. */
int  sfa_division_chain(sfa_ctx *ctx, const float *numerators, const float *denominators, float *q_chain, float *q_exact, unsigned char *admitted, size_t n);

/* ---- in-library kernel timing (HIP events on the context's stream) ---------------------------------
 * While enabled every SOR solve kernel launch is bracketed by an event pair on the launch stream. */
int  sfa_profile_enable(sfa_ctx *ctx, int on);
int  sfa_profile_read(sfa_ctx *ctx, int *n_sor_launches, double *sor_ms_total, double *sor_bytes_total);
/* the same for the data-term assembly kernel (pixels x data terms of the bracketed launches), and the name of the solver kernel shape the last
 * solve was launched with (so that a measurement can be matched to the kernel it was taken from) */
int  sfa_profile_read_kernels(sfa_ctx *ctx, int *n_assembly_launches, double *assembly_ms_total, double *assembly_pixel_terms, char *sor_kernel, int sor_kernel_len);
/* wall bracket on the stream: start/stop an event pair around arbitrary enqueued work */
int  sfa_timer_start(sfa_ctx *ctx);
int  sfa_timer_stop(sfa_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif
