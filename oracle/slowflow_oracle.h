/*
 * slowflow_oracle.h -- CPU restatement of the slowflow variational-refinement hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (slowflow_amd/, include/) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker.
 *
 * Plain scalar C, strict IEEE fp32 (build with -ffp-contract=off, no fast-math), every
 * function written from the equations of the reference and citing the reference file:line
 * it follows (paths relative to the reference root).  All planes are planar fp32,
 * row-major, explicit (w, h, stride); only the w valid columns of a row are defined on
 * output -- padding lanes (stride - w) are never read as meaningful and written as 0 or
 * left untouched.
 *
 * Pin status (see oracle/README.md and tests/test_oracle_pin.py):
 *   bit-exact against the compiled reference (oracle/_ref, built from the reference's own
 *   solver.c / image.c / variational_aux.c / penalty_functions headers):
 *     orc_sor_coupled, orc_convolve_{horiz,vert} (3/5-tap), orc_image_warp,
 *     orc_sub_laplacian, orc_dpsis_weight (output 0), orc_derivative_stack,
 *     orc_psi_deriv_{scalar,vec}, orc_psi_apply_vec, and -- end to end -- orc_variational_2frame
 *     (the reference's original two-frame variational(), variational.c:101);
 *   near-pinned (same operator in the reference's 2-frame variational_aux.c, different
 *   rounding order, <= few ulp): orc_smoothness (method 1), orc_add_data_and_match
 *   (normalised, ModL1);
 *   PARITY UNPINNED (restated from the source text only; variational_aux_mt.cpp /
 *   variational_mt.cpp need the absent GCO/OpenCV headers and cannot be built here):
 *     orc_add_data_and_match_ref, the unnormalised data-term branches, smoothing
 *     methods 0/2, orc_compute_one_level orchestration, orc_normalize, and the OpenCV
 *     defined pyramid arithmetic (orc_gaussian_blur_cv, orc_resize_linear_cv), and
 *     optimizeOcc: orc_occlusion_costs (restated) and orc_grid_cut (GCO v3.0 is not
 *     vendored; an exact fp64 minimum cut stands in for its two-label expansion).
 */
#ifndef SLOWFLOW_ORACLE_H
#define SLOWFLOW_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_REF 8

typedef struct {
    int id;        /* 0 quadratic, 2 lorentzian, 3 trunc mod-L1, 4 geman-mcclure, else mod-L1 */
    float eps;
    float trunc;
} orc_penalty;

/* mirrors the cfg keys read by variational_mt.cpp:173-192, 533-568 */
typedef struct {
    int   S;                  /* slow_flow_S; ref = S-1; frames = 2*ref+1 */
    int   one_direction;      /* slow_flow_method == "forward" */
    int   smoothing;          /* slow_flow_smoothing */
    int   dataterm_norm;      /* slow_flow_dataterm */
    int   niter_alter, niter_outer, niter_inner, niter_solver;
    float thres_outer, thres_inner;
    float sor_omega;
    float alpha, gamma, delta;
    orc_penalty robust_color, robust_grad, robust_reg;
    float rho[ORC_MAX_REF], omega[ORC_MAX_REF];
    int   hbit;               /* 16bit */
    float norm_avg[3], norm_std[3];
    int   occlusion_reasoning;
    int   layers;             /* slow_flow_layers */
    float p_scale;            /* slow_flow_p_scale */
    float presmooth_sigma;    /* >0: cfg sigma>0, value of slow_flow_sigma */
    float occlusion_penalty;  /* slow_flow_occlusion_penalty */
    float occlusion_alpha;    /* slow_flow_occlusion_alpha */
    int   niter_graphc;       /* slow_flow_niter_graphc */
    int   sor_order;          /* additive key slow_flow_sor_order: 0 lexicographic (the reference), 1 red_black (labelled mode, a different algorithm) */
} orc_params;

void orc_params_default(orc_params *p);

/* penalty_functions headers: psi'(x^2).  scalar overload = double inside, vec = pure fp32 */
float orc_psi_deriv_scalar(const orc_penalty *pen, float xsq);
float orc_psi_deriv_vec(const orc_penalty *pen, float xsq);
/* psi itself, v4sf overload (modified_l1_norm.h:24-26, quadratic_function.h:18-20, lorentzian.h:24-32,
 * trunc_modified_l1_norm.h:27-36, geman_mcclure.h:24-26) */
float orc_psi_apply_vec(const orc_penalty *pen, float xsq);

/* optimizeOcc (variational_aux_mt.cpp:758-887), first half: the two data costs per pixel.  succ / toref: nslots derivative
 * stacks (8 x 3 planes each, orc_derivative_stack order), masks: nslots raw warp masks.  PARITY UNPINNED (the file does
 * not compile here); psi is pinned. */
void orc_occlusion_costs(float *d0, float *d1, const float *masks, const float *succ, const float *toref, int ref,
                         const float *rho, const float *omega, float delta_over3, float gamma_over3, float penalty,
                         const orc_penalty *color, const orc_penalty *grad, int w, int h, int stride);
/* second half: GCO's two-label expansion = the exact minimum of sum_p D_{l_p}(p) + alpha * #{4-neighbours p,q: l_p != l_q}
 * (s-t minimum cut; Dinic, fp64).  occ[p] = 2*l_p - 1.  Returns the minimum energy.  GCO v3.0 is not vendored: PARITY
 * UNPINNED (ties may be broken differently). */
double orc_grid_cut(float *occ, const float *d0, const float *d1, float alpha, int w, int h, int stride);
/* energy of a labelling occ in {-1,+1} under the same model */
double orc_grid_cut_energy(const float *occ, const float *d0, const float *d1, float alpha, int w, int h, int stride);

/* image.c:400-526: 3-tap (order 1) / 5-tap (order 2) antisymmetric derivative filters.
 * order==2: c = [1/12,-8/12,-0,8/12,-1/12]; order==1: c = [-.5,-0,.5] */
void orc_convolve_horiz(float *dst, const float *src, int w, int h, int stride, int order);
void orc_convolve_vert(float *dst, const float *src, int w, int h, int stride, int order);

/* variational_aux_mt.cpp:722-756.  src/dst: 3 planes of h*stride each; mask may be NULL */
void orc_image_warp(float *dst3, float *mask, const float *src3, const float *wx, const float *wy,
                    int w, int h, int stride, int factor);

/* variational_mt.cpp:113-133: from I1 (=im1p) and I2 (=im2p) colour images compute
 * Iz = I1-I2, M = .5*(I2+I1), Ix,Iy = D5(M), Ixx,Ixy = D5(Ix), Iyy = D5y(Iy), Ixz,Iyz = D5(Iz).
 * every output = 3 planes.  order: Ix,Iy,Iz,Ixx,Ixy,Iyy,Ixz,Iyz (each 3*h*stride floats) */
void orc_derivative_stack(float *out8x3, const float *I1, const float *I2, int w, int h, int stride);

/* variational_aux_mt.cpp:673-719 (first output only; the two others are unused downstream) */
void orc_dpsis_weight(float *dst, const float *im3, int w, int h, int stride, float coef,
                      const float avg[3], const float std[3], int hbit);

/* variational_aux_mt.cpp:18-127 */
void orc_smoothness(int method, float *dst_horiz, float *dst_vert, const float *uu, const float *vv,
                    const float *dpsis, int w, int h, int stride, float alpha, const orc_penalty *reg);

/* variational_aux_mt.cpp:130-161 */
void orc_sub_laplacian(float *dst, const float *src, const float *wh, const float *wv, int w, int h, int stride);

/* variational_aux_mt.cpp:166-403 / 408-634.  D = 8 colour images (order as orc_derivative_stack);
 * chw[k] = channel-weight plane k, indexed with THIS level's y*stride+x (the reference walks the
 * level-0 weight planes linearly at every level, variational_aux_mt.cpp:177,366-371).
 * _ref returns -1 for the logic_error of :419 */
void orc_add_data_and_match(float *a11, float *a12, float *a22, float *b1, float *b2, const float *mask,
                            const float *du, const float *dv, const float *D8x3, const float *const chw[3],
                            int w, int h, int stride, float delta_over3, float gamma_over3, float s,
                            int dt_norm, const orc_penalty *color, const orc_penalty *grad);
int orc_add_data_and_match_ref(float *a11, float *a12, float *a22, float *b1, float *b2, const float *mask,
                               const float *du, const float *dv, const float *D8x3, const float *const chw[3],
                               int w, int h, int stride, float delta_over3, float gamma_over3, float s,
                               int dt_norm, const orc_penalty *color, const orc_penalty *grad);

/* solver.c:63-399 (fast solver arithmetic, raster order; a11/a12/a22 are overwritten by the
 * inverted 2x2 blocks) and solver.c:17-57 (readable) */
void orc_sor_coupled(float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2,
                     const float *sh, const float *sv, int w, int h, int stride, int iterations, float omega);
/* red-black ordering of the same point update: the CPU twin of the product's labelled `slow_flow_sor_order red_black` mode -- NOT the reference */
void orc_sor_red_black(float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2,
                       const float *sh, const float *sv, int w, int h, int stride, int iterations, float omega);
void orc_sor_coupled_readable(float *du, float *dv, const float *a11, const float *a12, const float *a22,
                              const float *b1, const float *b2, const float *sh, const float *sv,
                              int w, int h, int stride, int iterations, float omega);

/* variational_mt.cpp:17-85.  frames[f] = 3 planes.  writes avg/std (double->6 significant
 * digits->float round trip of :71-84,250-251 is applied by orc_normalize_publish) */
void orc_normalize(float **frames, int F, int w, int h, int stride, double avg[3], double std[3]);
void orc_normalize_publish(const double avg[3], const double std[3], float avg_f[3], float std_f[3]);

/* test hook: labels to use after the discrete step of alternation a (labels + a*h*stride), and the recorded relative energy excess of each
 * over the run's own exact minimum -- a minimum cut is not unique, so parity of the flow is checked under the SAME labelling */
void orc_force_labels(const float *labels, int n);
double orc_forced_gap(int alter);

/* variational_mt.cpp:293-320: occlusion / direction weighting of the 2*ref warp masks, in place */
void orc_mask_weight(float *masks, const float *occ, int ref, float data_norm, int one_direction, int w, int h, int stride);

/* variational_mt.cpp:169-493.  frames: 2*ref+1 colour images of this level; wx,wy in/out;
 * chw[3]: channel weight planes; occ: out occlusion plane (h*stride) or NULL; change[2] out */
/* records the values of the reference's per-iteration "avg change" lines of the next orc_compute_one_level calls: entries of 4 floats (0 inner / 1 outer, iteration, a, b) */
void orc_set_change_log(float *buf, int cap_entries);
int orc_change_log_count(void);
int orc_compute_one_level(const orc_params *p, float *wx, float *wy, float *const *frames,
                          const float *const chw[3], float *occ, int w, int h, int stride, float change[2]);

/* OpenCV-defined pyramid arithmetic, restated from the documented semantics (UNPINNED) */
void orc_gaussian_blur_cv(float *dst, const float *src, int w, int h, int stride, float sigma);
void orc_resize_linear_cv(float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride);
/* cv::resize(src, dst, Size(0,0), fx, fy, INTER_LINEAR) (slow_flow.cpp:552): source coordinate (dst + .5) / f - .5 */
void orc_resize_linear_fx(float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride, double fx, double fy);
int  orc_pyramid_sizes(int w, int h, int layers, float p_scale, int *ws, int *hs);
/* optional presmoothing of level 0 (cfg sigma > 0, variational_mt.cpp:590-597): gaussian_filter (image.c:310-348) applied with the
 * generic convolve_horiz / convolve_vert (image.c:537-644).  Pinned bit-exact against the compiled image.c. */
void orc_gaussian_presmooth(float *dst, const float *src, int w, int h, int stride, float sigma);

/* variational_mt.cpp:526-784 (chw == NULL: all ones, :534-539; as in the reference the level-0
 * weight planes are NOT rescaled per level) */
int orc_variational(const orc_params *p, float *wx, float *wy, float *const *frames, const float *const chw[3],
                    int w, int h, int stride, float change[2]);

/* ------------------------------------------------------------------------------------------
 * The reference's original two-frame refinement (epic_flow_extended/variational.c:101-143 with variational_aux.c): one
 * level, fixed modified-L1 penalties (eps 0.001), weights halved.  The reference file compiles here as is, so this
 * restatement is pinned END TO END, bit for bit, against the real `variational()` (tests/test_oracle_pin.py).
 * ---------------------------------------------------------------------------------------- */
typedef struct {   /* variational_params_t, variational.h:16-25 */
    float alpha, gamma, delta, sigma;
    int niter_outer, niter_inner, niter_solver;
    float sor_omega;
} orc_params_2f;
void orc_variational_2frame(float *wx, float *wy, const float *im1, const float *im2, const orc_params_2f *p, int w, int h, int stride);

#ifdef __cplusplus
}
#endif
#endif
