/*
 * slowflow_oracle.c -- CPU restatement of the slowflow variational-refinement hot path.
 * TEST INFRASTRUCTURE ONLY (see slowflow_oracle.h for the pin status of every function).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * The reference is compiled -O3 -msse4 without FMA or fast-math (CMakeLists.txt:6), i.e. every
 * fp32 expression is evaluated left-to-right with one rounding per operation; the restatement
 * keeps each expression's association exactly as written in the reference source.
 */
#include "slowflow_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float *plane_alloc(size_t n) {
    float *p = (float *)calloc(n ? n : 1, sizeof(float));
    if (!p) { fprintf(stderr, "slowflow_oracle: out of memory\n"); abort(); }
    return p;
}

void orc_params_default(orc_params *p) {
    /* slow_flow.cpp:64-128 (driver defaults) */
    memset(p, 0, sizeof(*p));
    p->S = 2; p->one_direction = 0; p->smoothing = 1; p->dataterm_norm = 1;
    p->niter_alter = 10; p->niter_outer = 10; p->niter_inner = 1; p->niter_solver = 30;
    p->thres_outer = 1e-5f; p->thres_inner = 1e-5f; p->sor_omega = 1.9f;
    p->alpha = 4.0f; p->gamma = 6.0f; p->delta = 1.0f;
    p->robust_color.id = 1; p->robust_color.eps = 0.001f; p->robust_color.trunc = 0.5f;
    p->robust_grad = p->robust_color; p->robust_reg = p->robust_color;
    for (int a = 0; a < ORC_MAX_REF; a++) { p->rho[a] = 1; p->omega[a] = 1; }
    p->omega[0] = 0; p->omega[1] = 2;
    p->hbit = 1;
    for (int k = 0; k < 3; k++) { p->norm_avg[k] = 0; p->norm_std[k] = 1; }
    p->occlusion_reasoning = 1;
    p->layers = 1; p->p_scale = 0.9f; p->presmooth_sigma = 0;
    p->occlusion_penalty = 0.1f; p->occlusion_alpha = 0.1f; p->niter_graphc = 10;   /* slow_flow.cpp:117-118 */
    p->sor_order = 0;
}

/* ------------------------------------------------------------------------------------------
 * penalty_functions headers -- psi'(x^2)
 * ---------------------------------------------------------------------------------------- */

/* scalar overloads: quadratic_function.h:23, modified_l1_norm.h:28-30 (double epsilon_sq member,
 * double sqrt), lorentzian.h:36-38, trunc_modified_l1_norm.h:40-45 (float epsilon_sq member and
 * float sqrt: all-float), geman_mcclure.h:28-32 */
float orc_psi_deriv_scalar(const orc_penalty *pen, float xsq) {
    const float e2f = pen->eps * pen->eps;     /* epsilon_sq(e*e): float product */
    const double e2d = (double)e2f;
    switch (pen->id) {
    case 0: return 1.0f;
    case 2: return (float)(1 / (2 * e2d + xsq));
    case 3:
        if (sqrtf(xsq) > pen->trunc) return 0;
        return 1 / (2 * sqrtf(xsq + e2f));
    case 4: {
        float tmp = (float)(e2d + xsq);
        tmp = tmp * tmp;
        return (float)((e2d + 2 * xsq) / tmp);
    }
    default: return (float)(1 / (2 * sqrt(xsq + e2d)));
    }
}

/* v4sf overloads: pure fp32, sqrtps/divps (modified_l1_norm.h:32-34, lorentzian.h:40-42,
 * trunc_modified_l1_norm.h:47-56, geman_mcclure.h:34-38) */
float orc_psi_deriv_vec(const orc_penalty *pen, float xsq) {
    const float e2 = pen->eps * pen->eps;
    switch (pen->id) {
    case 0: return 1.0f;
    case 2: return 1.0f / (2.0f * e2 + xsq);
    case 3: {
        float out = 1.0f / (2.0f * sqrtf(xsq + e2));
        if (sqrtf(xsq) > pen->trunc) out = 0;
        return out;
    }
    case 4: {
        float tmp = e2 + xsq;
        tmp = tmp * tmp;
        return (e2 + 2.0f * xsq) / tmp;
    }
    default: return 1.0f / (2.0f * sqrtf(xsq + e2));
    }
}

float orc_psi_apply_vec(const orc_penalty *pen, float xsq) {
    const float e2 = pen->eps * pen->eps;
    switch (pen->id) {
    case 0: return xsq;
    case 2: return (float)log(1 + 0.5 * (double)xsq / (double)e2);              /* double: epsilon_sq is a double member */
    case 3: {
        float out = sqrtf(xsq + e2);
        if (sqrtf(xsq) > pen->trunc) out = sqrtf(pen->trunc + e2);              /* sic: truncation, not its square */
        return out;
    }
    case 4: return xsq / ((xsq + 1.0f) * (xsq + 1.0f));
    default: return sqrtf(xsq + e2);
    }
}

/* ------------------------------------------------------------------------------------------
 * image.c -- derivative filters
 * ---------------------------------------------------------------------------------------- */

/* coefficients as convolution_new(order, half, even=0) builds them (image.c:363-366 with
 * variational_mt.cpp:570-573): coeffs[order-i] = +half[i]; coeffs[order+i] = -half[i] */
static void deriv_coeffs(int order, float c[5]) {
    if (order == 2) {
        const float half[3] = {0.0f, -8.0f / 12.0f, 1.0f / 12.0f};
        for (int i = 0; i <= 2; i++) { c[2 - i] = +half[i]; c[2 + i] = -half[i]; }
    } else {
        const float half[2] = {0.0f, -0.5f};
        for (int i = 0; i <= 1; i++) { c[1 - i] = +half[i]; c[1 + i] = -half[i]; }
    }
}

static inline int clampi(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }

/* image.c:460-526: replicate border on both sides (the shifted copies src_m1/src_m2/src_p1/src_p2) */
static void conv_horiz_fast(float *dst, const float *src, int w, int h, int stride, int order, const float c[5]) {
    for (int y = 0; y < h; y++) {
        const float *s = src + (size_t)y * stride;
        float *d = dst + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            if (order == 2) {
                const float m2 = s[clampi(x - 2, 0, w - 1)], m1 = s[clampi(x - 1, 0, w - 1)];
                const float p1 = s[clampi(x + 1, 0, w - 1)], p2 = s[clampi(x + 2, 0, w - 1)];
                d[x] = c[0] * m2 + c[1] * m1 + c[2] * s[x] + c[3] * p1 + c[4] * p2;   /* image.c:521 */
            } else {
                const float m1 = s[clampi(x - 1, 0, w - 1)], p1 = s[clampi(x + 1, 0, w - 1)];
                d[x] = c[0] * m1 + c[1] * s[x] + c[2] * p1;                            /* image.c:482 */
            }
        }
    }
}

void orc_convolve_horiz(float *dst, const float *src, int w, int h, int stride, int order) {
    float c[5];
    deriv_coeffs(order, c);
    conv_horiz_fast(dst, src, w, h, stride, order, c);
}

/* image.c:400-458: border rows fold the coefficients at run time in fp32 */
static void conv_vert_fast(float *dst, const float *src, int w, int h, int stride, int order, const float c[5]) {
    for (int y = 0; y < h; y++) {
        float *d = dst + (size_t)y * stride;
        const float *s0 = src + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            if (order == 2) {
                if (y == 0)                 /* image.c:434 */
                    d[x] = (c[0] + c[1] + c[2]) * s0[x] + c[3] * s0[x + stride] + c[4] * s0[x + 2 * stride];
                else if (y == 1)            /* image.c:439 */
                    d[x] = (c[0] + c[1]) * s0[x - stride] + c[2] * s0[x] + c[3] * s0[x + stride] + c[4] * s0[x + 2 * stride];
                else if (y == h - 2)        /* image.c:451 */
                    d[x] = c[0] * s0[x - 2 * stride] + c[1] * s0[x - stride] + c[2] * s0[x] + (c[3] + c[4]) * s0[x + stride];
                else if (y == h - 1)        /* image.c:455 */
                    d[x] = c[0] * s0[x - 2 * stride] + c[1] * s0[x - stride] + (c[2] + c[3] + c[4]) * s0[x];
                else                        /* image.c:446 */
                    d[x] = c[0] * s0[x - 2 * stride] + c[1] * s0[x - stride] + c[2] * s0[x] + c[3] * s0[x + stride] + c[4] * s0[x + 2 * stride];
            } else {
                if (y == 0)                 /* image.c:408 */
                    d[x] = (c[0] + c[1]) * s0[x] + c[2] * s0[x + stride];
                else if (y == h - 1)        /* image.c:420 */
                    d[x] = c[0] * s0[x - stride] + (c[1] + c[2]) * s0[x];
                else                        /* image.c:415 */
                    d[x] = c[0] * s0[x - stride] + c[1] * s0[x] + c[2] * s0[x + stride];
            }
        }
    }
}

void orc_convolve_vert(float *dst, const float *src, int w, int h, int stride, int order) {
    float c[5];
    deriv_coeffs(order, c);
    conv_vert_fast(dst, src, w, h, stride, order, c);
}

/* ------------------------------------------------------------------------------------------
 * variational_aux_mt.cpp:722-756 -- image_warp
 * ---------------------------------------------------------------------------------------- */
#define RECTIFY(a, b) (((a) < 0) ? (0) : (((a) < (b)-1) ? (a) : ((b)-1)))

void orc_image_warp(float *dst3, float *mask, const float *src3, const float *wx, const float *wy,
                    int w, int h, int stride, int factor) {
    const size_t plane = (size_t)stride * h;
    if (factor == 0) {                                        /* :723-728 */
        for (int k = 0; k < 3; k++)
            for (int y = 0; y < h; y++)
                memcpy(dst3 + k * plane + (size_t)y * stride, src3 + k * plane + (size_t)y * stride, sizeof(float) * w);
        /* the reference memset()s mask BYTES to 1 here (:726); unreachable from get_derivatives,
         * the oracle leaves mask untouched */
        return;
    }
    for (int j = 0; j < h; j++) {
        size_t offset = (size_t)j * stride;
        for (int i = 0; i < w; i++, offset++) {
            const float xx = i + factor * wx[offset];        /* :735 */
            const float yy = j + factor * wy[offset];
            const int x = (int)floor(xx);
            const int y = (int)floor(yy);
            const float dx = xx - x;
            const float dy = yy - y;
            if (mask) mask[offset] = (xx >= 0 && xx <= w - 1 && yy >= 0 && yy <= h - 1);
            const int x1 = RECTIFY(x, w), x2 = RECTIFY(x + 1, w);
            const int y1 = RECTIFY(y, h), y2 = RECTIFY(y + 1, h);
            for (int k = 0; k < 3; k++) {
                const float *s = src3 + k * plane;
                dst3[k * plane + offset] = s[(size_t)y1 * stride + x1] * (1.0f - dx) * (1.0f - dy)
                                         + s[(size_t)y1 * stride + x2] * dx * (1.0f - dy)
                                         + s[(size_t)y2 * stride + x1] * (1.0f - dx) * dy
                                         + s[(size_t)y2 * stride + x2] * dx * dy;          /* :748-753 */
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * variational_mt.cpp:113-133 -- one derivative stack
 * ---------------------------------------------------------------------------------------- */
enum { D_IX = 0, D_IY, D_IZ, D_IXX, D_IXY, D_IYY, D_IXZ, D_IYZ };

void orc_derivative_stack(float *out, const float *I1, const float *I2, int w, int h, int stride) {
    const size_t plane = (size_t)stride * h, cimg = 3 * plane;
    float *mean = plane_alloc(cimg);
    float *Iz = out + D_IZ * cimg;
    for (int k = 0; k < 3; k++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const size_t o = k * plane + (size_t)y * stride + x;
                mean[o] = 0.5f * (I2[o] + I1[o]);   /* :120 */
                Iz[o] = I1[o] - I2[o];              /* :122 */
            }
    for (int k = 0; k < 3; k++) {
        const size_t o = k * plane;
        orc_convolve_horiz(out + D_IX * cimg + o, mean + o, w, h, stride, 2);                  /* :127 */
        orc_convolve_vert(out + D_IY * cimg + o, mean + o, w, h, stride, 2);                   /* :128 */
        orc_convolve_horiz(out + D_IXX * cimg + o, out + D_IX * cimg + o, w, h, stride, 2);    /* :129 */
        orc_convolve_vert(out + D_IXY * cimg + o, out + D_IX * cimg + o, w, h, stride, 2);     /* :130 */
        orc_convolve_vert(out + D_IYY * cimg + o, out + D_IY * cimg + o, w, h, stride, 2);     /* :131 */
        orc_convolve_horiz(out + D_IXZ * cimg + o, Iz + o, w, h, stride, 2);                   /* :132 */
        orc_convolve_vert(out + D_IYZ * cimg + o, Iz + o, w, h, stride, 2);                    /* :133 */
    }
    free(mean);
}

/* ------------------------------------------------------------------------------------------
 * variational_aux_mt.cpp:673-719 -- local smoothness weight
 * ---------------------------------------------------------------------------------------- */
void orc_dpsis_weight(float *dst, const float *im3, int w, int h, int stride, float coef,
                      const float avg[3], const float std[3], int hbit) {
    const size_t plane = (size_t)stride * h;
    float *lum = plane_alloc(plane), *lx = plane_alloc(plane), *ly = plane_alloc(plane);
    const float *c1 = im3, *c2 = im3 + plane, *c3 = im3 + 2 * plane;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * stride + x;
            const float v = 0.299f * (c1[o] * std[0] + avg[0]) + 0.587f * (c2[o] * std[1] + avg[1]) + 0.114f * (c3[o] * std[2] + avg[2]);
            lum[o] = hbit ? v / 65535.0f : v / 255.0f;       /* :681,683 */
        }
    orc_convolve_horiz(lx, lum, w, h, stride, 2);
    orc_convolve_vert(ly, lum, w, h, stride, 2);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * stride + x;
            const float n = -coef * sqrtf(lx[o] * lx[o] + ly[o] * ly[o]);   /* :699 */
            dst[o] = 0.5f * expf(n);                                       /* :700 */
        }
    free(lum); free(lx); free(ly);
}

/* ------------------------------------------------------------------------------------------
 * variational_aux_mt.cpp:18-127 -- smoothness weights
 * ---------------------------------------------------------------------------------------- */
void orc_smoothness(int method, float *dst_horiz, float *dst_vert, const float *uu, const float *vv,
                    const float *dpsis, int w, int h, int stride, float alpha, const orc_penalty *reg) {
    const int s = stride;
    const size_t plane = (size_t)stride * h;
    float *ux1 = plane_alloc(plane), *uy1 = plane_alloc(plane), *vx1 = plane_alloc(plane), *vy1 = plane_alloc(plane);
    float *ux2 = plane_alloc(plane), *uy2 = plane_alloc(plane), *vx2 = plane_alloc(plane), *vy2 = plane_alloc(plane);
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w - 1; i++) {
            const size_t o = (size_t)j * s + i;
            ux1[o] = uu[o + 1] - uu[o];                      /* :27 */
            vx1[o] = vv[o + 1] - vv[o];
        }
    for (int j = 0; j < h - 1; j++)
        for (int i = 0; i < w; i++) {
            const size_t o = (size_t)j * s + i;
            uy1[o] = uu[o + s] - uu[o];                      /* :35 */
            vy1[o] = vv[o + s] - vv[o];
        }
    orc_convolve_horiz(ux2, uu, w, h, stride, 1);
    orc_convolve_horiz(vx2, vv, w, h, stride, 1);
    orc_convolve_vert(uy2, uu, w, h, stride, 1);
    orc_convolve_vert(vy2, vv, w, h, stride, 1);
    if (method <= 1) {
        for (int j = 0; j < h; j++) {
            for (int i = 0; i < w - 1; i++) {
                const size_t o = (size_t)j * s + i;
                float tmp = 0, tmp2 = 0;
                const float tmp_w = dpsis[o] + dpsis[o + 1];
                if (method == 1) {
                    tmp = 0.5f * (uy2[o] + uy2[o + 1]);      /* :57 */
                    tmp2 = 0.5f * (vy2[o] + vy2[o + 1]);
                }
                tmp = ux1[o] * ux1[o] + tmp * tmp;           /* :61 */
                tmp2 = vx1[o] * vx1[o] + tmp2 * tmp2;
                tmp = tmp + tmp2;
                dst_horiz[o] = tmp_w * alpha * orc_psi_deriv_scalar(reg, tmp);   /* :66 */
            }
            for (int i = w - 1; i < s; i++) dst_horiz[(size_t)j * s + i] = 0;    /* :68 */
        }
        for (int j = 0; j < h - 1; j++)
            for (int i = 0; i < w; i++) {
                const size_t o = (size_t)j * s + i;
                float tmp = 0, tmp2 = 0;
                const float tmp_w = dpsis[o] + dpsis[o + s];
                if (method == 1) {
                    tmp = 0.5f * (ux2[o] + ux2[o + s]);      /* :80 */
                    tmp2 = 0.5f * (vx2[o] + vx2[o + s]);
                }
                tmp = uy1[o] * uy1[o] + tmp * tmp;           /* :84 */
                tmp2 = vy1[o] * vy1[o] + tmp2 * tmp2;
                tmp = tmp + tmp2;
                dst_vert[o] = tmp_w * alpha * orc_psi_deriv_scalar(reg, tmp);    /* :89 */
            }
        for (int i = 0; i < s; i++) dst_vert[(size_t)(h - 1) * s + i] = 0;       /* :92 */
    } else {
        /* :96-116 as written, including the `float w` that shadows the image width inside the loop
         * body (:100), which makes the horizontal test `i < w - 1` compare against (weight - 1) */
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const size_t o = (size_t)j * s + i;
                float tmp = 0;
                float wgt = dpsis[o];
                if (i < wgt - 1) {
                    tmp += ux1[o] * ux1[o] + vx1[o] * vx1[o];
                    wgt += dpsis[o + 1];
                }
                if (j < h - 1) {
                    tmp += vy1[o] * vy1[o] + uy1[o] * uy1[o];
                    wgt += dpsis[o + s];
                }
                dst_horiz[o] = wgt * alpha * orc_psi_deriv_scalar(reg, tmp);
                dst_vert[o] = dst_horiz[o];
            }
    }
    free(ux1); free(uy1); free(vx1); free(vy1); free(ux2); free(uy2); free(vx2); free(vy2);
}

/* ------------------------------------------------------------------------------------------
 * variational_aux_mt.cpp:130-161 -- sub_laplacian (sequential scatter form, as the reference)
 * ---------------------------------------------------------------------------------------- */
void orc_sub_laplacian(float *dst, const float *src, const float *wh, const float *wv, int w, int h, int stride) {
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w - 1; i++) {
            const size_t o = (size_t)j * stride + i;
            const float tmp = wh[o] * (src[o + 1] - src[o]);     /* :138 */
            dst[o] += tmp;
            dst[o + 1] -= tmp;
        }
    for (int j = 0; j < h - 1; j++)
        for (int i = 0; i < w; i++) {
            const size_t o = (size_t)j * stride + i;
            const float tmp = wv[o] * (src[o + stride] - src[o]); /* :152 */
            dst[o] += tmp;
            dst[o + stride] -= tmp;
        }
}

/* ------------------------------------------------------------------------------------------
 * variational_aux_mt.cpp:166-403 -- successive-frames data term
 * ---------------------------------------------------------------------------------------- */
static const float datanorm = 0.1f * 0.1f;   /* variational_aux_mt.h:23 */

void orc_add_data_and_match(float *a11, float *a12, float *a22, float *b1, float *b2, const float *mask,
                            const float *du, const float *dv, const float *D, const float *const chw[3],
                            int w, int h, int stride, float delta_over3, float gamma_over3, float s,
                            int dt_norm, const orc_penalty *color, const orc_penalty *grad) {
    const size_t plane = (size_t)stride * h, cimg = 3 * plane;
    const float factor = s, factorp1 = s + 1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * stride + x;
            const float u = du[o], v = dv[o], m = mask[o];
            float wk[3], ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
            for (int k = 0; k < 3; k++) {
                const size_t ok = k * plane + o;
                wk[k] = chw[k][o];
                ix[k] = D[D_IX * cimg + ok]; iy[k] = D[D_IY * cimg + ok]; iz[k] = D[D_IZ * cimg + ok];
                ixx[k] = D[D_IXX * cimg + ok]; ixy[k] = D[D_IXY * cimg + ok]; iyy[k] = D[D_IYY * cimg + ok];
                ixz[k] = D[D_IXZ * cimg + ok]; iyz[k] = D[D_IYZ * cimg + ok];
            }
            float A11 = a11[o], A12 = a12[o], A22 = a22[o], B1 = b1[o], B2 = b2[o];
            if (delta_over3) {                                                  /* :189 */
                float r[3], tx[3], ty[3];
                for (int k = 0; k < 3; k++) {
                    r[k] = wk[k] * (iz[k] + ix[k] * factor * u + iy[k] * factor * v - ix[k] * factorp1 * u - iy[k] * factorp1 * v);  /* :190-192 */
                    tx[k] = factor * ix[k] - factorp1 * ix[k];                  /* :198,229 */
                    ty[k] = factor * iy[k] - factorp1 * iy[k];
                }
                if (!dt_norm) {
                    const float t = m * delta_over3 * orc_psi_deriv_vec(color, r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);   /* :196 */
                    for (int k = 0; k < 3; k++) {
                        const float t2 = t * wk[k];
                        A11 += t2 * tx[k] * tx[k];
                        A12 += t2 * tx[k] * ty[k];
                        A22 += t2 * ty[k] * ty[k];
                        B1 -= t2 * iz[k] * tx[k];
                        B2 -= t2 * iz[k] * ty[k];
                    }
                } else {
                    float n[3];
                    for (int k = 0; k < 3; k++) n[k] = tx[k] * tx[k] + ty[k] * ty[k] + datanorm;   /* :236-238 */
                    const float t = m * delta_over3 * orc_psi_deriv_vec(color, r[0] * r[0] / n[0] + r[1] * r[1] / n[1] + r[2] * r[2] / n[2]);   /* :240 */
                    for (int k = 0; k < 3; k++) {
                        float tk = t / n[k];
                        tk = tk * wk[k];
                        A11 += tk * tx[k] * tx[k];
                        A12 += tk * tx[k] * ty[k];
                        A22 += tk * ty[k] * ty[k];
                        B1 -= tk * iz[k] * tx[k];
                        B2 -= tk * iz[k] * ty[k];
                    }
                }
            }
            {   /* gradient constancy :269-364 */
                float r[6], X[3], Y[3], Z[3];
                for (int k = 0; k < 3; k++) {
                    r[2 * k] = wk[k] * (ixz[k] + ixx[k] * factor * u + ixy[k] * factor * v - ixx[k] * factorp1 * u - ixy[k] * factorp1 * v);
                    r[2 * k + 1] = wk[k] * (iyz[k] + ixy[k] * factor * u + iyy[k] * factor * v - ixy[k] * factorp1 * u - iyy[k] * factorp1 * v);
                    X[k] = factor * ixx[k] - factorp1 * ixx[k];
                    Y[k] = factor * iyy[k] - factorp1 * iyy[k];
                    Z[k] = factor * ixy[k] - factorp1 * ixy[k];
                }
                if (!dt_norm) {
                    const float t = m * gamma_over3 * orc_psi_deriv_vec(grad, r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);   /* :280 */
                    for (int k = 0; k < 3; k++) {
                        const float t2 = t * wk[k];
                        A11 += t2 * X[k] * X[k] + t2 * Z[k] * Z[k];
                        A12 += t2 * X[k] * Z[k] + t2 * Z[k] * Y[k];
                        A22 += t2 * Y[k] * Y[k] + t2 * Z[k] * Z[k];
                        B1 -= t2 * ixz[k] * X[k] + t2 * iyz[k] * Z[k];
                        B2 -= t2 * iyz[k] * Y[k] + t2 * ixz[k] * Z[k];
                    }
                } else {
                    float n[6];
                    for (int k = 0; k < 3; k++) {
                        n[2 * k] = X[k] * X[k] + Z[k] * Z[k] + datanorm;        /* :326-331 */
                        n[2 * k + 1] = Y[k] * Y[k] + Z[k] * Z[k] + datanorm;
                    }
                    const float t = m * gamma_over3 * orc_psi_deriv_vec(grad,
                        r[0] * r[0] / n[0] + r[1] * r[1] / n[1] + r[2] * r[2] / n[2] + r[3] * r[3] / n[3] + r[4] * r[4] / n[4] + r[5] * r[5] / n[5]);   /* :333 */
                    for (int k = 0; k < 3; k++) {
                        float ta = t / n[2 * k], tb = t / n[2 * k + 1];
                        ta = ta * wk[k];
                        tb = tb * wk[k];
                        A11 += ta * X[k] * X[k] + tb * Z[k] * Z[k];             /* :343-347 */
                        A12 += ta * X[k] * Z[k] + tb * Z[k] * Y[k];
                        A22 += tb * Y[k] * Y[k] + ta * Z[k] * Z[k];
                        B1 -= ta * ixz[k] * X[k] + tb * iyz[k] * Z[k];
                        B2 -= tb * iyz[k] * Y[k] + ta * ixz[k] * Z[k];
                    }
                }
            }
            a11[o] = A11; a12[o] = A12; a22[o] = A22; b1[o] = B1; b2[o] = B2;
        }
}

/* ------------------------------------------------------------------------------------------
 * variational_aux_mt.cpp:408-634 -- reference-frame data term
 * ---------------------------------------------------------------------------------------- */
int orc_add_data_and_match_ref(float *a11, float *a12, float *a22, float *b1, float *b2, const float *mask,
                               const float *du, const float *dv, const float *D, const float *const chw[3],
                               int w, int h, int stride, float delta_over3, float gamma_over3, float s,
                               int dt_norm, const orc_penalty *color, const orc_penalty *grad) {
    const size_t plane = (size_t)stride * h, cimg = 3 * plane;
    float factor = s;
    const float factorsq = factor * factor;                  /* :417 */
    if (s == 0) return -1;                                   /* :419-421 */
    if (s >= 0) factor = -factor;                            /* :424-425 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * stride + x;
            const float u = du[o], v = dv[o], m = mask[o];
            float wk[3], ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
            for (int k = 0; k < 3; k++) {
                const size_t ok = k * plane + o;
                wk[k] = chw[k][o];
                ix[k] = D[D_IX * cimg + ok]; iy[k] = D[D_IY * cimg + ok]; iz[k] = D[D_IZ * cimg + ok];
                ixx[k] = D[D_IXX * cimg + ok]; ixy[k] = D[D_IXY * cimg + ok]; iyy[k] = D[D_IYY * cimg + ok];
                ixz[k] = D[D_IXZ * cimg + ok]; iyz[k] = D[D_IYZ * cimg + ok];
            }
            float A11 = a11[o], A12 = a12[o], A22 = a22[o], B1 = b1[o], B2 = b2[o];
            if (delta_over3) {                                                  /* :439 */
                float r[3];
                for (int k = 0; k < 3; k++)
                    r[k] = wk[k] * (iz[k] + ix[k] * factor * u + iy[k] * factor * v);   /* :441-443 */
                if (!dt_norm) {
                    float t = m * delta_over3 * orc_psi_deriv_vec(color, r[0] * r[0] / factorsq + r[1] * r[1] / factorsq + r[2] * r[2] / factorsq);  /* :447 */
                    t /= factorsq;
                    float t2;
                    t2 = t * wk[0] * factor;                                    /* :450 */
                    B1 -= t2 * iz[0] * ix[0];
                    B2 -= t2 * iz[0] * iy[0];
                    t2 = t2 * factor;
                    A11 += t2 * ix[0] * ix[0];
                    A12 += t2 * ix[0] * iy[0];
                    A22 += t2 * iy[0] * iy[0];
                    t2 = t * factor * wk[1];                                    /* :458 */
                    B1 -= t2 * iz[1] * ix[1];
                    B2 -= t2 * iz[1] * iy[1];
                    t2 = t2 * factor;
                    A11 += t2 * ix[1] * ix[1];
                    A12 += t2 * ix[1] * iy[1];
                    A22 += t2 * iy[1] * iy[1];
                    t2 = t * factor * wk[2];                                    /* :466 */
                    B1 -= t2 * iz[2] * ix[2];
                    B2 -= t2 * iz[2] * iy[2];
                    t2 = t * factor;                                            /* :469 (sic: tmp, not tmp2) */
                    A11 += t2 * ix[2] * ix[2];
                    A12 += t2 * ix[2] * iy[2];
                    A22 += t2 * iy[2] * iy[2];
                } else {
                    float n[3];
                    for (int k = 0; k < 3; k++) n[k] = factorsq * ix[k] * ix[k] + factorsq * iy[k] * iy[k] + datanorm;   /* :475-477 */
                    const float t = m * delta_over3 * orc_psi_deriv_vec(color, r[0] * r[0] / n[0] + r[1] * r[1] / n[1] + r[2] * r[2] / n[2]);   /* :479 */
                    for (int k = 0; k < 3; k++) {
                        float tk = t / n[k];
                        tk = tk * wk[k] * factor;                               /* :484 */
                        B1 -= tk * iz[k] * ix[k];
                        B2 -= tk * iz[k] * iy[k];
                        tk = tk * factor;
                        A11 += tk * ix[k] * ix[k];
                        A12 += tk * ix[k] * iy[k];
                        A22 += tk * iy[k] * iy[k];
                    }
                }
            }
            {   /* gradient :511-593 */
                float r[6];
                for (int k = 0; k < 3; k++) {
                    r[2 * k] = wk[k] * (ixz[k] + ixx[k] * factor * u + ixy[k] * factor * v);       /* :511-516 */
                    r[2 * k + 1] = wk[k] * (iyz[k] + ixy[k] * factor * u + iyy[k] * factor * v);
                }
                if (!dt_norm) {
                    float t = m * gamma_over3 * orc_psi_deriv_vec(grad,
                        r[0] * r[0] / factorsq + r[1] * r[1] / factorsq + r[2] * r[2] / factorsq + r[3] * r[3] / factorsq + r[4] * r[4] / factorsq + r[5] * r[5] / factorsq);  /* :520-521 */
                    t /= factorsq;
                    for (int k = 0; k < 3; k++) {
                        float t2 = t * wk[k] * factor;                          /* :524 */
                        B1 -= t2 * ixx[k] * ixz[k] + t2 * ixy[k] * iyz[k];
                        B2 -= t2 * iyy[k] * iyz[k] + t2 * ixy[k] * ixz[k];
                        t2 = t2 * factor;
                        if (k == 0) {                                           /* :528-530 (sic: extra factorsq) */
                            A11 += t2 * factorsq * ixx[k] * ixx[k] + t2 * factorsq * ixy[k] * ixy[k];
                            A12 += t2 * factorsq * ixx[k] * ixy[k] + t2 * factorsq * ixy[k] * iyy[k];
                            A22 += t2 * factorsq * iyy[k] * iyy[k] + t2 * factorsq * ixy[k] * ixy[k];
                        } else {
                            A11 += t2 * ixx[k] * ixx[k] + t2 * ixy[k] * ixy[k];
                            A12 += t2 * ixx[k] * ixy[k] + t2 * ixy[k] * iyy[k];
                            A22 += t2 * iyy[k] * iyy[k] + t2 * ixy[k] * ixy[k];
                        }
                    }
                } else {
                    float n[6];
                    for (int k = 0; k < 3; k++) {
                        n[2 * k] = factorsq * ixx[k] * ixx[k] + factorsq * ixy[k] * ixy[k] + datanorm;      /* :549-554 */
                        n[2 * k + 1] = factorsq * iyy[k] * iyy[k] + factorsq * ixy[k] * ixy[k] + datanorm;
                    }
                    const float t = m * gamma_over3 * orc_psi_deriv_vec(grad,
                        r[0] * r[0] / n[0] + r[1] * r[1] / n[1] + r[2] * r[2] / n[2] + r[3] * r[3] / n[3] + r[4] * r[4] / n[4] + r[5] * r[5] / n[5]);   /* :556 */
                    for (int k = 0; k < 3; k++) {
                        float ta = t / n[2 * k], tb = t / n[2 * k + 1];
                        ta = ta * wk[k] * factor;                               /* :564-565 */
                        tb = tb * wk[k] * factor;
                        B1 -= ta * ixx[k] * ixz[k] + tb * ixy[k] * iyz[k];
                        B2 -= tb * iyy[k] * iyz[k] + ta * ixy[k] * ixz[k];
                        ta = ta * factor;
                        tb = tb * factor;
                        A11 += ta * ixx[k] * ixx[k] + tb * ixy[k] * ixy[k];
                        A12 += ta * ixx[k] * ixy[k] + tb * ixy[k] * iyy[k];
                        A22 += tb * iyy[k] * iyy[k] + ta * ixy[k] * ixy[k];
                    }
                }
            }
            a11[o] = A11; a12[o] = A12; a22[o] = A22; b1[o] = B1; b2[o] = B2;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * solver.c -- SOR
 * ---------------------------------------------------------------------------------------- */

/* solver.c:17-57 */
void orc_sor_coupled_readable(float *du, float *dv, const float *a11, const float *a12, const float *a22,
                              const float *b1, const float *b2, const float *sh, const float *sv,
                              int w, int h, int stride, int iterations, float omega) {
    for (int iter = 0; iter < iterations; iter++)
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const size_t o = (size_t)j * stride + i;
                float sigma_u = 0.0f, sigma_v = 0.0f, sum_dpsis = 0.0f;
                if (j > 0) {
                    sigma_u -= sv[o - stride] * du[o - stride];
                    sigma_v -= sv[o - stride] * dv[o - stride];
                    sum_dpsis += sv[o - stride];
                }
                if (i > 0) {
                    sigma_u -= sh[o - 1] * du[o - 1];
                    sigma_v -= sh[o - 1] * dv[o - 1];
                    sum_dpsis += sh[o - 1];
                }
                if (j < h - 1) {
                    sigma_u -= sv[o] * du[o + stride];
                    sigma_v -= sv[o] * dv[o + stride];
                    sum_dpsis += sv[o];
                }
                if (i < w - 1) {
                    sigma_u -= sh[o] * du[o + 1];
                    sigma_v -= sh[o] * dv[o + 1];
                    sum_dpsis += sh[o];
                }
                const float A11 = a11[o] + sum_dpsis, A12 = a12[o], A22 = a22[o] + sum_dpsis;
                const float det = A11 * A22 - A12 * A12;
                const float B1 = b1[o] - sigma_u, B2 = b2[o] - sigma_v;
                du[o] = (1.0f - omega) * du[o] + omega * (A22 * B1 - A12 * B2) / det;
                dv[o] = (1.0f - omega) * dv[o] + omega * (-A12 * B1 + A11 * B2) / det;
            }
}

/* solver.c:63-399.  Raster order; per pixel exactly the fast solver's operations:
 *   hl = sh[x-1] (f1: 0 at x=0, solver.c:82,94), right neighbour = du[x+1], 0 at x=w-1 (f2: :84,95)
 *   first sweep: dpsis = hl + sh (+ vt if y>0) (+ sv if y<h-1)  (:101,159,214)
 *                A11 = a22+dpsis, A22 = a11+dpsis, det = A11*A22 - a12*a12,
 *                a11 <- A11/det, a22 <- A22/det, a12 <- a12/(-det)      (:102-106)
 *   s1 = sh*du_r (+ vt*du_t) (+ sv*du_b) + b1                            (:108,166,221)
 *   x == 0: du += w*(a11*s1 + a12*s2 - du)                               (:110)
 *   x  > 0: B1 = hl*du_l + s1; du += w*(a11*B1 + a12*B2 - du)            (:113-116)
 * The right / top / bottom neighbours of a 4-pixel block are read before the block's own
 * updates (the v4sf part), the left one after: equivalent to plain raster order. */
void orc_sor_coupled(float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2,
                     const float *sh, const float *sv, int w, int h, int stride, int iterations, float omega) {
    if (w < 2 || h < 2 || iterations < 1) {              /* :66-69 */
        orc_sor_coupled_readable(du, dv, a11, a12, a22, b1, b2, sh, sv, w, h, stride, iterations, omega);
        return;
    }
    for (int iter = 0; iter < iterations; iter++)
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const size_t o = (size_t)j * stride + i;
                const float hl = i > 0 ? sh[o - 1] : 0.0f;
                const float hp = sh[o];
                if (iter == 0) {
                    float dpsis = hl + hp;
                    if (j > 0) dpsis = dpsis + sv[o - stride];
                    if (j < h - 1) dpsis = dpsis + sv[o];
                    const float A11 = a22[o] + dpsis, A22 = a11[o] + dpsis;
                    const float det = A11 * A22 - a12[o] * a12[o];
                    a11[o] = A11 / det;
                    a22[o] = A22 / det;
                    a12[o] = a12[o] / -det;
                }
                const float dur = i < w - 1 ? du[o + 1] : 0.0f;
                const float dvr = i < w - 1 ? dv[o + 1] : 0.0f;
                float s1 = hp * dur, s2 = hp * dvr;
                if (j > 0) { s1 = s1 + sv[o - stride] * du[o - stride]; s2 = s2 + sv[o - stride] * dv[o - stride]; }
                if (j < h - 1) { s1 = s1 + sv[o] * du[o + stride]; s2 = s2 + sv[o] * dv[o + stride]; }
                s1 = s1 + b1[o];
                s2 = s2 + b2[o];
                float B1 = s1, B2 = s2;
                if (i > 0) {
                    B1 = hl * du[o - 1] + s1;
                    B2 = hl * dv[o - 1] + s2;
                }
                du[o] += omega * (a11[o] * B1 + a12[o] * B2 - du[o]);
                dv[o] += omega * (a12[o] * B1 + a22[o] * B2 - dv[o]);
            }
}

/* RED-BLACK ordering of the same point update -- NOT the reference's algorithm (solver.c sweeps in raster order; SURVEY.md 0.1: after 30 sweeps a
 * red-black result is 1e-2 .. 1e-1 away from it).  It restates the labelled throughput / latency mode `slow_flow_sor_order red_black` of the
 * product so that the GPU kernels of that mode have a CPU twin: per sweep first every point with (x + y) even from the current values, then every
 * point with (x + y) odd; per point the fast solver's operations in the fast solver's order (as orc_sor_coupled), the 2x2 blocks inverted first. */
void orc_sor_red_black(float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2,
                       const float *sh, const float *sv, int w, int h, int stride, int iterations, float omega) {
    if (iterations < 1) return;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const size_t o = (size_t)j * stride + i;
            const float hl = i > 0 ? sh[o - 1] : 0.0f, hp = sh[o];
            float dpsis = hl + hp;
            if (j > 0) dpsis = dpsis + sv[o - stride];
            if (j < h - 1) dpsis = dpsis + sv[o];
            const float A11 = a22[o] + dpsis, A22 = a11[o] + dpsis;
            const float det = A11 * A22 - a12[o] * a12[o];
            a11[o] = A11 / det;
            a22[o] = A22 / det;
            a12[o] = a12[o] / -det;
        }
    for (int iter = 0; iter < iterations; iter++)
        for (int color = 0; color < 2; color++)
            for (int j = 0; j < h; j++)
                for (int i = (j + color) & 1; i < w; i += 2) {
                    const size_t o = (size_t)j * stride + i;
                    const float hl = i > 0 ? sh[o - 1] : 0.0f, hp = sh[o];
                    const float dur = i < w - 1 ? du[o + 1] : 0.0f, dvr = i < w - 1 ? dv[o + 1] : 0.0f;
                    float s1 = hp * dur, s2 = hp * dvr;
                    if (j > 0) { s1 = s1 + sv[o - stride] * du[o - stride]; s2 = s2 + sv[o - stride] * dv[o - stride]; }
                    if (j < h - 1) { s1 = s1 + sv[o] * du[o + stride]; s2 = s2 + sv[o] * dv[o + stride]; }
                    s1 = s1 + b1[o];
                    s2 = s2 + b2[o];
                    float B1 = s1, B2 = s2;
                    if (i > 0) { B1 = hl * du[o - 1] + s1; B2 = hl * dv[o - 1] + s2; }
                    du[o] += omega * (a11[o] * B1 + a12[o] * B2 - du[o]);
                    dv[o] += omega * (a12[o] * B1 + a22[o] * B2 - dv[o]);
                }
}

/* ------------------------------------------------------------------------------------------
 * variational_mt.cpp:17-85 -- normalize
 * ---------------------------------------------------------------------------------------- */
void orc_normalize(float **frames, int F, int w, int h, int stride, double avg[3], double std_dev[3]) {
    const size_t plane = (size_t)stride * h;
    for (int c = 0; c < 3; c++) { avg[c] = 0; std_dev[c] = 0; }
    for (int f = 0; f < F; f++) {
        double avg_frame[3] = {0, 0, 0}, sq_frame[3] = {0, 0, 0};
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++)
                for (int c = 0; c < 3; c++) {
                    const float v = frames[f][c * plane + (size_t)i * stride + j];
                    avg_frame[c] += v;
                    sq_frame[c] += v * v;            /* float product accumulated in double (:35) */
                }
        for (int c = 0; c < 3; c++) {
            avg[c] += avg_frame[c] / (h * w);
            std_dev[c] += sq_frame[c] / (h * w);
        }
    }
    for (int c = 0; c < 3; c++) {
        avg[c] /= F;
        std_dev[c] = sqrt((std_dev[c] / F) - avg[c] * avg[c]) / 255.0f;     /* :52 */
    }
    for (int f = 0; f < F; f++)
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++)
                for (int c = 0; c < 3; c++)
                    if (std_dev[c] > 0) {
                        float *p = &frames[f][c * plane + (size_t)i * stride + j];
                        *p = (float)((*p - avg[c]) / std_dev[c]);           /* :64-66 */
                    }
}

/* :71-84 publishes the six doubles through `ostream <<` (6 significant digits, %g style) and
 * :250-251 reads them back with atof into float */
void orc_normalize_publish(const double avg[3], const double std_dev[3], float avg_f[3], float std_f[3]) {
    char buf[64];
    for (int c = 0; c < 3; c++) {
        snprintf(buf, sizeof buf, "%g", avg[c]);
        avg_f[c] = (float)atof(buf);
        snprintf(buf, sizeof buf, "%g", std_dev[c]);
        std_f[c] = (float)atof(buf);
    }
}

/* ------------------------------------------------------------------------------------------
 * variational_aux_mt.cpp:758-887 -- optimizeOcc.  TEST INFRASTRUCTURE, parity unpinned (see header).
 * ---------------------------------------------------------------------------------------- */
#define ORC_DT_SCALE_GRAPHC 0.01f   /* variational_aux_mt.h:24 */

void orc_occlusion_costs(float *d0, float *d1, const float *masks, const float *succ, const float *toref, int ref,
                         const float *rho, const float *omega, float hd, float hg, float penalty,
                         const orc_penalty *color, const orc_penalty *grad, int w, int h, int stride) {
    const size_t plane = (size_t)stride * h, cimg = 3 * plane, stack = 8 * cimg;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * stride + x;
            float e[2] = {0.0f, 0.0f}, n[2] = {0.0f, 0.0f};
            for (int s = 0; s < 2 * ref; s++) {
                const float m = masks[s * plane + o];
                const float *S = succ + s * stack, *R = toref + s * stack;
                const float *iz = S + 2 * cimg + o, *ixz = S + 6 * cimg + o, *iyz = S + 7 * cimg + o;
                const float *izr = R + 2 * cimg + o, *ixzr = R + 6 * cimg + o, *iyzr = R + 7 * cimg + o;
                const int idx = (ref - s - 1 > s - ref) ? ref - s - 1 : s - ref;                                  /* :814 */
                float term = rho[idx] * hd * m * orc_psi_apply_vec(color, iz[0] * iz[0] + iz[plane] * iz[plane] + iz[2 * plane] * iz[2 * plane]);   /* :817 */
                term += rho[idx] * hg * m * orc_psi_apply_vec(grad, ixz[0] * ixz[0] + ixz[plane] * ixz[plane] + ixz[2 * plane] * ixz[2 * plane] +
                                                                        iyz[0] * iyz[0] + iyz[plane] * iyz[plane] + iyz[2 * plane] * iyz[2 * plane]);   /* :818-819 */
                term += omega[idx] * hd * m * orc_psi_apply_vec(color, izr[0] * izr[0] + izr[plane] * izr[plane] + izr[2 * plane] * izr[2 * plane]);   /* :822 */
                term += omega[idx] * hg * m * orc_psi_apply_vec(grad, ixzr[0] * ixzr[0] + ixzr[plane] * ixzr[plane] + ixzr[2 * plane] * ixzr[2 * plane] +
                                                                          iyzr[0] * iyzr[0] + iyzr[plane] * iyzr[plane] + iyzr[2 * plane] * iyzr[2 * plane]);   /* :823-827 */
                const int l = s >= ref ? 0 : 1;                                                                   /* :829-837 */
                e[l] += term;
                n[l] += m * (rho[idx] + rho[idx] + omega[idx] + omega[idx]);
            }
            for (int l = 0; l < 2; l++) {
                if (n[l] == 0) n[l] = 1;                                                                          /* :846-849 */
                const float c = ORC_DT_SCALE_GRAPHC * e[l] / n[l] + penalty * l;                                 /* :851 */
                (l ? d1 : d0)[o] = c;
            }
        }
}

double orc_grid_cut_energy(const float *occ, const float *d0, const float *d1, float alpha, int w, int h, int stride) {
    double E = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * stride + x;
            const int l = occ[o] > 0;
            E += l ? d1[o] : d0[o];
            if (x + 1 < w && (occ[o + 1] > 0) != l) E += alpha;
            if (y + 1 < h && (occ[o + stride] > 0) != l) E += alpha;
        }
    return E;
}

/* Dinic on the implicit grid graph.  Node p: source arc capacity max(D1-D0, 0) (cut when p takes label 1), sink arc
 * max(D0-D1, 0); four neighbour arcs of capacity alpha each way.  Label 1 = sink side = cannot be reached from the
 * source in the residual graph. */
typedef struct { int w, h; double *cs, *ct, *cn; int *level, *it; } cutg;   /* cn[4*p + d]: residual p -> neighbour d (0:+x 1:-x 2:+y 3:-y) */
static int cut_nb(const cutg *g, int p, int d) {
    const int x = p % g->w, y = p / g->w;
    switch (d) {
    case 0: return x + 1 < g->w ? p + 1 : -1;
    case 1: return x > 0 ? p - 1 : -1;
    case 2: return y + 1 < g->h ? p + g->w : -1;
    default: return y > 0 ? p - g->w : -1;
    }
}
static int cut_bfs(cutg *g, int *queue) {   /* levels from the source; returns 1 if the sink is reachable */
    const int N = g->w * g->h;
    int qh = 0, qt = 0, reach = 0;
    for (int p = 0; p < N; p++) {
        g->level[p] = -1;
        if (g->cs[p] > 0) { g->level[p] = 1; queue[qt++] = p; }
    }
    while (qh < qt) {
        const int p = queue[qh++];
        if (g->ct[p] > 0) reach = 1;
        for (int d = 0; d < 4; d++) {
            const int q = cut_nb(g, p, d);
            if (q >= 0 && g->cn[4 * p + d] > 0 && g->level[q] < 0) { g->level[q] = g->level[p] + 1; queue[qt++] = q; }
        }
    }
    return reach;
}
static double cut_dfs(cutg *g, int p, double f) {   /* iterative would be safer for huge grids; test sizes are small */
    if (g->ct[p] > 0) {
        const double a = f < g->ct[p] ? f : g->ct[p];
        g->ct[p] -= a;
        return a;
    }
    for (; g->it[p] < 4; g->it[p]++) {
        const int d = g->it[p], q = cut_nb(g, p, d);
        if (q < 0 || g->cn[4 * p + d] <= 0 || g->level[q] != g->level[p] + 1) continue;
        const double a = cut_dfs(g, q, f < g->cn[4 * p + d] ? f : g->cn[4 * p + d]);
        if (a > 0) {
            g->cn[4 * p + d] -= a;
            g->cn[4 * q + (d ^ 1)] += a;
            return a;
        }
    }
    return 0;
}
double orc_grid_cut(float *occ, const float *d0, const float *d1, float alpha, int w, int h, int stride) {
    const int N = w * h;
    cutg g = {w, h, calloc(N, sizeof(double)), calloc(N, sizeof(double)), calloc(4 * (size_t)N, sizeof(double)), malloc(N * sizeof(int)), malloc(N * sizeof(int))};
    int *queue = malloc(N * sizeof(int));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int p = y * w + x;
            const double u = (double)d1[(size_t)y * stride + x] - (double)d0[(size_t)y * stride + x];
            if (u > 0) g.cs[p] = u; else g.ct[p] = -u;
            for (int d = 0; d < 4; d++) g.cn[4 * p + d] = cut_nb(&g, p, d) >= 0 ? alpha : 0;
        }
    while (cut_bfs(&g, queue)) {
        memset(g.it, 0, N * sizeof(int));
        for (int p = 0; p < N; p++)
            while (g.cs[p] > 0 && g.level[p] == 1) {
                const double a = cut_dfs(&g, p, g.cs[p]);
                if (a <= 0) break;
                g.cs[p] -= a;
            }
    }
    cut_bfs(&g, queue);                                         /* level >= 0: reachable from the source = label 0 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) occ[(size_t)y * stride + x] = g.level[y * w + x] >= 0 ? -1.0f : 1.0f;   /* :876 */
    free(g.cs); free(g.ct); free(g.cn); free(g.level); free(g.it); free(queue);
    return orc_grid_cut_energy(occ, d0, d1, alpha, w, h, stride);
}

/* ------------------------------------------------------------------------------------------
 * variational_mt.cpp:169-493 -- one pyramid level
 * ---------------------------------------------------------------------------------------- */
/* Test hook for the discrete step: a minimum cut is not unique, so two exact solvers may return different labellings of equal energy and
 * the flows computed under them differ.  orc_force_labels hands orc_compute_one_level the labels to use after the discrete step of
 * alternation a = 1 .. n-1 (labels + a*h*stride; entry 0 unused); each alternation records the relative energy excess of the forced
 * labelling over this run's own exact minimum in gap[a].  NULL switches the hook off.  Single level, single thread. */
static const float *g_forced_labels = NULL;
static int g_forced_n = 0;
static double g_forced_gap[64];
void orc_force_labels(const float *labels, int n) {
    g_forced_labels = labels; g_forced_n = n < 64 ? n : 64;
    for (int i = 0; i < 64; i++) g_forced_gap[i] = 0;
}
double orc_forced_gap(int alter) { return alter >= 0 && alter < 64 ? g_forced_gap[alter] : 0; }

/* variational_mt.cpp:293-320: the occlusion / direction weighting of the warp masks, in place.
 * masks: 2*ref planes; occ: -1 / 0 / +1 per pixel; data_norm = sum_a (rho[a] + omega[a]) (:223-226) */
void orc_mask_weight(float *masks, const float *occ, int ref, float data_norm, int one_direction, int w, int h, int stride) {
    const size_t plane = (size_t)stride * h;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * stride + x;
            float factor = (occ[o] == 0.0f) ? 1.0f : 0.0f;                      /* :295 */
            factor = (1 + factor) * data_norm;                                  /* :297-300 */
            const float backward = ((occ[o] >= 0.0f) ? 1.0f : 0.0f) / factor;   /* :302 */
            const float forward = ((occ[o] <= 0.0f) ? 1.0f : 0.0f) / factor;    /* :303 */
            for (int s = one_direction ? ref : 0; s < 2 * ref; s++) {           /* :305-318 */
                float *m = masks + s * plane + o;
                if (s < ref) *m = 1.0f * backward * (*m);
                else         *m = 1.0f * forward * (*m);
            }
        }
}

/* the values of the reference's "inner it i avg change a,b" / "outer it i avg change a,b" lines (variational_mt.cpp:404-405, 431-432), recorded instead of
 * printed: entries of 4 floats (0 = inner / 1 = outer, iteration, a, b) into a caller's buffer (test infrastructure for the library's verbose lines) */
static float *g_change_log = 0;
static int g_change_cap = 0, g_change_n = 0;
void orc_set_change_log(float *buf, int cap_entries) { g_change_log = buf; g_change_cap = buf ? cap_entries : 0; g_change_n = 0; }
int orc_change_log_count(void) { return g_change_n; }
static void log_change(int kind, int it, float a, float b) {
    if (g_change_log && g_change_n < g_change_cap) { float *e = g_change_log + 4 * g_change_n++; e[0] = (float)kind; e[1] = (float)it; e[2] = a; e[3] = b; }
}

int orc_compute_one_level(const orc_params *p, float *wx, float *wy, float *const *frames,
                          const float *const chw[3], float *occ_out, int w, int h, int stride, float change[2]) {
    const int ref = p->S - 1;
    if (ref < 1 || ref > ORC_MAX_REF) return -2;
    const int nslots = 2 * ref;
    const size_t plane = (size_t)stride * h, cimg = 3 * plane, stack = 8 * cimg;
    const float gamma_over3 = p->gamma / 3.0f, delta_over3 = p->delta / 3.0f;   /* :548-549 */

    float *du = plane_alloc(plane), *dv = plane_alloc(plane), *old_du = plane_alloc(plane), *old_dv = plane_alloc(plane);
    float *sh = plane_alloc(plane), *sv = plane_alloc(plane), *uu = plane_alloc(plane), *vv = plane_alloc(plane);
    float *a11 = plane_alloc(plane), *a12 = plane_alloc(plane), *a22 = plane_alloc(plane), *b1 = plane_alloc(plane), *b2 = plane_alloc(plane);
    float *occ = plane_alloc(plane), *dpsis = plane_alloc(plane);
    float *w_s = plane_alloc(cimg), *w_sp1 = plane_alloc(cimg);
    float *mask = plane_alloc(nslots * plane);
    float *succ = plane_alloc(nslots * stack), *toref = plane_alloc(nslots * stack);

    if (p->one_direction || p->occlusion_reasoning)                              /* :219-220 */
        for (size_t i = 0; i < plane; i++) occ[i] = -1.0f;

    float data_norm = 0;                                                        /* :223-226 */
    for (int s = 0; s < ref; s++) data_norm += p->rho[s] + p->omega[s];

    orc_dpsis_weight(dpsis, frames[ref], w, h, stride, 5.0f, p->norm_avg, p->norm_std, p->hbit);   /* :257 */

    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uu[(size_t)y * stride + x] = wx[(size_t)y * stride + x];            /* :260-261 */
            vv[(size_t)y * stride + x] = wy[(size_t)y * stride + x];
        }

    float chg_x = 0, chg_y = 0;
    int rc = 0;
    for (int alter = 0; alter < p->niter_alter && !rc; alter++) {
        int need_derivs = 1;                                                    /* :266 */
        int occ_pending = alter > 0 && p->occlusion_reasoning && !p->one_direction;   /* :269-272, after get_derivatives (:266) */
        for (int outer = 0; outer < p->niter_outer; outer++) {
            if (outer > 0) need_derivs = 1;                                     /* :289-290 */
            if (need_derivs) {
                /* get_derivatives, variational_mt.cpp:87-166 */
                for (int s = p->one_direction ? ref : 0; s < nslots; s++) {
                    if (s < ref) {
                        orc_image_warp(w_s, mask + s * plane, frames[s], wx, wy, w, h, stride, s - ref);
                        orc_image_warp(w_sp1, NULL, frames[s + 1], wx, wy, w, h, stride, s - ref + 1);
                    } else {
                        orc_image_warp(w_s, NULL, frames[s], wx, wy, w, h, stride, s - ref);
                        orc_image_warp(w_sp1, mask + s * plane, frames[s + 1], wx, wy, w, h, stride, s - ref + 1);
                    }
                    orc_derivative_stack(succ + s * stack, w_s, w_sp1, w, h, stride);
                    if (s < ref) orc_derivative_stack(toref + s * stack, w_s, frames[ref], w, h, stride);      /* :139-141 */
                    else         orc_derivative_stack(toref + s * stack, frames[ref], w_sp1, w, h, stride);    /* :143-144 */
                }
                need_derivs = 0;
                if (occ_pending) {                                           /* optimizeOcc on the raw warp masks */
                    float *d0 = plane_alloc(plane), *d1 = plane_alloc(plane);
                    orc_occlusion_costs(d0, d1, mask, succ, toref, ref, p->rho, p->omega, delta_over3, gamma_over3, p->occlusion_penalty,
                                        &p->robust_color, &p->robust_grad, w, h, stride);
                    const double e_min = orc_grid_cut(occ, d0, d1, p->occlusion_alpha, w, h, stride);
                    if (g_forced_labels && alter < g_forced_n) {
                        /* test hook (orc_force_labels): the labelling comes from outside; how far it is from this run's own optimum
                         * is recorded, and the continuous optimisation goes on under it */
                        const float *fl = g_forced_labels + (size_t)alter * plane;
                        g_forced_gap[alter] = (orc_grid_cut_energy(fl, d0, d1, p->occlusion_alpha, w, h, stride) - e_min) / (fabs(e_min) > 1 ? fabs(e_min) : 1);
                        memcpy(occ, fl, plane * sizeof(float));
                    }
                    free(d0); free(d1);
                    occ_pending = 0;
                }
            }
            orc_mask_weight(mask, occ, ref, data_norm, p->one_direction, w, h, stride);   /* :293-320 */
            memset(du, 0, plane * sizeof(float));
            memset(dv, 0, plane * sizeof(float));

            for (int inner = 0; inner < p->niter_inner; inner++) {
                memcpy(old_du, du, plane * sizeof(float));
                memcpy(old_dv, dv, plane * sizeof(float));
                orc_smoothness(p->smoothing, sh, sv, uu, vv, dpsis, w, h, stride, p->alpha, &p->robust_reg);   /* :333 */
                memset(a11, 0, plane * sizeof(float)); memset(a12, 0, plane * sizeof(float)); memset(a22, 0, plane * sizeof(float));
                memset(b1, 0, plane * sizeof(float)); memset(b2, 0, plane * sizeof(float));
                for (int s = 0; s < ref && !rc; s++) {                           /* :343-361 */
                    if (!p->one_direction) {
                        if (p->rho[ref - 1 - s] > 0)
                            orc_add_data_and_match(a11, a12, a22, b1, b2, mask + s * plane, du, dv, succ + s * stack, chw, w, h, stride,
                                                   p->rho[ref - 1 - s] * delta_over3, p->rho[ref - 1 - s] * gamma_over3, (float)(s - ref),
                                                   p->dataterm_norm, &p->robust_color, &p->robust_grad);
                        if (p->omega[ref - 1 - s] > 0)
                            rc = orc_add_data_and_match_ref(a11, a12, a22, b1, b2, mask + s * plane, du, dv, toref + s * stack, chw, w, h, stride,
                                                            p->omega[ref - 1 - s] * delta_over3, p->omega[ref - 1 - s] * gamma_over3, (float)(s - ref),
                                                            p->dataterm_norm, &p->robust_color, &p->robust_grad);
                    }
                    if (p->rho[s] > 0)
                        orc_add_data_and_match(a11, a12, a22, b1, b2, mask + (ref + s) * plane, du, dv, succ + (ref + s) * stack, chw, w, h, stride,
                                               p->rho[s] * delta_over3, p->rho[s] * gamma_over3, (float)s,
                                               p->dataterm_norm, &p->robust_color, &p->robust_grad);
                    if (p->omega[s] > 0 && !rc)
                        rc = orc_add_data_and_match_ref(a11, a12, a22, b1, b2, mask + (ref + s) * plane, du, dv, toref + (ref + s) * stack, chw, w, h, stride,
                                                        p->omega[s] * delta_over3, p->omega[s] * gamma_over3, (float)(s + 1),
                                                        p->dataterm_norm, &p->robust_color, &p->robust_grad);
                }
                if (rc) break;
                orc_sub_laplacian(b1, uu, sh, sv, w, h, stride);                /* :364-365 */
                orc_sub_laplacian(b2, vv, sh, sv, w, h, stride);
                if (p->sor_order == 1) orc_sor_red_black(du, dv, a11, a12, a22, b1, b2, sh, sv, w, h, stride, p->niter_solver, p->sor_omega);   /* labelled mode, not the reference */
                else orc_sor_coupled(du, dv, a11, a12, a22, b1, b2, sh, sv, w, h, stride, p->niter_solver, p->sor_omega);   /* :368 */

                /* :371-402: padding lanes of du,dv are zeroed; the L1 change norms are fp32 running sums
                 * in raster order, four lanes of a block added left to right first.  The oracle sums the
                 * valid pixels in that order (padding lanes contribute +0 there). */
                float avg_du = 0, avg_dv = 0;
                for (int y = 0; y < h; y++) {
                    for (int xb = 0; xb < stride; xb += 4) {
                        float d[4], e[4];
                        for (int q = 0; q < 4; q++) {
                            const int x = xb + q;
                            const size_t o = (size_t)y * stride + x;
                            if (x < w) {
                                d[q] = fabsf(old_du[o] - du[o]);
                                e[q] = fabsf(old_dv[o] - dv[o]);
                                uu[o] = wx[o] + du[o];                          /* :396-397 */
                                vv[o] = wy[o] + dv[o];
                            } else { d[q] = 0; e[q] = 0; du[o] = 0; dv[o] = 0; }
                        }
                        avg_du += d[0] + d[1] + d[2] + d[3];
                        avg_dv += e[0] + e[1] + e[2] + e[3];
                    }
                }
                avg_du /= (h * w);
                avg_dv /= (h * w);
                log_change(0, inner, avg_du, avg_dv);                           /* :404-405 */
                if ((avg_du < avg_dv ? avg_dv : avg_du) < p->thres_inner) break;   /* :407 std::max(a, b) = (a < b) ? b : a: a NaN first argument stays */
            }
            if (rc) break;
            float avg_wx = 0, avg_wy = 0;                                       /* :412-425 */
            for (int y = 0; y < h; y++)
                for (int xb = 0; xb < stride; xb += 4) {
                    float d[4], e[4];
                    for (int q = 0; q < 4; q++) {
                        const int x = xb + q;
                        const size_t o = (size_t)y * stride + x;
                        if (x < w) { d[q] = fabsf(uu[o] - wx[o]); e[q] = fabsf(vv[o] - wy[o]); }
                        else { d[q] = 0; e[q] = 0; }
                    }
                    avg_wx += d[0] + d[1] + d[2] + d[3];
                    avg_wy += e[0] + e[1] + e[2] + e[3];
                }
            avg_wx /= (h * w);
            avg_wy /= (h * w);
            log_change(1, outer, avg_wx, avg_wy);                               /* :431-432 */
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    wx[(size_t)y * stride + x] = uu[(size_t)y * stride + x];    /* :428-429 */
                    wy[(size_t)y * stride + x] = vv[(size_t)y * stride + x];
                }
            chg_x = avg_wx; chg_y = avg_wy;
            if ((avg_wx < avg_wy ? avg_wy : avg_wx) < p->thres_outer) break;    /* :436 std::max, as above */
        }
    }
    if (change) { change[0] = chg_x; change[1] = chg_y; }
    if (occ_out) memcpy(occ_out, occ, plane * sizeof(float));
    free(du); free(dv); free(old_du); free(old_dv); free(sh); free(sv); free(uu); free(vv);
    free(a11); free(a12); free(a22); free(b1); free(b2); free(occ); free(dpsis);
    free(w_s); free(w_sp1); free(mask); free(succ); free(toref);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * OpenCV-defined arithmetic of the pyramid (variational_mt.cpp:607,611,672-673,711-712).
 * OpenCV is not vendored and absent: restated from its documented semantics.  UNPINNED.
 * ---------------------------------------------------------------------------------------- */

/* cv::GaussianBlur(src, dst, Size(0,0), sigma, sigma, BORDER_REPLICATE) on CV_32F:
 * ksize = cvRound(sigma*4*2+1)|1; kernel = getGaussianKernel(ksize, sigma, CV_32F): fp32 taps
 * exp(-x^2/(2 sigma^2)) (computed in double, stored float) normalised by the double reciprocal of
 * their float sum; separable, rows then columns, symmetric form k0*c + sum_j kj*(l_j + r_j) in fp32 */
void orc_gaussian_blur_cv(float *dst, const float *src, int w, int h, int stride, float sigma) {
    int ksize = ((int)lrint((double)sigma * 4 * 2 + 1)) | 1;
    const int r = ksize / 2;
    float *k = (float *)malloc(sizeof(float) * ksize);
    const double sigmaX = sigma, scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < ksize; i++) {
        const double x = i - (ksize - 1) * 0.5;
        k[i] = (float)exp(scale2X * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < ksize; i++) k[i] = (float)(k[i] * sum);
    float *tmp = plane_alloc((size_t)stride * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *s = src + (size_t)y * stride;
            float acc = k[r] * s[x];
            for (int j = 1; j <= r; j++)
                acc += k[r + j] * (s[clampi(x - j, 0, w - 1)] + s[clampi(x + j, 0, w - 1)]);
            tmp[(size_t)y * stride + x] = acc;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = k[r] * tmp[(size_t)y * stride + x];
            for (int j = 1; j <= r; j++)
                acc += k[r + j] * (tmp[(size_t)clampi(y - j, 0, h - 1) * stride + x] + tmp[(size_t)clampi(y + j, 0, h - 1) * stride + x]);
            dst[(size_t)y * stride + x] = acc;
        }
    free(tmp); free(k);
}

/* cv::resize(..., INTER_LINEAR) on CV_32F: fx = (float)((dx+0.5)*(sw/dw) - 0.5); sx = floor(fx);
 * fx -= sx; sx<0 -> (0, fx=0); sx>=sw-1 -> (sw-1, fx=0); horizontal pass on the two source rows
 * S[sx]*(1-fx) + S[sx+1]*fx, then vertical R0*(1-fy) + R1*fy, all fp32 */
static void resize_linear_scaled(float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride, double scale_x, double scale_y);
void orc_resize_linear_cv(float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride) {
    /* explicit dsize: inv_scale = dsize / ssize, scale = 1 / inv_scale (imgproc resize.cpp) */
    resize_linear_scaled(dst, dw, dh, dstride, src, sw, sh, sstride, (double)sw / dw, (double)sh / dh);
}
/* cv::resize(src, dst, Size(0,0), fx, fy, INTER_LINEAR) as the driver's input rescaling calls it (slow_flow.cpp:552): dsize =
 * (cvRound(sw*fx), cvRound(sh*fy)) is the caller's business; the source coordinate uses scale = 1/fx, NOT ssize/dsize (they differ when
 * sw*fx is not an integer).  OpenCV is not vendored: PARITY UNPINNED like the rest of the pyramid arithmetic. */
void orc_resize_linear_fx(float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride, double fx, double fy) {
    resize_linear_scaled(dst, dw, dh, dstride, src, sw, sh, sstride, 1.0 / fx, 1.0 / fy);
}
static void resize_linear_scaled(float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride, double scale_x, double scale_y) {
    int *xofs = (int *)malloc(sizeof(int) * dw);
    float *xa = (float *)malloc(sizeof(float) * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx; xa[dx] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        const float *r0 = src + (size_t)sy * sstride;
        const float *r1 = src + (size_t)(sy + 1 < sh ? sy + 1 : sy) * sstride;
        const float b0 = 1.f - fy, b1 = fy;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            const float a0 = 1.f - xa[dx], a1 = xa[dx];
            const float h0 = r0[sx] * a0 + r0[sx1] * a1;
            const float h1 = r1[sx] * a0 + r1[sx1] * a1;
            dst[(size_t)dy * dstride + dx] = h0 * b0 + h1 * b1;
        }
    }
    free(xofs); free(xa);
}

/* image.c:310-322 order of the presmoothing filter (only its order is used, for the size break) */
static int gaussian_filter_order(float sigma) {
    int order = (int)floor(3 * sigma) + 1;
    if (order == 0) order = 1;
    return order;
}

/* variational_mt.cpp:583-652: sizes floor((float)w*p) with the product rounded to fp32; levels stop
 * growing when the next one would be <= order+1 in either dimension.  returns the level count */
int orc_pyramid_sizes(int w, int h, int layers, float p_scale, int *ws, int *hs) {
    const float sigma = 1 / sqrtf(2 * p_scale);                                  /* :578 (C++: float sqrt overload) */
    const int order = gaussian_filter_order(sigma);
    int L = layers;
    for (int l = 0; l < layers; l++) {
        if (l == 0) { ws[0] = w; hs[0] = h; }
        else {
            ws[l] = (int)(float)floor(ws[l - 1] * p_scale);                      /* :609-611 */
            hs[l] = (int)(float)floor(hs[l - 1] * p_scale);
        }
        if (floor(ws[l] * p_scale) <= order + 1 || floor(hs[l] * p_scale) <= order + 1) {   /* :647 */
            L = l;
            break;
        }
    }
    return L;
}

static int stride_of(int w) { return ((w + 3) / 4) * 4; }                      /* image.c:25 */

/* image.c:310-348 + :351-361 + generic convolve_horiz/vert :537-644 -- the optional Gaussian
 * presmoothing of level 0 (cfg sigma > 0, variational_mt.cpp:590-597) */
void orc_gaussian_presmooth(float *dst, const float *src, int w, int h, int stride, float sigma) {
    const int order = gaussian_filter_order(sigma);
    const int n = 2 * order + 1;
    float *data = (float *)malloc(sizeof(float) * n), *coeffs = (float *)malloc(sizeof(float) * n), *accu = (float *)malloc(sizeof(float) * n);
    const float alpha = 1.0f / (2.0f * sigma * sigma);
    float sum = 0.0f;
    for (int i = -order; i <= order; i++) { data[i + order] = (float)exp(-i * i * alpha); sum += data[i + order]; }
    for (int i = 0; i < n; i++) data[i] /= sum;
    const float *half = data + order;
    for (int i = 0; i <= order; i++) coeffs[order - i] = coeffs[order + i] = half[i];
    float acc = 0.0f;
    for (int i = 0; i <= order; i++) { acc += coeffs[i]; accu[2 * order - i] = accu[i] = acc; }
    const float *coeff = coeffs + order, *coeff_accu = accu + order;
    const int i0 = -order, i1 = order;
    float *tmp = plane_alloc((size_t)stride * h);
    if (order <= 2) {                                           /* convolve_horiz / _vert dispatch on the order: image.c:529-535, 580-586 */
        conv_horiz_fast(tmp, src, w, h, stride, order, coeffs);
        conv_vert_fast(dst, tmp, w, h, stride, order, coeffs);
        free(tmp); free(data); free(coeffs); free(accu);
        return;
    }
    /* horizontal, image.c:545-578 */
    for (int j = 0; j < h; j++) {
        const float *al = src + (size_t)j * stride;
        float *o = tmp + (size_t)j * stride;
        const float *f0 = coeff + i0;
        int i;
        for (i = 0; i < -i0; i++) {
            float s = coeff_accu[-i - 1] * al[0];
            for (int ii = i1 + i; ii >= 0; ii--) s += coeff[ii - i] * al[ii];
            *o++ = s;
        }
        for (; i < w - i1; i++) {
            float s = 0;
            for (int ii = i1 - i0; ii >= 0; ii--) s += f0[ii] * al[ii];
            al++;
            *o++ = s;
        }
        for (; i < w; i++) {
            float s = coeff_accu[w - i] * al[w - i0 - 1 - i];
            for (int ii = w - i0 - 1 - i; ii >= 0; ii--) s += f0[ii] * al[ii];
            al++;
            *o++ = s;
        }
    }
    /* vertical, image.c:597-644 */
    {
        const float *in = tmp;
        const float *alast = in + (size_t)stride * (h - 1);
        const float *f0 = coeff + i0;
        int i;
        for (i = 0; i < -i0; i++) {
            const float fa = coeff_accu[-i - 1];
            const float *al = in + (size_t)i * stride;
            for (int j = 0; j < w; j++) {
                float s = fa * in[j];
                for (int ii = -i; ii <= i1; ii++) s += coeff[ii] * al[j + ii * stride];
                dst[(size_t)i * stride + j] = s;
            }
        }
        for (; i < h - i1; i++) {
            const float *al = in + (size_t)(i + i0) * stride;
            for (int j = 0; j < w; j++) {
                float s = 0;
                const float *al2 = al + j;
                for (int ii = 0; ii <= i1 - i0; ii++) { s += f0[ii] * al2[0]; al2 += stride; }
                dst[(size_t)i * stride + j] = s;
            }
        }
        for (; i < h; i++) {
            const float fa = coeff_accu[h - i];
            const float *al = in + (size_t)i * stride;
            for (int j = 0; j < w; j++) {
                float s = fa * alast[j];
                for (int ii = i0; ii <= h - 1 - i; ii++) s += coeff[ii] * al[j + ii * stride];
                dst[(size_t)i * stride + j] = s;
            }
        }
    }
    free(tmp); free(data); free(coeffs); free(accu);
}

/* ------------------------------------------------------------------------------------------
 * variational_mt.cpp:526-784 -- coarse-to-fine driver
 * ---------------------------------------------------------------------------------------- */
int orc_variational(const orc_params *p, float *wx, float *wy, float *const *frames, const float *const chw_in[3],
                    int w, int h, int stride, float change[2]) {
    const int ref = p->S - 1, F = 2 * ref + 1;
    if (ref < 1 || ref > ORC_MAX_REF || p->layers < 1 || p->layers > 64) return -2;
    int ws[64], hs[64];
    const int L = orc_pyramid_sizes(w, h, p->layers, p->p_scale, ws, hs);
    /* when the size break fires at level l the reference sets L = l: level l itself was built but is
     * not used (:647-651); the oracle does not build it */
    const int nbuilt = L;
    const float sigma = 1 / sqrtf(2 * p->p_scale);

    float *ones = NULL;
    const float *chw[3];
    if (chw_in) { chw[0] = chw_in[0]; chw[1] = chw_in[1]; chw[2] = chw_in[2]; }
    else {
        ones = plane_alloc((size_t)stride * h);
        for (size_t i = 0; i < (size_t)stride * h; i++) ones[i] = 1.0f;
        chw[0] = chw[1] = chw[2] = ones;
    }

    float ***pyr = (float ***)calloc(nbuilt, sizeof(float **));
    for (int l = 0; l < nbuilt; l++) {
        pyr[l] = (float **)calloc(F, sizeof(float *));
        const int lw = ws[l], lh = hs[l], ls = stride_of(lw);
        const size_t lplane = (size_t)(l == 0 ? stride : ls) * lh;
        for (int s = 0; s < F; s++) {
            pyr[l][s] = plane_alloc(3 * lplane);
            if (l == 0) {
                if (p->presmooth_sigma > 0) {
                    for (int k = 0; k < 3; k++) orc_gaussian_presmooth(pyr[0][s] + k * lplane, frames[s] + k * lplane, w, h, stride, p->presmooth_sigma);
                } else memcpy(pyr[0][s], frames[s], 3 * lplane * sizeof(float));       /* :599 */
            } else {
                const int pw = ws[l - 1], ph = hs[l - 1], ps = l - 1 == 0 ? stride : stride_of(pw);
                const size_t pplane = (size_t)ps * ph;
                float *blur = plane_alloc(pplane);
                for (int k = 0; k < 3; k++) {
                    orc_gaussian_blur_cv(blur, pyr[l - 1][s] + k * pplane, pw, ph, ps, sigma);   /* :607 */
                    orc_resize_linear_cv(pyr[l][s] + k * lplane, lw, lh, ls, blur, pw, ph, ps);  /* :611 */
                }
                free(blur);
            }
        }
    }

    float *wxl = wx, *wyl = wy;
    int cw = w, ch = h, cs = stride;
    if (L > 1) {                                                                /* :662-681 */
        const int lw = ws[L - 1], lh = hs[L - 1], ls = stride_of(lw);
        const float fx = (1.0f * lw) / w, fy = (1.0f * lh) / h;
        wxl = plane_alloc((size_t)ls * lh); wyl = plane_alloc((size_t)ls * lh);
        orc_resize_linear_cv(wxl, lw, lh, ls, wx, w, h, stride);
        orc_resize_linear_cv(wyl, lw, lh, ls, wy, w, h, stride);
        for (int y = 0; y < lh; y++) for (int x = 0; x < lw; x++) { wxl[(size_t)y * ls + x] *= fx; wyl[(size_t)y * ls + x] *= fy; }
        cw = lw; ch = lh; cs = ls;
    }
    int rc = 0;
    float chg[2] = {0, 0};
    for (int l = L - 1; l >= 0 && !rc; l--) {
        const int lw = ws[l], lh = hs[l], ls = l == 0 ? stride : stride_of(lw);
        if (l < L - 1) {                                                        /* :689-723 */
            float *tx = l > 0 ? plane_alloc((size_t)ls * lh) : wx;
            float *ty = l > 0 ? plane_alloc((size_t)ls * lh) : wy;
            const float fx = (1.0f * lw) / cw, fy = (1.0f * lh) / ch;
            orc_resize_linear_cv(tx, lw, lh, ls, wxl, cw, ch, cs);
            orc_resize_linear_cv(ty, lw, lh, ls, wyl, cw, ch, cs);
            for (int y = 0; y < lh; y++) for (int x = 0; x < lw; x++) { tx[(size_t)y * ls + x] *= fx; ty[(size_t)y * ls + x] *= fy; }
            free(wxl); free(wyl);
            wxl = tx; wyl = ty; cw = lw; ch = lh; cs = ls;
        }
        rc = orc_compute_one_level(p, wxl, wyl, pyr[l], chw, NULL, lw, lh, ls, chg);   /* :761 */
    }
    if (L > 1 && rc && wxl != wx) { free(wxl); free(wyl); }
    if (L == 0) rc = -3;
    if (change) { change[0] = chg[0]; change[1] = chg[1]; }
    for (int l = 0; l < nbuilt; l++) { for (int s = 0; s < F; s++) free(pyr[l][s]); free(pyr[l]); }
    free(pyr); free(ones);
    return rc;
}


/* ------------------------------------------------------------------------------------------
 * variational.c:19-143 + variational_aux.c -- the original two-frame refinement
 * ---------------------------------------------------------------------------------------- */
#define EPS2F (0.001f * 0.001f)          /* epsilon_color / _grad / _smooth, variational_aux.c:11-13 */

static void smoothness_2f(float *sh, float *sv, const float *uu, const float *vv, const float *dps, int w, int h, int s, float half_alpha) {
    const size_t plane = (size_t)s * h;
    float *ux2 = plane_alloc(plane), *uy2 = plane_alloc(plane), *vx2 = plane_alloc(plane), *vy2 = plane_alloc(plane);
    orc_convolve_horiz(ux2, uu, w, h, s, 1); orc_convolve_horiz(vx2, vv, w, h, s, 1);          /* variational_aux.c:110-113 */
    orc_convolve_vert(uy2, uu, w, h, s, 1);  orc_convolve_vert(vy2, vv, w, h, s, 1);
    memset(sh, 0, plane * sizeof(float)); memset(sv, 0, plane * sizeof(float));
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w - 1; i++) {                                                      /* :115-128 */
            const size_t o = (size_t)j * s + i;
            const float ux1 = uu[o + 1] - uu[o], vx1 = vv[o + 1] - vv[o];
            float tmp = 0.5f * (uy2[o] + uy2[o + 1]);
            const float uxsq = ux1 * ux1 + tmp * tmp;
            tmp = 0.5f * (vy2[o] + vy2[o + 1]);
            const float vxsq = vx1 * vx1 + tmp * tmp;
            tmp = uxsq + vxsq;
            sh[o] = (float)((dps[o] + dps[o + 1]) * half_alpha / sqrt(tmp + EPS2F));            /* double sqrt and division, :126 */
        }
    for (int j = 0; j < h - 1; j++)
        for (int i = 0; i < w; i++) {                                                          /* :130-146 */
            const size_t o = (size_t)j * s + i;
            const float uy1 = uu[o + s] - uu[o], vy1 = vv[o + s] - vv[o];
            float tmp = 0.5f * (ux2[o] + ux2[o + s]);
            const float uysq = uy1 * uy1 + tmp * tmp;
            tmp = 0.5f * (vx2[o] + vx2[o + s]);
            const float vysq = vy1 * vy1 + tmp * tmp;
            tmp = uysq + vysq;
            sv[o] = (float)((dps[o] + dps[o + s]) * half_alpha / sqrt(tmp + EPS2F));
        }
    free(ux2); free(uy2); free(vx2); free(vy2);
}

static void data_2f(float *a11, float *a12, float *a22, float *b1, float *b2, const float *mask, const float *du, const float *dv, const float *D,
                    int w, int h, int s, float hd, float hg) {
    const size_t plane = (size_t)s * h, cimg = 3 * plane;
    const float dn = 0.1f * 0.1f;                                                               /* datanorm, :10 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * s + x;
            float ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
            for (int k = 0; k < 3; k++) {
                ix[k] = D[D_IX * cimg + k * plane + o]; iy[k] = D[D_IY * cimg + k * plane + o]; iz[k] = D[D_IZ * cimg + k * plane + o];
                ixx[k] = D[D_IXX * cimg + k * plane + o]; ixy[k] = D[D_IXY * cimg + k * plane + o]; iyy[k] = D[D_IYY * cimg + k * plane + o];
                ixz[k] = D[D_IXZ * cimg + k * plane + o]; iyz[k] = D[D_IYZ * cimg + k * plane + o];
            }
            const float u = du[o], v = dv[o], m = mask[o];
            float A11 = 0, A12 = 0, A22 = 0, B1 = 0, B2 = 0;
            if (hd) {                                                                           /* :245-269 */
                float t[3], n[3];
                for (int k = 0; k < 3; k++) { t[k] = iz[k] + ix[k] * u + iy[k] * v; n[k] = ix[k] * ix[k] + iy[k] * iy[k] + dn; }
                const float q = m * hd / sqrtf(t[0] * t[0] / n[0] + t[1] * t[1] / n[1] + t[2] * t[2] / n[2] + EPS2F);
                for (int k = 0; k < 3; k++) {
                    const float tk = q / n[k];
                    A11 += tk * ix[k] * ix[k]; A12 += tk * ix[k] * iy[k]; A22 += tk * iy[k] * iy[k];
                    B1 -= tk * iz[k] * ix[k];  B2 -= tk * iz[k] * iy[k];
                }
            }
            float t[6], n[6];                                                                   /* :271-300 */
            for (int k = 0; k < 3; k++) {
                n[2 * k] = ixx[k] * ixx[k] + ixy[k] * ixy[k] + dn;
                n[2 * k + 1] = iyy[k] * iyy[k] + ixy[k] * ixy[k] + dn;
                t[2 * k] = ixz[k] + ixx[k] * u + ixy[k] * v;
                t[2 * k + 1] = iyz[k] + ixy[k] * u + iyy[k] * v;
            }
            const float q = m * hg / sqrtf(t[0] * t[0] / n[0] + t[1] * t[1] / n[1] + t[2] * t[2] / n[2] + t[3] * t[3] / n[3] + t[4] * t[4] / n[4] +
                                           t[5] * t[5] / n[5] + EPS2F);
            for (int k = 0; k < 3; k++) {
                const float ta = q / n[2 * k], tb = q / n[2 * k + 1];
                A11 += ta * ixx[k] * ixx[k] + tb * ixy[k] * ixy[k];
                A12 += ta * ixx[k] * ixy[k] + tb * ixy[k] * iyy[k];
                A22 += tb * iyy[k] * iyy[k] + ta * ixy[k] * ixy[k];
                B1 -= ta * ixx[k] * ixz[k] + tb * ixy[k] * iyz[k];
                B2 -= tb * iyy[k] * iyz[k] + ta * ixy[k] * ixz[k];
            }
            a11[o] = A11; a12[o] = A12; a22[o] = A22; b1[o] = B1; b2[o] = B2;
        }
}

void orc_variational_2frame(float *wx, float *wy, const float *im1, const float *im2, const orc_params_2f *p, int w, int h, int stride) {
    const size_t plane = (size_t)stride * h, cimg = 3 * plane;
    const float half_alpha = 0.5f * p->alpha, hg = p->gamma * 0.5f / 3.0f, hd = p->delta * 0.5f / 3.0f;     /* variational.c:113-115 */
    float *du = plane_alloc(plane), *dv = plane_alloc(plane), *mask = plane_alloc(plane), *sh = plane_alloc(plane), *sv = plane_alloc(plane);
    float *uu = plane_alloc(plane), *vv = plane_alloc(plane), *a11 = plane_alloc(plane), *a12 = plane_alloc(plane), *a22 = plane_alloc(plane);
    float *b1 = plane_alloc(plane), *b2 = plane_alloc(plane), *dps = plane_alloc(plane), *w_im2 = plane_alloc(cimg), *D = plane_alloc(8 * cimg);
    const float zero3[3] = {0, 0, 0}, one3[3] = {1, 1, 1};
    orc_dpsis_weight(dps, im1, w, h, stride, 5.0f, zero3, one3, 0);                                             /* :35 */
    for (int outer = 0; outer < p->niter_outer; outer++) {
        orc_image_warp(w_im2, mask, im2, wx, wy, w, h, stride, 1);                                              /* :41 */
        orc_derivative_stack(D, w_im2, im1, w, h, stride);                                                      /* :43: mean = (im2+im1)/2, dt = im2-im1 */
        memset(du, 0, plane * sizeof(float)); memset(dv, 0, plane * sizeof(float));
        memcpy(uu, wx, plane * sizeof(float)); memcpy(vv, wy, plane * sizeof(float));
        for (int inner = 0; inner < p->niter_inner; inner++) {
            smoothness_2f(sh, sv, uu, vv, dps, w, h, stride, half_alpha);                                       /* :54 */
            data_2f(a11, a12, a22, b1, b2, mask, du, dv, D, w, h, stride, hd, hg);                              /* :55 */
            orc_sub_laplacian(b1, wx, sh, sv, w, h, stride);                                                    /* :56-57 (wx, not uu) */
            orc_sub_laplacian(b2, wy, sh, sv, w, h, stride);
            orc_sor_coupled(du, dv, a11, a12, a22, b1, b2, sh, sv, w, h, stride, p->niter_solver, p->sor_omega);   /* :59 */
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const size_t o = (size_t)y * stride + x;
                    uu[o] = wx[o] + du[o]; vv[o] = wy[o] + dv[o];                                               /* :64-65 */
                }
        }
        memcpy(wx, uu, plane * sizeof(float)); memcpy(wy, vv, plane * sizeof(float));                           /* :70-71 */
    }
    free(du); free(dv); free(mask); free(sh); free(sv); free(uu); free(vv); free(a11); free(a12); free(a22); free(b1); free(b2); free(dps); free(w_im2); free(D);
}
