// kernels.hip -- pointwise / stencil kernels of the variational level (gfx950, wave64).
//
// Every kernel is HBM-bound fp32 planar work: one thread per pixel, x along the wave so that all
// plane accesses are coalesced 256-B rows; stencil neighbours come through L1/L2 (halo <= 2).
// Arithmetic is strict IEEE fp32 in the reference's association (no contraction, IEEE div/sqrt).
// Each __global__ cites the reference operator it implements.
#include "sfa_device.h"

#pragma clang fp contract(off)

namespace sfa {

// ---------------------------------------------------------------------------------------------------
// psi'(x^2): penalty_functions headers.  scalar overloads (double inside) and v4sf overloads (fp32).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float psi_vec(const PenaltyDev &p, float xsq) {
    const float e2 = p.eps * p.eps;
    switch (p.id) {
    case 0: return 1.0f;
    case 2: return __fdiv_rn(1.0f, 2.0f * e2 + xsq);                                   // lorentzian.h:40-42
    case 3: {                                                                           // trunc_modified_l1_norm.h:47-56
        float out = __fdiv_rn(1.0f, 2.0f * sqrt_rn(xsq + e2));
        if (sqrt_rn(xsq) > p.trunc) out = 0.0f;
        return out;
    }
    case 4: {                                                                           // geman_mcclure.h:34-38
        float t = e2 + xsq;
        t = t * t;
        return __fdiv_rn(e2 + 2.0f * xsq, t);
    }
    default: return __fdiv_rn(1.0f, 2.0f * sqrt_rn(xsq + e2));                       // modified_l1_norm.h:32-34
    }
}
__device__ __forceinline__ float psi_scalar(const PenaltyDev &p, float xsq) {
    const float e2f = p.eps * p.eps;
    const double e2d = (double)e2f;
    switch (p.id) {
    case 0: return 1.0f;
    case 2: return (float)(1.0 / (2.0 * e2d + (double)xsq));                            // lorentzian.h:36-38
    case 3:                                                                             // trunc_modified_l1_norm.h:40-45 (all float)
        if (sqrt_rn(xsq) > p.trunc) return 0.0f;
        return __fdiv_rn(1.0f, 2.0f * sqrt_rn(xsq + e2f));
    case 4: {                                                                           // geman_mcclure.h:28-32
        float t = (float)(e2d + (double)xsq);
        t = t * t;
        return (float)((e2d + (double)(2.0f * xsq)) / (double)t);
    }
    default: return (float)(1.0 / (2.0 * __dsqrt_rn((double)xsq + e2d)));               // modified_l1_norm.h:28-30
    }
}

// ---------------------------------------------------------------------------------------------------
// K1 image_warp (variational_aux_mt.cpp:722-756)
// ---------------------------------------------------------------------------------------------------
__global__ void k_warp(float *__restrict__ dst3, float *__restrict__ mask, const float *__restrict__ src3, const float *__restrict__ wx,
                       const float *__restrict__ wy, Geo g, int factor, long src_es) {
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    dst3 += b * g.es; wx += b * g.es; wy += b * g.es; src3 += b * src_es;
    const size_t o = (size_t)y * g.pitch + x;
    if (factor == 0) {                                               // :723-728
        for (int k = 0; k < 3; k++) dst3[k * g.pl + o] = src3[k * g.pl + o];
        return;
    }
    const float xx = x + factor * wx[o];                             // :735
    const float yy = y + factor * wy[o];
    const int xi = floor_to_int_x86(xx), yi = floor_to_int_x86(yy);                // :737-738
    const float dx = xx - xi, dy = yy - yi;
    if (mask) mask[b * g.es + o] = (xx >= 0 && xx <= g.w - 1 && yy >= 0 && yy <= g.h - 1) ? 1.0f : 0.0f;   // :742
    const int x1 = clampi(xi, 0, g.w - 1), x2 = clampi(xi + 1, 0, g.w - 1);
    const int y1 = clampi(yi, 0, g.h - 1), y2 = clampi(yi + 1, 0, g.h - 1);
    const float ax = 1.0f - dx, ay = 1.0f - dy;
    for (int k = 0; k < 3; k++) {
        const float *s = src3 + k * g.pl;
        dst3[k * g.pl + o] = s[(size_t)y1 * g.pitch + x1] * ax * ay + s[(size_t)y1 * g.pitch + x2] * dx * ay
                           + s[(size_t)y2 * g.pitch + x1] * ax * dy + s[(size_t)y2 * g.pitch + x2] * dx * dy;   // :748-753
    }
}
// all warps of one get_derivatives call in a single launch (grid z = window x job)
template <bool ALLJ>
__global__ void k_warp_jobs(WarpJobs J, float *__restrict__ base, const float *__restrict__ wx, const float *__restrict__ wy, Geo g) {
    const int b = ALLJ ? blockIdx.z : blockIdx.z / J.n, j0 = ALLJ ? 0 : blockIdx.z % J.n;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const long eb = b * g.es;
    const size_t o = (size_t)y * g.pitch + x;
    const float fx0 = wx[eb + o], fy0 = wy[eb + o];
    for (int j = j0; j <= (ALLJ ? J.n - 1 : j0); j++) {
        const int factor = J.job[j].factor;
        const float *src3 = base + eb + J.job[j].src_off;
        float *dst3 = base + eb + J.job[j].dst_off;
        const float xx = x + factor * fx0;                              // :735
        const float yy = y + factor * fy0;
        const int xi = floor_to_int_x86(xx), yi = floor_to_int_x86(yy);                // :737-738
        const float dx = xx - xi, dy = yy - yi;
        if (J.job[j].mask_off >= 0) base[eb + J.job[j].mask_off + o] = (xx >= 0 && xx <= g.w - 1 && yy >= 0 && yy <= g.h - 1) ? 1.0f : 0.0f;   // :742
        const int x1 = clampi(xi, 0, g.w - 1), x2 = clampi(xi + 1, 0, g.w - 1);
        const int y1 = clampi(yi, 0, g.h - 1), y2 = clampi(yi + 1, 0, g.h - 1);
        const float ax = 1.0f - dx, ay = 1.0f - dy;
        float out[3];                                                // all gathers before the stores (source and destination share `base`)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float *s = src3 + k * g.pl;
            out[k] = s[(size_t)y1 * g.pitch + x1] * ax * ay + s[(size_t)y1 * g.pitch + x2] * dx * ay
                   + s[(size_t)y2 * g.pitch + x1] * ax * dy + s[(size_t)y2 * g.pitch + x2] * dx * dy;               // :748-753
        }
#pragma unroll
        for (int k = 0; k < 3; k++) dst3[k * g.pl + o] = out[k];
    }
}
void launch_warp_jobs(sfa_ctx *c, const Geo &g, const WarpJobs &J, float *base, const float *wx, const float *wy) {
    if (J.n <= 0) return;
    // all jobs of a pixel in one thread (the flow is read once; 262 -> 250 us per 64-window launch) unless SFA_WARP_ALLJ=0 (one job per grid z, rounds 1-4)
    const bool allj = sw_int(Switches::WARP_ALLJ, 1) != 0;
    if (allj) hipLaunchKernelGGL(k_warp_jobs<true>, grid2d(g), block2d(), 0, c->stream, J, base, wx, wy, g);
    else hipLaunchKernelGGL(k_warp_jobs<false>, grid2d(g, J.n), block2d(), 0, c->stream, J, base, wx, wy, g);
}
void launch_warp(sfa_ctx *c, const Geo &g, float *dst3, float *mask, const float *src3, const float *wx, const float *wy, int factor, long src_es) {
    hipLaunchKernelGGL(k_warp, grid2d(g), block2d(), 0, c->stream, dst3, mask, src3, wx, wy, g, factor, src_es);
}

// ---------------------------------------------------------------------------------------------------
// K2/K3 derivative stack of get_derivatives (variational_mt.cpp:113-133)
//   stage 1: Iz = I1 - I2, Ix = D5x(M), Iy = D5y(M) with M = .5*(I2 + I1) formed on the fly
//   stage 2: Ixx = D5x(Ix), Ixy = D5y(Ix), Iyy = D5y(Iy), Ixz = D5x(Iz), Iyz = D5y(Iz)
// out24 plane order: Ix,Iy,Iz,Ixx,Ixy,Iyy,Ixz,Iyz, 3 channels each.
// ---------------------------------------------------------------------------------------------------

// One block = a 64x16 output tile of one (element, channel).  M and Iz are staged in LDS with a halo of 4, Ix/Iy are
// formed there with a halo of 2, and all eight planes leave in one pass: 2 plane reads + 8 plane writes per pixel
// instead of 5 + 8 through two kernels that lean on the caches for the vertical taps.  Border handling is the
// reference's (replicated columns, folded row coefficients), applied per stage exactly as the two-pass form does:
// clamped taps always land inside the image, hence inside the staged halo.
constexpr int DT_X = 64, DT_Y = 16, DT_H = 4, DT_W = DT_X + 2 * DT_H, DT_R = DT_Y + 2 * DT_H;
struct TileAcc {
    const float *t; int x0, y0;      // global coordinates of the tile's LDS origin
    __device__ __forceinline__ float operator()(int x, int y) const { return t[(y - y0) * DT_W + (x - x0)]; }
};
__global__ void __launch_bounds__(256) k_deriv_stack(float *__restrict__ out24, const float *__restrict__ I1, const float *__restrict__ I2, Geo g, long es1, long es2) {
    __shared__ float sM[DT_R * DT_W], sZ[DT_R * DT_W], sX[DT_R * DT_W], sY[DT_R * DT_W];
    const int b = blockIdx.z / 3, ch = blockIdx.z % 3;
    if (!elem_active(g, b)) return;
    const int x0 = blockIdx.x * DT_X - DT_H, y0 = blockIdx.y * DT_Y - DT_H;
    const float *a = I1 + b * es1 + ch * g.pl, *bb = I2 + b * es2 + ch * g.pl;
    for (int i = threadIdx.x; i < DT_R * DT_W; i += 256) {
        const int x = x0 + i % DT_W, y = y0 + i / DT_W;
        if (x >= 0 && x < g.w && y >= 0 && y < g.h) {
            const size_t o = (size_t)y * g.pitch + x;
            const float va = a[o], vb = bb[o];
            sM[i] = 0.5f * (vb + va);                                 // :120
            sZ[i] = va - vb;                                          // Iz  :122
        }
    }
    __syncthreads();
    const TileAcc m{sM, x0, y0};
    constexpr int W1 = DT_X + 4, R1 = DT_Y + 4;                       // halo of 2
    for (int i = threadIdx.x; i < R1 * W1; i += 256) {
        const int lx = 2 + i % W1, ly = 2 + i / W1;
        const int x = x0 + lx, y = y0 + ly;
        if (x >= 0 && x < g.w && y >= 0 && y < g.h) {
            sX[ly * DT_W + lx] = d5x(m, x, y, g.w);                   // Ix  :127
            sY[ly * DT_W + lx] = d5y(m, x, y, g.h);                   // Iy  :128
        }
    }
    __syncthreads();
    const TileAcc ix{sX, x0, y0}, iy{sY, x0, y0}, iz{sZ, x0, y0};
    float *o24 = out24 + b * g.es;
    const int tx = threadIdx.x & 63;
    const int x = x0 + DT_H + tx;
    if (x >= g.w) return;
#pragma unroll
    for (int k = 0; k < DT_Y / 4; k++) {
        const int y = y0 + DT_H + (threadIdx.x >> 6) + 4 * k;
        if (y >= g.h) break;
        const size_t o = (size_t)y * g.pitch + x;
        o24[(0 * 3 + ch) * g.pl + o] = ix(x, y);
        o24[(1 * 3 + ch) * g.pl + o] = iy(x, y);
        o24[(2 * 3 + ch) * g.pl + o] = iz(x, y);
        o24[(3 * 3 + ch) * g.pl + o] = d5x(ix, x, y, g.w);            // Ixx :129
        o24[(4 * 3 + ch) * g.pl + o] = d5y(ix, x, y, g.h);            // Ixy :130
        o24[(5 * 3 + ch) * g.pl + o] = d5y(iy, x, y, g.h);            // Iyy :131
        o24[(6 * 3 + ch) * g.pl + o] = d5x(iz, x, y, g.w);            // Ixz :132
        o24[(7 * 3 + ch) * g.pl + o] = d5y(iz, x, y, g.h);            // Iyz :133
    }
}
void launch_deriv_stack(sfa_ctx *c, const Geo &g, float *out24, const float *I1, const float *I2, long es1, long es2) {
    hipLaunchKernelGGL(k_deriv_stack, dim3((g.w + DT_X - 1) / DT_X, (g.h + DT_Y - 1) / DT_Y, g.nb * 3), dim3(256), 0, c->stream, out24, I1, I2, g, es1, es2);
}

// generic single-filter convolution (stage API: convolve_horiz / convolve_vert, image.c:400-526)
__global__ void k_convolve(float *__restrict__ dst, const float *__restrict__ src, Geo g, int order, int horiz, int nplanes) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    PlaneAcc s{src + b * g.es + pl * g.pl, g.pitch};
    float v;
    if (order == 2) v = horiz ? d5x(s, x, y, g.w) : d5y(s, x, y, g.h);
    else            v = horiz ? d3x(s, x, y, g.w) : d3y(s, x, y, g.h);
    dst[b * g.es + pl * g.pl + (size_t)y * g.pitch + x] = v;
}
void launch_convolve(sfa_ctx *c, const Geo &g, float *dst, const float *src, int order, int horiz, int nplanes) {
    hipLaunchKernelGGL(k_convolve, grid2d(g, nplanes), block2d(), 0, c->stream, dst, src, g, order, horiz, nplanes);
}

// ---------------------------------------------------------------------------------------------------
// K4 compute_dpsis_weight, first output (variational_aux_mt.cpp:673-719)
// ---------------------------------------------------------------------------------------------------
// The reference calls glibc's expf.  glibc (>= 2.27) is NOT correctly rounded (0.502 ulp: ~0.06 % of the arguments
// round the other way), so a correctly rounded exponential does not reproduce it.  This is glibc's published
// algorithm (sysdeps/ieee754/flt-32/e_expf.c, from Arm's optimized-routines: N = 32 table of 2^(i/N), degree-3
// polynomial, all in fp64), restated; tests/test_oracle_pin.py::test_expf_restatement checks the same restatement
// against libm on 5e5 arguments (0 mismatches).  T[i] = bits(2^(i/32)) - (i << 47).
__device__ const unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
__device__ __forceinline__ float expf_glibc(float x) {
    if (x < -0x1.9fe368p6f) return 0.0f;                      // underflow (not reached by the path: x in [-5*|grad lum|, 0])
    if (x > 0x1.62e42ep6f) return __builtin_inff();
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + Shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= Shift;
    const double r = z - kd;
    unsigned long long t = kExp2fTab[ki % 32];
    t += ki << (52 - 5);
    const double s = __longlong_as_double((long long)t);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}
struct LumAcc {
    const float *c1, *c2, *c3; int pitch; float a1, a2, a3, s1, s2, s3; int hbit;
    __device__ __forceinline__ float operator()(int x, int y) const {
        const size_t o = (size_t)y * pitch + x;
        const float v = 0.299f * (c1[o] * s1 + a1) + 0.587f * (c2[o] * s2 + a2) + 0.114f * (c3[o] * s3 + a3);   // :681,683
        return hbit ? __fdiv_rn(v, 65535.0f) : __fdiv_rn(v, 255.0f);
    }
};
// compute_dpsis_weight through an LDS tile of the luminance (64 x 16 pixels, halo 2).  Until round 5 every thread evaluated the luminance -- three loads, six operations and
// an IEEE division -- at each of its ten taps: 230 instructions and 30 loads per pixel for a 16-byte-per-pixel kernel (0.17 ms per 64-window level where the traffic takes
// 0.05).  Same values, same operations per value: bit-identical (test_dpsis_weight).
constexpr int DPS_TX = 64, DPS_TY = 16, DPS_W = DPS_TX + 4, DPS_R = DPS_TY + 4;
struct LumTile {
    const float *t; int x0, y0;                                  // tile entry (0, 0) = pixel (x0, y0)
    __device__ __forceinline__ float operator()(int x, int y) const { return t[(y - y0) * DPS_W + (x - x0)]; }
};
__global__ void __launch_bounds__(256) k_dpsis_tiled(float *__restrict__ dst, const float *__restrict__ im3, Geo g, long im_es, float coef, float a1, float a2, float a3,
                                                     float s1, float s2, float s3, int hbit) {
    __shared__ float tL[DPS_R * DPS_W];
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x0 = blockIdx.x * DPS_TX - 2, y0 = blockIdx.y * DPS_TY - 2;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const float *im = im3 + b * im_es;
    const LumAcc lum{im, im + g.pl, im + 2 * g.pl, g.pitch, a1, a2, a3, s1, s2, s3, hbit};
    for (int i = tid; i < DPS_R * DPS_W; i += 256) {
        const int gx = x0 + i % DPS_W, gy = y0 + i / DPS_W;
        if (gx >= 0 && gx < g.w && gy >= 0 && gy < g.h) tL[i] = lum(gx, gy);       // the taps are clamped to the image (d5x) / folded at its rows (d5y): nothing outside is read
    }
    __syncthreads();
    const LumTile L{tL, x0, y0};
    const int x = x0 + 2 + threadIdx.x;
    if (x >= g.w) return;
#pragma unroll
    for (int k = 0; k < DPS_TY / 4; k++) {
        const int y = y0 + 2 + threadIdx.y + 4 * k;
        if (y >= g.h) break;
        const float lx = d5x(L, x, y, g.w), ly = d5y(L, x, y, g.h);
        const float n = -coef * sqrt_rn(lx * lx + ly * ly);            // :699
        dst[b * g.es + (size_t)y * g.pitch + x] = 0.5f * expf_glibc(n);          // :700
    }
}
void launch_dpsis(sfa_ctx *c, const Geo &g, float *dst, const float *im3, long im_es, float coef, const float avg[3], const float stdv[3], int hbit) {
    hipLaunchKernelGGL(k_dpsis_tiled, dim3((g.w + DPS_TX - 1) / DPS_TX, (g.h + DPS_TY - 1) / DPS_TY, g.nb), dim3(64, 4), 0, c->stream, dst, im3, g, im_es, coef, avg[0], avg[1],
                       avg[2], stdv[0], stdv[1], stdv[2], hbit);
}

// ---------------------------------------------------------------------------------------------------
// K5 compute_smoothness (variational_aux_mt.cpp:18-127)
// ---------------------------------------------------------------------------------------------------
__global__ void k_smoothness(int method, float *__restrict__ sh, float *__restrict__ sv, const float *__restrict__ uu_, const float *__restrict__ vv_,
                             const float *__restrict__ dps_, Geo g, float alpha, PenaltyDev reg) {
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.pitch || y >= g.h) return;
    const size_t o = (size_t)y * g.pitch + x;
    sh += b * g.es; sv += b * g.es;
    if (x >= g.w) { sh[o] = 0.0f; sv[o] = 0.0f; return; }            // padding lanes stay zero
    PlaneAcc uu{uu_ + b * g.es, g.pitch}, vv{vv_ + b * g.es, g.pitch}, dps{dps_ + b * g.es, g.pitch};
    const int w = g.w, h = g.h;
    if (method <= 1) {
        float outh = 0.0f, outv = 0.0f;
        if (x < w - 1) {
            const float ux1 = uu(x + 1, y) - uu(x, y), vx1 = vv(x + 1, y) - vv(x, y);     // :27-28
            float t = 0.0f, t2 = 0.0f;
            if (method == 1) {
                t = 0.5f * (d3y(uu, x, y, h) + d3y(uu, x + 1, y, h));                      // :57
                t2 = 0.5f * (d3y(vv, x, y, h) + d3y(vv, x + 1, y, h));
            }
            t = ux1 * ux1 + t * t;                                                       // :61-64
            t2 = vx1 * vx1 + t2 * t2;
            t = t + t2;
            outh = (dps(x, y) + dps(x + 1, y)) * alpha * psi_scalar(reg, t);              // :66
        }
        if (y < h - 1) {
            const float uy1 = uu(x, y + 1) - uu(x, y), vy1 = vv(x, y + 1) - vv(x, y);     // :35-36
            float t = 0.0f, t2 = 0.0f;
            if (method == 1) {
                t = 0.5f * (d3x(uu, x, y, w) + d3x(uu, x, y + 1, w));                      // :80
                t2 = 0.5f * (d3x(vv, x, y, w) + d3x(vv, x, y + 1, w));
            }
            t = uy1 * uy1 + t * t;                                                       // :84-87
            t2 = vy1 * vy1 + t2 * t2;
            t = t + t2;
            outv = (dps(x, y) + dps(x, y + 1)) * alpha * psi_scalar(reg, t);              // :89
        }
        sh[o] = outh;                                                                    // :68 zero last column
        sv[o] = outv;                                                                    // :92 zero last row
    } else {
        // :96-116 as written: `float w` shadows the width, so the horizontal test compares x with (weight - 1)
        float t = 0.0f;
        float wgt = dps(x, y);
        if ((float)x < wgt - 1) {
            const float ux1 = uu(x + 1, y) - uu(x, y), vx1 = vv(x + 1, y) - vv(x, y);
            t += ux1 * ux1 + vx1 * vx1;
            wgt += dps(x + 1, y);
        }
        if (y < h - 1) {
            const float uy1 = uu(x, y + 1) - uu(x, y), vy1 = vv(x, y + 1) - vv(x, y);
            t += vy1 * vy1 + uy1 * uy1;
            wgt += dps(x, y + 1);
        }
        const float v = wgt * alpha * psi_scalar(reg, t);
        sh[o] = v;
        sv[o] = v;
    }
}
// methods 0 and 1 with uu, vv, dpsis staged in LDS (halo 1): ~21 neighbour reads per pixel come from the tile instead of L1
constexpr int SM_X = 64, SM_Y = 8, SM_W = SM_X + 2, SM_R = SM_Y + 2;
struct SmTile {
    const float *t; int x0, y0;
    __device__ __forceinline__ float operator()(int x, int y) const { return t[(y - y0) * SM_W + (x - x0)]; }
};
__global__ void __launch_bounds__(256) k_smoothness_tiled(int method, float *__restrict__ sh, float *__restrict__ sv, const float *__restrict__ uu_,
                                                          const float *__restrict__ vv_, const float *__restrict__ dps_, Geo g, float alpha, PenaltyDev reg) {
    __shared__ float tU[SM_R * SM_W], tV[SM_R * SM_W], tD[SM_R * SM_W];
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x0 = blockIdx.x * SM_X - 1, y0 = blockIdx.y * SM_Y - 1;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const float *pu = uu_ + b * g.es, *pv = vv_ + b * g.es, *pd = dps_ + b * g.es;
    for (int i = tid; i < SM_R * SM_W; i += 256) {
        const int gx = x0 + i % SM_W, gy = y0 + i / SM_W;
        if (gx >= 0 && gx < g.w && gy >= 0 && gy < g.h) {
            const size_t o = (size_t)gy * g.pitch + gx;
            tU[i] = pu[o]; tV[i] = pv[o]; tD[i] = pd[o];
        }
    }
    __syncthreads();
    const SmTile uu{tU, x0, y0}, vv{tV, x0, y0}, dps{tD, x0, y0};
    const int w = g.w, h = g.h;
    const int x = x0 + 1 + threadIdx.x;
    sh += b * g.es; sv += b * g.es;
    for (int k = 0; k < SM_Y / 4; k++) {
        const int y = y0 + 1 + threadIdx.y + 4 * k;
        if (x >= g.pitch || y >= h) continue;
        const size_t o = (size_t)y * g.pitch + x;
        if (x >= w) { sh[o] = 0.0f; sv[o] = 0.0f; continue; }        // padding lanes stay zero
        float outh = 0.0f, outv = 0.0f;
        if (x < w - 1) {
            const float ux1 = uu(x + 1, y) - uu(x, y), vx1 = vv(x + 1, y) - vv(x, y);     // :27-28
            float t = 0.0f, t2 = 0.0f;
            if (method == 1) {
                t = 0.5f * (d3y(uu, x, y, h) + d3y(uu, x + 1, y, h));                      // :57
                t2 = 0.5f * (d3y(vv, x, y, h) + d3y(vv, x + 1, y, h));
            }
            t = ux1 * ux1 + t * t;                                                       // :61-64
            t2 = vx1 * vx1 + t2 * t2;
            t = t + t2;
            outh = (dps(x, y) + dps(x + 1, y)) * alpha * psi_scalar(reg, t);              // :66
        }
        if (y < h - 1) {
            const float uy1 = uu(x, y + 1) - uu(x, y), vy1 = vv(x, y + 1) - vv(x, y);     // :35-36
            float t = 0.0f, t2 = 0.0f;
            if (method == 1) {
                t = 0.5f * (d3x(uu, x, y, w) + d3x(uu, x, y + 1, w));                      // :80
                t2 = 0.5f * (d3x(vv, x, y, w) + d3x(vv, x, y + 1, w));
            }
            t = uy1 * uy1 + t * t;                                                       // :84-87
            t2 = vy1 * vy1 + t2 * t2;
            t = t + t2;
            outv = (dps(x, y) + dps(x, y + 1)) * alpha * psi_scalar(reg, t);              // :89
        }
        sh[o] = outh;                                                                    // :68 zero last column
        sv[o] = outv;                                                                    // :92 zero last row
    }
}
void launch_smoothness(sfa_ctx *c, const Geo &g, int method, float *sh, float *sv, const float *uu, const float *vv, const float *dpsis, float alpha,
                       PenaltyDev reg) {
    if (method <= 1) {
        hipLaunchKernelGGL(k_smoothness_tiled, dim3((g.pitch + SM_X - 1) / SM_X, (g.h + SM_Y - 1) / SM_Y, g.nb), dim3(64, 4), 0, c->stream, method, sh, sv, uu, vv, dpsis,
                           g, alpha, reg);
        return;
    }
    dim3 grid((g.pitch + BX - 1) / BX, (g.h + BY - 1) / BY, g.nb);
    hipLaunchKernelGGL(k_smoothness, grid, block2d(), 0, c->stream, method, sh, sv, uu, vv, dpsis, g, alpha, reg);
}

// get_derivatives' warps (variational_mt.cpp:100-109) and compute_smoothness (:333) of the same flow field in one pass: both read wx, wy of every pixel, the
// smoothness kernel alone is bound by its two IEEE square-root / division chains per pixel (2.8 TB/s) and the warp by its traffic -- together the arithmetic runs
// under the gathers and the flow tile is read once.  Same per-pixel operations as k_smoothness_tiled and k_warp_jobs (bit-identical: test_fused_warp_smoothness).
// rows of the tile in flight (= waves per block); a thread works on SM_Y / WS_ROWS pixels one after the other.  Round 6: 8 (one pixel per thread: no gather of a
// second pixel queues behind the first one's stores in the wave's in-order memory counter; 4 until then)
#ifndef SFA_WS_ROWS
#define SFA_WS_ROWS 8
#endif
#ifndef SFA_WS_NT
#define SFA_WS_NT 1
#endif
constexpr int WS_ROWS = SFA_WS_ROWS;
static_assert(SM_Y % WS_ROWS == 0, "whole pixels per thread");
__global__ void __launch_bounds__(64 * WS_ROWS) k_warp_smooth(WarpJobs J, float *__restrict__ base, int method, float *__restrict__ sh, float *__restrict__ sv,
                                                     const float *__restrict__ uu_, const float *__restrict__ vv_, const float *__restrict__ dps_, Geo g, float alpha,
                                                     PenaltyDev reg) {
    __shared__ float tU[SM_R * SM_W], tV[SM_R * SM_W], tD[SM_R * SM_W];
    // what-if (-DSFA_WS_XCD=1, round 6; round 5 had measured the time only): a 1-D grid whose workgroup i works on tile (i % 8) * ceil(N / 8) + i / 8 of the row-major
    // (window, tile row, tile column) order, as k_assemble_images does -- every XCD walks a contiguous eighth of the tiles, so the third 128-byte line a shifted
    // 64-column gather footprint touches is its horizontal neighbour's line in the SAME L2
#if defined(SFA_WS_XCD) && SFA_WS_XCD
    int bx_, by_, b_;
    {
        const unsigned nx = (unsigned)((g.pitch + SM_X - 1) / SM_X), ny = (unsigned)((g.h + SM_Y - 1) / SM_Y), nxy = nx * ny, total = nxy * (unsigned)g.nb;
        const unsigned chunk = (total + 7u) / 8u;
        const unsigned j = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
        if (j >= total) return;
        b_ = (int)(j / nxy);
        const unsigned r = j - (unsigned)b_ * nxy;
        by_ = (int)(r / nx); bx_ = (int)(r - (unsigned)by_ * nx);
    }
    const int b = b_;
    if (!elem_active(g, b)) return;
    const int x0 = bx_ * SM_X - 1, y0 = by_ * SM_Y - 1;
#else
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x0 = blockIdx.x * SM_X - 1, y0 = blockIdx.y * SM_Y - 1;
#endif
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const long eb = b * g.es;
    // Round 6: every address of this kernel is a WAVE-UNIFORM base (window, plane, image set: scalar registers) + a 32-bit byte offset per lane (a plane is < 2^32 bytes:
    // the launcher checks it): the loads and stores take the scalar-base form.  Written with size_t indices the compiler formed every one of the 24 gather
    // addresses of a pixel in 64-bit VALU arithmetic -- per gather set 2 v_mad_i64_i32, 4 v_lshlrev_b64 and 18 v_lshl_add_u64: 127 VALU instructions and 13 registers less.
    // What bounds the kernel is neither that nor its traffic (what-if builds of this round, 128 windows, 650-680 us per launch as shipped before): the XCD-contiguous
    // tile order halves the fetched bytes (2.57 -> 1.42 GB: the over-fetch IS the cross-XCD duplicate of the gathers' third line) and leaves the time at 662; without
    // the smoothness weights' fp64 square roots and divisions it takes 823 (!); without the gathers 431; without the stores of the warped images 289 -- the
    // write path is what it waits for, and what moved it is how the stores leave (non-temporal, below) and that no load queues behind them (one pixel per thread)
    // timing-only what-if builds (-DSFA_X_WS=bits, wrong results, never shipped): 1 every tap of a gather reads ONE address (the pixel's own), 2 the smoothness weights
    // without their fp64 square root and division, 4 no stores of the warped images, 8 no gathers at all, 32 masks, sh, sv as non-temporal stores too
#ifndef SFA_X_WS
#define SFA_X_WS 0
#endif
    auto ld = [](const float *ubase, unsigned boff) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + boff); };
    auto st = [](float *ubase, unsigned boff, float v) {
        if (SFA_X_WS & 32) __builtin_nontemporal_store(v, reinterpret_cast<float *>(reinterpret_cast<char *>(ubase) + boff));      // what-if: masks, sh, sv as streaming stores too
        else *reinterpret_cast<float *>(reinterpret_cast<char *>(ubase) + boff) = v;
    };
    const unsigned pitch4 = 4u * (unsigned)g.pitch;
    const float *pu = uu_ + eb, *pv = vv_ + eb, *pd = dps_ + eb;
    for (int i = tid; i < SM_R * SM_W; i += 64 * WS_ROWS) {
        const int gx = x0 + i % SM_W, gy = y0 + i / SM_W;
        if (gx >= 0 && gx < g.w && gy >= 0 && gy < g.h) {
            const unsigned o4 = (unsigned)gy * pitch4 + 4u * (unsigned)gx;
            tU[i] = ld(pu, o4); tV[i] = ld(pv, o4); tD[i] = ld(pd, o4);
        }
    }
    __syncthreads();
    const SmTile uu{tU, x0, y0}, vv{tV, x0, y0}, dps{tD, x0, y0};
    const int w = g.w, h = g.h;
    const int x = x0 + 1 + threadIdx.x;
    sh += eb; sv += eb;
    for (int k = 0; k < SM_Y / WS_ROWS; k++) {
        const int y = y0 + 1 + threadIdx.y + WS_ROWS * k;
        if (x >= g.pitch || y >= h) continue;
        const unsigned o4 = (unsigned)y * pitch4 + 4u * (unsigned)x;
        if (x >= w) { st(sh, o4, 0.0f); st(sv, o4, 0.0f); continue; }        // padding lanes stay zero
        // ---- the warps of this pixel: the gathers of the first two jobs are issued here, the smoothness arithmetic runs while they are in flight ---------
        const float fx0 = uu(x, y), fy0 = vv(x, y);
        auto gather = [&](int j, float(&out)[3]) {
            const int factor = J.job[j].factor;
            const float *src3 = base + eb + J.job[j].src_off;
            const float xx = x + factor * fx0;                              // :735
            const float yy = y + factor * fy0;
            const int xi = floor_to_int_x86(xx), yi = floor_to_int_x86(yy);                // :737-738
            const float dx = xx - xi, dy = yy - yi;
            if (J.job[j].mask_off >= 0) st(base + eb + J.job[j].mask_off, o4, (xx >= 0 && xx <= g.w - 1 && yy >= 0 && yy <= g.h - 1) ? 1.0f : 0.0f);   // :742
            const int x1 = clampi(xi, 0, g.w - 1), x2 = clampi(xi + 1, 0, g.w - 1);
            const int y1 = clampi(yi, 0, g.h - 1), y2 = clampi(yi + 1, 0, g.h - 1);
            const float ax = 1.0f - dx, ay = 1.0f - dy;
            const unsigned r1 = (unsigned)y1 * pitch4, r2 = (unsigned)y2 * pitch4, c1 = 4u * (unsigned)x1, c2 = 4u * (unsigned)x2;
            const unsigned o11 = (SFA_X_WS & 1) ? o4 : r1 + c1, o12 = (SFA_X_WS & 1) ? o4 : r1 + c2, o21 = (SFA_X_WS & 1) ? o4 : r2 + c1, o22 = (SFA_X_WS & 1) ? o4 : r2 + c2;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float *sp = src3 + ch * g.pl;
                if (SFA_X_WS & 8) { out[ch] = ax * ay + dx * dy; continue; }
#if defined(SFA_WS_LDNT) && SFA_WS_LDNT      // what-if (round 6): the gathers as non-temporal loads
                auto ldn = [](const float *ubase, unsigned boff) { return __builtin_nontemporal_load(reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + boff)); };
                out[ch] = ldn(sp, o11) * ax * ay + ldn(sp, o12) * dx * ay + ldn(sp, o21) * ax * dy + ldn(sp, o22) * dx * dy;
#else
                out[ch] = ld(sp, o11) * ax * ay + ld(sp, o12) * dx * ay + ld(sp, o21) * ax * dy + ld(sp, o22) * dx * dy;               // :748-753
#endif
            }
        };
        auto put = [&](int j, const float(&out)[3]) {
            float *dst3 = base + eb + J.job[j].dst_off;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                if (SFA_X_WS & 4) continue;
                // Round 6: the warped images leave as NON-TEMPORAL stores (24 of the kernel's 40 written bytes per pixel; the data-term kernel reads them back by DMA,
                // from memory either way: 0.94 GB per 128-window launch).  Same box, whole path: 111.5 / 112.0 -> 110.6 / 110.6 ms per step together with the eight-row
                // block below (-DSFA_WS_NT=0: plain stores).  NOT for the solver's operand tile in k_assemble_images: the solver reads that from the caches (+3-5 % of
                // the whole path as non-temporal stores, -DSFA_ASM_NT=1)
                if (SFA_WS_NT) __builtin_nontemporal_store(out[ch], reinterpret_cast<float *>(reinterpret_cast<char *>(dst3 + ch * g.pl) + o4));
                else st(dst3 + ch * g.pl, o4, out[ch]);
            }
        };
        float out0[3] = {0, 0, 0}, out1[3] = {0, 0, 0};
        gather(0, out0);
        if (J.n > 1) gather(1, out1);
        // ---- compute_smoothness ----------------------------------------------------------------------------------------------------------------------
        float outh = 0.0f, outv = 0.0f;
        if (x < w - 1) {
            const float ux1 = uu(x + 1, y) - uu(x, y), vx1 = vv(x + 1, y) - vv(x, y);     // :27-28
            float t = 0.0f, t2 = 0.0f;
            if (method == 1) {
                t = 0.5f * (d3y(uu, x, y, h) + d3y(uu, x + 1, y, h));                      // :57
                t2 = 0.5f * (d3y(vv, x, y, h) + d3y(vv, x + 1, y, h));
            }
            t = ux1 * ux1 + t * t;                                                       // :61-64
            t2 = vx1 * vx1 + t2 * t2;
            t = t + t2;
            outh = (dps(x, y) + dps(x + 1, y)) * alpha * ((SFA_X_WS & 2) ? t : psi_scalar(reg, t));              // :66
        }
        if (y < h - 1) {
            const float uy1 = uu(x, y + 1) - uu(x, y), vy1 = vv(x, y + 1) - vv(x, y);     // :35-36
            float t = 0.0f, t2 = 0.0f;
            if (method == 1) {
                t = 0.5f * (d3x(uu, x, y, w) + d3x(uu, x, y + 1, w));                      // :80
                t2 = 0.5f * (d3x(vv, x, y, w) + d3x(vv, x, y + 1, w));
            }
            t = uy1 * uy1 + t * t;                                                       // :84-87
            t2 = vy1 * vy1 + t2 * t2;
            t = t + t2;
            outv = (dps(x, y) + dps(x, y + 1)) * alpha * ((SFA_X_WS & 2) ? t : psi_scalar(reg, t));              // :89
        }
        st(sh, o4, outh);                                                                // :68 zero last column
        st(sv, o4, outv);                                                                // :92 zero last row
        put(0, out0);
        if (J.n > 1) put(1, out1);
        for (int j = 2; j < J.n; j++) {                                                  // S >= 3: the further jobs one after the other
            float o3[3];
            gather(j, o3);
            put(j, o3);
        }
    }
}
// false: not this combination (the caller then launches the two kernels)
bool launch_warp_smooth(sfa_ctx *c, const Geo &g, const WarpJobs &J, float *base, const float *wx, const float *wy, int method, float *sh, float *sv,
                        const float *dpsis, float alpha, PenaltyDev reg) {
    const bool off = sw_given(Switches::NO_WARP_SMOOTH);
    if (off || method > 1 || J.n <= 0) return false;
    if ((unsigned long long)g.pl * 4ull >= (1ull << 32)) return false;      // the kernel's 32-bit byte offsets inside a plane (the two kernels it fuses address with 64 bits)
#if defined(SFA_WS_XCD) && SFA_WS_XCD
    const unsigned tiles = (unsigned)((g.pitch + SM_X - 1) / SM_X) * (unsigned)((g.h + SM_Y - 1) / SM_Y) * (unsigned)g.nb;
    hipLaunchKernelGGL(k_warp_smooth, dim3((tiles + 7) / 8 * 8), dim3(64, WS_ROWS), 0, c->stream, J, base, method, sh, sv, wx, wy, dpsis, g, alpha, reg);
#else
    hipLaunchKernelGGL(k_warp_smooth, dim3((g.pitch + SM_X - 1) / SM_X, (g.h + SM_Y - 1) / SM_Y, g.nb), dim3(64, WS_ROWS), 0, c->stream, J, base, method, sh, sv, wx, wy,
                       dpsis, g, alpha, reg);
#endif
    return true;
}

// ---------------------------------------------------------------------------------------------------
// sub_laplacian, gather form (variational_aux_mt.cpp:130-161).  The reference scatters edge by edge:
// all horizontal edges in raster order, then all vertical ones; pixel (x,y) therefore receives
//   b = (((b - th[x-1]) + th[x]) - tv[y-1]) + tv[y],   th[x] = wh[x]*(src[x+1]-src[x]), tv likewise,
// with absent border terms skipped (not added as zero).
// ---------------------------------------------------------------------------------------------------
template <class SrcAcc>
__device__ __forceinline__ float laplacian_gather(float bval, const SrcAcc &src, const SrcAcc &wh, const SrcAcc &wv, int x, int y, int w, int h) {
    const float c = src(x, y);
    if (x > 0) bval -= wh(x - 1, y) * (c - src(x - 1, y));
    if (x < w - 1) bval += wh(x, y) * (src(x + 1, y) - c);
    if (y > 0) bval -= wv(x, y - 1) * (c - src(x, y - 1));
    if (y < h - 1) bval += wv(x, y) * (src(x, y + 1) - c);
    return bval;
}
__global__ void k_sub_laplacian(float *__restrict__ dst, const float *__restrict__ src, const float *__restrict__ wh, const float *__restrict__ wv, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    PlaneAcc s{src + b * g.es, g.pitch}, h_{wh + b * g.es, g.pitch}, v_{wv + b * g.es, g.pitch};
    const size_t o = b * g.es + (size_t)y * g.pitch + x;
    dst[o] = laplacian_gather(dst[o], s, h_, v_, x, y, g.w, g.h);
}
void launch_sub_laplacian(sfa_ctx *c, const Geo &g, float *dst, const float *src, const float *wh, const float *wv) {
    hipLaunchKernelGGL(k_sub_laplacian, grid2d(g), block2d(), 0, c->stream, dst, src, wh, wv, g);
}

// ---------------------------------------------------------------------------------------------------
// The original two-frame refinement (epic_flow_extended/variational.c + variational_aux.c): smoothness and data term with
// its fixed modified-L1 penalties (eps = 0.001) in ITS operation order (the multi-frame class above rounds differently).
// ---------------------------------------------------------------------------------------------------
#define EPS2F (0.001f * 0.001f)     // epsilon_color / _grad / _smooth, variational_aux.c:11-13
__global__ void k_smoothness_2f(float *__restrict__ sh, float *__restrict__ sv, const float *__restrict__ uu_, const float *__restrict__ vv_,
                                const float *__restrict__ dps_, Geo g, float half_alpha) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.pitch || y >= g.h) return;
    const size_t o = (size_t)y * g.pitch + x;
    sh += b * g.es; sv += b * g.es;
    if (x >= g.w) { sh[o] = 0.0f; sv[o] = 0.0f; return; }
    PlaneAcc uu{uu_ + b * g.es, g.pitch}, vv{vv_ + b * g.es, g.pitch}, dps{dps_ + b * g.es, g.pitch};
    const int w = g.w, h = g.h;
    float outh = 0.0f, outv = 0.0f;
    if (x < w - 1) {                                                                     // variational_aux.c:115-128
        const float ux1 = uu(x + 1, y) - uu(x, y), vx1 = vv(x + 1, y) - vv(x, y);
        float tmp = 0.5f * (d3y(uu, x, y, h) + d3y(uu, x + 1, y, h));
        const float uxsq = ux1 * ux1 + tmp * tmp;
        tmp = 0.5f * (d3y(vv, x, y, h) + d3y(vv, x + 1, y, h));
        const float vxsq = vx1 * vx1 + tmp * tmp;
        tmp = uxsq + vxsq;
        outh = (float)((double)((dps(x, y) + dps(x + 1, y)) * half_alpha) / __dsqrt_rn((double)(tmp + EPS2F)));   // double sqrt and division (:126)
    }
    if (y < h - 1) {                                                                     // :130-146
        const float uy1 = uu(x, y + 1) - uu(x, y), vy1 = vv(x, y + 1) - vv(x, y);
        float tmp = 0.5f * (d3x(uu, x, y, w) + d3x(uu, x, y + 1, w));
        const float uysq = uy1 * uy1 + tmp * tmp;
        tmp = 0.5f * (d3x(vv, x, y, w) + d3x(vv, x, y + 1, w));
        const float vysq = vy1 * vy1 + tmp * tmp;
        tmp = uysq + vysq;
        outv = (float)((double)((dps(x, y) + dps(x, y + 1)) * half_alpha) / __dsqrt_rn((double)(tmp + EPS2F)));
    }
    sh[o] = outh;
    sv[o] = outv;
}
void launch_smoothness_2f(sfa_ctx *c, const Geo &g, float *sh, float *sv, const float *uu, const float *vv, const float *dpsis, float half_alpha) {
    dim3 grid((g.pitch + BX - 1) / BX, (g.h + BY - 1) / BY, g.nb);
    hipLaunchKernelGGL(k_smoothness_2f, grid, block2d(), 0, c->stream, sh, sv, uu, vv, dpsis, g, half_alpha);
}
// compute_data_and_match (variational_aux.c:215-302) followed by both sub_laplacian calls (variational.c:56-57, on wx / wy)
__global__ void __launch_bounds__(BX *BY) k_data_2f(const float *__restrict__ D, const float *__restrict__ mask, const float *__restrict__ du, const float *__restrict__ dv,
                                                     float *__restrict__ a11, float *__restrict__ a12, float *__restrict__ a22, float *__restrict__ b1, float *__restrict__ b2,
                                                     const float *__restrict__ wx, const float *__restrict__ wy, const float *__restrict__ sh, const float *__restrict__ sv,
                                                     Geo g, float hd, float hg) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const long eb = b * g.es;
    const size_t o = (size_t)y * g.pitch + x;
    const float *S = D + eb + o;
    float ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ix[k] = S[(0 * 3 + k) * g.pl]; iy[k] = S[(1 * 3 + k) * g.pl]; iz[k] = S[(2 * 3 + k) * g.pl];
        ixx[k] = S[(3 * 3 + k) * g.pl]; ixy[k] = S[(4 * 3 + k) * g.pl]; iyy[k] = S[(5 * 3 + k) * g.pl];
        ixz[k] = S[(6 * 3 + k) * g.pl]; iyz[k] = S[(7 * 3 + k) * g.pl];
    }
    const float u = du[eb + o], v = dv[eb + o], m = mask[eb + o];
    const float dn = 0.1f * 0.1f;                                                        // datanorm (:10)
    float A11 = 0.0f, A12 = 0.0f, A22 = 0.0f, B1 = 0.0f, B2 = 0.0f;
    if (hd) {                                                                            // :245-269
        float t[3], n[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { t[k] = iz[k] + ix[k] * u + iy[k] * v; n[k] = ix[k] * ix[k] + iy[k] * iy[k] + dn; }
        const float q = __fdiv_rn(m * hd, sqrt_rn(__fdiv_rn(t[0] * t[0], n[0]) + __fdiv_rn(t[1] * t[1], n[1]) + __fdiv_rn(t[2] * t[2], n[2]) + EPS2F));
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float tk = __fdiv_rn(q, n[k]);
            A11 += tk * ix[k] * ix[k]; A12 += tk * ix[k] * iy[k]; A22 += tk * iy[k] * iy[k];
            B1 -= tk * iz[k] * ix[k];  B2 -= tk * iz[k] * iy[k];
        }
    }
    float t[6], n[6];                                                                    // :271-300
#pragma unroll
    for (int k = 0; k < 3; k++) {
        n[2 * k] = ixx[k] * ixx[k] + ixy[k] * ixy[k] + dn;
        n[2 * k + 1] = iyy[k] * iyy[k] + ixy[k] * ixy[k] + dn;
        t[2 * k] = ixz[k] + ixx[k] * u + ixy[k] * v;
        t[2 * k + 1] = iyz[k] + ixy[k] * u + iyy[k] * v;
    }
    const float q = __fdiv_rn(m * hg, sqrt_rn(__fdiv_rn(t[0] * t[0], n[0]) + __fdiv_rn(t[1] * t[1], n[1]) + __fdiv_rn(t[2] * t[2], n[2]) + __fdiv_rn(t[3] * t[3], n[3]) +
                                              __fdiv_rn(t[4] * t[4], n[4]) + __fdiv_rn(t[5] * t[5], n[5]) + EPS2F));
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float ta = __fdiv_rn(q, n[2 * k]), tb = __fdiv_rn(q, n[2 * k + 1]);
        A11 += ta * ixx[k] * ixx[k] + tb * ixy[k] * ixy[k];
        A12 += ta * ixx[k] * ixy[k] + tb * ixy[k] * iyy[k];
        A22 += tb * iyy[k] * iyy[k] + ta * ixy[k] * ixy[k];
        B1 -= ta * ixx[k] * ixz[k] + tb * ixy[k] * iyz[k];
        B2 -= tb * iyy[k] * iyz[k] + ta * ixy[k] * ixz[k];
    }
    PlaneAcc U{wx + eb, g.pitch}, V{wy + eb, g.pitch}, H{sh + eb, g.pitch}, W{sv + eb, g.pitch};
    B1 = laplacian_gather(B1, U, H, W, x, y, g.w, g.h);
    B2 = laplacian_gather(B2, V, H, W, x, y, g.w, g.h);
    a11[eb + o] = A11; a12[eb + o] = A12; a22[eb + o] = A22; b1[eb + o] = B1; b2[eb + o] = B2;
}
void launch_data_2f(sfa_ctx *c, const Geo &g, const float *D, const float *mask, const float *du, const float *dv, float *a11, float *a12, float *a22, float *b1,
                    float *b2, const float *wx, const float *wy, const float *sh, const float *sv, float hd, float hg) {
    hipLaunchKernelGGL(k_data_2f, grid2d(g), block2d(), 0, c->stream, D, mask, du, dv, a11, a12, a22, b1, b2, wx, wy, sh, sv, g, hd, hg);
}

// ---------------------------------------------------------------------------------------------------
// mask weighting by occlusion / direction (variational_mt.cpp:293-320)
// ---------------------------------------------------------------------------------------------------
__global__ void k_mask_weight(float *__restrict__ masks, const float *__restrict__ occ, Geo g, float data_norm, int ref, int one_direction) {
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = (size_t)y * g.pitch + x;
    const float oc = occ[b * g.es + o];
    float factor = (oc == 0.0f) ? 1.0f : 0.0f;
    factor = (1 + factor) * data_norm;                                                   // :297
    const float backward = __fdiv_rn((oc >= 0.0f) ? 1.0f : 0.0f, factor);                // :302
    const float forward = __fdiv_rn((oc <= 0.0f) ? 1.0f : 0.0f, factor);                 // :303
    for (int s = one_direction ? ref : 0; s < 2 * ref; s++) {
        float *m = masks + b * g.es + s * g.pl + o;
        *m = (s < ref) ? 1.0f * backward * (*m) : 1.0f * forward * (*m);                 // :314,316
    }
}
void launch_mask_weight(sfa_ctx *c, const Geo &g, float *masks, const float *occ, float data_norm, int ref, int one_direction) {
    hipLaunchKernelGGL(k_mask_weight, grid2d(g), block2d(), 0, c->stream, masks, occ, g, data_norm, ref, one_direction);
}

__global__ void k_fill(float *p, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) p[i] = v;
}
void launch_fill(sfa_ctx *c, float *p, size_t n, float v) {
    if (!n) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, c->stream, p, n, v);
}

// ---------------------------------------------------------------------------------------------------
// K6 data-term assembly: all active add_data_and_match / add_data_and_match_ref terms
// (variational_aux_mt.cpp:166-403, 408-634; call order of variational_mt.cpp:343-361) accumulated in
// registers in the reference's order, then sub_laplacian(b1,uu), (b2,vv) (:364-365), one store per plane.
// ---------------------------------------------------------------------------------------------------
#define DATANORM (0.1f * 0.1f)   // variational_aux_mt.h:23


// Division by a denominator that two quotients share (r^2 / n and t / n in the normalised data terms): the hardware's own correctly-rounded chain
// (v_rcp, one Newton step, quotient, two residual corrections -- what __fdiv_rn expands to between its v_div_scale and v_div_fixup instructions) with the
// refined reciprocal computed once per denominator.
struct Recip { float b, r; };
__device__ __forceinline__ Recip recip_of(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return Recip{b, r};
}
__device__ __forceinline__ float div_by(float a, const Recip &d) {
    float q = a * d.r;
    float e = __builtin_fmaf(-d.b, q, a);
    q = __builtin_fmaf(e, d.r, q);
    e = __builtin_fmaf(-d.b, q, a);
    return __builtin_fmaf(e, d.r, q);
}
// When is that chain the IEEE quotient?  v_div_scale_f32 leaves numerator a and denominator b alone (and v_div_fmas / v_div_fixup are then a plain fma and
// the identity) unless: a or b is zero, exponent(a) - exponent(b) >= 96, b or 1 / b or a / b is denormal, or the biased exponent of a is <= 23 (CDNA3 ISA,
// V_DIV_SCALE_F32).  The guards below admit   b in [2^-27, 2^33)   (the data terms' denominators are sums of squares + 0.01: only the upper bound can fail) and
// a = +0  or  a in [2^-87, 2^53)   (for a = +0 the chain returns +0 like the division): exponent differences stay within (-121, 81).  The squared residuals
// are checked per pixel; the second numerator, the weight t = mask weight * (rho delta / 3 or rho gamma / 3) * psi'(s), is in range whenever the scalars are
// (launch_assemble_images, modified L1: hd, hg in [2^-16, 2^16], data_norm in [2^-8, 2^8], eps in [2^-20, 2^20]; the mask weights are 0, 1, 1 / data_norm or 1 / (2 data_norm),
// s < 6 * 2^53 / 0.01 by the first guard, so t = +0 or 2^-58 < t < 2^43; Lorentzian: hd, hg >= 2^-12 and eps in [2^-10, 2^10] give 2^-85 < t < 2^43) -- AssembleArgs::chain_ok, evaluated on the host.  A wave in which any live lane fails a guard takes the __fdiv_rn path for that group of quotients (wave-uniform branch): same bits either way,
// checked on the GPU over the boundary cases (tests/test_gpu_parity.py::test_shared_reciprocal_division_is_ieee).  SFA_EXACT_DIV_ONLY (build flag): always the slow path.
__device__ __forceinline__ bool wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }
__device__ __forceinline__ unsigned num_key(float a) { return __float_as_uint(a) - 1u; }                 // +0 -> 0xffffffff, otherwise monotone in a >= 0
constexpr unsigned kNumLo = 0x14000000u - 1u;                                                            // key of 2^-87 (biased exponent 40)
constexpr float kNumHi = 9007199254740992.0f, kDenHi = 8589934592.0f;                                    // 2^53, 2^33
__device__ __forceinline__ bool num_ok(float a) { return (int)__float_as_uint(a) >= 0 && num_key(a) >= kNumLo && a < kNumHi; }     // any float: negative values and -0 fail
__device__ __forceinline__ bool num_ok3(float a, float b, float c) {
    const unsigned ka = num_key(a), kb = num_key(b), kc = num_key(c);
    return min(ka, min(kb, kc)) >= kNumLo && fmaxf(a, fmaxf(b, c)) < kNumHi;
}
__device__ __forceinline__ bool den_ok3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)) < kDenHi; }
// SHDIV: a template parameter of the term functions -- on in the cfg-default instance of the fused kernel only (the generic instance is at its register cap:
// the reciprocals kept across the penalty evaluation made it spill)
#ifdef SFA_EXACT_DIV_ONLY
#define SFA_FAST_DIV(cond) false
#else
#define SFA_FAST_DIV(cond) (SHDIV && chain && wave_all(cond))
#endif

// test hook (sfa_division_chain): per element the shared-reciprocal quotient, the __fdiv_rn quotient and whether the guards admit the pair
__global__ void k_division_chain(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ q_chain, float *__restrict__ q_exact,
                                 unsigned char *__restrict__ admitted, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = a[i], y = b[i];
        q_chain[i] = div_by(x, recip_of(y));
        q_exact[i] = __fdiv_rn(x, y);
        admitted[i] = (num_ok(x) && den_ok3(y, y, y) && y >= 7.450580596923828125e-9f) ? 1 : 0;     // y >= 2^-27: the data terms' denominators are >= 0.01 by construction
    }
}
void launch_division_chain(sfa_ctx *c, const float *a, const float *b, float *q_chain, float *q_exact, unsigned char *admitted, size_t n) {
    hipLaunchKernelGGL(k_division_chain, dim3(2048), dim3(256), 0, c->stream, a, b, q_chain, q_exact, admitted, n);
}

struct Acc { float a11, a12, a22, b1, b2; };

struct Px {   // per-pixel inputs of one term
    float wk[3], ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
};

// ZUV: du = dv = 0 (first inner iteration): the flow products in the residuals are +-0 and only the squares of the residuals are used,
// so r = wk * Iz (etc.) gives the same bits as the full expression
template <bool ZUV = false, bool SHDIV = false>
__device__ __forceinline__ void term_succ(Acc &A, const Px &p, float m, float u, float v, float hd, float hg, float s, int dt_norm,
                                          const PenaltyDev &color, const PenaltyDev &grad, bool chain = false) {
    const float factor = s, factorp1 = s + 1;
    if (hd) {                                                                            // :189
        float r[3], tx[3], ty[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            r[k] = ZUV ? p.wk[k] * p.iz[k]
                       : p.wk[k] * (p.iz[k] + p.ix[k] * factor * u + p.iy[k] * factor * v - p.ix[k] * factorp1 * u - p.iy[k] * factorp1 * v);   // :190-192
            tx[k] = factor * p.ix[k] - factorp1 * p.ix[k];                               // :229-234
            ty[k] = factor * p.iy[k] - factorp1 * p.iy[k];
        }
        if (!dt_norm) {
            const float t = m * hd * psi_vec(color, r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);   // :196
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float t2 = t * p.wk[k];
                A.a11 += t2 * tx[k] * tx[k];
                A.a12 += t2 * tx[k] * ty[k];
                A.a22 += t2 * ty[k] * ty[k];
                A.b1 -= t2 * p.iz[k] * tx[k];
                A.b2 -= t2 * p.iz[k] * ty[k];
            }
        } else {
            float n[3];
#pragma unroll
            for (int k = 0; k < 3; k++) n[k] = tx[k] * tx[k] + ty[k] * ty[k] + DATANORM;   // :236-238
            const float a0 = r[0] * r[0], a1 = r[1] * r[1], a2 = r[2] * r[2];
            const bool f1 = SFA_FAST_DIV(num_ok3(a0, a1, a2) && den_ok3(n[0], n[1], n[2]));
            Recip d[3];
            float q0, q1, q2;
            if (f1) {
#pragma unroll
                for (int k = 0; k < 3; k++) d[k] = recip_of(n[k]);
                q0 = div_by(a0, d[0]); q1 = div_by(a1, d[1]); q2 = div_by(a2, d[2]);
            } else { q0 = __fdiv_rn(a0, n[0]); q1 = __fdiv_rn(a1, n[1]); q2 = __fdiv_rn(a2, n[2]); }
            const float t = m * hd * psi_vec(color, q0 + q1 + q2);                       // :240
            const bool f2 = f1;                                                         // t is in range by the launcher's parameter check (AssembleArgs::chain_ok)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float tk = f2 ? div_by(t, d[k]) : __fdiv_rn(t, n[k]);
                tk = tk * p.wk[k];
                A.a11 += tk * tx[k] * tx[k];                                             // :246-250
                A.a12 += tk * tx[k] * ty[k];
                A.a22 += tk * ty[k] * ty[k];
                A.b1 -= tk * p.iz[k] * tx[k];
                A.b2 -= tk * p.iz[k] * ty[k];
            }
        }
    }
    float r[6], X[3], Y[3], Z[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {                                                        // :269-276, 316-324
        r[2 * k] = ZUV ? p.wk[k] * p.ixz[k]
                       : p.wk[k] * (p.ixz[k] + p.ixx[k] * factor * u + p.ixy[k] * factor * v - p.ixx[k] * factorp1 * u - p.ixy[k] * factorp1 * v);
        r[2 * k + 1] = ZUV ? p.wk[k] * p.iyz[k]
                           : p.wk[k] * (p.iyz[k] + p.ixy[k] * factor * u + p.iyy[k] * factor * v - p.ixy[k] * factorp1 * u - p.iyy[k] * factorp1 * v);
        X[k] = factor * p.ixx[k] - factorp1 * p.ixx[k];
        Y[k] = factor * p.iyy[k] - factorp1 * p.iyy[k];
        Z[k] = factor * p.ixy[k] - factorp1 * p.ixy[k];
    }
    if (!dt_norm) {
        const float t = m * hg * psi_vec(grad, r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);   // :280
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float t2 = t * p.wk[k];
            A.a11 += t2 * X[k] * X[k] + t2 * Z[k] * Z[k];                                // :287-291
            A.a12 += t2 * X[k] * Z[k] + t2 * Z[k] * Y[k];
            A.a22 += t2 * Y[k] * Y[k] + t2 * Z[k] * Z[k];
            A.b1 -= t2 * p.ixz[k] * X[k] + t2 * p.iyz[k] * Z[k];
            A.b2 -= t2 * p.iyz[k] * Y[k] + t2 * p.ixz[k] * Z[k];
        }
    } else {
        float n[6];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            n[2 * k] = X[k] * X[k] + Z[k] * Z[k] + DATANORM;                             // :326-331
            n[2 * k + 1] = Y[k] * Y[k] + Z[k] * Z[k] + DATANORM;
        }
        float a[6], q[6];
#pragma unroll
        for (int k = 0; k < 6; k++) a[k] = r[k] * r[k];
        const bool f1 = SFA_FAST_DIV(num_ok3(a[0], a[1], a[2]) && num_ok3(a[3], a[4], a[5]) && den_ok3(n[0], n[1], n[2]) && den_ok3(n[3], n[4], n[5]));
        Recip d[6];
        if (f1) {
#pragma unroll
            for (int k = 0; k < 6; k++) { d[k] = recip_of(n[k]); q[k] = div_by(a[k], d[k]); }
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) q[k] = __fdiv_rn(a[k], n[k]);
        }
        const float t = m * hg * psi_vec(grad, q[0] + q[1] + q[2] + q[3] + q[4] + q[5]);   // :333
        const bool f2 = f1;                                                         // t is in range by the launcher's parameter check (AssembleArgs::chain_ok)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float ta = f2 ? div_by(t, d[2 * k]) : __fdiv_rn(t, n[2 * k]), tb = f2 ? div_by(t, d[2 * k + 1]) : __fdiv_rn(t, n[2 * k + 1]);
            ta = ta * p.wk[k];
            tb = tb * p.wk[k];
            A.a11 += ta * X[k] * X[k] + tb * Z[k] * Z[k];                                // :343-347
            A.a12 += ta * X[k] * Z[k] + tb * Z[k] * Y[k];
            A.a22 += tb * Y[k] * Y[k] + ta * Z[k] * Z[k];
            A.b1 -= ta * p.ixz[k] * X[k] + tb * p.iyz[k] * Z[k];
            A.b2 -= tb * p.iyz[k] * Y[k] + ta * p.ixz[k] * Z[k];
        }
    }
}

template <bool ZUV = false, bool SHDIV = false>
__device__ __forceinline__ void term_ref(Acc &A, const Px &p, float m, float u, float v, float hd, float hg, float s, int dt_norm,
                                         const PenaltyDev &color, const PenaltyDev &grad, bool chain = false) {
    float factor = s;
    const float factorsq = factor * factor;                                              // :417
    if (s >= 0) factor = -factor;                                                        // :424-425
    if (hd) {                                                                            // :439
        float r[3];
#pragma unroll
        for (int k = 0; k < 3; k++) r[k] = ZUV ? p.wk[k] * p.iz[k] : p.wk[k] * (p.iz[k] + p.ix[k] * factor * u + p.iy[k] * factor * v);   // :441-443
        if (!dt_norm) {
            float t = m * hd * psi_vec(color, __fdiv_rn(r[0] * r[0], factorsq) + __fdiv_rn(r[1] * r[1], factorsq) + __fdiv_rn(r[2] * r[2], factorsq));   // :447
            t = __fdiv_rn(t, factorsq);
            float t2;
            t2 = t * p.wk[0] * factor;                                                   // :450-456
            A.b1 -= t2 * p.iz[0] * p.ix[0];
            A.b2 -= t2 * p.iz[0] * p.iy[0];
            t2 = t2 * factor;
            A.a11 += t2 * p.ix[0] * p.ix[0];
            A.a12 += t2 * p.ix[0] * p.iy[0];
            A.a22 += t2 * p.iy[0] * p.iy[0];
            t2 = t * factor * p.wk[1];                                                   // :458-464
            A.b1 -= t2 * p.iz[1] * p.ix[1];
            A.b2 -= t2 * p.iz[1] * p.iy[1];
            t2 = t2 * factor;
            A.a11 += t2 * p.ix[1] * p.ix[1];
            A.a12 += t2 * p.ix[1] * p.iy[1];
            A.a22 += t2 * p.iy[1] * p.iy[1];
            t2 = t * factor * p.wk[2];                                                   // :466-472 (sic :469)
            A.b1 -= t2 * p.iz[2] * p.ix[2];
            A.b2 -= t2 * p.iz[2] * p.iy[2];
            t2 = t * factor;
            A.a11 += t2 * p.ix[2] * p.ix[2];
            A.a12 += t2 * p.ix[2] * p.iy[2];
            A.a22 += t2 * p.iy[2] * p.iy[2];
        } else {
            float n[3];
#pragma unroll
            for (int k = 0; k < 3; k++) n[k] = factorsq * p.ix[k] * p.ix[k] + factorsq * p.iy[k] * p.iy[k] + DATANORM;   // :475-477
            const float a0 = r[0] * r[0], a1 = r[1] * r[1], a2 = r[2] * r[2];
            const bool f1 = SFA_FAST_DIV(num_ok3(a0, a1, a2) && den_ok3(n[0], n[1], n[2]));
            Recip d[3];
            float q0, q1, q2;
            if (f1) {
#pragma unroll
                for (int k = 0; k < 3; k++) d[k] = recip_of(n[k]);
                q0 = div_by(a0, d[0]); q1 = div_by(a1, d[1]); q2 = div_by(a2, d[2]);
            } else { q0 = __fdiv_rn(a0, n[0]); q1 = __fdiv_rn(a1, n[1]); q2 = __fdiv_rn(a2, n[2]); }
            const float t = m * hd * psi_vec(color, q0 + q1 + q2);                       // :479
            const bool f2 = f1;                                                         // t is in range by the launcher's parameter check (AssembleArgs::chain_ok)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float tk = f2 ? div_by(t, d[k]) : __fdiv_rn(t, n[k]);
                tk = tk * p.wk[k] * factor;                                              // :484-490
                A.b1 -= tk * p.iz[k] * p.ix[k];
                A.b2 -= tk * p.iz[k] * p.iy[k];
                tk = tk * factor;
                A.a11 += tk * p.ix[k] * p.ix[k];
                A.a12 += tk * p.ix[k] * p.iy[k];
                A.a22 += tk * p.iy[k] * p.iy[k];
            }
        }
    }
    float r[6];
#pragma unroll
    for (int k = 0; k < 3; k++) {                                                        // :511-516
        r[2 * k] = ZUV ? p.wk[k] * p.ixz[k] : p.wk[k] * (p.ixz[k] + p.ixx[k] * factor * u + p.ixy[k] * factor * v);
        r[2 * k + 1] = ZUV ? p.wk[k] * p.iyz[k] : p.wk[k] * (p.iyz[k] + p.ixy[k] * factor * u + p.iyy[k] * factor * v);
    }
    if (!dt_norm) {
        float t = m * hg * psi_vec(grad, __fdiv_rn(r[0] * r[0], factorsq) + __fdiv_rn(r[1] * r[1], factorsq) + __fdiv_rn(r[2] * r[2], factorsq) +
                                             __fdiv_rn(r[3] * r[3], factorsq) + __fdiv_rn(r[4] * r[4], factorsq) + __fdiv_rn(r[5] * r[5], factorsq));   // :520-521
        t = __fdiv_rn(t, factorsq);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float t2 = t * p.wk[k] * factor;                                             // :524-526
            A.b1 -= t2 * p.ixx[k] * p.ixz[k] + t2 * p.ixy[k] * p.iyz[k];
            A.b2 -= t2 * p.iyy[k] * p.iyz[k] + t2 * p.ixy[k] * p.ixz[k];
            t2 = t2 * factor;
            if (k == 0) {                                                                // :528-530 (sic: extra factorsq)
                A.a11 += t2 * factorsq * p.ixx[k] * p.ixx[k] + t2 * factorsq * p.ixy[k] * p.ixy[k];
                A.a12 += t2 * factorsq * p.ixx[k] * p.ixy[k] + t2 * factorsq * p.ixy[k] * p.iyy[k];
                A.a22 += t2 * factorsq * p.iyy[k] * p.iyy[k] + t2 * factorsq * p.ixy[k] * p.ixy[k];
            } else {                                                                     // :536-538
                A.a11 += t2 * p.ixx[k] * p.ixx[k] + t2 * p.ixy[k] * p.ixy[k];
                A.a12 += t2 * p.ixx[k] * p.ixy[k] + t2 * p.ixy[k] * p.iyy[k];
                A.a22 += t2 * p.iyy[k] * p.iyy[k] + t2 * p.ixy[k] * p.ixy[k];
            }
        }
    } else {
        float n[6];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            n[2 * k] = factorsq * p.ixx[k] * p.ixx[k] + factorsq * p.ixy[k] * p.ixy[k] + DATANORM;       // :549-554
            n[2 * k + 1] = factorsq * p.iyy[k] * p.iyy[k] + factorsq * p.ixy[k] * p.ixy[k] + DATANORM;
        }
        float a[6], q[6];
#pragma unroll
        for (int k = 0; k < 6; k++) a[k] = r[k] * r[k];
        const bool f1 = SFA_FAST_DIV(num_ok3(a[0], a[1], a[2]) && num_ok3(a[3], a[4], a[5]) && den_ok3(n[0], n[1], n[2]) && den_ok3(n[3], n[4], n[5]));
        Recip d[6];
        if (f1) {
#pragma unroll
            for (int k = 0; k < 6; k++) { d[k] = recip_of(n[k]); q[k] = div_by(a[k], d[k]); }
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) q[k] = __fdiv_rn(a[k], n[k]);
        }
        const float t = m * hg * psi_vec(grad, q[0] + q[1] + q[2] + q[3] + q[4] + q[5]);   // :556
        const bool f2 = f1;                                                         // t is in range by the launcher's parameter check (AssembleArgs::chain_ok)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float ta = f2 ? div_by(t, d[2 * k]) : __fdiv_rn(t, n[2 * k]), tb = f2 ? div_by(t, d[2 * k + 1]) : __fdiv_rn(t, n[2 * k + 1]);
            ta = ta * p.wk[k] * factor;                                                  // :564-572
            tb = tb * p.wk[k] * factor;
            A.b1 -= ta * p.ixx[k] * p.ixz[k] + tb * p.ixy[k] * p.iyz[k];
            A.b2 -= tb * p.iyy[k] * p.iyz[k] + ta * p.ixy[k] * p.ixz[k];
            ta = ta * factor;
            tb = tb * factor;
            A.a11 += ta * p.ixx[k] * p.ixx[k] + tb * p.ixy[k] * p.ixy[k];
            A.a12 += ta * p.ixx[k] * p.ixy[k] + tb * p.ixy[k] * p.iyy[k];
            A.a22 += tb * p.iyy[k] * p.iyy[k] + ta * p.ixy[k] * p.ixy[k];
        }
    }
}

__global__ void __launch_bounds__(BX *BY) k_assemble(AssembleArgs a, const float *__restrict__ base, float *__restrict__ a11, float *__restrict__ a12,
                                                      float *__restrict__ a22, float *__restrict__ b1, float *__restrict__ b2, const float *__restrict__ du,
                                                      const float *__restrict__ dv, const float *__restrict__ uu, const float *__restrict__ vv,
                                                      const float *__restrict__ sh, const float *__restrict__ sv, Geo g) {
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = (size_t)y * g.pitch + x;
    const long eb = b * g.es;
    Acc A;
    if (a.accumulate) { A.a11 = a11[eb + o]; A.a12 = a12[eb + o]; A.a22 = a22[eb + o]; A.b1 = b1[eb + o]; A.b2 = b2[eb + o]; }
    else { A.a11 = A.a12 = A.a22 = A.b1 = A.b2 = 0.0f; }                                  // image_erase, variational_mt.cpp:336-340
    const float u = du[eb + o], v = dv[eb + o];
    Px p;
    if (a.chw) {
        // the reference walks the LEVEL-0 weight planes with this level's linear index (variational_aux_mt.cpp:177,366-371)
        const unsigned lin = (unsigned)y * (unsigned)a.lstride + (unsigned)x;          // < 2^31: checked where the weights are attached
        const unsigned r0 = lin / (unsigned)a.chw_stride0, c0 = lin % (unsigned)a.chw_stride0;
        const float *cw = a.chw + b * a.chw_es + r0 * a.chw_pitch + c0;
        p.wk[0] = cw[0]; p.wk[1] = cw[a.chw_pl]; p.wk[2] = cw[2 * a.chw_pl];
    } else { p.wk[0] = p.wk[1] = p.wk[2] = 1.0f; }
    for (int t = 0; t < a.n; t++) {
        const Term &T = a.t[t];
        const float *S = base + eb + T.stack_off + o;
        const float m = base[eb + T.mask_off + o];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            p.ix[k] = S[(0 * 3 + k) * g.pl]; p.iy[k] = S[(1 * 3 + k) * g.pl]; p.iz[k] = S[(2 * 3 + k) * g.pl];
            p.ixx[k] = S[(3 * 3 + k) * g.pl]; p.ixy[k] = S[(4 * 3 + k) * g.pl]; p.iyy[k] = S[(5 * 3 + k) * g.pl];
            p.ixz[k] = S[(6 * 3 + k) * g.pl]; p.iyz[k] = S[(7 * 3 + k) * g.pl];
        }
        if (T.is_ref) term_ref(A, p, m, u, v, T.hd, T.hg, T.s, a.dt_norm, a.color, a.grad);
        else          term_succ(A, p, m, u, v, T.hd, T.hg, T.s, a.dt_norm, a.color, a.grad);
    }
    if (a.do_laplacian) {                                                                // variational_mt.cpp:364-365
        PlaneAcc U{uu + eb, g.pitch}, V{vv + eb, g.pitch}, H{sh + eb, g.pitch}, W{sv + eb, g.pitch};
        A.b1 = laplacian_gather(A.b1, U, H, W, x, y, g.w, g.h);
        A.b2 = laplacian_gather(A.b2, V, H, W, x, y, g.w, g.h);
    }
    a11[eb + o] = A.a11; a12[eb + o] = A.a12; a22[eb + o] = A.a22; b1[eb + o] = A.b1; b2[eb + o] = A.b2;
}
void launch_assemble(sfa_ctx *c, const Geo &g, const AssembleArgs &a, const float *base, float *a11, float *a12, float *a22, float *b1, float *b2,
                     const float *du, const float *dv, const float *uu, const float *vv, const float *sh, const float *sv) {
    hipLaunchKernelGGL(k_assemble, grid2d(g), block2d(), 0, c->stream, a, base, a11, a12, a22, b1, b2, du, dv, uu, vv, sh, sv, g);
}

// ---------------------------------------------------------------------------------------------------
// K6 fused: the derivative stack of every term is formed in LDS from its image pair (stage 1/2 of k_deriv_stack)
// and consumed on the spot by the term arithmetic above -- the 24-plane stacks never exist in HBM.  One block = a
// 64 x TY pixel tile of one element; per term M (halo 4) of the three channels is staged, then Ix/Iy/Iz (halo 2), then
// each thread runs the term for its pixel(s): three barriers per term.  Mask weights (variational_mt.cpp:293-320) are applied
// on the fly from the occlusion plane.  Same operations in the same order as the unfused kernels: bit-identical.
// ---------------------------------------------------------------------------------------------------
constexpr int AT_W1 = DT_X + 4;                                      // halo-2 planes
struct Tile1Acc {     // a staged plane of DT_W columns whose first entry is pixel (x0, y0)
    const float *t; int x0, y0;
    __device__ __forceinline__ float operator()(int x, int y) const { return t[(y - y0) * DT_W + (x - x0)]; }
};
struct Tile2Acc {
    const float *t; int x0, y0;      // global coordinates of the halo-2 origin
    __device__ __forceinline__ float operator()(int x, int y) const { return t[(y - y0) * AT_W1 + (x - x0)]; }
};
// interior taps at fixed LDS offsets (the generic accessors cost more in index arithmetic than the filter itself)
template <int W>
__device__ __forceinline__ float d5x_in(const float *t, int c) { return tap5(t[c - 2], t[c - 1], t[c], t[c + 1], t[c + 2]); }
template <int W>
__device__ __forceinline__ float d5y_in(const float *t, int c) { return tap5(t[c - 2 * W], t[c - W], t[c], t[c + W], t[c + 2 * W]); }

// Debug instrumentation (off in the product; tools/asm_timing.py): wave cycles of k_assemble_images by phase.  AT_MARK(i) charges the time since the
// previous mark to slot i: 0 per-pixel term arithmetic, 1 wait for the staged planes, 2 image DMA issue, 3 its vmcnt wait, 4 barrier behind it,
// 5 in-place conversion, 6 barrier before stage 1, 7 stage 1, 8 barrier behind it, 9 barrier before the epilogue, 10 epilogue per pixel,
// 11 barrier behind the operand tile, 12 diagonal stores, 13 prologue.
#ifdef SFA_ASM_TIMING
__device__ unsigned long long g_asm_timing[16];
} // namespace sfa
extern "C" int sfa_debug_asm_timing(unsigned long long *out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(sfa::g_asm_timing), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sfa::g_asm_timing), sizeof(unsigned long long) * 16);
}
namespace sfa {
// the stamp is fenced on both sides: nothing of the phase before it may sink below it, nothing of the phase behind it may rise above it (without the fences the
// scheduler moved the per-pixel arithmetic across the marks and the table charged it to the barriers)
#define AT_MARK(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); at_acc[i] += n_ - at_t; at_t = n_; } while (0)
#else
#define AT_MARK(i)
#endif
// LDS-DMA issued from inline assembly (global_load_lds_dwordx4: every lane's 16 bytes land at M0 + 16 * lane).  The builtin form made the compiler's wait-count
// pass put an `s_waitcnt vmcnt(0)` INSIDE the issue loops (a register of the address arithmetic had a load pending), so every wave-instruction waited for the
// one before it: one exposed memory round trip per piece, 30 % of the kernel's wave time (round-4 ISA reading).  The assembly form is invisible to that pass:
// nothing waits until the explicit `dma_wait` below.  Untracked DMA is safe next to tracked loads: vmcnt retires in order, so a compiler-made wait for an older
// load is unaffected and one for a younger load over-waits.  M0 is in the clobber list: the compiler may use it itself (dynamic register indexing, its own LDS-DMA).
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p; }
// (clang warns that M0 is a reserved register: it is -- never allocated --, and naming it is exactly the point: the statement redefines it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// the same with a wave-uniform 64-bit base and a 32-bit byte offset per lane (no 64-bit address arithmetic in the vector unit)
#if defined(SFA_ASM_DMA_NT) && SFA_ASM_DMA_NT      // what-if (round 6): the image quads as non-temporal loads (they stream through the L2 the solver's operands sit in)
#define SFA_DMA_NT_ " nt"
#else
#define SFA_DMA_NT_ ""
#endif
__device__ __forceinline__ void dma16s(const float *sbase, unsigned voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" SFA_DMA_NT_ ::"v"(voff), "s"(sbase), "s"(lds_byte) : "memory", "m0");
}
__device__ __forceinline__ void dma16(const float *gsrc, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte) : "memory", "m0");
}
#pragma clang diagnostic pop

// TY rows x 64 columns per block of NT threads (NT/64 rows in flight, TY*64/NT pixels per thread).
// Column borders: the staged planes carry REPLICATED columns outside the image, so the clamped taps of image.c:501-516
// become fixed LDS offsets.  Row borders use folded coefficients (different expressions, image.c:433-457): rows are
// wave-uniform here, so that is a scalar branch.
// FAST = 1: the cfg's defaults as compile-time constants (normalised data term, modified-L1 penalties, no channel weights): the same arithmetic with the
// penalty switch and the weight planes folded away -- fewer live registers, no spills (a spill reload is a vector-memory instruction, and a wait for it is a
// wait for every DMA issued before it).  FAST = 2: the same with Lorentzian penalties for both data terms (BASELINE config 5).  FAST = 0: everything at run time.
// timing-only what-if builds (-DSFA_X_AI=bits, wrong results by construction, never shipped; tools/README.md): 1 no image DMA, 2 no conversion pass,
// 4 no stage 1, 8 single reads instead of the second-derivative taps, 16 no term arithmetic, 32 no epilogue DMA, 64 no operand stores, 128 no prologue / mask loads
#ifndef SFA_X_AI
#define SFA_X_AI 0
#endif
// wave priority of the staging phases (DMA issue, conversion, stage 1, epilogue) against the per-pixel arithmetic (priority 0).  The blocks of a CU (two then, three now) drift into
// step (both stage, then both compute); with the short staging phases preferred a block gets through them while the other one computes: 1174 -> 1146 us per launch
// (priorities 1, 2, 3 measured alike; 0 = off)
#ifndef SFA_PRIO_STAGE
#define SFA_PRIO_STAGE 1
#endif
#ifndef SFA_TILE_FAST
#define SFA_TILE_FAST 1
#endif
#define SFA_PRIO(p) do { if (SFA_PRIO_STAGE) __builtin_amdgcn_s_setprio(p); } while (0)
// Terms staged per round: 1 (48 KB of LDS: three blocks per CU = six waves per SIMD, for which the kernel stays within 80 registers -- the default since the
// tap addresses stopped occupying 45 of them) or 2 (rounds 2-4: a pair shares one DMA wait and one barrier and an image both terms use is fetched once, but
// 76 KB of LDS allow two blocks per CU only).  Measured, 64 windows, mean of the levels: 1 083 -> 990 us per launch
#ifndef SFA_ASM_PAIR
#define SFA_ASM_PAIR 0
#endif
#ifndef SFA_ASM_DMA_BARRIER
#define SFA_ASM_DMA_BARRIER 1
#endif
#ifndef SFA_ASM_TY
#define SFA_ASM_TY 8
#endif
#ifndef SFA_ASM_STAGGER
#define SFA_ASM_STAGGER 0
#endif
#ifndef SFA_ASM_STAGGER_MODE
#define SFA_ASM_STAGGER_MODE 0
#endif
constexpr int kAsmPair = SFA_ASM_PAIR, kAsmMinWaves = SFA_ASM_PAIR ? 4 : 6, kAsmTY = SFA_ASM_TY, kAsmNT = 64 * SFA_ASM_TY;      // tile rows, threads (one row per wave)
struct XcdTiles { int nx, ny, chunk; };      // tile columns, tile rows, ceil(tiles of the launch / 8)
template <int TY, int NT, int MINB, bool ZUV, int FAST, bool XT>
__global__ void __launch_bounds__(NT, MINB) k_assemble_images(AssembleArgs a, const float *__restrict__ base, float *__restrict__ a11, float *__restrict__ a12,
                                                              float *__restrict__ a22, float *__restrict__ b1, float *__restrict__ b2,
                                                              const float *__restrict__ du, const float *__restrict__ dv, const float *__restrict__ uu,
                                                              const float *__restrict__ vv, const float *__restrict__ sh, const float *__restrict__ sv,
                                                              const float *__restrict__ occ, Geo g, XcdTiles xt) {
    constexpr int TR = TY + 2 * DT_H, AT_R1 = TY + 4, NR = NT / 64, NP = TY / NR;
    // one LDS block: staged planes during the terms, the operand tile of the solver afterwards
    // Ix, Iy planes (halo 2 in y) share M's COLUMN geometry since round 4 (index j = column x0 + j, 72 wide): stage 1 then works on M's own aligned quads and reads
    // its column taps as aligned 16-byte rows (they were 2 x 8 bytes at half the LDS rate on planes shifted by two columns)
    constexpr int AW1 = DT_W;
    constexpr int NM = TR * DT_W, N1 = AT_R1 * AW1;
    // kAsmPair: terms are staged two at a time (one exposed global-load latency and one barrier less per pair); default: one term per round (three blocks per CU)
    constexpr int ZOFF = kAsmPair ? 6 * NM : 3 * NM, XOFF = 2 * ZOFF;           // start of the Iz planes, of the Ix planes
    __shared__ __attribute__((aligned(16))) float lds[XOFF + 6 * N1];
    float(*sM2)[NM] = reinterpret_cast<float(*)[NM]>(lds);                      // [2 terms][3 ch] M  = (I1+I2)/2, halo 4 (rows x DT_W)
    float(*sZ2)[NM] = reinterpret_cast<float(*)[NM]>(lds + ZOFF);               // [2 terms][3 ch] Iz = I1-I2, same geometry (aligned 16-byte rows)
    float(*sX)[N1] = reinterpret_cast<float(*)[N1]>(lds + XOFF);                // Ix, Iy of the term in work: halo 2 (rows x AW1)
    float(*sY)[N1] = reinterpret_cast<float(*)[N1]>(lds + XOFF + 3 * N1);
    // the smoothness-side operands of the epilogue (uu, vv, sh, sv with a halo of one) take the place of the M planes once the last term's Ix, Iy exist
    constexpr int GR = TY + 2, NGQ = 4 * GR * (DT_W / 4), NGF = 4 * GR * DT_W;
    static_assert(NGF <= ZOFF, "the epilogue's operand planes replace the M planes");
    static_assert(NGF + TY * (67 * 8 + 69 * 2) <= XOFF + 6 * N1, "operand tile must fit the staging block");
    static_assert(DT_W % 4 == 0 && AW1 % 4 == 0 && NM % 4 == 0 && N1 % 4 == 0, "16-byte LDS rows");
#ifdef SFA_ASM_TIMING
    unsigned long long at_acc[14] = {0}, at_t = __builtin_readcyclecounter();
#endif
    // Tile of this block.  The hardware hands consecutive workgroups of a launch to the 8 XCDs in turn (workgroup i -> XCD i % 8, each with an L2 of its
    // own), so with the plain (x, y, window) grid a tile's horizontal neighbours -- which share two of the four 128-byte lines of every staged row -- and its
    // vertical neighbours (8 of 16 staged rows) are fetched through OTHER L2s: 2.9 GB crossed the fabric per 64-window launch for 1.4 GB of unique
    // inputs.  XT: a 1-D grid whose workgroup i works on tile (i % 8) * ceil(N / 8) + i / 8 of the row-major (window, tile row, tile column) order: every XCD
    // walks a contiguous eighth of the tiles, its CUs hold neighbouring tiles at any time, and the shared lines are L2 hits.
    int bx_, by_, b_;
    if (XT) {
        const unsigned nxy = (unsigned)xt.nx * (unsigned)xt.ny;
        const unsigned j = (blockIdx.x & 7u) * (unsigned)xt.chunk + (blockIdx.x >> 3);
        if (j >= nxy * (unsigned)g.nb) return;
        b_ = (int)(j / nxy);
        const unsigned r = j - (unsigned)b_ * nxy;
        by_ = (int)(r / (unsigned)xt.nx);
        bx_ = (int)(r - (unsigned)by_ * (unsigned)xt.nx);
    } else { bx_ = blockIdx.x; by_ = blockIdx.y; b_ = blockIdx.z; }
    const int bx = __builtin_amdgcn_readfirstlane(bx_), by = __builtin_amdgcn_readfirstlane(by_), b = __builtin_amdgcn_readfirstlane(b_);
    // reset the solver's progress words and ticket -- for EVERY window of the launch: the solver that follows runs all of them, also the
    // passengers whose operands this launch leaves alone (Geo::active), and draws its tickets from window 0's block whoever is active
    if (a.op.sa && bx == 0 && by == 0) {
        for (int i = threadIdx.x; i < a.op.ntasks; i += NT) a.op.flags[(size_t)b * a.op.ntasks + i] = 0;
        if (b == 0 && threadIdx.x == 0) a.op.flags[(size_t)a.op.nb * a.op.ntasks] = 0;
    }
    if (!elem_active(g, b)) return;
#if SFA_ASM_STAGGER
    // what-if (round 6): the blocks of the launch's FIRST residency round (three per CU) start a third of a block's life apart, so that one block of a CU stages
    // while the others compute; every later block starts when an earlier one ends and inherits its phase.  SFA_ASM_STAGGER = units of ~6 400 cycles per slot
    if (XT && blockIdx.x < 8u * 32u * 3u) {
        const unsigned j = blockIdx.x >> 3;                                     // index inside the XCD
        const unsigned slot = SFA_ASM_STAGGER_MODE ? j % 3u : (j >> 5) % 3u;
        for (unsigned s_ = 0; s_ < slot * SFA_ASM_STAGGER; s_++) __builtin_amdgcn_s_sleep(100);
    }
#endif
    const int x0 = bx * DT_X - DT_H, y0 = by * TY - DT_H;                       // origin of the halo-4 tile
    // The whole staged tile (halo 4) lies inside the image -- no replicated columns, no folded border rows, no skipped rows: 84 % of the tiles at 1024x436.
    // The conversion pass and stage 1 then run without the per-item index arithmetic and checks (the item -> plane / row / quad divisions were a fifth of
    // their instructions): wave-uniform branch, same values
    const bool tile_in = SFA_TILE_FAST && x0 >= 0 && x0 + DT_W <= g.w && y0 >= 0 && y0 + TR <= g.h;
    const long eb = b * g.es;
    const int tx = threadIdx.x & 63;
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);            // one row per wave
    const int x = x0 + DT_H + tx;
    const int dt_norm = FAST ? 1 : a.dt_norm;
    PenaltyDev pcolor = a.color, pgrad = a.grad;
    if (FAST) { pcolor.id = FAST; pgrad.id = FAST; }
    Acc A[NP];
    float u[NP], v[NP], wk[NP][3], fwd[NP], bwd[NP], oc[NP];
    bool ok[NP];
    // Every per-pixel global load of the prologue is only ISSUED here; the values are first looked at behind the wait for the first image DMA, so their
    // round trip runs beside it (before: a wait of its own in front of the DMA issue, 12 % of the wave time).
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const int y = y0 + DT_H + ty + NR * k;
        ok[k] = x < g.w && y < g.h;
        A[k].a11 = A[k].a12 = A[k].a22 = A[k].b1 = A[k].b2 = 0.0f;                    // image_erase, variational_mt.cpp:336-340
        u[k] = v[k] = 0.0f; fwd[k] = bwd[k] = 0.0f; oc[k] = 0.0f;
        wk[k][0] = wk[k][1] = wk[k][2] = 1.0f;
        if (!ok[k]) continue;
        const size_t o = (size_t)y * g.pitch + x;
        if (!ZUV) { u[k] = du[eb + o]; v[k] = dv[eb + o]; }
        if (!FAST && a.chw) {                                                          // see k_assemble
            const unsigned lin = (unsigned)y * (unsigned)a.lstride + (unsigned)x;      // < 2^31: checked where the weights are attached
            const unsigned r0 = lin / (unsigned)a.chw_stride0, c0 = lin % (unsigned)a.chw_stride0;
            const float *cw = a.chw + b * a.chw_es + r0 * a.chw_pitch + c0;
            wk[k][0] = cw[0]; wk[k][1] = cw[a.chw_pl]; wk[k][2] = cw[2 * a.chw_pl];
        }
        if (!(SFA_X_AI & 128)) oc[k] = occ[eb + o];                                   // k_mask_weight
    }
    constexpr int QM = DT_W / 4, Q1 = AW1 / 4;                       // float4 quads per staged row
    // global -> LDS by DMA like the image quads: rows y-1 .. y+TY of the four planes, the 18 aligned quads that cover columns x0 .. x0+71.  Quads outside
    // the planes are skipped; the border rules of the gather never read them.
    const bool need_g = a.do_laplacian || a.op.sa;
    auto stage_epilogue_operands = [&]() {
        if (!need_g) return;
        for (int i0 = 0; i0 < NGQ; i0 += NT) {
            const int item = i0 + (int)threadIdx.x;
            const int q = item % QM, row = (item / QM) % GR, pl = item / (QM * GR);
            const int gy = y0 + DT_H - 1 + row, gx = x0 + 4 * q;
            const float *src = pl == 0 ? uu : pl == 1 ? vv : pl == 2 ? sh : sv;
            if (!(SFA_X_AI & 32) && item < NGQ && gy >= 0 && gy < g.h && gx >= 0 && gx + 3 < g.pitch && (pl >= 2 || a.do_laplacian))
                dma16(src + eb + (size_t)gy * g.pitch + gx, lds_addr(lds + 4 * (i0 + 64 * ty)));
        }
    };
    AT_MARK(13);
    if (a.n == 0) stage_epilogue_operands();
    float mk2[2][NP];                                                  // the raw warp masks of the staged pair of terms
    for (int t = 0; t < a.n; t++) {
        const Term &T = a.t[t];
        const int ub = kAsmPair ? t & 1 : 0;                       // staging buffer of this term
        float(*sM)[NM] = sM2 + 3 * ub;
        float(*sZ)[NM] = sZ2 + 3 * ub;
        if (ub == 0) {
        AT_MARK(0);
        if (t > 0) __syncthreads();                                // the staged planes are free again
        AT_MARK(1);
        SFA_PRIO(SFA_PRIO_STAGE);
        // stage 0 for this term and the next: M and Iz (halo 4) of the three channels, one float4 of a row per item; columns outside
        // the image are replicated (clamped source column), rows outside are never read
        // The raw image quads go global -> LDS by DMA (no register staging: at this kernel's VGPR budget a register pipeline spills, and the
        // plain load -> convert -> store loop exposed one HBM round trip per item).  Four image sets of three planes: I1, I2 of this term into the
        // M / Iz planes of buffer 0, those of the next term into buffer 1; a set's quad order == its LDS order, so a wave-instruction's 64 quads
        // land on 64 consecutive 16-byte slots.  An image that both terms use (the reference frame: I2 of the backward pair is I1 of the forward
        // pair) is fetched once.  A second pass converts in place.
        const int npair = kAsmPair && t + 1 < a.n ? 2 : 1;
        constexpr int NSQ = 3 * TR * QM, NSI = (NSQ + 63) / 64;        // quads / wave-instructions per image set
        const Term &T0 = a.t[t], &T1 = a.t[t + npair - 1];
        // where the next term's I1 / I2 come from: 0 this term's I1, 1 this term's I2, 2 own fetch
        const int from_a = npair == 2 ? (T1.i1_off == T0.i1_off ? 0 : T1.i1_off == T0.i2_off ? 1 : 2) : 0;
        const int from_b = npair == 2 ? (T1.i2_off == T0.i1_off ? 0 : T1.i2_off == T0.i2_off ? 1 : 2) : 0;
        const long set_src[4] = {T0.i1_off, T0.i2_off, T1.i1_off, T1.i2_off};
        const int set_dst[4] = {0, ZOFF, 3 * NM, ZOFF + 3 * NM};
        // one wave-instruction per part and set; a part's quad -> (row, channel) arithmetic is the same for every set, so it runs once per part (it was a
        // third of this phase's instructions when it ran per set), and the address is a wave-uniform base per set + one 32-bit lane offset
        const float *set_base[4];
#pragma unroll
        for (int set = 0; set < 4; set++) set_base[set] = base + eb + set_src[set];
        const bool own_a = npair == 2 && from_a == 2, own_b = npair == 2 && from_b == 2;
        for (int part = ty; part < NSI; part += NR) {
            const int k = part * 64 + tx;
            const int q = k % QM, ly = (k / QM) % TR, ch = k / (QM * TR);
            const int gy = y0 + ly, gx = x0 + 4 * q;
            if (!(SFA_X_AI & 1) && k < NSQ && gy >= 0 && gy < g.h && gx >= 0 && gx + 3 < g.w) {
                const unsigned po = 4u * ((unsigned)ch * (unsigned)g.pl + (unsigned)gy * (unsigned)g.pitch + (unsigned)gx);   // offset inside one image set: < 12 * plane bytes < 2^32 (launcher)
                const unsigned ld = lds_addr(lds + 256 * part);
                dma16s(set_base[0], po, ld + 4u * set_dst[0]);
                dma16s(set_base[1], po, ld + 4u * set_dst[1]);
                if (own_a) dma16s(set_base[2], po, ld + 4u * set_dst[2]);
                if (own_b) dma16s(set_base[3], po, ld + 4u * set_dst[3]);
            }
        }
        // the masks of the staged terms: issued BEHIND the DMA (the asm statements above are memory barriers to the compiler), looked at behind the one wait.
        // In front of it their destination registers were reused as the don't-care high half of a 64-bit multiply-add in the issue loop, and the wait-count
        // pass answered with an `s_waitcnt vmcnt(0)` inside the loop: a full memory round trip before the first DMA piece (12 % of the wave time in the phase stamps)
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const size_t o = (size_t)(y0 + DT_H + ty + NR * k) * g.pitch + x;
            mk2[0][k] = ok[k] && !(SFA_X_AI & 128) ? base[eb + T0.mask_off + o] : 0.0f;
            mk2[1][k] = npair == 2 ? (ok[k] && !(SFA_X_AI & 128) ? base[eb + T1.mask_off + o] : 0.0f) : mk2[0][k];
        }
        AT_MARK(2);
        // everything issued so far has landed behind this wait: the DMA pieces and the prologue's / the masks' loads (tied to it through the operand list,
        // so that no use of them can be scheduled, and waited for, in front of the DMA issue)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NP; k++) {
            asm volatile("" : "+v"(mk2[0][k]), "+v"(mk2[1][k]), "+v"(oc[k]));
            if (!FAST) asm volatile("" : "+v"(wk[k][0]), "+v"(wk[k][1]), "+v"(wk[k][2]));
            if (!ZUV) asm volatile("" : "+v"(u[k]), "+v"(v[k]));
        }
        AT_MARK(3);
        // Round 6, measured and NOT kept (-DSFA_ASM_DMA_BARRIER=0): the conversion pass below touches exactly the quads this wave's own DMA brought (a wave issues the
        // parts ty, ty + NR, ... and converts the items ty * 64 + tx + i * NT: the same 16-byte slots, in the tile-interior path and in the border path alike), so the
        // wave's own `s_waitcnt vmcnt(0)` above is all the conversion needs and this barrier -- one of four per term -- can go.  Bit-identical (60 parity tests) and
        // worth nothing: 110.0 / 109.3 ms per bench step without it against 109.8 / 109.5 with it, same box: the waves of a block reach the barrier in front of
        // stage 1 together either way.
        if (SFA_ASM_DMA_BARRIER || kAsmPair) __syncthreads();      // the DMA of every wave has landed
        AT_MARK(4);
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < NP; k++) {                          // k_mask_weight (variational_mt.cpp:293-320)
                float factor = (oc[k] == 0.0f) ? 1.0f : 0.0f;
                factor = (1 + factor) * a.data_norm;
                bwd[k] = __fdiv_rn((oc[k] >= 0.0f) ? 1.0f : 0.0f, factor);
                fwd[k] = __fdiv_rn((oc[k] <= 0.0f) ? 1.0f : 0.0f, factor);
            }
        }
        if (tile_in) {                                             // a staged quad's LDS slot is 16 bytes * its item number in every plane set: no index arithmetic at all
            for (int item = threadIdx.x; item < ((SFA_X_AI & 2) ? 0 : NSQ); item += NT) {
                auto quad = [&](int lds_off) { return *reinterpret_cast<const float4 *>(lds + lds_off + 4 * item); };
                auto convert = [&](const float4 &va, const float4 &vb, int lds_off) {       // Iz also on the four rows nobody reads: cheaper than knowing the row
                    *reinterpret_cast<float4 *>(lds + lds_off + 4 * item) =
                        make_float4(0.5f * (vb.x + va.x), 0.5f * (vb.y + va.y), 0.5f * (vb.z + va.z), 0.5f * (vb.w + va.w));        // variational_mt.cpp:120
                    *reinterpret_cast<float4 *>(lds + ZOFF + lds_off + 4 * item) = make_float4(va.x - vb.x, va.y - vb.y, va.z - vb.z, va.w - vb.w);   // :122
                };
                const float4 a0 = quad(0), b0 = quad(ZOFF);
                if (npair == 2) {
                    const float4 a1 = from_a == 0 ? a0 : from_a == 1 ? b0 : quad(3 * NM);
                    const float4 b1 = from_b == 0 ? a0 : from_b == 1 ? b0 : quad(ZOFF + 3 * NM);
                    convert(a1, b1, 3 * NM);
                }
                convert(a0, b0, 0);
            }
        } else
        for (int item = threadIdx.x; item < ((SFA_X_AI & 2) ? 0 : NSQ); item += NT) {     // both terms of the pair in one item: the shared image is read before it is overwritten
            const int q = item % QM, ly = (item / QM) % TR, ch = item / (QM * TR);
            const int gy = y0 + ly, gx = x0 + 4 * q;
            if (gy < 0 || gy >= g.h) continue;
            const bool interior = gx >= 0 && gx + 3 < g.w;
            const float *row = base + eb + ch * g.pl + (size_t)gy * g.pitch;
            auto fetch = [&](long src_off, int lds_off) {          // columns outside the image: replicated (clamped source column)
                if (interior) return *reinterpret_cast<const float4 *>(lds + lds_off + 4 * item);
                const float *r = row + src_off;
                return make_float4(r[clampi(gx, 0, g.w - 1)], r[clampi(gx + 1, 0, g.w - 1)], r[clampi(gx + 2, 0, g.w - 1)], r[clampi(gx + 3, 0, g.w - 1)]);
            };
            auto convert = [&](const float4 &va, const float4 &vb, int lds_off) {
                *reinterpret_cast<float4 *>(lds + lds_off + 4 * item) =
                    make_float4(0.5f * (vb.x + va.x), 0.5f * (vb.y + va.y), 0.5f * (vb.z + va.z), 0.5f * (vb.w + va.w));        // variational_mt.cpp:120
                if (ly >= 2 && ly < TR - 2)                         // Iz is read two rows around the tile only (M four: Ix, Iy exist on the halo-2 rows)
                    *reinterpret_cast<float4 *>(lds + ZOFF + lds_off + 4 * item) = make_float4(va.x - vb.x, va.y - vb.y, va.z - vb.z, va.w - vb.w);   // :122
            };
            const float4 a0 = fetch(T0.i1_off, 0), b0 = fetch(T0.i2_off, ZOFF);
            if (npair == 2) {
                const float4 a1 = from_a == 0 ? a0 : from_a == 1 ? b0 : fetch(T1.i1_off, 3 * NM);
                const float4 b1 = from_b == 0 ? a0 : from_b == 1 ? b0 : fetch(T1.i2_off, ZOFF + 3 * NM);
                convert(a1, b1, 3 * NM);
            }
            convert(a0, b0, 0);
        }
        AT_MARK(5);
        }
        AT_MARK(0);
        __syncthreads();                                           // staged planes complete / the previous term is done with Ix, Iy
        AT_MARK(6);
        SFA_PRIO(SFA_PRIO_STAGE);
        // stage 1: Ix, Iy on the halo-2 rows, one aligned quad of M's columns per item.  The row taps of the quad's four columns are the 8 values around it (the two
        // outermost quads of a row reach into the neighbouring rows' ends: their outer two columns are never read by anybody), the column taps four aligned quads
        if (tile_in) {
            // items in plane-major order (channel, row, quad): with 72 = 4 * 18 columns the quad's M index is 4 * item + 2 rows + (NM - N1) * channel and its
            // Ix / Iy index 4 * item
            static_assert(DT_W == 4 * Q1 && AW1 == DT_W && N1 == 4 * AT_R1 * Q1, "linear item numbering of stage 1");
            for (int item = threadIdx.x; item < ((SFA_X_AI & 4) ? 0 : AT_R1 * 3 * Q1); item += NT) {
                const int ch = (item >= AT_R1 * Q1 ? 1 : 0) + (item >= 2 * AT_R1 * Q1 ? 1 : 0);
                const float *Mq = sM[0] + 4 * item + 2 * DT_W + (NM - N1) * ch;
                const float2 lo = *reinterpret_cast<const float2 *>(Mq - 2), hi = *reinterpret_cast<const float2 *>(Mq + 4);
                const float4 mid = *reinterpret_cast<const float4 *>(Mq);
                const float m2[8] = {lo.x, lo.y, mid.x, mid.y, mid.z, mid.w, hi.x, hi.y};
                float X[4], Y[4];
#pragma unroll
                for (int e = 0; e < 4; e++) X[e] = tap5(m2[e], m2[e + 1], m2[e + 2], m2[e + 3], m2[e + 4]);                      // :127
                const float4 r0 = *reinterpret_cast<const float4 *>(Mq - 2 * DT_W), r1 = *reinterpret_cast<const float4 *>(Mq - DT_W);
                const float4 r3 = *reinterpret_cast<const float4 *>(Mq + DT_W), r4 = *reinterpret_cast<const float4 *>(Mq + 2 * DT_W);
                Y[0] = tap5(r0.x, r1.x, m2[2], r3.x, r4.x); Y[1] = tap5(r0.y, r1.y, m2[3], r3.y, r4.y);                         // :128
                Y[2] = tap5(r0.z, r1.z, m2[4], r3.z, r4.z); Y[3] = tap5(r0.w, r1.w, m2[5], r3.w, r4.w);
                *reinterpret_cast<float4 *>(&sX[0][4 * item]) = make_float4(X[0], X[1], X[2], X[3]);
                *reinterpret_cast<float4 *>(&sY[0][4 * item]) = make_float4(Y[0], Y[1], Y[2], Y[3]);
            }
        } else
        for (int item = threadIdx.x; item < ((SFA_X_AI & 4) ? 0 : AT_R1 * 3 * Q1); item += NT) {
            const int q = item % Q1, ch = (item / Q1) % 3, ly = item / (3 * Q1);
            const int gy = y0 + 2 + ly, gx = x0 + 4 * q;
            if (gy < 0 || gy >= g.h) continue;
            const float *M = sM[ch];
            const int c = (ly + 2) * DT_W + 4 * q;                 // M index of column gx
            const bool y_in = gy >= 2 && gy + 2 < g.h;
            float m2[8];                                            // row gy, columns gx-2 .. gx+5
            {
                const float2 lo = *reinterpret_cast<const float2 *>(M + c - 2), hi = *reinterpret_cast<const float2 *>(M + c + 4);
                const float4 mid = *reinterpret_cast<const float4 *>(M + c);
                m2[0] = lo.x; m2[1] = lo.y; m2[2] = mid.x; m2[3] = mid.y; m2[4] = mid.z; m2[5] = mid.w; m2[6] = hi.x; m2[7] = hi.y;
            }
            float X[4], Y[4];
            if (gx >= 0 && gx + 3 < g.w) {
#pragma unroll
                for (int e = 0; e < 4; e++) X[e] = tap5(m2[e], m2[e + 1], m2[e + 2], m2[e + 3], m2[e + 4]);                      // :127
            } else {                                                // X(clamp(x), y) for the columns outside the image
#pragma unroll
                for (int e = 0; e < 4; e++) X[e] = d5x_in<DT_W>(M, c + e + (clampi(gx + e, 0, g.w - 1) - (gx + e)));
            }
            if (y_in) {                                             // tap5's own order of operations, one aligned row of the quad's columns at a time  (:128)
                const float4 r0 = *reinterpret_cast<const float4 *>(M + c - 2 * DT_W), r1 = *reinterpret_cast<const float4 *>(M + c - DT_W);
                const float4 r3 = *reinterpret_cast<const float4 *>(M + c + DT_W), r4 = *reinterpret_cast<const float4 *>(M + c + 2 * DT_W);
                Y[0] = tap5(r0.x, r1.x, m2[2], r3.x, r4.x); Y[1] = tap5(r0.y, r1.y, m2[3], r3.y, r4.y);
                Y[2] = tap5(r0.z, r1.z, m2[4], r3.z, r4.z); Y[3] = tap5(r0.w, r1.w, m2[5], r3.w, r4.w);
            } else {
                const TileAcc m4{M, x0, y0};
#pragma unroll
                for (int e = 0; e < 4; e++) Y[e] = d5y(m4, clampi(gx + e, 0, g.w - 1), gy, g.h);
            }
            *reinterpret_cast<float4 *>(&sX[ch][ly * AW1 + 4 * q]) = make_float4(X[0], X[1], X[2], X[3]);
            *reinterpret_cast<float4 *>(&sY[ch][ly * AW1 + 4 * q]) = make_float4(Y[0], Y[1], Y[2], Y[3]);
        }
        AT_MARK(7);
        __syncthreads();
        AT_MARK(8);
        SFA_PRIO(0);
        // the last term's arithmetic runs beside the DMA of the epilogue's operands: no vector-memory instruction may follow until the wait behind the loop
        if (t == a.n - 1) stage_epilogue_operands();
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int y = y0 + DT_H + ty + NR * k;
            if (y >= g.h) break;
            const bool y_in = y >= 2 && y + 2 < g.h;
            // The lane's column is made opaque once per term: every tap address below is then computed inside the term loop from ONE register (the planes
            // above 64 KB -- Iy, part of Ix -- are out of reach of a 16-bit LDS offset from the block's start, and left to itself the compiler kept ~45 registers
            // of per-plane, per-clamped-row addresses of both the interior and the border path live across the whole term loop: tools/isa_live.py), and the
            // interior path's reads are offsets from a base in the middle of the block (cb: the Iz planes' start + the pixel), all below 48 KB
            int txl = tx;
            asm volatile("" : "+v"(txl));
            const int c = (ty + NR * k + 2) * AW1 + (txl + 4);    // halo-2 rows (Ix, Iy), M's columns
            const int cz = (ty + NR * k + 4) * DT_W + (txl + 4);  // halo-4 plane (Iz)
            static_assert(AW1 == DT_W, "cz = c + 2 rows");
            int cb = ZOFF + c;
            asm volatile("" : "+v"(cb));
            const float *const pX = lds + cb + (XOFF - ZOFF), *const pY = pX + 3 * N1, *const pZ = lds + cb + 3 * ub * NM + 2 * DT_W;   // sX[0] + c, sY[0] + c, sZ[0] + cz
            Px p;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) p.wk[ch] = FAST ? 1.0f : wk[k][ch];
            if (SFA_X_AI & 8) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    p.ixy[ch] = sX[ch][c + 1]; p.iyy[ch] = sY[ch][c + 1]; p.iyz[ch] = sZ[ch][cz + 1];
                    p.ix[ch] = sX[ch][c]; p.iy[ch] = sY[ch][c]; p.iz[ch] = sZ[ch][cz]; p.ixx[ch] = sX[ch][c - 1]; p.ixz[ch] = sZ[ch][cz - 1];
                }
            } else if (y_in) {
                // All 69 taps of the pixel (per channel: a row and a column of Ix and of Iz, a column of Iy) are READ first and the filters run behind a scheduling
                // fence.  Left to itself the compiler read two taps, waited, used them, read the next two (82 LDS instructions per term, nearly each with a wait
                // of its own, in a kernel with then 4 waves per SIMD): the per-pixel phase was 36 % of a wave's life, and most of that LDS latency (round-4 phase stamps).
                // Two channels' taps in flight at most (all three at once: 69 registers of taps, spills at this kernel's 128).
                float xr[3][5], xc[3][4], yc[3][5], zr[3][5], zc[3][4];
                auto read_taps = [&](int ch) {
                    const float *X = pX + ch * N1, *Y = pY + ch * N1, *Z = pZ + ch * NM;
#pragma unroll
                    for (int j = 0; j < 5; j++) xr[ch][j] = X[j - 2];
                    xc[ch][0] = X[-2 * AW1]; xc[ch][1] = X[-AW1]; xc[ch][2] = X[AW1]; xc[ch][3] = X[2 * AW1];
#pragma unroll
                    for (int j = 0; j < 5; j++) yc[ch][j] = Y[(j - 2) * AW1];
#pragma unroll
                    for (int j = 0; j < 5; j++) zr[ch][j] = Z[j - 2];
                    zc[ch][0] = Z[-2 * DT_W]; zc[ch][1] = Z[-DT_W]; zc[ch][2] = Z[DT_W]; zc[ch][3] = Z[2 * DT_W];
                };
                auto filters = [&](int ch) {
                    p.ix[ch] = xr[ch][2]; p.iy[ch] = yc[ch][2]; p.iz[ch] = zr[ch][2];
                    p.ixx[ch] = tap5(xr[ch][0], xr[ch][1], xr[ch][2], xr[ch][3], xr[ch][4]);     // :129
                    p.ixy[ch] = tap5(xc[ch][0], xc[ch][1], xr[ch][2], xc[ch][2], xc[ch][3]);     // :130
                    p.iyy[ch] = tap5(yc[ch][0], yc[ch][1], yc[ch][2], yc[ch][3], yc[ch][4]);     // :131
                    p.ixz[ch] = tap5(zr[ch][0], zr[ch][1], zr[ch][2], zr[ch][3], zr[ch][4]);     // :132
                    p.iyz[ch] = tap5(zc[ch][0], zc[ch][1], zr[ch][2], zc[ch][2], zc[ch][3]);     // :133
                };
                read_taps(0); read_taps(1);
                __builtin_amdgcn_sched_barrier(0);
                filters(0);
                __builtin_amdgcn_sched_barrier(0);
                read_taps(2);
                __builtin_amdgcn_sched_barrier(0);
                filters(1); filters(2);
            } else {
                const int xl = x0 + DT_H + txl, xc_ = xl < g.w ? xl : g.w - 1;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const Tile1Acc X{sX[ch], x0, y0 + 2}, Y{sY[ch], x0, y0 + 2};               // (DT_W wide, like M)
                    const TileAcc Z{sZ[ch], x0, y0};
                    p.ixy[ch] = d5y(X, xc_, y, g.h);
                    p.iyy[ch] = d5y(Y, xc_, y, g.h);
                    p.iyz[ch] = d5y(Z, xc_, y, g.h);
                    p.ix[ch] = sX[ch][c]; p.iy[ch] = sY[ch][c]; p.iz[ch] = sZ[ch][cz];
                    p.ixx[ch] = d5x_in<AW1>(sX[ch], c);                                // :129
                    p.ixz[ch] = d5x_in<DT_W>(sZ[ch], cz);                              // :132
                }
            }
            if (!ok[k]) continue;
            float m = mk2[ub][k];
            if (!a.one_direction || !T.backward) m = T.backward ? 1.0f * bwd[k] * m : 1.0f * fwd[k] * m;   // :314,316
            if (SFA_X_AI & 16) {
                for (int ch = 0; ch < 3; ch++) {
                    A[k].a11 += p.ix[ch] + p.ixx[ch] + m; A[k].a12 += p.iy[ch] + p.ixy[ch]; A[k].a22 += p.iz[ch] + p.iyy[ch]; A[k].b1 += p.ixz[ch]; A[k].b2 += p.iyz[ch];
                }
            } else if (T.is_ref) term_ref<ZUV, FAST != 0>(A[k], p, m, u[k], v[k], T.hd, T.hg, T.s, dt_norm, pcolor, pgrad, a.chain_ok != 0);
            else          term_succ<ZUV, FAST != 0>(A[k], p, m, u[k], v[k], T.hd, T.hg, T.s, dt_norm, pcolor, pgrad, a.chain_ok != 0);
        }
    }
    // row strides chosen for the anti-diagonal read-out below (entry 64*rl + dl of a 65-wide row puts the TY rows of a diagonal into one bank group:
    // PMC of round 2 showed 37 % of this kernel's LDS cycles in bank conflicts): with 67 float4 / 69 float2 per row the 16 lanes of a b128 pass
    // (32 of a b64 pass) land on distinct banks
    float4(*tA)[67] = reinterpret_cast<float4(*)[67]>(lds + NGF);               // [TY][67] each; live after the last barrier below
    float4(*tB)[67] = tA + TY;
    float2(*tX)[69] = reinterpret_cast<float2(*)[69]>(tB + TY);
    AT_MARK(0);
    if (need_g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                            // the DMA has landed / every thread is done with the staged planes
    }
    AT_MARK(9);
    SFA_PRIO(SFA_PRIO_STAGE);
    const Tile1Acc U{lds, x0, y0 + DT_H - 1}, V{lds + GR * DT_W, x0, y0 + DT_H - 1}, Hh{lds + 2 * GR * DT_W, x0, y0 + DT_H - 1},
        Wv{lds + 3 * GR * DT_W, x0, y0 + DT_H - 1};
#pragma unroll
    for (int k = 0; k < NP; k++) {
        if (!ok[k]) continue;
        const int yl = ty + NR * k, y = y0 + DT_H + yl;
        const size_t o = (size_t)y * g.pitch + x;
        if (a.do_laplacian) {                                                            // variational_mt.cpp:364-365
            A[k].b1 = laplacian_gather(A[k].b1, U, Hh, Wv, x, y, g.w, g.h);
            A[k].b2 = laplacian_gather(A[k].b2, V, Hh, Wv, x, y, g.w, g.h);
        }
        if (!a.op.sa) {
            a11[eb + o] = A[k].a11; a12[eb + o] = A[k].a12; a22[eb + o] = A[k].a22; b1[eb + o] = A[k].b1; b2[eb + o] = A[k].b2;
            continue;
        }
        // k_sor_prepare's per-pixel part (solver.c:101-106,159,214): neighbour weights, inverted 2x2 block
        const float hp = Hh(x, y);
        const float hl = x > 0 ? Hh(x - 1, y) : 0.0f;                                    // f1[0] = 0, solver.c:82
        const float vp = Wv(x, y);
        const float vt = y > 0 ? Wv(x, y - 1) : 0.0f;
        float dpsis = hl + hp;
        if (y > 0) dpsis = dpsis + vt;
        if (y < g.h - 1) dpsis = dpsis + vp;
        const float m12 = A[k].a12;
        const float A11 = A[k].a22 + dpsis, A22 = A[k].a11 + dpsis;                      // solver.c:102
        const float det = A11 * A22 - m12 * m12;
        const float i11 = __fdiv_rn(A11, det), i22 = __fdiv_rn(A22, det), i12 = __fdiv_rn(m12, -det);   // solver.c:104-106
        tA[yl][tx] = make_float4(i11, i12, i22, vt);
        tB[yl][tx] = make_float4(A[k].b1, A[k].b2, hp, y < g.h - 1 ? vp : 0.0f);
        tX[yl][tx] = make_float2(u[k], v[k]);
    }
    if (!a.op.sa) return;
    AT_MARK(10);
    __syncthreads();
    AT_MARK(11);
    // the tile's anti-diagonals: TY consecutive entries of the diagonal-major operand planes each
    const int c0 = x0 + DT_H, r0 = y0 + DT_H;
    for (int item = threadIdx.x; item < (DT_X + TY - 1) * TY; item += NT) {
        const int dl = item / TY, rl = item % TY, cl = dl - rl;
        if (cl < 0 || cl >= DT_X) continue;
        const int r = r0 + rl, c = c0 + cl;
        if (r >= g.h || c >= g.w) continue;
        const size_t e = (size_t)b * a.op.ent + (size_t)(c + r + a.op.G) * a.op.RP + (r + a.op.G);
        if (SFA_X_AI & 64) continue;
        const float2 xv = tX[rl][cl];
        unsigned long long xu;
        __builtin_memcpy(&xu, &xv, 8);
#if defined(SFA_ASM_NT) && SFA_ASM_NT      // what-if (round 6): the operand tile leaves as non-temporal stores
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        const float4 qa = tA[rl][cl], qb = tB[rl][cl];
        __builtin_nontemporal_store((v4f_){qa.x, qa.y, qa.z, qa.w}, reinterpret_cast<v4f_ *>(a.op.sa + e));
        __builtin_nontemporal_store((v4f_){qb.x, qb.y, qb.z, qb.w}, reinterpret_cast<v4f_ *>(a.op.sb + e));
        __builtin_nontemporal_store(xu, a.op.x + e);
#else
        a.op.sa[e] = tA[rl][cl];
        a.op.sb[e] = tB[rl][cl];
        a.op.x[e] = xu;
#endif
    }
#ifdef SFA_ASM_TIMING
    AT_MARK(12);
    // a sample of the blocks only: with every wave adding its 14 slots the atomics on 14 addresses were the kernel (0.85 s per bench step instead of 0.07)
    if ((threadIdx.x & 63) == 0 && (bx + 3 * by + 7 * b) % 61 == 0)
        for (int i = 0; i < 14; i++) atomicAdd(&g_asm_timing[i], at_acc[i]);
#endif
}
int launch_assemble_images(sfa_ctx *c, const Geo &g, const AssembleArgs &a_in, const float *base, float *a11, float *a12, float *a22, float *b1, float *b2,
                           const float *du, const float *dv, const float *uu, const float *vv, const float *sh, const float *sv, const float *occ) {
    AssembleArgs a = a_in;
    // the kernel addresses the three planes of an image set with a 32-bit byte offset (dma16s): 12 bytes x plane entries must fit (357 Mpx per plane)
    if ((unsigned long long)g.pl * 12ull >= (1ull << 32)) return set_error(c, SFA_ERR_ARG, "k_assemble_images: plane of %ld entries is beyond the kernel's 32-bit offsets", g.pl);
    const bool prof = c->profile && c->ev2_used + 2 <= c->ev2.size();
    if (prof) (void)hipEventRecord(c->ev2[c->ev2_used], c->stream);
    // the cfg's defaults (slow_flow_dataterm 1, modified-L1 penalties -- every id select_robust_function maps to the default class,
    // variational_aux_mt.cpp:909-925 --, no channel weights) take the instance with those choices folded in
    auto is_modl1 = [](int id) { return id != 0 && id != 2 && id != 3 && id != 4; };
    const bool foldable = a.dt_norm == 1 && !a.chw && !sw_given(Switches::ASSEMBLE_GENERIC);
    const int fast = !foldable ? 0 : (is_modl1(a.color.id) && is_modl1(a.grad.id)) ? 1 : (a.color.id == 2 && a.grad.id == 2) ? 2 : 0;
    {   // the shared-reciprocal divisions' precondition on the scalars (see recip_of / div_by): t = mask weight * hd (hg) * psi'(s) must be +0 or in [2^-87, 2^53).
        // modified L1: psi' = 1 / (2 sqrt(s + eps^2)) in (2^-33, 2^19] for eps in [2^-20, 2^20] and s < 2^63; Lorentzian: psi' = 1 / (2 eps^2 + s) in (2^-64, 2^19] for eps
        // in [2^-10, 2^10]; the mask weight is 0, 1, 1 / data_norm or 1 / (2 data_norm).
        auto in = [](float v, float lo, float hi) { return v >= lo && v <= hi; };
        const float elo = fast == 2 ? 1.0f / 1024 : 1.0f / 1048576, ehi = fast == 2 ? 1024.0f : 1048576.0f, hlo = fast == 2 ? 1.0f / 4096 : 1.0f / 65536;
        bool okp = fast != 0 && in(a.data_norm, 1.0f / 256, 256.0f) && in(a.color.eps, elo, ehi) && in(a.grad.eps, elo, ehi);
        for (int t = 0; t < a.n && okp; t++) okp = (a.t[t].hd == 0.0f || in(a.t[t].hd, hlo, 65536.0f)) && (a.t[t].hg == 0.0f || in(a.t[t].hg, hlo, 65536.0f));
        a.chain_ok = okp && !sw_given(Switches::EXACT_DIV) ? 1 : 0;
    }
    const dim3 grid_((g.w + DT_X - 1) / DT_X, (g.h + kAsmTY - 1) / kAsmTY, g.nb);
    // the XCD-contiguous tile order (see the kernel) from 8 workgroups per XCD on; SFA_ASM_XCD=0: the plain grid, for A/B measurements
    const bool xcd_env = sw_int(Switches::ASM_XCD, 1) != 0;
    const long ntiles = (long)grid_.x * grid_.y * grid_.z;
    const bool xcd = xcd_env && ntiles >= 64 && ntiles < (1l << 30);
    const XcdTiles xt{(int)grid_.x, (int)grid_.y, (int)((ntiles + 7) / 8)};
    const dim3 grid1_(8u * (unsigned)xt.chunk, 1, 1);
#define SFA_LAUNCH_AI(ZUV_, FAST_)                                                                                                                      \
    do {                                                                                                                                                \
        if (xcd) hipLaunchKernelGGL((k_assemble_images<kAsmTY, kAsmNT, kAsmMinWaves, ZUV_, FAST_, true>), grid1_, dim3(kAsmNT), 0, c->stream, a, base, a11, a12, a22, b1, b2, du, dv, uu, vv, sh, sv, occ, g, xt); \
        else hipLaunchKernelGGL((k_assemble_images<kAsmTY, kAsmNT, kAsmMinWaves, ZUV_, FAST_, false>), grid_, dim3(kAsmNT), 0, c->stream, a, base, a11, a12, a22, b1, b2, du, dv, uu, vv, sh, sv, occ, g, xt); \
    } while (0)
    // 64 x 8 tiles, 512 threads, <= 80 VGPRs and 48 KB of LDS (three blocks per CU; rounds 2 - mid-4: 128 VGPRs, 76 KB, two blocks).  Measured and dropped: 64 x 12 on 768 threads, 64 x 16 with two pixels per thread (209 VGPRs, one block per CU: slower),
    // 8 x 256 threads, 16 x 1024 threads (spills at its 128-register cap)
    if (a.zero_duv) { if (fast == 1) SFA_LAUNCH_AI(true, 1); else if (fast == 2) SFA_LAUNCH_AI(true, 2); else SFA_LAUNCH_AI(true, 0); }
    else            { if (fast == 1) SFA_LAUNCH_AI(false, 1); else if (fast == 2) SFA_LAUNCH_AI(false, 2); else SFA_LAUNCH_AI(false, 0); }
#undef SFA_LAUNCH_AI
    if (prof) {
        (void)hipEventRecord(c->ev2[c->ev2_used + 1], c->stream);
        c->ev2_used += 2;
        c->asm_pixel_terms += (double)g.w * g.h * g.nb * a.n;
    }
    return SFA_OK;
}

// ---------------------------------------------------------------------------------------------------
// K8 flow update + L1 change norms (variational_mt.cpp:371-402, 412-429).  The reference keeps a sequential
// fp32 running sum; here: fp64 wave __shfl reduction -> block -> one double pair per block, summed by the
// last stage in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// block of BX*BY threads = BY waves; returns the block sum in thread 0
__device__ __forceinline__ void block_sum2(double &a, double &b) {
    __shared__ double sm[2 * BY];
    a = wave_sum(a);
    b = wave_sum(b);
    const int wid = threadIdx.y, lane = threadIdx.x;
    if (lane == 0) { sm[wid] = a; sm[BY + wid] = b; }
    __syncthreads();
    if (wid == 0 && lane == 0) {
        double sa = 0, sb = 0;
        for (int i = 0; i < BY; i++) { sa += sm[i]; sb += sm[BY + i]; }
        a = sa; b = sb;
    }
}

__global__ void __launch_bounds__(BX *BY) k_update_inner(float *__restrict__ uu, float *__restrict__ vv, const float *__restrict__ wx, const float *__restrict__ wy,
                                                          const float *__restrict__ du, const float *__restrict__ dv, const float *__restrict__ odu,
                                                          const float *__restrict__ odv, double *__restrict__ partial, float *__restrict__ dfa, float *__restrict__ dfb, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x;
    double sa = 0, sb = 0;
    if (elem_active(g, b) && x < g.w)
        for (int y = blockIdx.y * BY + threadIdx.y; y < g.h; y += gridDim.y * BY) {
            const size_t o = b * g.es + (size_t)y * g.pitch + x;
            const float d = du[o], e = dv[o];
            sa += (double)fabsf(odu[o] - d);                                             // :389-393
            sb += (double)fabsf(odv[o] - e);
            if (dfa) { dfa[o] = fabsf(odu[o] - d); dfb[o] = fabsf(odv[o] - e); }         // the per-pixel terms of the norms, for k_exact_norms
            uu[o] = wx[o] + d;                                                           // :396-397
            vv[o] = wy[o] + e;
        }
    block_sum2(sa, sb);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * blk] = sa; partial[2 * blk + 1] = sb;
    }
}
// the solver's result read straight from its diagonal-major x plane (k_sor_finish folded in)
__global__ void __launch_bounds__(BX *BY) k_update_inner_x(float *__restrict__ uu, float *__restrict__ vv, const float *__restrict__ wx, const float *__restrict__ wy,
                                                            const unsigned long long *__restrict__ xs, long ent, int RP, int G, const float *__restrict__ odu,
                                                            const float *__restrict__ odv, float *__restrict__ du_out, float *__restrict__ dv_out,
                                                            double *__restrict__ partial, float *__restrict__ dfa, float *__restrict__ dfb, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x;
    double sa = 0, sb = 0;
    if (elem_active(g, b) && x < g.w)
        for (int y = blockIdx.y * BY + threadIdx.y; y < g.h; y += gridDim.y * BY) {
            const size_t o = b * g.es + (size_t)y * g.pitch + x;
            const unsigned long long xv = xs[(size_t)b * ent + (size_t)(x + y + G) * RP + (y + G)];
            const float d = __uint_as_float((unsigned)(xv & 0xffffffffu)), e = __uint_as_float((unsigned)(xv >> 32));
            const float od = odu ? odu[o] : 0.0f, oe = odv ? odv[o] : 0.0f;
            sa += (double)fabsf(od - d);                                                 // :389-393
            sb += (double)fabsf(oe - e);
            if (dfa) { dfa[o] = fabsf(od - d); dfb[o] = fabsf(oe - e); }
            uu[o] = wx[o] + d;                                                           // :396-397
            vv[o] = wy[o] + e;
            if (du_out) { du_out[o] = d; dv_out[o] = e; }
        }
    block_sum2(sa, sb);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * blk] = sa; partial[2 * blk + 1] = sb;
    }
}
// last inner iteration of an outer one, solver result still in its x plane: flow update (:396-397) and the outer change norms and
// wx <- uu (:412-429) in one pass; the inner norms are not formed (nobody reads them after the last inner iteration)
// The x plane is diagonal-major: a row-major reader would touch one 8-byte entry per 5 KB.  A block therefore takes 64x16 tiles, reads them
// along the anti-diagonals (16 consecutive entries = 128 B each, as k_sor_prepare writes them) into LDS and works row-major from there.
template <bool FUSE>
__global__ void __launch_bounds__(BX *BY) k_update_outer_x(float *__restrict__ uu, float *__restrict__ vv, float *__restrict__ wx, float *__restrict__ wy,
                                                            const unsigned long long *__restrict__ xs, long ent, int RP, int G, double *__restrict__ partial,
                                                            double *__restrict__ out, unsigned *__restrict__ done, float *__restrict__ dfa, float *__restrict__ dfb, Geo g) {
    __shared__ unsigned long long tX[16][67];                    // 67: the 16 rows of an anti-diagonal land on distinct banks (65 put them all on one: 69 % conflict cycles)
    const int b = blockIdx.z;
    const int tid = threadIdx.y * BX + threadIdx.x;
    const int c0 = blockIdx.x * 64;
    double sa = 0, sb = 0;
    if (!elem_active(g, b)) return;                              // a passenger: its result words keep their last values
    {
        for (int r0 = blockIdx.y * 16; r0 < g.h; r0 += gridDim.y * 16) {
            __syncthreads();
            for (int item = tid; item < (64 + 16 - 1) * 16; item += BX * BY) {
                const int dl = item / 16, rl = item % 16, cl = dl - rl;
                if (cl < 0 || cl >= 64) continue;
                const int r = r0 + rl, c = c0 + cl;
                if (r < g.h && c < g.w) tX[rl][cl] = xs[(size_t)b * ent + (size_t)(c + r + G) * RP + (r + G)];
            }
            __syncthreads();
            const int x = c0 + threadIdx.x;
            if (x < g.w)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int rl = threadIdx.y + 4 * k, y = r0 + rl;
                    if (y >= g.h) break;
                    const size_t o = b * g.es + (size_t)y * g.pitch + x;
                    const unsigned long long xv = tX[rl][threadIdx.x];
                    const float d = __uint_as_float((unsigned)(xv & 0xffffffffu)), e = __uint_as_float((unsigned)(xv >> 32));
                    const float ox = wx[o], oy = wy[o];
                    const float u = ox + d, v = oy + e;                                   // :396-397
                    sa += (double)fabsf(u - ox);                                         // :415-419
                    sb += (double)fabsf(v - oy);
                    if (uu) { uu[o] = u; vv[o] = v; }                                     // (null: the caller reads wx, wy in their place -- one inner iteration)
                    if (dfa) { dfa[o] = fabsf(u - ox); dfb[o] = fabsf(v - oy); }         // the per-pixel terms of the norms, for k_exact_break
#if defined(SFA_UPD_NT) && SFA_UPD_NT      // what-if (round 6): the new flow leaves as non-temporal stores
                    __builtin_nontemporal_store(u, wx + o); __builtin_nontemporal_store(v, wy + o);
#else
                    wx[o] = u; wy[o] = v;                                                 // :428-429
#endif
                }
        }
    }
    block_sum2(sa, sb);
    if (!FUSE) {
        if (tid == 0) {
            const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            partial[2 * blk] = sa; partial[2 * blk + 1] = sb;
        }
        return;
    }
    // FUSE (launches of a few windows): the window's LAST block to finish sums the window's partials -- k_reduce_partials folded in (a launch of its own per iteration:
    // 4.6 us, a quarter of what this kernel takes for a lone window), in that kernel's order: 256 strided sums, then the tree.  The partials cross XCDs, whose L2s are
    // not coherent: they are written through and read past the L2 (agent-scope atomics = sc1), and the ticket is drawn behind the stores' completion (vmcnt counts
    // stores on gfx9).  Not for large batches: 256 tickets per window on one word and a wait for the block's stores in front of each cost the 128-window launch
    // 30 % (the whole path 1.5 %).
    __shared__ unsigned s_last;
    const int per_elem = gridDim.x * gridDim.y;
    if (tid == 0) {
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        __hip_atomic_store(&partial[2 * blk], sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partial[2 * blk + 1], sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(&done[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(per_elem - 1);
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(&done[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    double ra = 0, rb = 0;
    for (int i = tid; i < per_elem; i += BX * BY) {
        ra += __hip_atomic_load(&partial[2 * ((size_t)b * per_elem + i)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rb += __hip_atomic_load(&partial[2 * ((size_t)b * per_elem + i) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static_assert(BX * BY == 256, "k_reduce_partials' order");
    double *s0 = reinterpret_cast<double *>(&tX[0][0]), *s1 = s0 + 256;          // the tile is dead: 2 x 256 doubles of its 16 x 67
    __syncthreads();
    s0[tid] = ra; s1[tid] = rb;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { s0[tid] += s0[tid + off]; s1[tid] += s1[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) { out[2 * b] = s0[0]; out[2 * b + 1] = s1[0]; }
}
__global__ void __launch_bounds__(BX *BY) k_update_outer(float *__restrict__ wx, float *__restrict__ wy, const float *__restrict__ uu, const float *__restrict__ vv,
                                                          double *__restrict__ partial, float *__restrict__ dfa, float *__restrict__ dfb, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x;
    double sa = 0, sb = 0;
    if (elem_active(g, b) && x < g.w)
        for (int y = blockIdx.y * BY + threadIdx.y; y < g.h; y += gridDim.y * BY) {
            const size_t o = b * g.es + (size_t)y * g.pitch + x;
            const float u = uu[o], v = vv[o];
            sa += (double)fabsf(u - wx[o]);                                              // :415-419
            sb += (double)fabsf(v - wy[o]);
            if (dfa) { dfa[o] = fabsf(u - wx[o]); dfb[o] = fabsf(v - wy[o]); }
            wx[o] = u;                                                                   // :428-429
            wy[o] = v;
        }
    block_sum2(sa, sb);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * blk] = sa; partial[2 * blk + 1] = sb;
    }
}
// one block per batch element sums that element's partials in a fixed order
__global__ void k_reduce_partials(const double *__restrict__ partial, int per_elem, double *__restrict__ out, WMask active,
                                  const unsigned long long *__restrict__ amask) {
    const int b = blockIdx.x;
    if (!elem_active(active, amask, b)) return;                                            // the result words of passengers keep their last values
    double sa = 0, sb = 0;
    for (int i = threadIdx.x; i < per_elem; i += 256) { sa += partial[2 * ((size_t)b * per_elem + i)]; sb += partial[2 * ((size_t)b * per_elem + i) + 1]; }
    __shared__ double s0[256], s1[256];
    s0[threadIdx.x] = sa; s1[threadIdx.x] = sb;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { s0[threadIdx.x] += s0[threadIdx.x + off]; s1[threadIdx.x] += s1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * b] = s0[0]; out[2 * b + 1] = s1[0]; }
}

// The outer break on the device (variational_mt.cpp:431-436): the norms of the windows that ran this outer iteration, and the windows that go on.
// One wave per mask word; lane = window within the word.  `last` keeps every window's norms of ITS last iteration (what the caller gets back as the change).
// `unsure` (or null): the windows whose norm lies within `band` (sfa_internal.h: break_band -- the bound on the difference between the two summations at this
// image size) of the threshold.  The reference forms the norms as fp32 running sums in raster order (variational_mt.cpp:412-425); the fp64 tree sum here differs
// from that by ~1e-5 of its value typically and by at most `band`, so outside the band the two decide alike, and inside it k_exact_break repeats the reference's
// own summation.  Their bits stay set here.
__global__ void __launch_bounds__(64) k_outer_threshold(const double *__restrict__ red, double *__restrict__ last, unsigned long long *__restrict__ amask,
                                                        WMask active, int nb, double npx, float thres, unsigned long long *__restrict__ unsure, double band) {
    const int word = blockIdx.x, b = 64 * word + (int)threadIdx.x;
    const unsigned long long cur = amask[word] & active.w[word];
    bool met = false, close = false;
    if (b < nb && ((cur >> threadIdx.x) & 1ull)) {
        const double a = red[2 * b] / npx, d = red[2 * b + 1] / npx;
        last[2 * b] = a; last[2 * b + 1] = d;
        // std::max(a, d) as the reference writes it (:436): (a < d) ? d : a -- a NaN norm in `a` stays a NaN, the comparison is false and the window
        // keeps iterating (fmaxf would drop the NaN and could declare the window converged)
        const float fa = (float)a, fd = (float)d;
        const float mx = (fa < fd) ? fd : fa;
        met = thres > 0.0f && mx < thres;
        const double dm = (a < d) ? d : a;
        close = unsure && thres > 0.0f && fabs(dm - (double)thres) <= band * (double)thres;
        if (close) met = false;
    }
    const unsigned long long m = __ballot(met), u = __ballot(close);
    if (threadIdx.x == 0) {
        amask[word] = amask[word] & ~m;
        if (unsure) unsure[word] = u;
    }
}
// The reference's own norms for the windows k_outer_threshold could not decide (variational_mt.cpp:412-436): blocks of four pixels in raster order, the four
// |differences| of a block added left to right, the block added to an fp32 running sum; then the fp32 division by height * width and max() as the reference writes
// it.  One wave per window: the 64 lanes form 64 block sums at a time, the running sum walks through them lane by lane (v_readlane + one dependent v_add_f32 per
// block: ~0.45 ms for 1024 x 436 -- which is why only the undecided windows come here, a handful per run).  dfa, dfb: the per-pixel |differences| left by the update.
__device__ __forceinline__ void exact_norms(const float *__restrict__ pa, const float *__restrict__ pb, const Geo &g, int lane, float &fa, float &fd) {
    const int nblk = (g.w + 3) / 4;                                  // blocks of a row that hold a pixel (the reference's stride may hold more: they add +0)
    float acc_a = 0.0f, acc_b = 0.0f;
    for (int y = 0; y < g.h; y++)
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int blk = b0 + lane, x = 4 * blk;
            float sa = 0.0f, sb = 0.0f;
            if (blk < nblk) {
                const float4 qa = *reinterpret_cast<const float4 *>(pa + (size_t)y * g.pitch + x), qb = *reinterpret_cast<const float4 *>(pb + (size_t)y * g.pitch + x);
                const float a0 = qa.x, a1 = x + 1 < g.w ? qa.y : 0.0f, a2 = x + 2 < g.w ? qa.z : 0.0f, a3 = x + 3 < g.w ? qa.w : 0.0f;
                const float c0 = qb.x, c1 = x + 1 < g.w ? qb.y : 0.0f, c2 = x + 2 < g.w ? qb.z : 0.0f, c3 = x + 3 < g.w ? qb.w : 0.0f;
                sa = ((a0 + a1) + a2) + a3;
                sb = ((c0 + c1) + c2) + c3;
            }
            const int n = min(64, nblk - b0);
            for (int i = 0; i < n; i++) {
                acc_a = acc_a + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sa), i));
                acc_b = acc_b + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sb), i));
            }
        }
    const float npx = (float)(g.h * g.w);
    fa = __fdiv_rn(acc_a, npx); fd = __fdiv_rn(acc_b, npx);
}
__global__ void __launch_bounds__(64) k_exact_break(const float *__restrict__ dfa, const float *__restrict__ dfb, double *__restrict__ last,
                                                    unsigned long long *__restrict__ amask, const unsigned long long *__restrict__ unsure, Geo g, float thres) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!((unsure[b >> 6] >> (b & 63)) & 1ull)) return;
    float fa, fd;
    exact_norms(dfa + b * g.es, dfb + b * g.es, g, lane, fa, fd);
    if (lane == 0) {
        last[2 * b] = (double)fa; last[2 * b + 1] = (double)fd;
        const float mx = (fa < fd) ? fd : fa;                      // std::max(a, b) = (a < b) ? b : a
        if (mx < thres) atomicAnd(&amask[b >> 6], ~(1ull << (b & 63)));
    }
}
// the same sums for the inner break (:371-407), which is taken on the host: the norms of the windows named in `which`, as the reference forms them, into out[2 b], out[2 b + 1]
__global__ void __launch_bounds__(64) k_exact_norms(const float *__restrict__ dfa, const float *__restrict__ dfb, float *__restrict__ out, WMask which, Geo g) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!which.test(b)) return;
    float fa, fd;
    exact_norms(dfa + b * g.es, dfb + b * g.es, g, lane, fa, fd);
    if (lane == 0) { out[2 * b] = fa; out[2 * b + 1] = fd; }
}
void launch_exact_norms(sfa_ctx *c, const Geo &g, const float *dfa, const float *dfb, const WMask &which, float *out) {
    hipLaunchKernelGGL(k_exact_norms, dim3(g.nb), dim3(64), 0, c->stream, dfa, dfb, out, which, g);
}
__global__ void k_set_mask(unsigned long long *amask, WMask v) { for (int i = 0; i < kMaskWords; i++) amask[i] = v.w[i]; }
static unsigned long long *unsure_of(sfa_ctx *c) { return c->d_last->unsure; }
void launch_outer_threshold(sfa_ctx *c, const Geo &g, const double *red, float thres, const float *dfa, const float *dfb) {
    const bool exact = dfa && thres > 0.0f;
    hipLaunchKernelGGL(k_outer_threshold, dim3(kMaskWords), dim3(64), 0, c->stream, red, c->d_last->last, c->d_amask, g.active, g.nb, (double)g.h * g.w, thres,
                       exact ? unsure_of(c) : (unsigned long long *)nullptr, break_band(g.w, g.h));
    if (exact) hipLaunchKernelGGL(k_exact_break, dim3(g.nb), dim3(64), 0, c->stream, dfa, dfb, c->d_last->last, c->d_amask, unsure_of(c), g, thres);
}
void launch_set_mask(sfa_ctx *c, const WMask &v) { hipLaunchKernelGGL(k_set_mask, dim3(1), dim3(1), 0, c->stream, c->d_amask, v); }

// partial-sum scratch lives behind the result words in ctx->d_red: [0, 2*kMaxBatch) results, then partials
static double *partials_of(sfa_ctx *c) { return c->d_red + 2 * kMaxBatch; }
// reduction grids: column strips of BX, at most kRedRows row-blocks that stride over the rows
constexpr int kRedRows = 16;
static inline dim3 red_grid(const Geo &g, int z) { return dim3((g.w + BX - 1) / BX, std::min((g.h + BY - 1) / BY, kRedRows), z); }

void launch_update_inner(sfa_ctx *c, const Geo &g, float *uu, float *vv, const float *wx, const float *wy, const float *du, const float *dv,
                         const float *old_du, const float *old_dv, double *red, float *dfa, float *dfb) {
    dim3 grid = red_grid(g, g.nb);
    const int per_elem = grid.x * grid.y;
    hipLaunchKernelGGL(k_update_inner, grid, block2d(), 0, c->stream, uu, vv, wx, wy, du, dv, old_du, old_dv, partials_of(c), dfa, dfb, g);
    hipLaunchKernelGGL(k_reduce_partials, dim3(g.nb), dim3(256), 0, c->stream, partials_of(c), per_elem, red, g.active, g.amask);
}
void launch_update_inner_x(sfa_ctx *c, const Geo &g, float *uu, float *vv, const float *wx, const float *wy, const SorOperandOut &x, const float *old_du,
                           const float *old_dv, float *du_out, float *dv_out, double *red, float *dfa, float *dfb) {
    dim3 grid = red_grid(g, g.nb);
    const int per_elem = grid.x * grid.y;
    hipLaunchKernelGGL(k_update_inner_x, grid, block2d(), 0, c->stream, uu, vv, wx, wy, x.x, x.ent, x.RP, x.G, old_du, old_dv, du_out, dv_out, partials_of(c), dfa, dfb, g);
    hipLaunchKernelGGL(k_reduce_partials, dim3(g.nb), dim3(256), 0, c->stream, partials_of(c), per_elem, red, g.active, g.amask);
}
constexpr int kFuseReduceWindows = 4;      // launches of up to that many windows sum their partials in k_update_outer_x itself
void launch_update_outer_x(sfa_ctx *c, const Geo &g, float *uu, float *vv, float *wx, float *wy, const SorOperandOut &x, double *red, float *dfa, float *dfb) {
    dim3 grid((g.w + 63) / 64, std::min((g.h + 15) / 16, kRedRows), g.nb);
    const int per_elem = grid.x * grid.y;
    unsigned *done = c->d_last->done;
    if (g.nb <= kFuseReduceWindows) {
        hipLaunchKernelGGL(k_update_outer_x<true>, grid, block2d(), 0, c->stream, uu, vv, wx, wy, x.x, x.ent, x.RP, x.G, partials_of(c), red, done, dfa, dfb, g);
        return;
    }
    hipLaunchKernelGGL(k_update_outer_x<false>, grid, block2d(), 0, c->stream, uu, vv, wx, wy, x.x, x.ent, x.RP, x.G, partials_of(c), red, done, dfa, dfb, g);
    hipLaunchKernelGGL(k_reduce_partials, dim3(g.nb), dim3(256), 0, c->stream, partials_of(c), per_elem, red, g.active, g.amask);
}
void launch_update_outer(sfa_ctx *c, const Geo &g, float *wx, float *wy, const float *uu, const float *vv, double *red, float *dfa, float *dfb) {
    dim3 grid = red_grid(g, g.nb);
    const int per_elem = grid.x * grid.y;
    hipLaunchKernelGGL(k_update_outer, grid, block2d(), 0, c->stream, wx, wy, uu, vv, partials_of(c), dfa, dfb, g);
    hipLaunchKernelGGL(k_reduce_partials, dim3(g.nb), dim3(256), 0, c->stream, partials_of(c), per_elem, red, g.active, g.amask);
}

__global__ void k_copy_planes(float *__restrict__ dst, const float *__restrict__ src, Geo g, int nplanes, long dst_es, long src_es) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = pl * g.pl + (size_t)y * g.pitch + x;
    dst[b * dst_es + o] = src[b * src_es + o];
}
void launch_copy_planes(sfa_ctx *c, const Geo &g, float *dst, const float *src, int nplanes, long dst_es, long src_es) {
    hipLaunchKernelGGL(k_copy_planes, grid2d(g, nplanes), block2d(), 0, c->stream, dst, src, g, nplanes, dst_es, src_es);
}

// image_erase / fill_n on whole planes (padding lanes included) of every active element
__global__ void k_fill_planes(float *__restrict__ p, Geo g, int nplanes, float v) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.pitch || y >= g.h) return;
    p[b * g.es + pl * g.pl + (size_t)y * g.pitch + x] = v;
}
void launch_fill_planes(sfa_ctx *c, const Geo &g, float *p, int nplanes, float v) {
    dim3 grid((g.pitch + BX - 1) / BX, (g.h + BY - 1) / BY, g.nb * nplanes);
    hipLaunchKernelGGL(k_fill_planes, grid, block2d(), 0, c->stream, p, g, nplanes, v);
}
void launch_zero_planes(sfa_ctx *c, const Geo &g, float *p, int nplanes) { launch_fill_planes(c, g, p, nplanes, 0.0f); }

// image_mul_scalar (image.c:49-57)
__global__ void k_scale_plane(float *__restrict__ p, Geo g, float s) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    p[b * g.es + (size_t)y * g.pitch + x] *= s;
}
void launch_scale_plane(sfa_ctx *c, const Geo &g, float *p, float s) {
    hipLaunchKernelGGL(k_scale_plane, grid2d(g), block2d(), 0, c->stream, p, g, s);
}

// ---------------------------------------------------------------------------------------------------
// K9 pyramid: cv::GaussianBlur(Size(0,0), sigma, BORDER_REPLICATE) and cv::resize(INTER_LINEAR) on CV_32F
// (variational_mt.cpp:607,611,672-673,711-712).  OpenCV is not vendored: documented semantics restated
// (symmetric separable fp32 taps, rows then columns; half-pixel-centre bilinear), parity unpinned.
// ---------------------------------------------------------------------------------------------------
struct Taps { float k[17]; int r; };

__global__ void k_gauss_h(float *__restrict__ dst, const float *__restrict__ src, Geo g, int nplanes, Taps t) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const float *s = src + b * g.es + pl * g.pl + (size_t)y * g.pitch;
    float acc = t.k[t.r] * s[x];
    for (int j = 1; j <= t.r; j++) acc += t.k[t.r + j] * (s[clampi(x - j, 0, g.w - 1)] + s[clampi(x + j, 0, g.w - 1)]);
    dst[b * g.es + pl * g.pl + (size_t)y * g.pitch + x] = acc;
}
__global__ void k_gauss_v(float *__restrict__ dst, const float *__restrict__ src, Geo g, int nplanes, Taps t) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const float *s = src + b * g.es + pl * g.pl;
    float acc = t.k[t.r] * s[(size_t)y * g.pitch + x];
    for (int j = 1; j <= t.r; j++)
        acc += t.k[t.r + j] * (s[(size_t)clampi(y - j, 0, g.h - 1) * g.pitch + x] + s[(size_t)clampi(y + j, 0, g.h - 1) * g.pitch + x]);
    dst[b * g.es + pl * g.pl + (size_t)y * g.pitch + x] = acc;
}
void launch_gauss_blur(sfa_ctx *c, const Geo &g, float *dst, float *tmp, const float *src, int nplanes, const float *taps, int radius) {
    Taps t;
    t.r = radius;
    for (int i = 0; i < 2 * radius + 1; i++) t.k[i] = taps[i];
    hipLaunchKernelGGL(k_gauss_h, grid2d(g, nplanes), block2d(), 0, c->stream, tmp, src, g, nplanes, t);
    hipLaunchKernelGGL(k_gauss_v, grid2d(g, nplanes), block2d(), 0, c->stream, dst, tmp, g, nplanes, t);
}

__global__ void k_resize(float *__restrict__ dst, int dw, int dh, int dpitch, long dpl, long des, const float *__restrict__ src, int sw, int sh, int spitch,
                         long spl, long ses, int nplanes, double scale_x, double scale_y, float post_scale) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int dx = blockIdx.x * BX + threadIdx.x, dy = blockIdx.y * BY + threadIdx.y;
    if (dx >= dw || dy >= dh) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
    const int sx1 = sx + 1 < sw ? sx + 1 : sx, sy1 = sy + 1 < sh ? sy + 1 : sy;
    const float *s = src + b * ses + pl * spl;
    const float *r0 = s + (size_t)sy * spitch, *r1 = s + (size_t)sy1 * spitch;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float h0 = r0[sx] * a0 + r0[sx1] * a1;
    const float h1 = r1[sx] * a0 + r1[sx1] * a1;
    float v = h0 * b0 + h1 * b1;
    if (post_scale != 1.0f) v *= post_scale;                                             // image_mul_scalar after the flow resize (:679-680,716-717)
    dst[b * des + pl * dpl + (size_t)dy * dpitch + dx] = v;
}
void launch_resize_scaled(sfa_ctx *c, float *dst, int dw, int dh, int dpitch, long dpl, long des, const float *src, int sw, int sh, int spitch, long spl,
                          long ses, int nplanes, int nb, float post_scale, double scale_x, double scale_y) {
    dim3 grid((dw + BX - 1) / BX, (dh + BY - 1) / BY, nb * nplanes);
    hipLaunchKernelGGL(k_resize, grid, block2d(), 0, c->stream, dst, dw, dh, dpitch, dpl, des, src, sw, sh, spitch, spl, ses, nplanes, scale_x, scale_y, post_scale);
}
void launch_resize(sfa_ctx *c, float *dst, int dw, int dh, int dpitch, long dpl, long des, const float *src, int sw, int sh, int spitch, long spl, long ses,
                   int nplanes, int nb, float post_scale) {
    // cv::resize with an explicit dsize: scale = ssize / dsize (imgproc resize.cpp: inv_scale_x = dsize.width / ssize.width)
    launch_resize_scaled(c, dst, dw, dh, dpitch, dpl, des, src, sw, sh, spitch, spl, ses, nplanes, nb, post_scale, (double)sw / dw, (double)sh / dh);
}

// The flow field's two planes one level up (variational_mt.cpp:703-717): k_resize's arithmetic for both planes of a pixel at once -- the column's source index and
// weight once per thread, the rows' once per block (LDS) -- four rows per thread.  As two launches of k_resize (an fp64 coordinate pair per pixel and plane) the step
// ran at 1.9 TB/s.
constexpr int kRfRows = 16;
__global__ void __launch_bounds__(BX * 4) k_resize_flow(float *__restrict__ dx_, float *__restrict__ dy_, int dw, int dh, int dpitch, long des, const float *__restrict__ sx_,
                                                         const float *__restrict__ sy_, int sw, int sh, int spitch, long ses, double scale_x, double scale_y, float post_x,
                                                         float post_y) {
    __shared__ int s_sy[kRfRows], s_sy1[kRfRows];
    __shared__ float s_fy[kRfRows];
    const int b = blockIdx.z;
    const int dx = blockIdx.x * BX + threadIdx.x, dy0 = blockIdx.y * kRfRows;
    if (threadIdx.y == 0 && threadIdx.x < kRfRows) {
        const int dy = dy0 + threadIdx.x;
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        s_sy[threadIdx.x] = sy; s_sy1[threadIdx.x] = sy + 1 < sh ? sy + 1 : sy; s_fy[threadIdx.x] = fy;
    }
    __syncthreads();
    if (dx >= dw) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
    const float a0 = 1.f - fx, a1 = fx;
    const float *px = sx_ + b * ses, *py = sy_ + b * ses;
#pragma unroll
    for (int k = 0; k < kRfRows / 4; k++) {
        const int ry = threadIdx.y + 4 * k, dy = dy0 + ry;
        if (dy >= dh) break;
        const size_t o0 = (size_t)s_sy[ry] * spitch, o1 = (size_t)s_sy1[ry] * spitch;
        const float fy = s_fy[ry], b0 = 1.f - fy, b1 = fy;
        const float x00 = px[o0 + sx], x01 = px[o0 + sx1], x10 = px[o1 + sx], x11 = px[o1 + sx1];
        const float y00 = py[o0 + sx], y01 = py[o0 + sx1], y10 = py[o1 + sx], y11 = py[o1 + sx1];
        float vx = (x00 * a0 + x01 * a1) * b0 + (x10 * a0 + x11 * a1) * b1;
        float vy = (y00 * a0 + y01 * a1) * b0 + (y10 * a0 + y11 * a1) * b1;
        if (post_x != 1.0f) vx *= post_x;                                              // image_mul_scalar after the flow resize (:679-680,716-717)
        if (post_y != 1.0f) vy *= post_y;
        const size_t o = b * des + (size_t)dy * dpitch + dx;
        dx_[o] = vx; dy_[o] = vy;
    }
}
void launch_resize_flow(sfa_ctx *c, float *dstx, float *dsty, int dw, int dh, int dpitch, long des, const float *srcx, const float *srcy, int sw, int sh, int spitch,
                        long ses, int nb, float post_x, float post_y) {
    const dim3 grid((dw + BX - 1) / BX, (dh + kRfRows - 1) / kRfRows, nb);
    hipLaunchKernelGGL(k_resize_flow, grid, dim3(BX, 4), 0, c->stream, dstx, dsty, dw, dh, dpitch, des, srcx, srcy, sw, sh, spitch, ses, (double)sw / dw, (double)sh / dh,
                       post_x, post_y);
}

// One pyramid step in one pass (variational_mt.cpp:607,611): GaussianBlur then resize of `nplanes` planes per window.  A block owns a
// 64 x td tile of the DESTINATION (td = 32 when the footprint fits LDS: the row pass runs over the footprint plus 2r rows, 1.2x the tile at 32 rows
// against 1.9x at 8); the source footprint (+ blur radius, replicated at the image border) is staged in LDS,
// blurred along rows, then along columns, and sampled bilinearly -- the same operations in the same order as k_gauss_h,
// k_gauss_v, k_resize, without the two intermediate images.
// Round 5: the footprint starts at a source column that is a multiple of 4 (up to three columns left of what the tile needs), so the staging is plain 16-byte
// copies (the scatter of quads into an unaligned footprint was 43 % LDS bank-conflict cycles), and the column pass writes over the staged footprint, which is dead by
// then: 40 KB per block instead of 54, four blocks per CU instead of two.
#ifndef SFA_PYR_GROUP
#define SFA_PYR_GROUP 4
#endif
constexpr int kPyrRB = 5;                                            // rows per item of the column pass
#ifndef SFA_PYR_NT
#define SFA_PYR_NT 256
#endif
constexpr int kPyrNT = SFA_PYR_NT, kPyrNY = kPyrNT / 64;          // threads per block (what-if, round 6: 512 = eight waves on a tile)
template <int R>
__global__ void __launch_bounds__(kPyrNT) k_pyr_down(float *__restrict__ dst, int dw, int dh, int dpitch, long dpl, long des, const float *__restrict__ src, int sw, int sh,
                                                  int spitch, long spl, long ses, int nplanes, double scale_x, double scale_y, Taps t, int CM, int RM, int td) {
    extern __shared__ __attribute__((aligned(16))) float pyr_lds[];
    constexpr int r = R;                                              // compile-time radius: the tap loops unroll, the row buffer stays in registers
    const int CS = (CM + 2 * r + 3) / 4 * 4 + 4, RS = RM + 2 * r;       // CM, CS multiples of 4; CS leaves room for the aligned quads of the row pass
    float *S = pyr_lds, *Hb = S + RS * CS, *V = S;                      // V (RM x CM) lies on S (RS x CS): S is dead once the row pass has run
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int dx0 = blockIdx.x * 64, dy0 = blockIdx.y * td;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    // first source column / row any pixel of the tile samples (k_resize's own coordinate arithmetic)
    int mx0 = (int)floorf((float)((dx0 + 0.5) * scale_x - 0.5)), my0 = (int)floorf((float)((dy0 + 0.5) * scale_y - 0.5));
    mx0 = mx0 < 0 ? 0 : mx0; my0 = my0 < 0 ? 0 : my0;
    const int sxa = (mx0 - r) & ~3;                                  // source column of S column 0 (two's complement: & ~3 floors negative columns too)
    mx0 = sxa + r;                                                   // source column of tile column 0: <= the first one sampled, by at most 3
    const float *s = src + b * ses + pl * spl;
    // staging: S column c is source column sxa + c (replicated outside the image), one aligned 16-byte load and one 16-byte LDS store per item.
    // (row, quad) of a thread's items advance incrementally: one division per thread instead of one per item.
    {
        const int NQ = CS / 4;
        int j = tid / NQ, qa = tid % NQ;
        const int dj = kPyrNT / NQ, dq = kPyrNT % NQ;
        // a thread's items four at a time: every global load of the group is issued before the first value is stored (one item at a time -- load, wait,
        // LDS store the next load could not pass -- a thread's 4-5 items were as many memory round trips in a row)
        constexpr int G = SFA_PYR_GROUP;
        while (j < RS) {
            float4 v[G];
            int jj[G], qq[G];
#pragma unroll
            for (int i = 0; i < G; i++) {
                jj[i] = j; qq[i] = qa;
                if (j < RS) {
                    const float *row = s + (size_t)clampi(my0 - r + j, 0, sh - 1) * spitch;
                    const int gx = sxa + 4 * qa;
                    if (gx >= 0 && gx + 3 < sw) v[i] = *reinterpret_cast<const float4 *>(row + gx);
                    else v[i] = make_float4(row[clampi(gx, 0, sw - 1)], row[clampi(gx + 1, 0, sw - 1)], row[clampi(gx + 2, 0, sw - 1)], row[clampi(gx + 3, 0, sw - 1)]);
                }
                qa += dq; j += dj;
                if (qa >= NQ) { qa -= NQ; j++; }
            }
#pragma unroll
            for (int i = 0; i < G; i++)
                if (jj[i] < RS) *reinterpret_cast<float4 *>(S + jj[i] * CS + 4 * qq[i]) = v[i];
        }
    }
    __syncthreads();
    // k_gauss_h on the rows of the footprint, four columns per item: the 4 + 2r inputs come as aligned float4 reads
    const int CQ = CM / 4, djq = kPyrNT / CQ, dcq = kPyrNT % CQ;
    for (int j = tid / CQ, cq = tid % CQ; j < RS;) {
        const int c = 4 * cq;
        constexpr int nq = (4 + 2 * r + 3) / 4;
        float in[4 * nq];                                            // columns c - r .. of the tile  (S column c + r is tile column c)
        const float4 *row4 = reinterpret_cast<const float4 *>(S + j * CS + c);     // c is a multiple of 4, CS too
#pragma unroll
        for (int q = 0; q < nq; q++) { const float4 v = row4[q]; in[4 * q] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w; }
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float acc = t.k[r] * in[e + r];
#pragma unroll
            for (int q = 1; q <= r; q++) acc += t.k[r + q] * (in[e + r - q] + in[e + r + q]);
            out[e] = acc;
        }
        *reinterpret_cast<float4 *>(Hb + j * CM + c) = make_float4(out[0], out[1], out[2], out[3]);
        cq += dcq; j += djq;
        if (cq >= CQ) { cq -= CQ; j++; }
    }
    __syncthreads();
    // k_gauss_v, four columns of kPyrRB consecutive rows per item: the rows' 2r + kPyrRB inputs are read once (one row per item: 2r + 1 reads per output row --
    // the column pass was 45 % of the kernel's LDS traffic)
    constexpr int RB = kPyrRB;
    for (int item = tid; item < (RM + RB - 1) / RB * CQ; item += kPyrNT) {
        const int jg = item / CQ, c = 4 * (item % CQ), j0 = RB * jg;
        float4 in[RB + 2 * r];
#pragma unroll
        for (int i = 0; i < RB + 2 * r; i++) in[i] = *reinterpret_cast<const float4 *>(Hb + min(j0 + i, RS - 1) * CM + c);
#pragma unroll
        for (int o = 0; o < RB; o++) {
            if (j0 + o >= RM) break;
            const float4 ctr = in[o + r];
            float4 acc = make_float4(t.k[r] * ctr.x, t.k[r] * ctr.y, t.k[r] * ctr.z, t.k[r] * ctr.w);
#pragma unroll
            for (int q = 1; q <= r; q++) {
                const float4 up = in[o + r - q], dn = in[o + r + q];
                const float kq = t.k[r + q];
                acc.x += kq * (up.x + dn.x); acc.y += kq * (up.y + dn.y); acc.z += kq * (up.z + dn.z); acc.w += kq * (up.w + dn.w);
            }
            *reinterpret_cast<float4 *>(V + (j0 + o) * CM + c) = acc;
        }
    }
    // the tile's rows: source rows and weight (k_resize's arithmetic), once per block
    int *s_sy = reinterpret_cast<int *>(Hb + RS * CM), *s_sy1 = s_sy + td;
    float *s_fy = reinterpret_cast<float *>(s_sy1 + td);
    if (tid < td) {
        const int dy = dy0 + tid;
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        s_sy[tid] = (sy - my0) * CM; s_sy1[tid] = ((sy + 1 < sh ? sy + 1 : sy) - my0) * CM; s_fy[tid] = fy;
    }
    __syncthreads();
    const int dx = dx0 + threadIdx.x;
    if (dx >= dw) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);                 // k_resize
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
    const float a0 = 1.f - fx, a1 = fx;
    for (int k = 0; k < td / kPyrNY; k++) {
        const int ry = threadIdx.y + kPyrNY * k, dy = dy0 + ry;
        if (dy >= dh) break;
        const float *r0 = V + s_sy[ry] - mx0, *r1 = V + s_sy1[ry] - mx0;
        const float fy = s_fy[ry], b0 = 1.f - fy, b1 = fy;
        const float h0 = r0[sx] * a0 + r0[sx1] * a1;
        const float h1 = r1[sx] * a0 + r1[sx1] * a1;
        dst[b * des + pl * dpl + (size_t)dy * dpitch + dx] = h0 * b0 + h1 * b1;
    }
}
// returns false if the footprint does not fit LDS (very small p_scale): the caller then runs the separate kernels
bool launch_pyr_down(sfa_ctx *c, float *dst, int dw, int dh, int dpitch, long dpl, long des, const float *src, int sw, int sh, int spitch, long spl, long ses,
                     int nplanes, int nb, const float *taps, int radius) {
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    const int CM = (((int)ceil(64 * scale_x) + 2) + 3 + 3) / 4 * 4, CS = (CM + 2 * radius + 3) / 4 * 4 + 4;   // + 3: the footprint starts at a multiple of 4
    int td = 32, RM = 0, RS = 0;
    size_t lds = 0;
    for (;; td /= 2) {                                               // the tallest tile whose footprint fits
        RM = (int)ceil(td * scale_y) + 2; RS = RM + 2 * radius;
        lds = (size_t)(RS * CS + RS * CM + 3 * td) * sizeof(float);  // the column pass writes over the staged footprint (RM * CM <= RS * CS); + the rows' sampling table
        if (lds <= 40 * 1024 || td == 8) break;                     // 40 KB: four blocks per CU
    }
    if (lds > 60 * 1024 || radius > 8) return false;
    Taps t;
    t.r = radius;
    for (int i = 0; i < 2 * radius + 1; i++) t.k[i] = taps[i];
    const dim3 grid((dw + 63) / 64, (dh + td - 1) / td, nb * nplanes), block(64, kPyrNY);
#define SFA_PYR(RR) case RR: hipLaunchKernelGGL(k_pyr_down<RR>, grid, block, lds, c->stream, dst, dw, dh, dpitch, dpl, des, src, sw, sh, spitch, spl, ses, nplanes, scale_x, scale_y, t, CM, RM, td); break
    switch (radius) {
        SFA_PYR(1); SFA_PYR(2); SFA_PYR(3); SFA_PYR(4); SFA_PYR(5); SFA_PYR(6); SFA_PYR(7); SFA_PYR(8);
    default: return false;
    }
#undef SFA_PYR
    return true;
}

// optional level-0 Gaussian presmoothing (cfg sigma > 0, variational_mt.cpp:590-597): gaussian_filter
// (image.c:310-348) + the generic convolve_horiz / convolve_vert (image.c:537-644), whose border handling
// uses the accumulated coefficients.
struct PreTaps { float c[33]; float accu[33]; int order; };
__global__ void k_presmooth_h(float *__restrict__ dst, const float *__restrict__ src, Geo g, int nplanes, PreTaps t) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    if (i >= g.w || j >= g.h) return;
    const float *row = src + b * g.es + pl * g.pl + (size_t)j * g.pitch;
    const int i0 = -t.order, i1 = t.order;
    const float *coeff = t.c + t.order, *coeff_accu = t.accu + t.order;
    float sum;
    if (i < -i0) {                                                                       // image.c:550-556
        sum = coeff_accu[-i - 1] * row[0];
        for (int ii = i1 + i; ii >= 0; ii--) sum += coeff[ii - i] * row[ii];
    } else if (i < g.w - i1) {                                                           // image.c:558-565
        const float *al = row + (i + i0);
        sum = 0;
        for (int ii = i1 - i0; ii >= 0; ii--) sum += t.c[ii] * al[ii];
    } else {                                                                             // image.c:567-574
        const float *al = row + (i + i0);
        sum = coeff_accu[g.w - i] * al[g.w - i0 - 1 - i];
        for (int ii = g.w - i0 - 1 - i; ii >= 0; ii--) sum += t.c[ii] * al[ii];
    }
    dst[b * g.es + pl * g.pl + (size_t)j * g.pitch + i] = sum;
}
__global__ void k_presmooth_v(float *__restrict__ dst, const float *__restrict__ src, Geo g, int nplanes, PreTaps t) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;   // j = column, i = row
    if (j >= g.w || i >= g.h) return;
    const float *in = src + b * g.es + pl * g.pl;
    const int i0 = -t.order, i1 = t.order, st = g.pitch;
    const float *coeff = t.c + t.order, *coeff_accu = t.accu + t.order;
    float sum;
    if (i < -i0) {                                                                       // image.c:600-609
        sum = coeff_accu[-i - 1] * in[j];
        for (int ii = -i; ii <= i1; ii++) sum += coeff[ii] * in[(size_t)(i + ii) * st + j];
    } else if (i < g.h - i1) {                                                           // image.c:615-626
        sum = 0;
        for (int ii = 0; ii <= i1 - i0; ii++) sum += t.c[ii] * in[(size_t)(i + i0 + ii) * st + j];
    } else {                                                                             // image.c:631-640
        sum = coeff_accu[g.h - i] * in[(size_t)(g.h - 1) * st + j];
        for (int ii = i0; ii <= g.h - 1 - i; ii++) sum += coeff[ii] * in[(size_t)(i + ii) * st + j];
    }
    dst[b * g.es + pl * g.pl + (size_t)i * st + j] = sum;
}
// filters of order 1 and 2 take the 3 / 5-tap routines of image.c (:529-535, :580-586 dispatch on conv->order) -- replicated columns,
// folded row coefficients -- with the Gaussian's coefficients
struct Coef5 { float c[5]; };
__global__ void k_presmooth_small(float *__restrict__ dst, const float *__restrict__ src, Geo g, int nplanes, Coef5 k, int order, int horiz) {
    const int b = blockIdx.z / nplanes, pl = blockIdx.z % nplanes;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const float *s = src + b * g.es + pl * g.pl;
    const int w = g.w, h = g.h;
    auto f = [&](int xx, int yy) { return s[(size_t)yy * g.pitch + xx]; };
    const float *c = k.c;
    float v;
    if (horiz) {
        if (order == 2) v = c[0] * f(clampi(x - 2, 0, w - 1), y) + c[1] * f(clampi(x - 1, 0, w - 1), y) + c[2] * f(x, y) + c[3] * f(clampi(x + 1, 0, w - 1), y) +
                            c[4] * f(clampi(x + 2, 0, w - 1), y);                        // image.c:521
        else            v = c[0] * f(clampi(x - 1, 0, w - 1), y) + c[1] * f(x, y) + c[2] * f(clampi(x + 1, 0, w - 1), y);   // image.c:482
    } else if (order == 2) {                                                             // image.c:433-457
        if (y == 0) v = (c[0] + c[1] + c[2]) * f(x, 0) + c[3] * f(x, 1) + c[4] * f(x, 2);
        else if (y == 1) v = (c[0] + c[1]) * f(x, 0) + c[2] * f(x, 1) + c[3] * f(x, 2) + c[4] * f(x, 3);
        else if (y == h - 2) v = c[0] * f(x, y - 2) + c[1] * f(x, y - 1) + c[2] * f(x, y) + (c[3] + c[4]) * f(x, y + 1);
        else if (y == h - 1) v = c[0] * f(x, y - 2) + c[1] * f(x, y - 1) + (c[2] + c[3] + c[4]) * f(x, y);
        else v = c[0] * f(x, y - 2) + c[1] * f(x, y - 1) + c[2] * f(x, y) + c[3] * f(x, y + 1) + c[4] * f(x, y + 2);
    } else {                                                                             // image.c:407-422
        if (y == 0) v = (c[0] + c[1]) * f(x, 0) + c[2] * f(x, 1);
        else if (y == h - 1) v = c[0] * f(x, y - 1) + (c[1] + c[2]) * f(x, y);
        else v = c[0] * f(x, y - 1) + c[1] * f(x, y) + c[2] * f(x, y + 1);
    }
    dst[b * g.es + pl * g.pl + (size_t)y * g.pitch + x] = v;
}
void launch_presmooth(sfa_ctx *c, const Geo &g, float *dst, float *tmp, const float *src, int nplanes, float sigma) {
    PreTaps t;
    int order = (int)floor(3 * sigma) + 1;                                               // image.c:320
    if (order == 0) order = 1;
    if (order > 16) order = 16;
    t.order = order;
    const int n = 2 * order + 1;
    float data[33];
    const float alpha = 1.0f / (2.0f * sigma * sigma);
    float sum = 0.0f;
    for (int i = -order; i <= order; i++) { data[i + order] = (float)exp(-i * i * alpha); sum += data[i + order]; }   // image.c:332-335
    for (int i = 0; i < n; i++) data[i] /= sum;
    const float *half = data + order;
    for (int i = 0; i <= order; i++) t.c[order - i] = t.c[order + i] = half[i];          // image.c:355-357
    float acc = 0.0f;
    for (int i = 0; i <= order; i++) { acc += t.c[i]; t.accu[2 * order - i] = t.accu[i] = acc; }   // image.c:358-361
    if (order <= 2) {
        Coef5 k;
        for (int i = 0; i < 5; i++) k.c[i] = i < n ? t.c[i] : 0.0f;
        hipLaunchKernelGGL(k_presmooth_small, grid2d(g, nplanes), block2d(), 0, c->stream, tmp, src, g, nplanes, k, order, 1);
        hipLaunchKernelGGL(k_presmooth_small, grid2d(g, nplanes), block2d(), 0, c->stream, dst, tmp, g, nplanes, k, order, 0);
        return;
    }
    hipLaunchKernelGGL(k_presmooth_h, grid2d(g, nplanes), block2d(), 0, c->stream, tmp, src, g, nplanes, t);
    hipLaunchKernelGGL(k_presmooth_v, grid2d(g, nplanes), block2d(), 0, c->stream, dst, tmp, g, nplanes, t);
}

// ---------------------------------------------------------------------------------------------------
// normalize (variational_mt.cpp:17-85): per-channel sum and sum of squares (float product, double sum)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BX *BY) k_norm_sums(const float *__restrict__ frames3, double *__restrict__ partial, Geo g) {
    const int ch = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x;
    double sa = 0, sb = 0;
    if (x < g.w)
        for (int y = blockIdx.y * BY + threadIdx.y; y < g.h; y += gridDim.y * BY) {
            const float v = frames3[ch * g.pl + (size_t)y * g.pitch + x];
            sa += (double)v;                                                             // :32-34
            sb += (double)(v * v);                                                       // :35-37
        }
    block_sum2(sa, sb);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const size_t blk = ((size_t)ch * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * blk] = sa; partial[2 * blk + 1] = sb;
    }
}
void launch_normalize_sums(sfa_ctx *c, const Geo &g, const float *frames3, double *red) {
    dim3 grid = red_grid(g, 3);
    hipLaunchKernelGGL(k_norm_sums, grid, block2d(), 0, c->stream, frames3, partials_of(c), g);
    hipLaunchKernelGGL(k_reduce_partials, dim3(3), dim3(256), 0, c->stream, partials_of(c), (int)(grid.x * grid.y), red, WMask::first(3), (const unsigned long long *)nullptr);
}
__global__ void k_norm_apply(float *__restrict__ frames3, Geo g, double a0, double a1, double a2, double s0, double s1, double s2) {
    const int ch = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const double a = ch == 0 ? a0 : (ch == 1 ? a1 : a2), s = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
    if (!(s > 0)) return;                                                                // :64-66
    float *p = frames3 + ch * g.pl + (size_t)y * g.pitch + x;
    *p = (float)(((double)*p - a) / s);
}
void launch_normalize_apply(sfa_ctx *c, const Geo &g, float *frames3, const double avg[3], const double stdv[3]) {
    dim3 grid((g.w + BX - 1) / BX, (g.h + BY - 1) / BY, 3);
    hipLaunchKernelGGL(k_norm_apply, grid, block2d(), 0, c->stream, frames3, g, avg[0], avg[1], avg[2], stdv[0], stdv[1], stdv[2]);
}

}  // namespace sfa
