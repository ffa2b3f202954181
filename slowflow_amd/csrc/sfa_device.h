// sfa_device.h -- device-side helpers shared by the kernel translation units (kernels.hip, occlusion.hip):
// launch shapes, the reference's derivative filters with its border handling, IEEE sqrt.
#pragma once
#include "sfa_internal.h"

#pragma clang fp contract(off)

namespace sfa {

#define BX 64
#define BY 4

static inline dim3 grid2d(const Geo &g, int zmul = 1) { return dim3((g.w + BX - 1) / BX, (g.h + BY - 1) / BY, g.nb * zmul); }
static inline dim3 block2d() { return dim3(BX, BY, 1); }

// window b takes part: its bit in what the host knows AND in what the device knows (one word of each is looked at)
__device__ __forceinline__ bool elem_active(const WMask &active, const unsigned long long *amask, int b) {
    const unsigned long long m = amask ? (active.w[b >> 6] & amask[b >> 6]) : active.w[b >> 6];
    return (m >> (b & 63)) & 1ull;
}
__device__ __forceinline__ bool elem_active(const Geo &g, int b) { return elem_active(g.active, g.amask, b); }
__device__ __forceinline__ int clampi(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }
// `int x = floor(xx)` as the reference's x86 build evaluates it (variational_aux_mt.cpp:737-738: cvttsd2si): a value outside the int range -- a refinement that
// diverges, e.g. under an ill-posed parameter set -- and NaN give INT_MIN, the instruction's "integer indefinite".  The GPU's own conversion saturates (+huge ->
// INT_MAX), and `x + 1` then overflows: the compiler, entitled to assume it does not, had turned clamp(x + 1, 0, w - 1) into a select that let INT_MIN through as
// a column index -- a memory access fault (found by tools/fuzz_parity.py, round 6; the form of rounds 1-5 had it too).  With this conversion x + 1 cannot overflow,
// and a diverged window reads what the reference's would: column / row 0.
__device__ __forceinline__ int floor_to_int_x86(float v) {
    const float f = floorf(v);
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (-2147483647 - 1);
}

// derivative filter taps as convolution_new builds them (image.c:363-366, variational_mt.cpp:570-573)
#define C5_0 (1.0f / 12.0f)
#define C5_1 (-8.0f / 12.0f)
#define C5_2 (-0.0f)
#define C5_3 (8.0f / 12.0f)
#define C5_4 (-(1.0f / 12.0f))
#define C3_0 (-0.5f)
#define C3_1 (-0.0f)
#define C3_2 (0.5f)

// image.c:521
__device__ __forceinline__ float tap5(float m2, float m1, float c, float p1, float p2) {
#if defined(SFA_X_NO_CENTRE_TAP)
    return C5_0 * m2 + C5_1 * m1 + C5_3 * p1 + C5_4 * p2;      // timing what-if only: the reference multiplies the centre by -0.0 and adds it
#else
    return C5_0 * m2 + C5_1 * m1 + C5_2 * c + C5_3 * p1 + C5_4 * p2;
#endif
}

// horizontal 5-tap at (x,y) of a plane accessor F(x,y); replicate border (image.c:501-516)
template <class F>
__device__ __forceinline__ float d5x(F f, int x, int y, int w) {
    return tap5(f(clampi(x - 2, 0, w - 1), y), f(clampi(x - 1, 0, w - 1), y), f(x, y), f(clampi(x + 1, 0, w - 1), y), f(clampi(x + 2, 0, w - 1), y));
}
// vertical 5-tap with the run-time folded border coefficients (image.c:433-457)
template <class F>
__device__ __forceinline__ float d5y(F f, int x, int y, int h) {
    if (y == 0) return (C5_0 + C5_1 + C5_2) * f(x, 0) + C5_3 * f(x, 1) + C5_4 * f(x, 2);
    if (y == 1) return (C5_0 + C5_1) * f(x, 0) + C5_2 * f(x, 1) + C5_3 * f(x, 2) + C5_4 * f(x, 3);
    if (y == h - 2) return C5_0 * f(x, y - 2) + C5_1 * f(x, y - 1) + C5_2 * f(x, y) + (C5_3 + C5_4) * f(x, y + 1);
    if (y == h - 1) return C5_0 * f(x, y - 2) + C5_1 * f(x, y - 1) + (C5_2 + C5_3 + C5_4) * f(x, y);
    return tap5(f(x, y - 2), f(x, y - 1), f(x, y), f(x, y + 1), f(x, y + 2));
}
// 3-tap (image.c:482, 407-422)
template <class F>
__device__ __forceinline__ float d3x(F f, int x, int y, int w) {
    return C3_0 * f(clampi(x - 1, 0, w - 1), y) + C3_1 * f(x, y) + C3_2 * f(clampi(x + 1, 0, w - 1), y);
}
template <class F>
__device__ __forceinline__ float d3y(F f, int x, int y, int h) {
    if (y == 0) return (C3_0 + C3_1) * f(x, 0) + C3_2 * f(x, 1);
    if (y == h - 1) return C3_0 * f(x, y - 1) + (C3_1 + C3_2) * f(x, y);
    return C3_0 * f(x, y - 1) + C3_1 * f(x, y) + C3_2 * f(x, y + 1);
}

// IEEE-correct fp32 square root (sqrtps in the reference).  NOT __fsqrt_rn: in this toolchain that is the native
// approximate instruction (__clang_hip_math.h:302); __builtin_sqrtf is correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }

struct PlaneAcc {
    const float *p; int pitch;
    __device__ __forceinline__ float operator()(int x, int y) const { return p[(size_t)y * pitch + x]; }
};


}  // namespace sfa
