// sor_device.h -- device helpers shared by the solver kernels (sor.hip, sor_chain.hip): the SOR point update in the fast solver's
// operation order (solver.c:337-343), DPP lane shifts, write-through / bypassing accesses for the hand-over between workgroups,
// buffer descriptors, bounded polls.
#pragma once
#include "sfa_internal.h"
#include "sfa_device.h"

#pragma clang fp contract(off)

namespace sfa {

constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ float2 u2f(unsigned long long v) { return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))); }
__device__ __forceinline__ unsigned long long f2u(float a, float b) { return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32); }

// The SOR point update on (du, dv) pairs, the fast solver's operations in the fast solver's order (solver.c:337-343):
//   s = ((hp*x_right + vt*x_top) + vp*x_bottom) + b;  B = hl*x_left + s;  x += w*((a11*B1 + a12*B2) - x) (and the dv row)
// written on 2-vectors so that it compiles to v_pk_mul_f32 / v_pk_add_f32 (one rounding per operation, no FMA).
// Absent neighbours are not skipped with selects: their edge weight is exactly 0 (vt at row 0, vp at the last row, hl at
// column 0 -- k_sor_prepare) and their value is a finite 0 (zero guards), so the term contributes +-0 and the sum is
// unchanged (IEEE ==; at most the sign of an exact zero differs).  Likewise a point outside the image has all-zero
// operands and a zero "self", so its update is exactly 0 without forcing it.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f f2v(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ v2f sor_point(v2f self, v2f right, v2f top, v2f bottom, v2f left, float hl, const float4 &SA, const float4 &SB, float omega) {
    // SA = (inv11, inv12, inv22, vt)   SB = (b1, b2, hp, vp)
    // written on the register pairs the loads deliver, with explicit broadcasts: the packed ops then select their halves
    // (op_sel) instead of copying scalars into fresh pairs -- 8 VALU instructions less per step of the band kernel
    const v2f SAxy = {SA.x, SA.y}, SAzw = {SA.z, SA.w}, SBxy = {SB.x, SB.y}, SBzw = {SB.z, SB.w};
#ifdef SFA_X_NOARITH      // timing experiment only: one dependent operation instead of fourteen
    return self + right * SAxy + top * SBxy + bottom * SAzw + left * hl;
#endif
    v2f s = __builtin_shufflevector(SBzw, SBzw, 0, 0) * right;
    s = s + __builtin_shufflevector(SAzw, SAzw, 1, 1) * top;
    s = s + __builtin_shufflevector(SBzw, SBzw, 1, 1) * bottom;
    s = s + SBxy;
    const v2f B = hl * left + s;
    const v2f t = SAxy * __builtin_shufflevector(B, B, 0, 0) + __builtin_shufflevector(SAxy, SAzw, 1, 2) * __builtin_shufflevector(B, B, 1, 1);
    return self + omega * (t - self);
}

// value of lane-1 (lane 0 receives `fill`): DPP wave_shr:1, no LDS
__device__ __forceinline__ float lane_shr1(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

// lane 0 <- lane X of the same register (X < 16): one DPP row shift confined to row 0; the other lanes are don't-care
template <int X>
__device__ __forceinline__ int row0_from(int v) {
    if constexpr (X == 0) return v;
    else return __builtin_amdgcn_mov_dpp(v, 0x100 + X /* row_shl:X: lane i <- lane i+X */, 0x1, 0xf, false);   // other lanes: undefined
}
__device__ __forceinline__ int lane0_from(int v, int X) {        // X is a constant after unrolling
    switch (X) {
        case 0: return row0_from<0>(v);   case 1: return row0_from<1>(v);   case 2: return row0_from<2>(v);   case 3: return row0_from<3>(v);
        case 4: return row0_from<4>(v);   case 5: return row0_from<5>(v);   case 6: return row0_from<6>(v);   case 7: return row0_from<7>(v);
        case 8: return row0_from<8>(v);   case 9: return row0_from<9>(v);   case 10: return row0_from<10>(v); case 11: return row0_from<11>(v);
        case 12: return row0_from<12>(v); case 13: return row0_from<13>(v); case 14: return row0_from<14>(v); default: return row0_from<15>(v);
    }
}
// value of lane-1; lane 0 keeps lane 0 of `fillvec`
__device__ __forceinline__ float lane_shr1_vec(float v, int fillvec) {
    return __int_as_float(__builtin_amdgcn_update_dpp(fillvec, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

__device__ __forceinline__ unsigned long long ld_x(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // global_load_dwordx2 sc1
}
__device__ __forceinline__ void st_x(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               // global_store_dwordx2 sc1 (write-through)
}
__device__ __forceinline__ unsigned ld_flag(const unsigned *p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Buffer addressing for the band kernel's streams: a wave-uniform descriptor per plane, the lane part of the address in one CONSTANT 32-bit
// voffset and the per-step advance in the scalar soffset -- no per-step 64-bit vector address arithmetic (flat loads cost a v_lshl_add_u64 each).
// Lanes that must not take part in an access get an out-of-range voffset: the range check drops their load (returns 0) or store, no exec mask.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr unsigned kOobOffset = 0xfffffff0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *base, unsigned long long bytes) {
    const unsigned long long p = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xffffff00ull ? 0xffffff00ull : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ float4 bload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ unsigned long long bload8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return (unsigned long long)v.x | ((unsigned long long)v.y << 32);
}
__device__ __forceinline__ float2 bload8f(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ void bstore8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float a, float b) {
    v2u v = {__float_as_uint(a), __float_as_uint(b)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// bounded relaxed poll of one progress word (wave-uniform); returns the value seen (>= target) or 0xffffffff on give-up
__device__ __forceinline__ unsigned wait_ge(const unsigned *p, unsigned target, unsigned *err) {
    unsigned spins = 0, limit = kSpinLimit;
    for (;;) {
        const unsigned v = ld_flag(p);
        if (v >= target) return v;
        if (spins == 0) {                                   // slow path only: err[1] != 0 shortens the bound (test hook sfa_ctx_set_wait_bound)
            const unsigned o = ld_flag(err + 1);
            if (o) limit = o;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0 || spins > limit) {
            const unsigned e = ld_flag(err);
            if (e || spins > limit) {
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0xffffffffu;
            }
        }
    }
}

}  // namespace sfa
