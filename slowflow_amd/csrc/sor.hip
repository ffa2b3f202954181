// sor.hip -- sor_coupled (solver.c:63-399) on gfx950.
//
// The reference sweeps in raster order: x(c,r)^k needs x(c-1,r)^k, x(c,r-1)^k (already updated) and
// x(c+1,r)^(k-1), x(c,r+1)^(k-1) (old).  A red-black sweep is a different algorithm (1e-2 off after 30
// sweeps); instead the SAME dependency graph is executed as a pipeline of hyperplanes:
//
//   * iteration k of row r is given the skewed row index rho = r + k.  64 consecutive rho form a band;
//     task (band b, iteration k) is ONE wavefront, lane l <-> rho = 64 b + l, i.e. row r = 64 b + l - k.
//   * at local step s the lane works on column c = s - l, so the whole wave sits on one anti-diagonal
//     d = c + r = s + 64 b - k of the image: all its operands are 64 CONSECUTIVE entries of the
//     diagonal-major planes built by k_sor_prepare (entry (d, r) at (d+G)*RP + r+G) -> every load and
//     store of the sweep is a coalesced 256-B..1-KB wave access.
//   * with the skew, every dependency of (b,k) points to (b,k-1), (b-1,k) or (b-1,k-1): the left
//     neighbour is the lane's own previous result, the top neighbour lane l-1's previous result (DPP
//     wave_shr:1), right/bottom/self are iteration k-1 values read back from the in-place x plane.
//   * tasks hand over through write-through (sc1) stores + one progress word per task, polled with sc1
//     loads (agent scope; placement independent); tickets are drawn from an atomic counter in an order
//     in which every dependency has a smaller ticket, so the pipeline cannot deadlock whatever the
//     dispatch order or residency.  Every spin is bounded.
//
// Per point the arithmetic is the fast solver's, operation for operation (solver.c:183-196), so the
// result is numerically identical to the reference's lexicographic sweep.
#include "sfa_internal.h"

#pragma clang fp contract(off)

namespace sfa {

constexpr int CH = 8;               // steps per hand-over chunk (divides 64)
constexpr unsigned kSpinLimit = 1u << 22;

struct SorArgs {
    const float4 *sa;               // (inv11, inv12, inv22, b1)
    const float4 *sb;               // (b2, hp, vp, vt)
    unsigned long long *x;          // (du, dv) pairs, in place
    unsigned *flags;                // [nb][K][NB] chunks completed; flags[nb*ntasks] = ticket
    const int2 *order;              // ticket/nb -> (b, k)
    unsigned *err;
    long ent;                       // entries per batch element
    int W, H, K, NB, RP, G, NS, NCH, ntasks, nb;
    float omega;
};

__device__ __forceinline__ float2 u2f(unsigned long long v) { return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))); }
__device__ __forceinline__ unsigned long long f2u(float a, float b) { return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32); }

// value of lane-1 (lane 0 receives `fill`): DPP wave_shr:1, no LDS
__device__ __forceinline__ float lane_shr1(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

__device__ __forceinline__ unsigned long long ld_x(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // global_load_dwordx2 sc1
}
__device__ __forceinline__ void st_x(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               // global_store_dwordx2 sc1 (write-through)
}

// bounded relaxed poll of one progress word (wave-uniform)
__device__ __forceinline__ bool wait_ge(const unsigned *p, unsigned target, unsigned *err) {
    unsigned spins = 0;
    for (;;) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
            const unsigned e = __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (e || spins > kSpinLimit) {
                if (threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

__global__ void __launch_bounds__(64) k_sor_solve(SorArgs a) {
    const int lane = threadIdx.x;
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(a.flags + (size_t)a.nb * a.ntasks, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)(a.nb * a.ntasks)) return;
    const int job = t % a.nb, idx = t / a.nb;
    const int2 bk = a.order[idx];
    const int b = bk.x, k = bk.y;

    const float4 *__restrict__ SA = a.sa + (size_t)job * a.ent;
    const float4 *__restrict__ SB = a.sb + (size_t)job * a.ent;
    unsigned long long *X = a.x + (size_t)job * a.ent;
    unsigned *jflags = a.flags + (size_t)job * a.ntasks;
    unsigned *myflag = jflags + k * a.NB + b;
    const unsigned *f_prev = jflags + (k - 1) * a.NB + b;     // (b, k-1), valid if k > 0
    const unsigned *f_up = jflags + k * a.NB + (b - 1);       // (b-1, k), valid if b > 0

    const int W = a.W, H = a.H, RP = a.RP;
    const float omega = a.omega;
    const int r0 = 64 * b - k;
    const int r = r0 + lane;
    const bool row_ok = r >= 0 && r < H;
    const bool top_ok = r > 0, bot_ok = r < H - 1;
    // entry of this lane at step s: d = s + r0  ->  e = (s + r0 + G)*RP + r + G
    size_t e = (size_t)(r0 + a.G) * RP + (size_t)(r + a.G);

    float2 self = make_float2(0.f, 0.f);    // x^(k-1)(c, r): the previous step's right neighbour
    float2 xl = make_float2(0.f, 0.f);      // x^k(c-1, r): own previous result
    float2 xprev = make_float2(0.f, 0.f);   // own previous result, source of the next step's top via DPP
    float hl = 0.f;                         // hp(c-1, r)
    bool first = true;

    for (int ch = 0; ch < a.NCH; ch++) {
        // ---- wait for the producers of this chunk ---------------------------------------------------
        if (k > 0 && !wait_ge(f_prev, (unsigned)(ch + 1), a.err)) return;
        if (b > 0) {
            const int need = min(ch + 1 + 64 / CH, a.NCH);
            if (!wait_ge(f_up, (unsigned)need, a.err)) return;
        }
        if (first) { self = u2f(ld_x(X + e)); first = false; }
        // ---- issue every load of the chunk -----------------------------------------------------------
        float4 sa[CH], sb[CH];
        unsigned long long xr[CH], xb[CH], xt0[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const size_t ej = e + (size_t)j * RP;
            sa[j] = SA[ej];
            sb[j] = SB[ej];
            xr[j] = ld_x(X + ej + RP);            // (c+1, r)   old
            xb[j] = ld_x(X + ej + RP + 1);        // (c, r+1)   old
            xt0[j] = 0;
            if (b > 0 && lane == 0) xt0[j] = ld_x(X + ej - RP - 1);   // (c, r-1) of band b-1, new
        }
        // ---- CH dependent steps ------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int s = ch * CH + j;
            const int c = s - lane;
            const bool valid = row_ok && (unsigned)c < (unsigned)W;
            const float2 right = u2f(xr[j]), bottom = u2f(xb[j]), t0 = u2f(xt0[j]);
            float2 top;
            top.x = lane_shr1(xprev.x, t0.x);
            top.y = lane_shr1(xprev.y, t0.y);
            const float a11 = sa[j].x, a12 = sa[j].y, a22 = sa[j].z, b1 = sa[j].w;
            const float b2 = sb[j].x, hp = sb[j].y, vp = sb[j].z, vt = sb[j].w;
            float s1 = hp * right.x, s2 = hp * right.y;                                   // solver.c:337-338
            if (top_ok) { s1 = s1 + vt * top.x; s2 = s2 + vt * top.y; }
            if (bot_ok) { s1 = s1 + vp * bottom.x; s2 = s2 + vp * bottom.y; }
            s1 = s1 + b1;
            s2 = s2 + b2;
            float B1 = s1, B2 = s2;
            if (c > 0) { B1 = hl * xl.x + s1; B2 = hl * xl.y + s2; }                      // solver.c:340-341
            float2 xn;
            xn.x = self.x + omega * (a11 * B1 + a12 * B2 - self.x);                       // solver.c:342
            xn.y = self.y + omega * (a12 * B1 + a22 * B2 - self.y);                       // solver.c:343
            if (valid) st_x(X + e, f2u(xn.x, xn.y));
            xl = xn;
            xprev = xn;
            hl = hp;
            self = right;
            e += RP;
        }
        // ---- publish: every store of this wave drained, then one relaxed agent-scope flag store --------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(myflag, (unsigned)(ch + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------
// prepare: row-major planes -> diagonal-major operands; first-sweep 2x2 block inversion (solver.c:183-188
// with the row variants :101,159,214); zero guards; reset the progress words.
// ---------------------------------------------------------------------------------------------------
struct PrepArgs {
    float4 *sa; float4 *sb; unsigned long long *x; unsigned *flags;
    const float *du, *dv, *b1, *b2, *sh, *sv;
    float *a11, *a12, *a22;
    long ent, es;
    int W, H, RP, ND, G, pitch, ntasks, nb, inv_out;
};
__global__ void __launch_bounds__(256) k_sor_prepare(PrepArgs p) {
    const int job = blockIdx.z;
    const int rr = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dd = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i = threadIdx.x; i < p.ntasks; i += 256) p.flags[(size_t)job * p.ntasks + i] = 0;
        if (job == 0 && threadIdx.x == 0) p.flags[(size_t)p.nb * p.ntasks] = 0;      // ticket
    }
    if (rr >= p.RP || dd >= p.ND) return;
    const int r = rr - p.G, d = dd - p.G, c = d - r;
    const size_t e = (size_t)job * p.ent + (size_t)dd * p.RP + rr;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = make_float4(0.f, 0.f, 0.f, 0.f);
    float xu = 0.f, xv = 0.f;
    if (r >= 0 && r < p.H && c >= 0 && c < p.W) {
        const size_t o = (size_t)job * p.es + (size_t)r * p.pitch + c;
        const float hp = p.sh[o];
        const float hl = c > 0 ? p.sh[o - 1] : 0.0f;                                      // f1[0] = 0, solver.c:82
        const float vp = p.sv[o];
        const float vt = r > 0 ? p.sv[o - p.pitch] : 0.0f;
        float dpsis = hl + hp;                                                            // solver.c:101,159,214
        if (r > 0) dpsis = dpsis + vt;
        if (r < p.H - 1) dpsis = dpsis + vp;
        const float m12 = p.a12[o];
        const float A11 = p.a22[o] + dpsis, A22 = p.a11[o] + dpsis;                       // solver.c:102
        const float det = A11 * A22 - m12 * m12;
        const float i11 = __fdiv_rn(A11, det), i22 = __fdiv_rn(A22, det), i12 = __fdiv_rn(m12, -det);   // solver.c:104-106
        A = make_float4(i11, i12, i22, p.b1[o]);
        B = make_float4(p.b2[o], hp, vp, vt);
        xu = p.du[o];
        xv = p.dv[o];
        if (p.inv_out) { p.a11[o] = i11; p.a12[o] = i12; p.a22[o] = i22; }
    }
    p.sa[e] = A;
    p.sb[e] = B;
    p.x[e] = f2u(xu, xv);
}

__global__ void k_sor_finish(float *__restrict__ du, float *__restrict__ dv, const unsigned long long *__restrict__ x, long ent, long es, int W, int H,
                             int RP, int G, int pitch) {
    const int job = blockIdx.z;
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y * 4 + threadIdx.y;
    if (c >= W || r >= H) return;
    const float2 v = u2f(x[(size_t)job * ent + (size_t)(c + r + G) * RP + (r + G)]);
    const size_t o = (size_t)job * es + (size_t)r * pitch + c;
    du[o] = v.x;
    dv[o] = v.y;
}

// tiny systems: the reference itself falls back to the readable solver (solver.c:66-69, 17-57)
__global__ void k_sor_readable(float *du_, float *dv_, const float *a11_, const float *a12_, const float *a22_, const float *b1_, const float *b2_,
                               const float *sh_, const float *sv_, long es, int w, int h, int stride, int iterations, float omega) {
    if (threadIdx.x != 0) return;
    const long eb = (long)blockIdx.x * es;
    float *du = du_ + eb, *dv = dv_ + eb;
    const float *a11 = a11_ + eb, *a12 = a12_ + eb, *a22 = a22_ + eb, *b1 = b1_ + eb, *b2 = b2_ + eb, *sh = sh_ + eb, *sv = sv_ + eb;
    for (int iter = 0; iter < iterations; iter++)
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const size_t o = (size_t)j * stride + i;
                float sigma_u = 0.0f, sigma_v = 0.0f, sum_dpsis = 0.0f;
                if (j > 0) { sigma_u -= sv[o - stride] * du[o - stride]; sigma_v -= sv[o - stride] * dv[o - stride]; sum_dpsis += sv[o - stride]; }
                if (i > 0) { sigma_u -= sh[o - 1] * du[o - 1]; sigma_v -= sh[o - 1] * dv[o - 1]; sum_dpsis += sh[o - 1]; }
                if (j < h - 1) { sigma_u -= sv[o] * du[o + stride]; sigma_v -= sv[o] * dv[o + stride]; sum_dpsis += sv[o]; }
                if (i < w - 1) { sigma_u -= sh[o] * du[o + 1]; sigma_v -= sh[o] * dv[o + 1]; sum_dpsis += sh[o]; }
                const float A11 = a11[o] + sum_dpsis, A12 = a12[o], A22 = a22[o] + sum_dpsis;
                const float det = A11 * A22 - A12 * A12;
                const float B1 = b1[o] - sigma_u, B2 = b2[o] - sigma_v;
                du[o] = (1.0f - omega) * du[o] + __fdiv_rn(omega * (A22 * B1 - A12 * B2), det);
                dv[o] = (1.0f - omega) * dv[o] + __fdiv_rn(omega * (-A12 * B1 + A11 * B2), det);
            }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
int SorWorkspace::configure(sfa_ctx *c, int w_, int h_, int K_, int nb_) {
    if (ctx == c && w == w_ && h == h_ && K == K_ && nb == nb_) return SFA_OK;
    ctx = c; w = w_; h = h_; K = K_; nb = nb_;
    NB = (h + K - 1 + 63) / 64;
    G = K + 64;
    RP = round_up(h + 2 * G, 16);
    NS = w + 63;
    NCH = (NS + CH - 1) / CH;
    ND = w + 64 * NB + CH + 2 * G;
    ntasks = NB * K;
    ent = (long)ND * RP;
    SFA_TRY(sa.alloc(c, (size_t)nb * ent * sizeof(float4)));
    SFA_TRY(sb.alloc(c, (size_t)nb * ent * sizeof(float4)));
    SFA_TRY(x.alloc(c, (size_t)nb * ent * sizeof(unsigned long long)));
    SFA_TRY(flags.alloc(c, ((size_t)nb * ntasks + 16) * sizeof(unsigned)));
    SFA_TRY(order.alloc(c, (size_t)ntasks * sizeof(int2)));
    // ticket order: ascending 3*b + k; every dependency ((b,k-1): -1, (b-1,k): -3, (b-1,k-1): -4) is earlier
    std::vector<int2> ord;
    ord.reserve(ntasks);
    for (int key = 0; key <= 3 * (NB - 1) + (K - 1); key++)
        for (int b = 0; b < NB; b++) {
            const int k = key - 3 * b;
            if (k >= 0 && k < K) ord.push_back(make_int2(b, k));
        }
    SFA_HIP(c, hipMemcpyAsync(order.p, ord.data(), ord.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    SFA_HIP(c, hipStreamSynchronize(c->stream));
    return SFA_OK;
}

int sor_run(sfa_ctx *c, SorWorkspace &ws, const Geo &g, float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2,
            const float *sh, const float *sv, int K, float omega, bool inv_out) {
    if (g.w < 2 || g.h < 2 || K < 1) {                                                    // solver.c:66-69
        hipLaunchKernelGGL(k_sor_readable, dim3(g.nb), dim3(64), 0, c->stream, du, dv, a11, a12, a22, b1, b2, sh, sv, g.es, g.w, g.h, g.pitch, K, omega);
        return SFA_OK;
    }
    SFA_TRY(ws.configure(c, g.w, g.h, K, g.nb));
    PrepArgs p;
    p.sa = (float4 *)ws.sa.p; p.sb = (float4 *)ws.sb.p; p.x = (unsigned long long *)ws.x.p; p.flags = (unsigned *)ws.flags.p;
    p.du = du; p.dv = dv; p.b1 = b1; p.b2 = b2; p.sh = sh; p.sv = sv; p.a11 = a11; p.a12 = a12; p.a22 = a22;
    p.ent = ws.ent; p.es = g.es; p.W = g.w; p.H = g.h; p.RP = ws.RP; p.ND = ws.ND; p.G = ws.G; p.pitch = g.pitch;
    p.ntasks = ws.ntasks; p.nb = g.nb; p.inv_out = inv_out ? 1 : 0;
    hipLaunchKernelGGL(k_sor_prepare, dim3((ws.RP + 63) / 64, (ws.ND + 3) / 4, g.nb), dim3(256), 0, c->stream, p);

    SorArgs a;
    a.sa = p.sa; a.sb = p.sb; a.x = p.x; a.flags = p.flags; a.order = (const int2 *)ws.order.p; a.err = c->d_err;
    a.ent = ws.ent; a.W = g.w; a.H = g.h; a.K = K; a.NB = ws.NB; a.RP = ws.RP; a.G = ws.G; a.NS = ws.NS; a.NCH = ws.NCH;
    a.ntasks = ws.ntasks; a.nb = g.nb; a.omega = omega;
    const bool prof = c->profile && c->ev_used + 2 <= c->ev.size();
    if (prof) hipEventRecord(c->ev[c->ev_used], c->stream);
    hipLaunchKernelGGL(k_sor_solve, dim3(g.nb * ws.ntasks), dim3(64), 0, c->stream, a);
    if (prof) {
        hipEventRecord(c->ev[c->ev_used + 1], c->stream);
        c->ev_used += 2;
        c->sor_bytes += (44.0 * K + 12.0) * (double)g.w * g.h * g.nb;
    }
    hipLaunchKernelGGL(k_sor_finish, dim3((g.w + 63) / 64, (g.h + 3) / 4, g.nb), dim3(64, 4), 0, c->stream, du, dv, p.x, ws.ent, g.es, g.w, g.h, ws.RP,
                       ws.G, g.pitch);
    SFA_HIP(c, hipGetLastError());
    return SFA_OK;
}

}  // namespace sfa
