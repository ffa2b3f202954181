// sor.hip -- sor_coupled (solver.c:63-399) on gfx950.
//
// The reference sweeps in raster order: x(c,r)^k needs x(c-1,r)^k, x(c,r-1)^k (already updated) and
// x(c+1,r)^(k-1), x(c,r+1)^(k-1) (old).  A red-black sweep is a different algorithm (1e-2 off after 30
// sweeps); instead the SAME dependency graph is executed as a pipeline of hyperplanes:
//
//   * iteration k of row r is given the skewed row index rho = r + k.  64 consecutive rho form a band;
//     a task (band b, group g) is ONE wavefront, lane l <-> rho = 64 b + l, that carries F consecutive
//     iterations k0 = g F .. k0+F-1 FUSED: row r = 64 b - k0 + l - f for fused index f.
//   * at local step s the lane works on column c = s - l - f, so for each f the whole wave sits on one
//     anti-diagonal d = c + r of the image: all its operands are 64 CONSECUTIVE entries of the
//     diagonal-major planes built by k_sor_prepare (entry (d, r) at (d+G)*RP + r+G) -> every load and
//     store of the sweep is a coalesced 256-B..1-KB wave access.
//   * with the skew, every dependency of (b,g) points to (b,g-1) or (b-1,g): the left neighbour is the
//     lane's own previous result, the top neighbour lane l-1's previous result (DPP wave_shr:1), right /
//     bottom / self of f >= 1 are the previous step's f-1 results (registers), and only f = 0 reads
//     iteration k0-1 values back from the in-place x plane: x traffic and operand HBM traffic drop by F.
//   * tasks hand over through write-through (sc1) stores + one progress word per task, polled with sc1
//     loads (agent scope; placement independent); tickets are drawn from an atomic counter in an order
//     in which every dependency has a smaller ticket, so the pipeline cannot deadlock whatever the
//     dispatch order or residency.  Every spin is bounded.
//
// Per point the arithmetic is the fast solver's, operation for operation (solver.c:183-196), so the
// result is numerically identical to the reference's lexicographic sweep.
#include <type_traits>

#include "sfa_internal.h"
#include "sfa_device.h"
#include "sor_device.h"

#pragma clang fp contract(off)

namespace sfa {

#ifndef SFA_BAND_STARTLAG
#define SFA_BAND_STARTLAG 2          // macro chunks a band starts behind the strictly needed progress of the band above (see the prologue of band_wave)
#endif
#ifndef SFA_PUBLISH_VMCNT
#define SFA_PUBLISH_VMCNT (2 * CH - 1)
#endif

#ifndef SFA_PROGRESS_STRIDE
#define SFA_PROGRESS_STRIDE 32
#endif
constexpr int kProgressStride = SFA_PROGRESS_STRIDE;      // words between two progress words of the task / band kernels

struct SorArgs {
    const float4 *sa;               // (inv11, inv12, inv22, vt)
    const float4 *sb;               // (b1, b2, hp, vp)
    unsigned long long *x;          // (du, dv) pairs, in place
    unsigned *flags;                // [nb][NG][NB] chunks completed; flags[nb*ntasks] = ticket
    const int2 *order;              // ticket/nb -> (band, group)
    unsigned *err;
    long ent;                       // entries per batch element
    int W, H, K, NB, NG, RP, G, NS, NCH, ntasks, nb;
    int nwords;                     // progress words per window (padded)
    float omega;
    WMask active; const unsigned long long *amask;   // windows of this launch (Geo::active, Geo::amask): the others' tasks return at once
};


// One wave = task (band b, group g): 64 consecutive skewed rows rho = 64 b + l, iterations k0 = g*F .. k0+F-1 FUSED
// (the host picks an F that divides K).
// Lane l, fused index f, step s: row r = 64 b - k0 + l - f, column c = s - l - f, diagonal d = s + 64 b - k0 - 2 f.
// Within a step the F updates are independent; they consume the previous step's results:
//   left   = own result of f            top    = lane l-1's result of f      (DPP)
//   bottom = own result of f-1          right  = lane l-1's result of f-1    (the same DPP value as top of f-1)
//   self   = the previous step's right
// and for f = 0 right / bottom / self are iteration k0-1 values read from the in-place x plane.  Only the last fused
// iterate is stored by all lanes; lane 63 also stores its intermediate iterates, in place, where lane 0 of band b+1
// reads its "lane -1" values.  Operands of f >= 1 are the entries that lane l-f loaded 2f steps earlier: L2 hits.
//
// Operands are consumed from a register ring of CH step-slots; slot j is refilled for the next chunk right after step j
// has used it (prefetch distance = one chunk).  x values of the next chunk may only be fetched once the producers'
// progress words allow it; those words are polled asynchronously (never a stalling load) and a task starts LAG chunks
// behind its producers so that, at equal speed, the prefetch condition keeps holding; if it does not, the chunk falls
// back to a blocking wait.  A chunk is published one chunk late behind a counted s_waitcnt (no drain of the prefetch).
template <int F, int CH>
struct SorWave {
    // long chunks (CH = 16, the single-solve shape): a stage starts ONE chunk behind its producers and publishes the previous chunk a quarter
    // into the current one -- with 30 stages in a row the start-up lag of the pipeline is half of a lone solve's critical path
    // (single 1024x436 solve: CH 8 / LAG 3 0.88 ms, CH 16 / LAG 3 0.83, LAG 1 0.72, + early publication 0.6x; CH 4 1.37, CH 32 spills)
    static constexpr int LAG = CH >= 16 ? 1 : 3;
    static constexpr int PUB = CH >= 16 ? CH / 4 : 0;          // steps into a chunk at which the previous chunk is published (0: at its end, by the caller)
    // wave-uniform state
    const float4 *pa[F];            // SA + U - f*FOFF
    const float4 *pb[F];
    unsigned long long *px;         // X + U
    long STEP;
    int lane, W, ch0;               // ch0: first step of the current chunk
    bool row_ok[F], top_ok[F], bot_ok[F], has_up;
    float omega;
    // per-lane carried state
    float2 res[F], selfv[F];
    float hl[F];
    // operand ring
    float4 sa[F][CH], sb[F][CH];
    unsigned long long xr[CH], xb[CH], tv, tv_next;
    long tv_off;
    bool tv_lane;

    __device__ __forceinline__ void load_x_slot(int j, long step_off) {
        xr[j] = ld_x(px + step_off + STEP + lane);                 // (c+1, r)   iteration k0-1
        xb[j] = ld_x(px + step_off + STEP + 1 + lane);             // (c, r+1)   iteration k0-1
    }
    __device__ __forceinline__ void load_s_slot(int j, long step_off) {
#pragma unroll
        for (int f = 0; f < F; f++) { sa[f][j] = pa[f][step_off + lane]; sb[f][j] = pb[f][step_off + lane]; }
    }

    // CH dependent steps.  REFILL: operand slots are refilled for the next chunk; PRE: so are the x slots.
    template <bool REFILL, bool PRE>
    __device__ __forceinline__ void chunk(unsigned *myflag = nullptr, unsigned done = 0) {
        const unsigned long long tv_cur = tv;
        if (PRE && tv_lane) tv = ld_x(px + (long)CH * STEP + tv_off);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int s = ch0 + j;
            float2 sh[F];
#pragma unroll
            for (int f = 0; f < F; f++) {
                float2 t0;
                t0.x = __int_as_float(__builtin_amdgcn_readlane((int)(unsigned)(tv_cur & 0xffffffffu), f * CH + j));
                t0.y = __int_as_float(__builtin_amdgcn_readlane((int)(unsigned)(tv_cur >> 32), f * CH + j));
                sh[f].x = lane_shr1(res[f].x, t0.x);
                sh[f].y = lane_shr1(res[f].y, t0.y);
            }
            float2 nres[F];
            const float2 right0 = u2f(xr[j]), bottom0 = u2f(xb[j]);
#pragma unroll
            for (int f = 0; f < F; f++) {
                const int c = s - lane - f;
                const bool valid = row_ok[f] && (unsigned)c < (unsigned)W;
                const float2 right = f == 0 ? right0 : sh[f > 0 ? f - 1 : 0];
                const float2 bottom = f == 0 ? bottom0 : res[f > 0 ? f - 1 : 0];
                const v2f xn = sor_point(f2v(selfv[f]), f2v(right), f2v(sh[f]), f2v(bottom), f2v(res[f]), hl[f], sa[f][j], sb[f][j], omega);
                nres[f] = make_float2(xn.x, xn.y);
                unsigned long long *dst = px - (long)f * (2 * STEP + 1) + lane;
                if (f == F - 1) { if (valid) st_x(dst, f2u(xn.x, xn.y)); }
                else if (lane == 63 && valid) st_x(dst, f2u(xn.x, xn.y));               // intermediate iterate for band b+1's lane 0
                hl[f] = sb[f][j].z;
                selfv[f] = right;
            }
#pragma unroll
            for (int f = 0; f < F; f++) res[f] = nres[f];
            if (REFILL) {
                load_s_slot(j, (long)CH * STEP);
                if (PRE) load_x_slot(j, (long)CH * STEP);
            }
#pragma unroll
            for (int f = 0; f < F; f++) { pa[f] += STEP; pb[f] += STEP; }
            px += STEP;
            if (REFILL && PUB > 0 && j == PUB - 1) {
                // the previous chunk's stores are older than the 2 * PUB operand loads issued since (in-order vmcnt): this counted wait covers
                // them without draining the prefetch
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PUB - 1) : "memory");
                if (done > 0 && lane == 0) __hip_atomic_store(myflag, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        ch0 += CH;
    }
};

template <int F, int CH>
__global__ void __launch_bounds__(64) k_sor_solve(SorArgs a) {
    const int lane = threadIdx.x;
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(a.flags + (size_t)a.nb * a.nwords, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)(a.nb * a.ntasks)) return;
    const int job = t % a.nb, idx = t / a.nb;
    if (!elem_active(a.active, a.amask, job)) return;        // a passenger: none of its tasks runs, so none of them waits
    const int2 bg = a.order[idx];
    const int b = __builtin_amdgcn_readfirstlane(bg.x), g = __builtin_amdgcn_readfirstlane(bg.y);
    const int k0 = g * F;
    // one progress word per 128-byte line (kProgressStride words apart): packed, the words of a solve share a few lines and every publication
    // queues behind the others (sor_chain.hip measured that)
    unsigned *jflags = a.flags + (size_t)job * a.nwords;
    unsigned *myflag = jflags + (size_t)(g * a.NB + b) * kProgressStride;
    const unsigned *f_prev = jflags + (size_t)((g - 1) * a.NB + b) * kProgressStride;     // (b, g-1), valid if g > 0
    const unsigned *f_up = jflags + (size_t)(g * a.NB + (b - 1)) * kProgressStride;       // (b-1, g), valid if b > 0
    const int NCH = a.NCH, RP = a.RP;
    const int r0 = 64 * b - k0;
    const long FOFF = 2L * RP + 1;
    const long U = (long)(r0 + a.G) * RP + (r0 + a.G);         // entry of (step 0, f = 0, lane 0)

    SorWave<F, CH> w;
    w.lane = lane; w.W = a.W; w.ch0 = 0; w.STEP = RP; w.omega = a.omega; w.has_up = b > 0;
#pragma unroll
    for (int f = 0; f < F; f++) {
        w.pa[f] = a.sa + (size_t)job * a.ent + U - f * FOFF;
        w.pb[f] = a.sb + (size_t)job * a.ent + U - f * FOFF;
        const int r = r0 + lane - f;
        w.row_ok[f] = r >= 0 && r < a.H;
        w.top_ok[f] = r > 0;
        w.bot_ok[f] = r < a.H - 1;
        w.res[f] = make_float2(0.f, 0.f); w.selfv[f] = make_float2(0.f, 0.f); w.hl[f] = 0.f;
    }
    w.px = a.x + (size_t)job * a.ent + U;
    // lane t = f*CH + j fetches lane 0's top value of (f, step j of the chunk): entry U(s) - f*FOFF - RP - 1
    w.tv_lane = b > 0 && lane < F * CH;
    w.tv_off = (long)(lane % CH) * RP - (long)(lane / CH) * FOFF - RP - 1;
    w.tv = 0;

    unsigned known_prev = 0, known_up = 0;
    // what chunk `ch` needs from its producers (in completed chunks)
    auto need_prev = [&](int ch) { return (unsigned)min(ch + 1 + (F - 1 + CH - 1) / CH, NCH); };
    auto need_up = [&](int ch) { return (unsigned)min(ch + 1 + (64 + CH - 1) / CH, NCH); };
    auto ready = [&](int ch) { return (g == 0 || known_prev >= need_prev(ch)) && (b == 0 || known_up >= need_up(ch)); };

    // ---- prologue: start LAG chunks behind the producers, fetch chunk 0 ----------------------------------
    if (g > 0) { known_prev = wait_ge(f_prev, need_prev(SorWave<F, CH>::LAG), a.err); if (known_prev == 0xffffffffu) return; }
    if (b > 0) { known_up = wait_ge(f_up, need_up(SorWave<F, CH>::LAG), a.err); if (known_up == 0xffffffffu) return; }
    w.selfv[0] = u2f(ld_x(w.px + lane));
#pragma unroll
    for (int j = 0; j < CH; j++) { w.load_s_slot(j, (long)j * RP); w.load_x_slot(j, (long)j * RP); }
    if (w.tv_lane) w.tv = ld_x(w.px + w.tv_off);

    // progress words are polled asynchronously: the load issued at the top of chunk ch is consumed at the top of chunk
    // ch+1, so a poll never stalls the wave (the view of the producers is one chunk stale)
    unsigned pend_prev = 0, pend_up = 0;
    for (int ch = 0; ch + 1 < NCH; ch++) {
        if (g > 0) known_prev = max(known_prev, (unsigned)__builtin_amdgcn_readfirstlane(pend_prev));
        if (b > 0) known_up = max(known_up, (unsigned)__builtin_amdgcn_readfirstlane(pend_up));
        if (g > 0) pend_prev = __hip_atomic_load(f_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b > 0) pend_up = __hip_atomic_load(f_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool pre = ready(ch + 1);
        if (pre) w.template chunk<true, true>(myflag, (unsigned)ch);
        else     w.template chunk<true, false>(myflag, (unsigned)ch);
        // ---- publish the PREVIOUS chunk.  Its stores are older than the >= 2*CH operand loads this chunk's body has
        // issued since (in-order vmcnt), so this counted wait covers them without draining the prefetch.  (Long chunks did it PUB steps in.) ----
        if (SorWave<F, CH>::PUB == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SFA_PUBLISH_VMCNT) : "memory");
            if (ch > 0 && lane == 0) __hip_atomic_store(myflag, (unsigned)ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!pre) {
            // the producers were not far enough ahead: wait, then fetch the next chunk's x values
            if (g > 0) { known_prev = wait_ge(f_prev, need_prev(ch + 1), a.err); if (known_prev == 0xffffffffu) return; }
            if (b > 0) { known_up = wait_ge(f_up, need_up(ch + 1), a.err); if (known_up == 0xffffffffu) return; }
#pragma unroll
            for (int j = 0; j < CH; j++) w.load_x_slot(j, (long)j * RP);
            if (w.tv_lane) w.tv = ld_x(w.px + w.tv_off);
        }
    }
    w.template chunk<false, false>();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(myflag, (unsigned)NCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// k_sor_band: ALL K sweeps of one band in one workgroup.
//
// Workgroup = band b (64 skewed rows); its NW = K/F wavefronts are the pipeline stages: wave w carries the fused
// iterations k0 = w F .. k0+F-1 exactly as a k_sor_solve task does, but hands its last iterate to wave w+1 through an
// LDS ring (one 512-B slot per step) instead of the x plane, and learns the progress of wave w-1 from an LDS word.
// Only three things touch HBM: the operand planes (wave 0 pulls them in, waves 1.. hit L1/L2 a few steps later),
// the initial x (wave 0) / final x (wave NW-1), and lane 63's iterates of every sweep, which lane 0 of band b+1
// needs ("edge" rows, write-through + progress words as in k_sor_solve).  HBM traffic per solve drops from
// (44 K + 12) to about 52 bytes per pixel: the sweep is no longer bandwidth bound.
// Bands depend only on the band above; tickets are handed out band-major, so any residency is deadlock free.
// ---------------------------------------------------------------------------------------------------
struct BandArgs {
    const float4 *sa; const float4 *sb; unsigned long long *x;
    unsigned long long *edge;      // [nb][NB][K][Wp]  lane-63 iterates of every sweep
    unsigned *gflags;              // [nb][NB][NW] chunks whose edge stores are complete; gflags[nb*NB*NW] = ticket
    unsigned *err;
    long ent, edge_job;            // entries per batch element (diag planes / edge rows)
    int W, H, K, NB, NW, RP, G, NS, NCH, nb, Wp, EP;
    int lead;                      // steps a stage may run ahead of the next one (<= ring slots)
    float omega;
    WMask active; const unsigned long long *amask;   // windows of this launch (Geo::active, Geo::amask): the others' bands return at once
};

// What a band hands to the band below -- lane 63's iterates and the progress word -- leaves as write-through (sc1) stores.  sor_chain.hip found that ONE
// small sc1 store per 4-step interval from any wave holds up its CU's load stream (~520 cycles) and moved to no-return atomics; here a wave publishes twice
// per 12-step macro chunk and the same change measured nothing (1024x436x30, batches 8 / 16 / 32 / 64: 1255 / 1311 / 1372 / 2044 us with atomics against
// 1251 / 1300 / 1362 / 2038 with the stores), so the form the hand-off rules are documented for stays.  -DSFA_BAND_ATOMICS: the comparison build.
__device__ __forceinline__ void publish_x(unsigned long long *p, unsigned long long v) {
#ifdef SFA_BAND_ATOMICS
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    st_x(p, v);
#endif
}
__device__ __forceinline__ void publish_word(unsigned *p, unsigned v) {
#ifdef SFA_BAND_ATOMICS
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__device__ __forceinline__ bool wait_lds_ge(unsigned *p, unsigned target, unsigned *err) {
    unsigned spins = 0;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 4095u) == 0) {
            if (ld_flag(err) || spins > (kSpinLimit << 2)) {
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

#ifdef SFA_BAND_TIMING
__device__ unsigned long long g_band_timing[16 * 16 * 6];
} // namespace sfa
extern "C" int sfa_debug_band_timing(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sfa::g_band_timing), sizeof(unsigned long long) * 16 * 16 * 6); }
namespace sfa {
#endif
// ROLE of a wave in the band pipeline: 0 first stage (reads the initial x), 1 middle, 2 last (stores the final x),
// 3 the only stage (K == F).  The role is a template parameter so that the per-step code carries no role branches.
// Two granularities: operands are refilled and the LDS hand-over to the next stage happens every CH steps; the
// (slower) HBM hand-over to the band below -- progress words, lane-0 values -- every MC steps.
//
// Instruction diet of the step (the kernel is issue bound: PMC of round 2 showed ~195 instructions per wave-step, 42 of them the packed
// arithmetic of the three point updates, and the slowest SIMD -- three waves -- busy issuing for the whole step):
//   * all global streams go through buffer descriptors with a constant per-lane voffset and the step in the scalar soffset (no 64-bit
//     vector address arithmetic); lanes that sit out get an out-of-range voffset instead of an exec mask;
//   * lane 0's "lane -1" values (the band above's lane 63) reach the DPP shifts as their `old` operand straight from a broadcast LDS read
//     (BC shapes) instead of one row-shift DPP per value and step;
//   * the only lane-masked regions left in a step (lane 63's edge staging, the window's extra positions) sit behind the arithmetic and in
//     front of plain LDS / memory instructions, so no hazard nops are needed between an exec write and the next step's DPP block.
// FP: sweeps of the PREVIOUS stage (stages may differ in width: k_sor_band_mixed); k0: first sweep of this stage
template <int F, int CH, int MC, int RING, int ROLE, int FP = F>
__device__ __forceinline__ void band_wave(const BandArgs &a, unsigned long long (*ring)[RING][64], unsigned *lprog, unsigned char *win, unsigned long long (*estage)[MC],
                                          unsigned long long *tvbuf, int job, int b, int wave, int lane, int k0) {
    constexpr bool FIRST = ROLE == 0 || ROLE == 3, LASTW = ROLE == 2 || ROLE == 3;
    constexpr int NQ = MC / CH;
    const int W = a.W, H = a.H, RP = a.RP, NMC = a.NCH, NSP = a.NS;   // NS: steps padded to a multiple of MC
    const float omega = a.omega;
    const int r0 = 64 * b - k0;
    const long FOFF = 2L * RP + 1;
    const long U0 = (long)(r0 + a.G) * RP + (r0 + a.G);
    // Only sweep f = 0 reads its operands from HBM.  Sweep f at step s works on what f = 0 worked on at step s - 2f, one
    // row up per sweep (FOFF): the wave keeps its last 2(F-1) operand rows in LDS and sweeps f >= 1 read them back at
    // lane offset -f.  Rows r0-(F-1) .. r0-1 (positions 0 .. F-2 of a window row) come through a second, (F-1)-lane load per plane.
    constexpr int NSLOT = F > 1 ? 2 * (F - 1) : 1, WP = 64 + F - 1;      // window rows (steps) / positions per row
    static_assert(MC % NSLOT == 0 && MC % CH == 0 && F * MC <= 64 && (F + 1) * CH <= 64, "band shape: window rows, chunks and the edge / lane-0 fetch lanes must fit");
#ifdef SFA_X_NOLOAD       // timing experiment only: zero records, the range check drops every operand load
    const __amdgpu_buffer_rsrc_t rA = plane_rsrc(a.sa + (size_t)job * a.ent, 0);
    const __amdgpu_buffer_rsrc_t rB = plane_rsrc(a.sb + (size_t)job * a.ent, 0);
#else
    const __amdgpu_buffer_rsrc_t rA = plane_rsrc(a.sa + (size_t)job * a.ent, (unsigned long long)a.ent * 16);
    const __amdgpu_buffer_rsrc_t rB = plane_rsrc(a.sb + (size_t)job * a.ent, (unsigned long long)a.ent * 16);
#endif
    const __amdgpu_buffer_rsrc_t rX = plane_rsrc(a.x + (size_t)job * a.ent, (unsigned long long)a.ent * 8);
    float4 *winA = reinterpret_cast<float4 *>(win), *winB = winA + NSLOT * WP;   // [NSLOT][WP] each
    const unsigned vA = lane * 16u, vX = lane * 8u;                 // constant lane parts of every stream
    const bool ex_lane = F > 1 && lane < F - 1;                     // lanes that also fetch the extra positions (entries U0-(F-1) .. U0-1)
    const unsigned vE = ex_lane ? lane * 16u : kOobOffset;
    const unsigned st16 = RP * 16u, st8 = RP * 8u;
    unsigned so16 = __builtin_amdgcn_readfirstlane((unsigned)(U0 * 16)), so8 = __builtin_amdgcn_readfirstlane((unsigned)(U0 * 8));   // + step * st{16,8}
    bool row_ok_last = false;
    float2 res[F], selfv[F];
    float hl[F];
#pragma unroll
    for (int f = 0; f < F; f++) { res[f] = make_float2(0.f, 0.f); selfv[f] = make_float2(0.f, 0.f); hl[f] = 0.f; }
    if (F > 1) {
        // window rows of the steps before the start: columns < 0 (zero guards in HBM) except at the extra positions, whose
        // rows sit further up the diagonal (lane < 0: column s - lane can be >= 0 for s < 0)
        for (int i = lane; i < 2 * NSLOT * WP; i += 64) winA[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 1; t <= NSLOT; t++) {
            const float4 ea = bload16(rA, vE, so16 - (F - 1) * 16u - t * st16), eb = bload16(rB, vE, so16 - (F - 1) * 16u - t * st16);
            if (ex_lane) { winA[((NSLOT - t) % NSLOT) * WP + lane] = ea; winB[((NSLOT - t) % NSLOT) * WP + lane] = eb; }
        }
    }
    { const int r = r0 + lane - (F - 1); row_ok_last = r >= 0 && r < H; }
    // final iterate (last stage): entry of (step s, sweep F-1) = U0 + s*RP - (F-1)*FOFF + lane; lanes outside the image are sent out of range
    const unsigned so8_last = __builtin_amdgcn_readfirstlane((unsigned)((U0 - (long)(F - 1) * FOFF) * 8));
    unsigned long long *e_mine = a.edge + (size_t)job * a.edge_job + ((size_t)b * a.K + k0) * a.Wp + a.EP;
    const unsigned long long *e_up = a.edge + (size_t)job * a.edge_job + ((size_t)(b - 1) * a.K + k0) * a.Wp + a.EP;
    unsigned *gmine = a.gflags + (((size_t)job * a.NB + b) * a.NW + wave) * kProgressStride;
    const unsigned *g_up = a.gflags + (((size_t)job * a.NB + (b - 1)) * a.NW + wave) * kProgressStride;      // (b-1, w)
    const unsigned *g_up2 = g_up - kProgressStride;                                                           // (b-1, w-1)
#ifdef SFA_X_NOBAND       // timing experiment only: bands do not wait for each other
    const bool has_up = false, publishes = false;
#else
    const bool has_up = b > 0, publishes = b + 1 < a.NB;
#endif
    // lane t = fi*TN + j fetches lane 0's "lane -1" value of step j of the chunk: fi = 0: right of f = 0 (sweep
    // k0-1, column s+1); fi = f+1: top of f (sweep k0+f, column s-f)
    // BC (all shapes with (F+1)*CH <= 64 lanes): the fetch covers one CH-step chunk; the fetched values are parked in LDS (tvbuf, two chunks) and every
    // step reads its F+1 values back with wave-uniform (broadcast) addresses into the registers the DPP shifts then write into: lane 0
    // keeps the broadcast value, lanes 1.. receive their neighbour's.  Other shapes: one fetch per macro chunk, v_readlane per value.
    constexpr bool BC = (F + 1) * CH <= 64;
    constexpr int TN = BC ? CH : MC;                             // steps covered by one fetch
    const int tfi = lane / TN, tj = lane % TN;
    const bool tv_lane = has_up && lane < (F + 1) * TN && (tfi > 0 || !FIRST);
    const long tv_off = tfi == 0 ? (long)(-1) * a.Wp + tj + 1 : (long)(tfi - 1) * a.Wp + tj - (tfi - 1);
    if (BC && lane < (F + 1) * TN) tvbuf[lane] = 0ull;       // no band above: every "lane -1" value is a zero

    float4 sa0[CH], sb0[CH];
    float4 exa[CH], exb[CH];
    unsigned long long xb[CH], xr[CH], tv = 0;
    unsigned known_up = 0, known_up2 = 0, pend_up = 0, pend_up2 = 0;
    auto need_up = [&](int m) { return (unsigned)min(m + 1 + (64 + MC - 1) / MC, NMC); };
    auto need_up2 = [&](int m) { return (unsigned)min(m + 1 + (64 + FP + MC - 1) / MC, NMC); };     // the band above's PREVIOUS stage: its last sweep is FP - 1 columns behind
    auto ready = [&](int m) { return !has_up || (known_up >= need_up(m) && (FIRST || known_up2 >= need_up2(m))); };

#ifdef SFA_BAND_TIMING
    unsigned long long t_begin = __builtin_readcyclecounter(), t_up = 0, t_down = 0, t_band = 0, t_tmp = 0, t_first = 0;
#define SFA_T0() t_tmp = __builtin_readcyclecounter()
#define SFA_T1(acc) acc += __builtin_readcyclecounter() - t_tmp
#else
#define SFA_T0()
#define SFA_T1(acc)
#endif
    // ---- prologue ---------------------------------------------------------------------------------------
    if (has_up) {
        // start one macro chunk further behind the band above than strictly needed: its progress is seen one macro chunk
        // stale and published one late; at equal speed the lag of the start is the lag of the whole run
        known_up = wait_ge(g_up, need_up(SFA_BAND_STARTLAG), a.err);
        if (known_up == 0xffffffffu) return;
        if (!FIRST) { known_up2 = wait_ge(g_up2, need_up2(SFA_BAND_STARTLAG), a.err); if (known_up2 == 0xffffffffu) return; }
    }
#pragma unroll
    for (int j = 0; j < CH; j++) {
        sa0[j] = bload16(rA, vA, so16 + j * st16);
        sb0[j] = bload16(rB, vA, so16 + j * st16);
        if (F > 1) { exa[j] = bload16(rA, vE, so16 - (F - 1) * 16u + j * st16); exb[j] = bload16(rB, vE, so16 - (F - 1) * 16u + j * st16); }
    }
    if (FIRST) {
        selfv[0] = u2f(bload8(rX, vX, so8));
#pragma unroll
        for (int j = 0; j < CH; j++) {
            xr[j] = bload8(rX, vX, so8 + (j + 1) * st8);
            xb[j] = bload8(rX, vX, so8 + (j + 1) * st8 + 8u);
        }
    } else if (has_up && lane == 0) selfv[0] = u2f(ld_x(e_up - a.Wp));                  // x^(k0-1)(0, r0): band above, lane 63
    if (tv_lane) tv = ld_x(e_up + tv_off);
    // selfv[0] is the one prologue load whose first use lies inside the loop and that the loop then redefines with arithmetic: left pending, the compiler
    // guards that use with s_waitcnt vmcnt(0) in the first step of EVERY chunk (it cannot count the loads of later iterations), which drains the
    // operand prefetch once per CH steps.  Resolve it here, once.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(selfv[0].x), "+v"(selfv[0].y));

    int s0 = 0;
#ifdef SFA_BAND_TIMING
    t_first = __builtin_readcyclecounter();
#endif
    for (int m = 0; m < NMC; m++) {
        const bool last = m + 1 >= NMC;
        bool pre = false;
        if (has_up) {
            known_up = max(known_up, (unsigned)__builtin_amdgcn_readfirstlane(pend_up));
            if (!FIRST) known_up2 = max(known_up2, (unsigned)__builtin_amdgcn_readfirstlane(pend_up2));
            pre = !last && ready(m + 1);
        }
        unsigned long long tv_cur = tv;
        if (!BC && pre && tv_lane) tv = ld_x(e_up + (s0 + MC) + tv_off);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            unsigned long long *tvb = tvbuf;                              // one chunk's values: each chunk writes (in order, this wave only) before it reads
            if (BC) {                                            // this chunk's lane-0 values into LDS; fetch the next chunk's (the whole macro chunk is published)
                if (tv_lane) tvb[lane] = tv;
                if ((q < NQ - 1 || pre) && tv_lane) tv = ld_x(e_up + (s0 + CH) + tv_off);
            }
            // the band above: look at its progress words as late as possible (one chunk before the next macro chunk decides on them)
            if (q == NQ - 1 && has_up && !last) {
                pend_up = __hip_atomic_load(g_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!FIRST) pend_up2 = __hip_atomic_load(g_up2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!FIRST) {                                        // the previous stage: its last iterate through the LDS ring
                const unsigned need = (unsigned)min(s0 + CH - 1 + FP, NSP);
                SFA_T0();
                if (!wait_lds_ge(&lprog[wave - 1], need, a.err)) return;
                SFA_T1(t_up);
#pragma unroll
                for (int j = 0; j < CH; j++) xb[j] = ring[wave - 1][(s0 + j + FP - 1) & (RING - 1)][lane];
            }
            if (!LASTW) {                                        // back-pressure: do not overrun slots wave+1 has not read
                const int need = s0 + CH - 1 - a.lead - F + 2;
                SFA_T0();
                if (need > 0 && !wait_lds_ge(&lprog[wave + 1], (unsigned)need, a.err)) return;
                SFA_T1(t_down);
            }
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const int s = s0 + j, jj = q * CH + j;
                float2 sh[F], right0, bottom0;
                bottom0 = u2f(xb[j]);
                if (BC) {
                    // lane 0's neighbours: one broadcast read per value (the same address in every lane); the shifts write into these registers
                    float2 fl[F + 1];
#pragma unroll
                    for (int fi = FIRST ? 1 : 0; fi <= F; fi++) fl[fi] = u2f(tvb[fi * TN + j]);
                    if (FIRST) right0 = u2f(xr[j]);
                    else {
                        right0.x = lane_shr1(bottom0.x, fl[0].x);
                        right0.y = lane_shr1(bottom0.y, fl[0].y);
                    }
#pragma unroll
                    for (int f = 0; f < F; f++) {
                        sh[f].x = lane_shr1(res[f].x, fl[f + 1].x);
                        sh[f].y = lane_shr1(res[f].y, fl[f + 1].y);
                    }
                } else {
                    const int tvx = (int)(unsigned)(tv_cur & 0xffffffffu), tvy = (int)(unsigned)(tv_cur >> 32);
                    if (FIRST) right0 = u2f(xr[j]);
                    else {
                        right0.x = lane_shr1(bottom0.x, __int_as_float(__builtin_amdgcn_readlane(tvx, jj)));
                        right0.y = lane_shr1(bottom0.y, __int_as_float(__builtin_amdgcn_readlane(tvy, jj)));
                    }
#pragma unroll
                    for (int f = 0; f < F; f++) {
                        sh[f].x = lane_shr1(res[f].x, __int_as_float(__builtin_amdgcn_readlane(tvx, (f + 1) * MC + jj)));
                        sh[f].y = lane_shr1(res[f].y, __int_as_float(__builtin_amdgcn_readlane(tvy, (f + 1) * MC + jj)));
                    }
                }
                // operands of the trailing sweeps: the window row written 2f steps ago, f positions down
                float4 oa[F], ob[F];
                oa[0] = sa0[j]; ob[0] = sb0[j];
#pragma unroll
                for (int f = 1; f < F; f++) {
                    const int slot = (jj - 2 * f + 4 * NSLOT) % NSLOT;      // window row of step s - 2f (MC is a multiple of NSLOT: jj and s agree modulo NSLOT)
#ifdef SFA_X_NOWIN        // timing experiment only
                    oa[f] = sa0[j]; ob[f] = sb0[j]; (void)slot;
#else
                    oa[f] = winA[slot * WP + lane + (F - 1 - f)];
                    ob[f] = winB[slot * WP + lane + (F - 1 - f)];
#endif
                }
                float2 nres[F];
#pragma unroll
                for (int f = 0; f < F; f++) {
                    const float2 right = f == 0 ? right0 : sh[f > 0 ? f - 1 : 0];
                    const float2 bottom = f == 0 ? bottom0 : res[f > 0 ? f - 1 : 0];
                    const v2f xn = sor_point(f2v(selfv[f]), f2v(right), f2v(sh[f]), f2v(bottom), f2v(res[f]), hl[f], oa[f], ob[f], omega);
                    nres[f] = make_float2(xn.x, xn.y);
                    hl[f] = ob[f].z;
                    selfv[f] = right;
                }
#pragma unroll
                for (int f = 0; f < F; f++) res[f] = nres[f];
                // ---- the step's lane-masked work, behind the arithmetic and in front of the plain LDS / memory instructions -------------------
                // lane 63's iterate of every sweep is band b+1's lane-0 input (zeros outside the image land in the row pads): staged in LDS,
                // stored once per macro chunk
                if (publishes && lane == 63) {
#pragma unroll
                    for (int f = 0; f < F; f++) estage[f][jj] = f2u(res[f].x, res[f].y);
                }
#ifndef SFA_X_NOWIN
                if (F > 1) {                                      // this step's operands become window row s
                    const int slot = jj % NSLOT;
                    if (ex_lane) { winA[slot * WP + lane] = exa[j]; winB[slot * WP + lane] = exb[j]; }
                    winA[slot * WP + lane + (F - 1)] = sa0[j];
                    winB[slot * WP + lane + (F - 1)] = sb0[j];
                }
#endif
                if (!LASTW) ring[wave][s & (RING - 1)][lane] = f2u(res[F - 1].x, res[F - 1].y);
                // refill slot j for step s + CH (beyond the last step this reads zero guards)
                sa0[j] = bload16(rA, vA, so16 + CH * st16);
                sb0[j] = bload16(rB, vA, so16 + CH * st16);
                if (F > 1) { exa[j] = bload16(rA, vE, so16 - (F - 1) * 16u + CH * st16); exb[j] = bload16(rB, vE, so16 - (F - 1) * 16u + CH * st16); }
                if (FIRST) {
                    xr[j] = bload8(rX, vX, so8 + (CH + 1) * st8);
                    xb[j] = bload8(rX, vX, so8 + (CH + 1) * st8 + 8u);
                }
                // the final iterate leaves behind this step's refills: vmcnt retires in order, so a store queued ahead of the operand
                // loads would have to be acknowledged before they count as arrived (one step of prefetch depth lost to the write latency)
                if (LASTW) bstore8(rX, (row_ok_last && (unsigned)(s - lane - (F - 1)) < (unsigned)W) ? vX : kOobOffset, so8_last + (unsigned)s * st8, res[F - 1].x, res[F - 1].y);
                so16 += st16;
                so8 += st8;
            }
            s0 += CH;
            // tell wave+1 / wave-1 (LDS words are written in order behind the ring writes)
            __hip_atomic_store(&lprog[wave], (unsigned)s0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // band b+1 (HBM): the previous macro chunk's edge store is older than the CH * 2 operand loads issued since (in-order
            // vmcnt), so this counted wait covers it without draining the prefetch; its progress word goes out now, CH steps late
            if (q == 0 && publishes && m > 0) {
                // INVARIANT (checked by tools/check_publish_vmcnt.py on the ISA): at least SFA_PUBLISH_MIN VMEM instructions are issued between the
                // edge store of the previous macro chunk and this wait in every role of every shape
                if (CH >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else         asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (lane == 0) publish_word(gmine, (unsigned)m);
            }
        }
        // lane 63's iterates of this macro chunk: sweep f, steps s0-MC .. s0-1 are MC consecutive columns of edge row f, one
        // 64-byte piece per sweep in a single store (instead of F stores per step)
        if (publishes && lane < F * MC) {
            const int fi = lane / MC, j = lane % MC;
            publish_x(e_mine + (long)fi * a.Wp + (s0 - MC - 63 - fi + j), estage[fi][j]);
        }
        if (has_up && !last && !pre) {
            SFA_T0();
            known_up = wait_ge(g_up, need_up(m + 1), a.err);
            if (known_up == 0xffffffffu) return;
            if (!FIRST) { known_up2 = wait_ge(g_up2, need_up2(m + 1), a.err); if (known_up2 == 0xffffffffu) return; }
            if (tv_lane) tv = ld_x(e_up + s0 + tv_off);
            SFA_T1(t_band);
        }
    }
    if (publishes) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) publish_word(gmine, (unsigned)NMC);
    }
#ifdef SFA_BAND_TIMING
    if (job == 0 && lane == 0 && b < 16 && wave < 16) {
        unsigned long long *o = g_band_timing + (b * 16 + wave) * 6;
        o[0] = t_begin; o[1] = __builtin_readcyclecounter(); o[2] = t_up; o[3] = t_down; o[4] = t_band; o[5] = t_first;
    }
#endif
}

template <int F, int MAXW, int CH, int MC, int RING>
__global__ void __launch_bounds__(MAXW * 64) k_sor_band(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NW = a.NW;                                          // K / F pipeline stages (waves)
    unsigned long long(*ring)[RING][64] = reinterpret_cast<unsigned long long(*)[RING][64]>(smem);   // [NW-1][RING][64]
    unsigned *lprog = reinterpret_cast<unsigned *>(smem + (size_t)(NW - 1) * RING * 64 * 8);          // steps completed by each wave
    constexpr int WINB = F > 1 ? 2 * (F - 1) * (64 + F - 1) * 32 : 0;                                // operand window per wave (bytes)
    unsigned char *win0 = smem + (size_t)(NW - 1) * RING * 64 * 8 + 256;                             // 16-byte aligned behind the progress words
    unsigned long long(*est0)[MC] = reinterpret_cast<unsigned long long(*)[MC]>(win0 + (size_t)NW * WINB);   // [NW][F][MC] edge staging
    constexpr int TVB = (F + 1) * CH <= 64 ? (F + 1) * CH : 0;                                        // lane-0 values of one chunk per wave (BC shapes)
    unsigned long long *tvb0 = reinterpret_cast<unsigned long long *>(est0 + (size_t)NW * F);         // [NW][TVB]
    unsigned &s_ticket = lprog[NW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.gflags + (size_t)a.nb * a.NB * NW * kProgressStride, 1u);
    if (threadIdx.x < NW) lprog[threadIdx.x] = 0;
    __syncthreads();
    const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket);
    if (t >= (unsigned)(a.nb * a.NB)) return;
    const int job = t % a.nb, b = t / a.nb;                     // band-major tickets
    if (!elem_active(a.active, a.amask, job)) return;
    if (NW == 1)               band_wave<F, CH, MC, RING, 3>(a, ring, lprog, win0 + (size_t)wave * WINB, est0 + (size_t)wave * F, tvb0 + (size_t)wave * TVB, job, b, wave, lane, wave * F);
    else if (wave == 0)        band_wave<F, CH, MC, RING, 0>(a, ring, lprog, win0 + (size_t)wave * WINB, est0 + (size_t)wave * F, tvb0 + (size_t)wave * TVB, job, b, wave, lane, wave * F);
    else if (wave == NW - 1)   band_wave<F, CH, MC, RING, 2>(a, ring, lprog, win0 + (size_t)wave * WINB, est0 + (size_t)wave * F, tvb0 + (size_t)wave * TVB, job, b, wave, lane, wave * F);
    else                       band_wave<F, CH, MC, RING, 1>(a, ring, lprog, win0 + (size_t)wave * WINB, est0 + (size_t)wave * F, tvb0 + (size_t)wave * TVB, job, b, wave, lane, wave * F);
}

// Stages of two widths: NA stages of FA sweeps followed by NB_ stages of FB sweeps (FA >= FB, NA*FA + NB_*FB = K).  With 8 stages the waves of a workgroup sit
// two per SIMD; 6 x 4 + 2 x 3 = 30 sweeps puts 8, 8, 7, 7 sweeps on the four SIMDs where the uniform 6 x 5 shape puts 10, 10, 5, 5.
template <int FA, int NA, int FB, int NB_, int CH, int MC, int RING>
__global__ void __launch_bounds__((NA + NB_) * 64) k_sor_band_mixed(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NA + NB_;
    static_assert(FA >= FB && NB_ >= 1 && NA >= 2, "shape");
    unsigned long long(*ring)[RING][64] = reinterpret_cast<unsigned long long(*)[RING][64]>(smem);   // [NW-1][RING][64]
    unsigned *lprog = reinterpret_cast<unsigned *>(smem + (size_t)(NW - 1) * RING * 64 * 8);
    constexpr int WINA = 2 * (FA - 1) * (64 + FA - 1) * 32, WINB_ = FB > 1 ? 2 * (FB - 1) * (64 + FB - 1) * 32 : 0;
    unsigned char *win0 = smem + (size_t)(NW - 1) * RING * 64 * 8 + 256;
    unsigned long long(*est0)[MC] = reinterpret_cast<unsigned long long(*)[MC]>(win0 + (size_t)NA * WINA + (size_t)NB_ * WINB_);   // [NW][FA][MC]
    constexpr int TVB = (FA + 1) * CH;
    unsigned long long *tvb0 = reinterpret_cast<unsigned long long *>(est0 + (size_t)NW * FA);                                   // [NW][TVB]
    unsigned &s_ticket = lprog[NW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.gflags + (size_t)a.nb * a.NB * NW * kProgressStride, 1u);
    if (threadIdx.x < NW) lprog[threadIdx.x] = 0;
    __syncthreads();
    const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket);
    if (t >= (unsigned)(a.nb * a.NB)) return;
    const int job = t % a.nb, b = t / a.nb;
    if (!elem_active(a.active, a.amask, job)) return;
    unsigned char *win = wave < NA ? win0 + (size_t)wave * WINA : win0 + (size_t)NA * WINA + (size_t)(wave - NA) * WINB_;
    unsigned long long(*est)[MC] = est0 + (size_t)wave * FA;
    unsigned long long *tvb = tvb0 + (size_t)wave * TVB;
    const int k0 = wave < NA ? wave * FA : NA * FA + (wave - NA) * FB;
    if (wave == 0)                   band_wave<FA, CH, MC, RING, 0, FA>(a, ring, lprog, win, est, tvb, job, b, wave, lane, k0);
    else if (wave < NA)              band_wave<FA, CH, MC, RING, 1, FA>(a, ring, lprog, win, est, tvb, job, b, wave, lane, k0);
    else if (wave == NA && NB_ == 1) band_wave<FB, CH, MC, RING, 2, FA>(a, ring, lprog, win, est, tvb, job, b, wave, lane, k0);
    else if (wave == NA)             band_wave<FB, CH, MC, RING, 1, FA>(a, ring, lprog, win, est, tvb, job, b, wave, lane, k0);
    else if (wave == NW - 1)         band_wave<FB, CH, MC, RING, 2, FB>(a, ring, lprog, win, est, tvb, job, b, wave, lane, k0);
    else                             band_wave<FB, CH, MC, RING, 1, FB>(a, ring, lprog, win, est, tvb, job, b, wave, lane, k0);
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
    WMask active; const unsigned long long *amask;
};
// One block = a 64-column x 16-row tile: row-major reads coalesced along the columns, the 10 operand floats staged in
// LDS, then written along the tile's anti-diagonals: 16 consecutive entries (256 B of SA/SB) per diagonal.  Only valid
// (c,r) entries are written; the zero guards are set once when the workspace is (re)shaped and never touched again.
constexpr int PT_C = 64, PT_R = 16;
__global__ void __launch_bounds__(256) k_sor_prepare(PrepArgs p) {
    __shared__ float4 tA[PT_R][PT_C + 2];            // odd (row stride - 1): the 16 rows of an anti-diagonal on distinct bank groups
    __shared__ float4 tB[PT_R][PT_C + 2];
    __shared__ float2 tX[PT_R][PT_C + 3];
    const int job = blockIdx.z;
    const int c0 = blockIdx.x * PT_C, r0 = blockIdx.y * PT_R;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i = threadIdx.x; i < p.ntasks; i += 256) p.flags[(size_t)job * p.ntasks + i] = 0;
        if (job == 0 && threadIdx.x == 0) p.flags[(size_t)p.nb * p.ntasks] = 0;      // ticket
    }
    if (!elem_active(p.active, p.amask, job)) return;                   // passengers: progress words reset, operands left alone
#pragma unroll
    for (int i = 0; i < PT_R / 4; i++) {
        const int rl = ty + 4 * i, r = r0 + rl, c = c0 + tx;
        if (r < p.H && c < p.W) {
            const size_t o = (size_t)job * p.es + (size_t)r * p.pitch + c;
            const float hp = p.sh[o];
            const float hl = c > 0 ? p.sh[o - 1] : 0.0f;                                  // f1[0] = 0, solver.c:82
            const float vp = p.sv[o];
            const float vt = r > 0 ? p.sv[o - p.pitch] : 0.0f;
            float dpsis = hl + hp;                                                        // solver.c:101,159,214
            if (r > 0) dpsis = dpsis + vt;
            if (r < p.H - 1) dpsis = dpsis + vp;
            const float m12 = p.a12[o];
            const float A11 = p.a22[o] + dpsis, A22 = p.a11[o] + dpsis;                   // solver.c:102
            const float det = A11 * A22 - m12 * m12;
            const float i11 = __fdiv_rn(A11, det), i22 = __fdiv_rn(A22, det), i12 = __fdiv_rn(m12, -det);   // solver.c:104-106
            tA[rl][tx] = make_float4(i11, i12, i22, vt);                                  // vt = 0 at row 0
            tB[rl][tx] = make_float4(p.b1[o], p.b2[o], hp, r < p.H - 1 ? vp : 0.0f);      // no bottom edge in the last row
            tX[rl][tx] = make_float2(p.du[o], p.dv[o]);
            if (p.inv_out) { p.a11[o] = i11; p.a12[o] = i12; p.a22[o] = i22; }
        }
    }
    __syncthreads();
    // tile diagonals: dl = cl + rl in [0, PT_C + PT_R - 2]; 16 row slots per diagonal
    for (int item = threadIdx.x; item < (PT_C + PT_R - 1) * PT_R; item += 256) {
        const int dl = item / PT_R, rl = item % PT_R, cl = dl - rl;
        if (cl < 0 || cl >= PT_C) continue;
        const int r = r0 + rl, c = c0 + cl;
        if (r >= p.H || c >= p.W) continue;
        const size_t e = (size_t)job * p.ent + (size_t)(c + r + p.G) * p.RP + (r + p.G);
        p.sa[e] = tA[rl][cl];
        p.sb[e] = tB[rl][cl];
        const float2 xv = tX[rl][cl];
        p.x[e] = f2u(xv.x, xv.y);
    }
}

__global__ void k_sor_finish(float *__restrict__ du, float *__restrict__ dv, const unsigned long long *__restrict__ x, long ent, long es, int W, int H,
                             int RP, int G, int pitch, WMask active, const unsigned long long *__restrict__ amask) {
    const int job = blockIdx.z;
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y * 4 + threadIdx.y;
    if (c >= W || r >= H || !elem_active(active, amask, job)) return;
    const float2 v = u2f(x[(size_t)job * ent + (size_t)(c + r + G) * RP + (r + G)]);
    const size_t o = (size_t)job * es + (size_t)r * pitch + c;
    du[o] = v.x;
    dv[o] = v.y;
}

// tiny systems: the reference itself falls back to the readable solver (solver.c:66-69, 17-57)
__global__ void k_sor_readable(float *du_, float *dv_, const float *a11_, const float *a12_, const float *a22_, const float *b1_, const float *b2_,
                               const float *sh_, const float *sv_, long es, int w, int h, int stride, int iterations, float omega,
                               WMask active, const unsigned long long *amask) {
    if (threadIdx.x != 0 || !elem_active(active, amask, blockIdx.x)) return;
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
// LABELLED MODE `slow_flow_sor_order red_black` -- a DIFFERENT ALGORITHM, never the default and never presented as matching the
// reference: per sweep all points with (x + y) even are updated from the current values, then all points with (x + y) odd.  The point
// update is the fast solver's (solver.c:183-196), operation for operation, so the kernels have an exact CPU twin among the test checkers;
// the ORDER differs from solver.c, and with it the result (1e-2 .. 1e-1 after 30 sweeps).  What it buys: every point of a
// colour is independent, so one frame pair is not bound by the W + H + 2K step critical path of the raster order.
// Row-major planes as the stage API has them; one launch per colour pass.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rb_invert(float *__restrict__ a11, float *__restrict__ a12, float *__restrict__ a22, const float *__restrict__ sh,
                                                   const float *__restrict__ sv, Geo g) {
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = b * g.es + (size_t)y * g.pitch + x;
    const float hp = sh[o], hl = x > 0 ? sh[o - 1] : 0.0f;
    float dpsis = hl + hp;                                                               // solver.c:101,159,214
    if (y > 0) dpsis = dpsis + sv[o - g.pitch];
    if (y < g.h - 1) dpsis = dpsis + sv[o];
    const float m12 = a12[o];
    const float A11 = a22[o] + dpsis, A22 = a11[o] + dpsis;                               // solver.c:102
    const float det = A11 * A22 - m12 * m12;
    a11[o] = __fdiv_rn(A11, det); a22[o] = __fdiv_rn(A22, det); a12[o] = __fdiv_rn(m12, -det);   // solver.c:104-106
}
__global__ void __launch_bounds__(256) k_rb_pass(float *__restrict__ du, float *__restrict__ dv, const float *__restrict__ i11, const float *__restrict__ i12,
                                                 const float *__restrict__ i22, const float *__restrict__ b1, const float *__restrict__ b2,
                                                 const float *__restrict__ sh, const float *__restrict__ sv, Geo g, int color, float omega) {
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x = 2 * (blockIdx.x * 64 + threadIdx.x) + ((y + color) & 1);
    if (x >= g.w || y >= g.h) return;
    const size_t o = b * g.es + (size_t)y * g.pitch + x;
    const float hp = sh[o], hl = x > 0 ? sh[o - 1] : 0.0f;
    const float dur = x < g.w - 1 ? du[o + 1] : 0.0f, dvr = x < g.w - 1 ? dv[o + 1] : 0.0f;
    float s1 = hp * dur, s2 = hp * dvr;                                                   // solver.c:108,166,221
    if (y > 0) { const float vt = sv[o - g.pitch]; s1 = s1 + vt * du[o - g.pitch]; s2 = s2 + vt * dv[o - g.pitch]; }
    if (y < g.h - 1) { const float vp = sv[o]; s1 = s1 + vp * du[o + g.pitch]; s2 = s2 + vp * dv[o + g.pitch]; }
    s1 = s1 + b1[o];
    s2 = s2 + b2[o];
    float B1 = s1, B2 = s2;
    if (x > 0) { B1 = hl * du[o - 1] + s1; B2 = hl * dv[o - 1] + s2; }                   // solver.c:113-114
    const float u = du[o], v = dv[o];
    du[o] = u + omega * (i11[o] * B1 + i12[o] * B2 - u);                                  // solver.c:115-116
    dv[o] = v + omega * (i12[o] * B1 + i22[o] * B2 - v);
}
// The same algorithm as the kernel north_star names: LDS tiles with stencil halos, several sweeps per tile visit.  A block owns a 64 x 16 tile and works on
// the region around it that T sweeps can reach (halo 2 T: every colour pass consumes one ring): the (du, dv) pairs of the region sit in LDS, the seven
// operands of every cell a thread owns (six cells of each colour) in its registers, loaded once per visit.  Colour pass p updates the cells whose four
// neighbours are still exact (p + 1 cells away from every region edge that is not an image border), with k_rb_pass's operations in k_rb_pass's order, so a
// halo cell gets the very bits the neighbouring block computes for it and the whole solve is bit-identical to the one-launch-per-colour form and to the CPU
// twin.  K sweeps = ceil(K / T) launches instead of 2 K; the iterate ping-pongs between the caller's planes and a scratch pair (a visit reads its halos
// from the previous visit's output).
template <int T>
__global__ void __launch_bounds__(256) k_rb_tile(const float *__restrict__ du_in, const float *__restrict__ dv_in, long es_in, int pitch_in,
                                                 float *__restrict__ du_out, float *__restrict__ dv_out, long es_out, int pitch_out,
                                                 const float *__restrict__ i11, const float *__restrict__ i12, const float *__restrict__ i22,
                                                 const float *__restrict__ b1, const float *__restrict__ b2, const float *__restrict__ sh, const float *__restrict__ sv,
                                                 Geo g, int nsweeps, float omega) {
    constexpr int HAL = 2 * T, RW = 64 + 2 * HAL, RH = 16 + 2 * HAL, HALF = RW / 2, NC = HALF * RH, PER = (NC + 255) / 256;
    __shared__ float2 xs[RH][RW + 1];
    const int b = blockIdx.z;
    if (!elem_active(g, b)) return;
    const int tid = threadIdx.x;
    const int x0 = (int)blockIdx.x * 64 - HAL, y0 = (int)blockIdx.y * 16 - HAL;
    const float *dui = du_in + b * es_in, *dvi = dv_in + b * es_in;
    for (int idx = tid; idx < RW * RH; idx += 256) {
        const int rx = idx % RW, ry = idx / RW, gx = x0 + rx, gy = y0 + ry;
        float2 v = make_float2(0.f, 0.f);
        if (gx >= 0 && gx < g.w && gy >= 0 && gy < g.h) { const size_t o = (size_t)gy * pitch_in + gx; v = make_float2(dui[o], dvi[o]); }
        xs[ry][rx] = v;
    }
    // the cells this thread owns, colour by colour, and their operands
    struct Cell { float i11, i12, i22, b1, b2, hp, hl, vp, vt; };
    Cell cell[2][PER];
    const size_t eb = (size_t)b * g.es;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int n = j * 256 + tid, ry = n / HALF, xh = n - ry * HALF, gy = y0 + ry;
            const int rx = 2 * xh + ((c + x0 + gy) & 1), gx = x0 + rx;
            Cell q = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (n < NC && gx >= 0 && gx < g.w && gy >= 0 && gy < g.h) {
                const size_t o = eb + (size_t)gy * g.pitch + gx;
                q.i11 = i11[o]; q.i12 = i12[o]; q.i22 = i22[o]; q.b1 = b1[o]; q.b2 = b2[o];
                q.hp = sh[o]; q.hl = gx > 0 ? sh[o - 1] : 0.0f;
                q.vp = sv[o]; q.vt = gy > 0 ? sv[o - g.pitch] : 0.0f;
            }
            cell[c][j] = q;
        }
    __syncthreads();
    const bool open_l = x0 > 0, open_r = x0 + RW < g.w, open_t = y0 > 0, open_b = y0 + RH < g.h;      // region edges with image behind them
    for (int p = 0; p < 2 * nsweeps; p++) {
        const int c = p & 1;
        const int lo_x = open_l ? p + 1 : 0, hi_x = open_r ? RW - 2 - p : RW - 1, lo_y = open_t ? p + 1 : 0, hi_y = open_b ? RH - 2 - p : RH - 1;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int n = j * 256 + tid, ry = n / HALF, xh = n - ry * HALF, gy = y0 + ry;
            const int rx = 2 * xh + ((c + x0 + gy) & 1), gx = x0 + rx;
            if (n >= NC || gx < 0 || gx >= g.w || gy < 0 || gy >= g.h || rx < lo_x || rx > hi_x || ry < lo_y || ry > hi_y) continue;
            const Cell &q = c == 0 ? cell[0][j] : cell[1][j];
            const float2 self = xs[ry][rx];
            const float2 r = gx < g.w - 1 ? xs[ry][rx + 1] : make_float2(0.f, 0.f);
            float s1 = q.hp * r.x, s2 = q.hp * r.y;                                                  // solver.c:108,166,221
            if (gy > 0) { const float2 t = xs[ry - 1][rx]; s1 = s1 + q.vt * t.x; s2 = s2 + q.vt * t.y; }
            if (gy < g.h - 1) { const float2 t = xs[ry + 1][rx]; s1 = s1 + q.vp * t.x; s2 = s2 + q.vp * t.y; }
            s1 = s1 + q.b1;
            s2 = s2 + q.b2;
            float B1 = s1, B2 = s2;
            if (gx > 0) { const float2 l = xs[ry][rx - 1]; B1 = q.hl * l.x + s1; B2 = q.hl * l.y + s2; }   // solver.c:113-114
            xs[ry][rx] = make_float2(self.x + omega * (q.i11 * B1 + q.i12 * B2 - self.x), self.y + omega * (q.i12 * B1 + q.i22 * B2 - self.y));   // solver.c:115-116
        }
        __syncthreads();
    }
    float *duo = du_out + b * es_out, *dvo = dv_out + b * es_out;
    for (int idx = tid; idx < 64 * 16; idx += 256) {
        const int lx = idx & 63, ly = idx >> 6, gx = (int)blockIdx.x * 64 + lx, gy = (int)blockIdx.y * 16 + ly;
        if (gx < g.w && gy < g.h) { const float2 v = xs[HAL + ly][HAL + lx]; const size_t o = (size_t)gy * pitch_out + gx; duo[o] = v.x; dvo[o] = v.y; }
    }
}

int sor_rb_run(sfa_ctx *c, const Geo &g, float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2, const float *sh,
               const float *sv, int K, float omega) {
    if (K < 1) return SFA_OK;
    const dim3 blk(64, 4);
    hipLaunchKernelGGL(k_rb_invert, dim3((g.w + 63) / 64, (g.h + 3) / 4, g.nb), blk, 0, c->stream, a11, a12, a22, sh, sv, g);
    const bool prof = c->profile && c->ev_used + 2 <= c->ev.size();
    if (prof) (void)hipEventRecord(c->ev[c->ev_used], c->stream);
    const int mode = sw_int(Switches::RB_TILE, 5);      // sweeps per tile visit (3 or 5); 0: one launch per colour pass (the round-2 form)
#ifndef SFA_RELEASE
    if (mode == 0) {
        const dim3 grid((g.w + 127) / 128, (g.h + 3) / 4, g.nb);
        for (int k = 0; k < K; k++)
            for (int color = 0; color < 2; color++)
                hipLaunchKernelGGL(k_rb_pass, grid, blk, 0, c->stream, du, dv, a11, a12, a22, b1, b2, sh, sv, g, color, omega);
    } else
#endif
    {
        const int T = mode == 3 ? 3 : 5;
        const long pl = (long)g.pitch * g.h;
        const size_t need = (size_t)2 * g.nb * pl * sizeof(float);
        if (need > c->rb_tmp_bytes) {
            SFA_HIP(c, hipStreamSynchronize(c->stream));                                   // nobody may still be reading the old scratch
            if (c->rb_tmp) (void)hipFree(c->rb_tmp);
            c->rb_tmp = nullptr; c->rb_tmp_bytes = 0;
            SFA_HIP(c, hipMalloc(&c->rb_tmp, need));
            c->rb_tmp_bytes = need;
        }
        float *tu = static_cast<float *>(c->rb_tmp), *tv = tu + (size_t)g.nb * pl;
        const dim3 grid((g.w + 63) / 64, (g.h + 15) / 16, g.nb);
        bool in_tmp = false;                                                              // where the current iterate lives
        for (int k = 0; k < K; k += T) {
            const int ns = std::min(T, K - k);
            const float *iu = in_tmp ? tu : du, *iv = in_tmp ? tv : dv;
            float *ou = in_tmp ? du : tu, *ov = in_tmp ? dv : tv;
            const long ei = in_tmp ? pl : g.es, eo = in_tmp ? g.es : pl;
#ifndef SFA_RELEASE
            if (T == 3) hipLaunchKernelGGL((k_rb_tile<3>), grid, dim3(256), 0, c->stream, iu, iv, ei, g.pitch, ou, ov, eo, g.pitch, a11, a12, a22, b1, b2, sh, sv, g, ns, omega);
            else
#endif
                        hipLaunchKernelGGL((k_rb_tile<5>), grid, dim3(256), 0, c->stream, iu, iv, ei, g.pitch, ou, ov, eo, g.pitch, a11, a12, a22, b1, b2, sh, sv, g, ns, omega);
            in_tmp = !in_tmp;
        }
        if (in_tmp) {                                                                     // an odd number of visits: the result comes home
            Geo gc = g;
            launch_copy_planes(c, gc, du, tu, 1, g.es, pl);
            launch_copy_planes(c, gc, dv, tv, 1, g.es, pl);
        }
    }
    if (prof) {
        (void)hipEventRecord(c->ev[c->ev_used + 1], c->stream);
        c->ev_used += 2;
        c->sor_bytes += (44.0 * K + 12.0) * (double)g.w * g.h * g.nb;
    }
    SFA_HIP(c, hipGetLastError());
    return SFA_OK;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// sor_chain.hip: the few-windows pipeline (groups of stages per workgroup, I/O wave)
bool chain_shape(int id, int K, int *KG, int *NW, int *FMAX);
int chain_ch();
int chain_shift(int id);
void chain_kernel_name(int id, int NG, char *buf, size_t n);
int chain_flag_stride();
int chain_ah();
int sor_chain_launch(sfa_ctx *c, SorWorkspace &ws, const Geo &g, int K, float omega);

// which chain shape (sor_chain.hip: kChainShapes) solves nb systems of K sweeps; 0 = the band / task kernels below
static int chain_choice(int K, int nb, int NBands) {
    int id = 0;
    if (sw_given(Switches::SOR_CHAIN)) id = sw_int(Switches::SOR_CHAIN, 0);
    else if (sw_given(Switches::SOR_BAND) || sw_given(Switches::SOR_F)) id = 0;          // an explicit choice of the other kernels
    else {
        // default: by the number of bands in the launch.  Measured at 1024x436, K = 30 (8 bands per window), us per launch, with the operand ring
        // (sor_chain.hip): shape 6 (5 stages of 1 sweep) / 5 (5 x 2) / 3 (5 x 3) / the band kernel --
        //   1 window 268 / 385 / 529 / 582 (task kernel), 4: 282 / 392 / 534 / 896, 8: 407 / 414 / 551 / 1243, 16: 615 / 592 / 575 / 1291,
        //   32: 1106 / 840 / 853 / 1347, 64: 2140 / 1614 / 1502 / 2005.
        // Few bands: many short groups per band (the start-up skew of a band is what a lone solve waits for); many bands: few long groups (fewer
        // hand-overs, operands fetched from memory once per 15 sweeps).  A shape whose sweeps per group do not divide K falls back to the next one.
        const int bands = nb * NBands;
        // Round 4: shape 11 (six stages of 3,3,3,2,2,2 sweeps: with the two I/O waves eight waves, two per SIMD, at most 5 sweeps on a SIMD where 5 x 3 puts 6)
        //   16 windows 582 -> 523, 32: 859 -> 770, 64: 1515 -> 1408 (shape 12 = 2,2,2,3,3,3: 553 / 813 / 1470).
        //   shape 16 = shape 6 with one-interval poll / publication lags (sor_chain.hip kChainShapes): faster up to four windows, slower from eight on
        // Round 5: shape 14 (seven stages of 2,2,2,2,2,2,3 sweeps on nine waves: the first stage -- the one with the operand loads and the ring fill -- carries two
        //   sweeps and shares a SIMD with a 2-sweep stage, the 3-sweep LAST stage sits beside the two I/O waves: at most 4 sweeps on a SIMD where shape 11 puts 5)
        //   10 windows 455 -> 425 (1 x 5: 432), 12: 465 -> 430, 16: 480 -> 440, 32: 717 -> 670, 64: 1274 -> 1179-1200; 8 windows: 419 against 407 for 1 x 5
        static const int lone[] = {16, 6, 1, 2, 5, 3, 0}, few[] = {6, 1, 2, 5, 3, 0}, many[] = {14, 11, 3, 5, 2, 6, 1, 0};
        int KG, NW, FMAX;
        for (const int *cand = bands <= 32 ? lone : bands <= 72 ? few : many; *cand; cand++)          // 64 bands: 1 x 5 407 us, seven stages 419; 80 bands: 432 against 425
            if (chain_shape(*cand, K, &KG, &NW, &FMAX)) { id = *cand; break; }
    }
    int KG, NW, FMAX;
    if (id > 0 && !chain_shape(id, K, &KG, &NW, &FMAX)) id = 0;
    return id;
}
// band kernel (all K sweeps of a band in one workgroup): F fused sweeps per wave, NW = K/F waves; 0 = not applicable
static int band_shape(int K, int nb) {
#ifdef SFA_RELEASE
    (void)K; (void)nb;
    return 0;                                                 // release build: the task kernel is the one fallback (the band kernels are not compiled)
#endif
    // default: batches (>= 8 systems in lockstep: the two kernels tie at 8, and the band kernel leaves most CUs to a second stream) take
    // the band kernel, single solves the task kernel whose K stages
    // spread over K CUs (shorter critical path); SFA_SOR_BAND = 0 (never) / 1..3 (always, that many fused sweeps)
    int F = nb >= 8 ? 43 : 0;                                // round 2: 8 stages of 4,4,4,4,4,4,3,3 sweeps (two waves and 8,8,7,7 sweeps per SIMD) for K = 30, else the 6-stage shape
                                                             // (5 sweeps per wave), which with buffer addressing beats the 10-stage one by 5-7 %
    F = sw_int(Switches::SOR_BAND, F);
    if (F == 43) { if (K == 30) return 43; F = 5; }           // 6 x 4 + 2 x 3 sweeps: K = 30 only; otherwise the uniform shapes
    if (F <= 0 || F > 6 || F == 4) return 0;
    auto fits = [&](int f) { return f >= 1 && K % f == 0 && K / f <= (f == 6 ? 5 : f == 5 ? 6 : f == 3 ? 10 : 16); };
    if (fits(F)) return F;
    if (fits(5)) return 5;
    if (fits(3)) return 3;
    if (fits(2)) return 2;
    if (fits(1)) return 1;
    return 0;
}
// steps per operand refill / LDS hand-over (= depth of the operand prefetch) and per HBM hand-over to the band below.  Three steps
// of prefetch hide the Infinity-Cache latency of the first-touch operand rows under load (2 -> 3: -6 .. -10 % solve time);
// for the default F = 3 only: (F + 1) * MC lanes fetch the band above's values (F = 5: MC <= 10), and the 15 / 16-wave shapes of
// F = 2 / 1 have 128 registers per lane, which the deeper ring does not fit.
#ifndef SFA_BAND_CH3
#define SFA_BAND_CH3 4
#endif
#ifndef SFA_BAND_CH2
#define SFA_BAND_CH2 2
#endif
#ifndef SFA_BAND_CH5
#define SFA_BAND_CH5 4
#endif
constexpr int band_ch(int F) { return F == 3 ? SFA_BAND_CH3 : F == 5 ? SFA_BAND_CH5 : F == 6 ? 5 : F == 2 ? SFA_BAND_CH2 : 2; }
#ifndef SFA_BAND_MC3
#define SFA_BAND_MC3 12
#endif
#ifndef SFA_BAND_MC2
#define SFA_BAND_MC2 8
#endif
#ifndef SFA_BAND_MC5
#define SFA_BAND_MC5 8
#endif
constexpr int band_mc(int F) { return F == 3 ? SFA_BAND_MC3 : F == 2 ? SFA_BAND_MC2 : F == 5 ? SFA_BAND_MC5 : F == 6 ? 10 : 8; }
static int band_ring(int F) { return F == 2 ? 8 : 16; }      // LDS ring slots per wave pair, power of two
static size_t band_window(int F) { return F > 1 ? (size_t)2 * (F - 1) * (64 + F - 1) * 32 : 0; }   // operand window bytes per wave

// fused iterations per wave / steps per hand-over chunk (env SFA_SOR_F, SFA_SOR_CH override the default)
static void sor_shape(int K, int nwaves1, int &F, int &CHK) {
    // few waves (a single solve): the pipeline is latency bound, one iteration per wave is the shortest critical path;
    // many waves (batches): HBM bound, fusing two iterations halves the operand and x traffic
    F = nwaves1 >= 900 ? 2 : 1; CHK = F == 1 ? 16 : 8;       // a lone solve: long chunks, short start lag (SorWave::LAG / PUB)
    F = sw_int(Switches::SOR_F, F);
    CHK = sw_int(Switches::SOR_CH, CHK);
    if (!((F == 1 && (CHK == 8 || CHK == 16)) || (F == 2 && (CHK == 8 || CHK == 4)) || (F == 3 && CHK == 4))) { F = 2; CHK = 8; }
    if (K % F != 0) { F = 1; CHK = 16; }                     // the fused kernel carries exactly F iterations per group
}

int SorWorkspace::configure(sfa_ctx *c, int w_, int h_, int K_, int nb_) {
    int F_, CH_;
    const int chain_ = chain_choice(K_, nb_, (h_ + K_ - 1 + 63) / 64);
    const int band_ = chain_ ? 0 : band_shape(K_, nb_);
    if (chain_) { F_ = 0; CH_ = chain_ch(); }
    else if (band_ == 43) { F_ = 4; CH_ = 12; }              // mixed shape: widest stage 4 sweeps, macro chunk 12
    else if (band_) { F_ = band_; CH_ = band_mc(band_); }
    else sor_shape(K_, nb_ * ((h_ + K_ - 1 + 63) / 64) * K_, F_, CH_);
    if (ctx == c && w == w_ && h == h_ && K == K_ && nb == nb_ && F == F_ && CHK == CH_ && band == band_ && chain == chain_) return SFA_OK;
    ctx = c; w = w_; h = h_; K = K_; nb = nb_; F = F_; CHK = CH_; band = band_; chain = chain_;
    NB = (h + K - 1 + 63) / 64;
    if (chain) {
        // sor_chain.hip: NG groups of KG sweeps per band, one workgroup each; NCH = chunks per stage, NS = barrier intervals per workgroup
        int KG, NW, FMAX;
        chain_shape(chain, K, &KG, &NW, &FMAX);
        const int CH = chain_ch(), AH = chain_ah(), LEAD = AH + 1;
        NG = K / KG;
        // guard entries around the planes: at least K + 64; round 6: a multiple of 8, so that the TY = 8 consecutive entries of an anti-diagonal that a tile of
        // k_assemble_images writes (rows 8 by .. 8 by + 7: entry (r + G) of its diagonal) are ONE aligned 128-byte line of the SA / SB planes instead of a 96 + 32-byte
        // straddle of two lines that the tile below completes from another CU (-DSFA_GUARD_ALIGN=1: K + 64 as before)
#ifndef SFA_GUARD_ALIGN
#define SFA_GUARD_ALIGN 8
#endif
        G = round_up(K + 64, SFA_GUARD_ALIGN);
        RP = round_up(h + 2 * G, 16);
        NCH = round_up((w + 64 + KG - NW + 2 * CH + FMAX + CH - 1) / CH + chain_shift(chain), 4);     // + the chunks by which the groups start early
        NS = round_up(NCH + LEAD + NW + 2, AH);
        ND = NS * CH + 64 * NB + G + 32;                      // diagonals: the I/O wave reads the x plane up to interval NS, the bands sit 64 rows apart
        ntasks = NB * NG;                                     // workgroups per window
        nwords = ntasks * chain_flag_stride();                // progress words per window (one 128-byte line per workgroup); the ticket follows the last window's
        ent = (long)ND * RP;
        EP = 256;
        Wp = round_up(EP + NS * CH + 64 + 2 * K + 64, 8);
        edge_job = (long)NB * K * Wp;
        SFA_TRY(edge.alloc(c, (size_t)nb * edge_job * sizeof(unsigned long long)));
        SFA_HIP(c, hipMemsetAsync(edge.p, 0, (size_t)nb * edge_job * sizeof(unsigned long long), c->stream));
        SFA_TRY(sa.alloc(c, (size_t)nb * ent * sizeof(float4)));
        SFA_TRY(sb.alloc(c, (size_t)nb * ent * sizeof(float4)));
        SFA_TRY(x.alloc(c, (size_t)nb * ent * sizeof(unsigned long long)));
        SFA_TRY(flags.alloc(c, ((size_t)nb * nwords + 16) * sizeof(unsigned)));
        SFA_TRY(order.alloc(c, (size_t)NB * NG * sizeof(int2)));
        SFA_HIP(c, hipMemsetAsync(sa.p, 0, (size_t)nb * ent * sizeof(float4), c->stream));
        SFA_HIP(c, hipMemsetAsync(sb.p, 0, (size_t)nb * ent * sizeof(float4), c->stream));
        SFA_HIP(c, hipMemsetAsync(x.p, 0, (size_t)nb * ent * sizeof(unsigned long long), c->stream));
        std::vector<int2> ord;                               // ticket order: ascending 3 b + g; (b-1,g), (b,g-1), (b-1,g-1) all come earlier
        ord.reserve(NB * NG);
        for (int key = 0; key <= 3 * (NB - 1) + (NG - 1); key++)
            for (int b = 0; b < NB; b++) {
                const int g = key - 3 * b;
                if (g >= 0 && g < NG) ord.push_back(make_int2(b, g));
            }
        SFA_HIP(c, hipMemcpyAsync(order.p, ord.data(), ord.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
        SFA_HIP(c, hipStreamSynchronize(c->stream));
        return SFA_OK;
    }
    NG = band == 43 ? 8 : (K + F - 1) / F;
    G = K + 64;
    RP = round_up(h + 2 * G, 16);
    NS = w + 63 + F - 1;
    NCH = (NS + CHK - 1) / CHK;
    if (band) NS = NCH * CHK;                                   // the band kernel runs whole macro chunks
    ND = w + 64 * NB + 2 * CHK + F + 2 * G + 8;
    ntasks = NB * NG;                                           // band kernel: NG = NW waves per band workgroup
    nwords = ntasks * kProgressStride;
    ent = (long)ND * RP;
    if (band) {
        EP = 72;                                                    // columns s-63-f >= -66 of the unconditional lane-63 stores
        Wp = round_up(EP + NS + 2 * CHK + 8, 8);
        edge_job = (long)NB * K * Wp;
        SFA_TRY(edge.alloc(c, (size_t)nb * edge_job * sizeof(unsigned long long)));
        SFA_HIP(c, hipMemsetAsync(edge.p, 0, (size_t)nb * edge_job * sizeof(unsigned long long), c->stream));
    }
    SFA_TRY(sa.alloc(c, (size_t)nb * ent * sizeof(float4)));
    SFA_TRY(sb.alloc(c, (size_t)nb * ent * sizeof(float4)));
    SFA_TRY(x.alloc(c, (size_t)nb * ent * sizeof(unsigned long long)));
    SFA_TRY(flags.alloc(c, ((size_t)nb * nwords + 16) * sizeof(unsigned)));
    SFA_TRY(order.alloc(c, (size_t)ntasks * sizeof(int2)));
    // guards (entries outside the image) must read as zero and are never written afterwards
    SFA_HIP(c, hipMemsetAsync(sa.p, 0, (size_t)nb * ent * sizeof(float4), c->stream));
    SFA_HIP(c, hipMemsetAsync(sb.p, 0, (size_t)nb * ent * sizeof(float4), c->stream));
    SFA_HIP(c, hipMemsetAsync(x.p, 0, (size_t)nb * ent * sizeof(unsigned long long), c->stream));
    // ticket order: ascending 3*b + g; every dependency ((b,g-1): -1, (b-1,g): -3) has a smaller ticket
    std::vector<int2> ord;
    ord.reserve(ntasks);
    for (int key = 0; key <= 3 * (NB - 1) + (NG - 1); key++)
        for (int b = 0; b < NB; b++) {
            const int g = key - 3 * b;
            if (g >= 0 && g < NG) ord.push_back(make_int2(b, g));
        }
    SFA_HIP(c, hipMemcpyAsync(order.p, ord.data(), ord.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    SFA_HIP(c, hipStreamSynchronize(c->stream));
    return SFA_OK;
}

int sor_operand_target(sfa_ctx *c, SorWorkspace &ws, const Geo &g, int K, SorOperandOut *out) {
    if (g.w < 2 || g.h < 2 || K < 1) return set_error(c, SFA_ERR_ARG, "sor_operand_target: system too small for the pipelined solver");
    SFA_TRY(ws.configure(c, g.w, g.h, K, g.nb));
    out->sa = (float4 *)ws.sa.p; out->sb = (float4 *)ws.sb.p; out->x = (unsigned long long *)ws.x.p; out->flags = (unsigned *)ws.flags.p;
    out->ent = ws.ent; out->RP = ws.RP; out->G = ws.G; out->ntasks = ws.nwords; out->nb = g.nb;      // (the producer of the operands resets that many progress words per window)
    return SFA_OK;
}

static int sor_launch_solve(sfa_ctx *c, SorWorkspace &ws, const Geo &g, float *du, float *dv, int K, float omega);

int sor_run(sfa_ctx *c, SorWorkspace &ws, const Geo &g, float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2,
            const float *sh, const float *sv, int K, float omega, bool inv_out) {
    if (g.w < 2 || g.h < 2 || K < 1) {                                                    // solver.c:66-69
        hipLaunchKernelGGL(k_sor_readable, dim3(g.nb), dim3(64), 0, c->stream, du, dv, a11, a12, a22, b1, b2, sh, sv, g.es, g.w, g.h, g.pitch, K, omega,
                           g.active, g.amask);
        return SFA_OK;
    }
    SFA_TRY(ws.configure(c, g.w, g.h, K, g.nb));
    PrepArgs p;
    p.sa = (float4 *)ws.sa.p; p.sb = (float4 *)ws.sb.p; p.x = (unsigned long long *)ws.x.p; p.flags = (unsigned *)ws.flags.p;
    p.du = du; p.dv = dv; p.b1 = b1; p.b2 = b2; p.sh = sh; p.sv = sv; p.a11 = a11; p.a12 = a12; p.a22 = a22;
    p.ent = ws.ent; p.es = g.es; p.W = g.w; p.H = g.h; p.RP = ws.RP; p.ND = ws.ND; p.G = ws.G; p.pitch = g.pitch;
    p.ntasks = ws.nwords; p.nb = g.nb; p.inv_out = inv_out ? 1 : 0; p.active = g.active; p.amask = g.amask;
    hipLaunchKernelGGL(k_sor_prepare, dim3((g.w + PT_C - 1) / PT_C, (g.h + PT_R - 1) / PT_R, g.nb), dim3(256), 0, c->stream, p);
    return sor_launch_solve(c, ws, g, du, dv, K, omega);
}

int sor_run_prepared(sfa_ctx *c, SorWorkspace &ws, const Geo &g, float *du, float *dv, int K, float omega) {
    if (ws.ctx != c || ws.w != g.w || ws.h != g.h || ws.K != K || ws.nb != g.nb) return set_error(c, SFA_ERR_ARG, "sor_run_prepared: workspace not configured for this system");
    return sor_launch_solve(c, ws, g, du, dv, K, omega);
}

static int sor_launch_solve(sfa_ctx *c, SorWorkspace &ws, const Geo &g, float *du, float *dv, int K, float omega) {
    struct { float4 *sa, *sb; unsigned long long *x; unsigned *flags; } p{(float4 *)ws.sa.p, (float4 *)ws.sb.p, (unsigned long long *)ws.x.p, (unsigned *)ws.flags.p};
    SorArgs a;
    a.sa = p.sa; a.sb = p.sb; a.x = p.x; a.flags = p.flags; a.order = (const int2 *)ws.order.p; a.err = c->d_err;
    a.ent = ws.ent; a.W = g.w; a.H = g.h; a.K = K; a.NB = ws.NB; a.NG = ws.NG; a.RP = ws.RP; a.G = ws.G; a.NS = ws.NS; a.NCH = ws.NCH;
    a.ntasks = ws.ntasks; a.nwords = ws.nwords; a.nb = g.nb; a.omega = omega; a.active = g.active; a.amask = g.amask;
    const bool prof = c->profile && c->ev_used + 2 <= c->ev.size();
    if (prof) (void)hipEventRecord(c->ev[c->ev_used], c->stream);
    if (ws.chain) chain_kernel_name(ws.chain, ws.NG, c->sor_kernel, sizeof c->sor_kernel);
    else if (ws.band == 43) snprintf(c->sor_kernel, sizeof c->sor_kernel, "k_sor_band_mixed<4,6,3,2,4,12,16>");
    else if (ws.band) snprintf(c->sor_kernel, sizeof c->sor_kernel, "k_sor_band<%d>", ws.F);
    else snprintf(c->sor_kernel, sizeof c->sor_kernel, "k_sor_solve<%d,%d>", ws.F, ws.CHK);
    if (ws.chain) {
        SFA_TRY(sor_chain_launch(c, ws, g, K, omega));
    }
#ifndef SFA_RELEASE
    else if (ws.band) {
        BandArgs ba;
        ba.sa = p.sa; ba.sb = p.sb; ba.x = p.x; ba.edge = (unsigned long long *)ws.edge.p; ba.gflags = p.flags; ba.err = c->d_err;
        ba.ent = ws.ent; ba.edge_job = ws.edge_job; ba.W = g.w; ba.H = g.h; ba.K = K; ba.NB = ws.NB; ba.NW = ws.NG; ba.RP = ws.RP; ba.G = ws.G;
        ba.lead = band_ring(ws.F);
        if (sw_given(Switches::SOR_LEAD)) ba.lead = std::max(band_ch(ws.F) + 2, std::min(sw_int(Switches::SOR_LEAD, 0), band_ring(ws.F)));
        ba.NS = ws.NS; ba.NCH = ws.NCH; ba.nb = g.nb; ba.Wp = ws.Wp; ba.EP = ws.EP; ba.omega = omega; ba.active = g.active; ba.amask = g.amask;
        const dim3 bgrid(g.nb * ws.NB), bblock(ws.NG * 64);
        const size_t tvb = (ws.F + 1) * band_ch(ws.F) <= 64 ? (size_t)(ws.F + 1) * band_ch(ws.F) * 8 : 0;            // lane-0 values, one chunk per wave
        const size_t lds = (size_t)(ws.NG - 1) * band_ring(ws.F) * 64 * 8 + 256 + ws.NG * band_window(ws.F) + (size_t)ws.NG * ws.F * band_mc(ws.F) * 8 + ws.NG * tvb;
        if (ws.band == 43) {
            const size_t lds43 = (size_t)7 * 16 * 64 * 8 + 256 + 6 * (size_t)(2 * 3 * 67 * 32) + 2 * (size_t)(2 * 2 * 66 * 32) + (size_t)8 * 4 * 12 * 8 + (size_t)8 * 5 * 4 * 8;
            hipLaunchKernelGGL((k_sor_band_mixed<4, 6, 3, 2, 4, 12, 16>), bgrid, bblock, lds43, c->stream, ba);
        } else if (ws.F == 6) hipLaunchKernelGGL((k_sor_band<6, 5, band_ch(6), band_mc(6), 16>), bgrid, bblock, lds, c->stream, ba);
        else if (ws.F == 5) hipLaunchKernelGGL((k_sor_band<5, 6, band_ch(5), band_mc(5), 16>), bgrid, bblock, lds, c->stream, ba);
        else if (ws.F == 3) hipLaunchKernelGGL((k_sor_band<3, 10, band_ch(3), band_mc(3), 16>), bgrid, bblock, lds, c->stream, ba);
        else if (ws.F == 2) hipLaunchKernelGGL((k_sor_band<2, 16, band_ch(2), band_mc(2), 8>), bgrid, bblock, lds, c->stream, ba);
        else                hipLaunchKernelGGL((k_sor_band<1, 16, band_ch(1), band_mc(1), 16>), bgrid, bblock, lds, c->stream, ba);
    }
#endif
    else {
    const dim3 sgrid(g.nb * ws.ntasks), sblock(64);
    // (release build: sor_shape's two defaults -- one sweep per wave in chunks of 16 for few waves, two in chunks of 8 for many)
    if (ws.F == 1 && ws.CHK == 16)      hipLaunchKernelGGL((k_sor_solve<1, 16>), sgrid, sblock, 0, c->stream, a);
    SFA_FULL(else if (ws.F == 1)                 hipLaunchKernelGGL((k_sor_solve<1, 8>), sgrid, sblock, 0, c->stream, a);)
    else if (ws.F == 2 && ws.CHK == 8)  hipLaunchKernelGGL((k_sor_solve<2, 8>), sgrid, sblock, 0, c->stream, a);
    SFA_FULL(else if (ws.F == 2)                 hipLaunchKernelGGL((k_sor_solve<2, 4>), sgrid, sblock, 0, c->stream, a);)
    SFA_FULL(else if (ws.F == 3)                 hipLaunchKernelGGL((k_sor_solve<3, 4>), sgrid, sblock, 0, c->stream, a);)
    else return set_error(c, SFA_ERR_ARG, "sor_launch_solve: no task kernel of %d sweeps per wave in chunks of %d in this build", ws.F, ws.CHK);
    }
    if (prof) {
        (void)hipEventRecord(c->ev[c->ev_used + 1], c->stream);
        c->ev_used += 2;
        c->sor_bytes += (44.0 * K + 12.0) * (double)g.w * g.h * g.nb;
    }
    if (du)
        hipLaunchKernelGGL(k_sor_finish, dim3((g.w + 63) / 64, (g.h + 3) / 4, g.nb), dim3(64, 4), 0, c->stream, du, dv, p.x, ws.ent, g.es, g.w, g.h, ws.RP,
                           ws.G, g.pitch, g.active, g.amask);
    SFA_HIP(c, hipGetLastError());
    return SFA_OK;
}

}  // namespace sfa
