// sor_chain.hip -- sor_coupled (solver.c:63-399) for FEW systems per launch (1 .. 32 frame windows): the same hyperplane pipeline as
// sor.hip (reference raster order, IEEE == results), cut into many small workgroups so that a lone solve spreads over the chip and
// every wave carries few instructions per step.
//
// Geometry (sor.hip's): skewed rows rho = r + k; a band = 64 rho = one wavefront; a STAGE = F consecutive sweeps fused in one wave;
// at local step s the lane works on column c = s - lane - f of row r = 64 b - k0 + lane - f; operands live in diagonal-major planes.
//
// What is different here:
//   * the K sweeps of a band are split into NG GROUPS of KG sweeps; a workgroup = (window, band b, group g) = NW compute waves (the
//     group's stages) + ONE I/O wave.  Consecutive stages of a group hand the iterate over through LDS rings; consecutive groups of a
//     band through the in-place x plane; bands through the edge rows (lane 63's iterate of every sweep), as in sor.hip.
//   * GLOBAL TIME.  Stage st (first sweep k0) works at global time t on local step s = t - O, O = k0 - st + 1.  With that offset a
//     stage consumes at time t exactly what its predecessor produced at time t: ring slot = t mod ring size for writer and reader, so
//     chunk boundaries (CH steps) are the same for every stage and all LDS addresses of a chunk are lane * 8 + immediate.
//   * LOCKSTEP BY BARRIER.  The waves of a workgroup meet at one s_barrier per chunk; compute wave w runs chunk I - LEAD - w in
//     interval I.  No LDS progress words, no polling in the compute waves: their step is the operand refill (buffer loads), three LDS
//     reads, the DPP shifts, the packed point updates and one or two LDS writes.
//   * THE I/O WAVE owns everything that crosses workgroups: it polls the progress words of the three workgroups this one depends on
//     ((b-1,g), (b-1,g-1), (b,g-1)), loads the band above's lane-63 values and the previous group's iterate AH intervals ahead
//     (register FIFO, sc1 loads), parks them in LDS for the compute waves, stores this workgroup's lane-63 values and last iterate
//     (sc1 write-through) and publishes ONE progress word per workgroup behind a counted s_waitcnt vmcnt (its own VMEM stream only:
//     the compute waves' operand prefetch is never drained by a publication).
//   * every stage starts in the zero guards (s < 0), so "self", "left weight" and the first results need no special start-up.
//
// Deadlock freedom: tickets are drawn in ascending 3 b + g, every dependency has a smaller ticket; only the I/O wave ever waits for
// another workgroup, and a wait that gives up (bounded) poisons the run (error word) but still walks through every barrier.
#include <atomic>
#include "sfa_internal.h"
#include "sfa_device.h"
#include "sor_device.h"

#pragma clang fp contract(off)

namespace sfa {

struct ChainArgs {
    const float4 *sa, *sb;
    unsigned long long *x, *edge;
    unsigned *gflags;              // [nb][NB][NG] intervals published; gflags[nb*NB*NG] = ticket
    const int2 *order;             // ticket / nb -> (band, group)
    unsigned *err;
    long ent, edge_job;            // entries per window: operand planes / edge rows
    unsigned long long edge_bytes, flag_bytes;
    int W, H, K, NB, NG, RP, G, nb, Wp, EP;
    int nch;                       // chunks per stage (even)
    int NI;                        // barrier intervals per workgroup (multiple of AH)
    float omega;
    WMask active; const unsigned long long *amask;   // windows of this launch (Geo::active, Geo::amask): the others' workgroups return at once
};

// one progress word per workgroup, each on a 128-byte line of its own: every OUT wave updates its word once per interval (atomic) and up to
// three IN waves poll it; packed 4 bytes apart the words of a whole solve sat on three lines and every atomic queued behind all the others
constexpr int kFlagStride = 32;      // words
constexpr unsigned kAuxSc1 = 16;   // raw buffer builtins: aux bit 4 = sc1 (bit 0 sc0, bit 1 nt)

__device__ __forceinline__ unsigned long long bload8_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kAuxSc1);
    return (unsigned long long)v.x | ((unsigned long long)v.y << 32);
}
__device__ __forceinline__ void bstore8_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned long long v) {
    v2u t = {(unsigned)v, (unsigned)(v >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, kAuxSc1);
}
__device__ __forceinline__ unsigned bload4_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, kAuxSc1); }
__device__ __forceinline__ void bstore4_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, kAuxSc1); }

// Everything that LEAVES a workgroup goes out as no-return buffer atomics (8-byte swaps for data, a 4-byte unsigned max for the progress word).
// Measured on MI355X (tools/chain_timing.py, -DSFA_X_NOIO_POLL builds): ONE write-through store of 4 or 8 bytes per lane (sc1, nt or sc0 sc1)
// per interval from any wave of a CU holds up that CU's whole load stream for ~520 cycles (the three stages' operand prefetch: 941 -> 1527-1656
// cycles per 4-step chunk); plain stores, 16-byte sc1 stores, bypassing loads and no-return atomics cost the other waves nothing (1007-1014).
// Rows of the iterate start at odd entries for half of the stages, so 16-byte stores do not fit them; atomics execute at the memory side
// ("drop the line"), are counted in vmcnt in issue order like stores, and a consumer's sc1 loads behind the progress word see them.
// The clang builtins cover only a few buffer atomics: inline asm (no result register, so nothing for the compiler's waitcnt pass to track).
struct RsrcWords { v4u w; };
__device__ __forceinline__ RsrcWords rsrc_words(const void *base, unsigned long long bytes) {
    const unsigned long long p = (unsigned long long)base;
    RsrcWords r;
    r.w.x = __builtin_amdgcn_readfirstlane((unsigned)p);
    r.w.y = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) & 0xffffu;
    r.w.z = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xffffff00ull ? 0xffffff00ull : bytes));
    r.w.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void batomic_swap8(const RsrcWords &r, unsigned voff, unsigned soff, unsigned long long v) {
#ifdef SFA_X_OUT_NOATOM   // timing experiment only
    asm volatile("" ::"v"(v), "v"(voff), "s"(r.w), "s"(soff)); return;
#endif
    asm volatile("buffer_atomic_swap_x2 %0, %1, %2, %3 offen" ::"v"(v), "v"(voff), "s"(r.w), "s"(soff) : "memory");
}
__device__ __forceinline__ void batomic_umax4(const RsrcWords &r, unsigned voff, unsigned soff, unsigned v) {
#ifdef SFA_X_OUT_NOATOM   // timing experiment only
    asm volatile("" ::"v"(v), "v"(voff), "s"(r.w), "s"(soff)); return;
#endif
    asm volatile("buffer_atomic_umax %0, %1, %2, %3 offen" ::"v"(v), "v"(voff), "s"(r.w), "s"(soff) : "memory");
}

// one barrier interval: the LDS writes of the interval are complete before any wave passes
#define SFA_CHAIN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#ifdef SFA_CHAIN_TIMING
// per-phase wave cycles (s_memtime) of the first workgroups of window 0: [wg < 64][wave < 16][16 words]
__device__ unsigned long long g_chain_timing[64 * 16 * 16];
#define SFA_CT_STAMP(t) do { t = __builtin_readcyclecounter(); } while (0)
// stamped barrier: time spent waiting at the barrier goes to `acc`
#define SFA_CHAIN_BARRIER_T(acc) do { unsigned long long _t0, _t1; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); _t0 = __builtin_readcyclecounter(); \
    asm volatile("s_barrier" ::: "memory"); _t1 = __builtin_readcyclecounter(); acc += _t1 - _t0; } while (0)
#define SFA_CHAIN_BARRIER_T2(acc, accw) do { unsigned long long _t0, _t1, _tw = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); _t0 = __builtin_readcyclecounter(); \
    asm volatile("s_barrier" ::: "memory"); _t1 = __builtin_readcyclecounter(); acc += _t1 - _t0; accw += _t0 - _tw; } while (0)
#else
#define SFA_CHAIN_BARRIER_T(acc) SFA_CHAIN_BARRIER()
#define SFA_CHAIN_BARRIER_T2(acc, accw) SFA_CHAIN_BARRIER()
#endif

#ifndef SFA_CHAIN_OPRING
#define SFA_CHAIN_OPRING 1
#endif
// chunks by which a group with an operand ring starts early: the first stage's first step then sits at column <= 1 - KG - ... (s_start <= -1 - 4 SHIFT)
__host__ __device__ constexpr int chain_start_shift(int KG) { return KG <= 10 ? 2 : 4; }
#ifndef SFA_CHAIN_PF
#define SFA_CHAIN_PF 2
#endif
#ifndef SFA_CHAIN_PFL
#define SFA_CHAIN_PFL 3
#endif
#ifndef SFA_CHAIN_PF1
#define SFA_CHAIN_PF1 4                                              // = CH: one-sweep stages read a whole chunk's rows at its top (1: the step-ahead form)
#endif
// shape of a workgroup: NA stages of FA sweeps, then NB_ stages of FB sweeps
template <int FA, int NA, int FB, int NB_>
struct ChainShape {
    static constexpr int NW = NA + NB_, KG = NA * FA + NB_ * FB, FMAX = FA > FB ? FA : FB;
    __host__ __device__ static constexpr int Fw(int w) { return w < NA ? FA : FB; }
    __host__ __device__ static constexpr int kw(int w) { return w < NA ? w * FA : NA * FA + (w - NA) * FB; }   // sweeps of the group in front of wave w
    // Read-ahead of the operand ring (chain_compute PF): steps between the LDS read of a ring-fed sweep's operand row and its use.  Every stage of >= 2 sweeps:
    // the first stage 2 (its own trailing sweeps read rows the wave wrote itself at most 2 kappa steps ago), the later stages 3 (their rows are at least an
    // interval old); every stage of one sweep (the lone solve): the whole chunk at its top (PF1 = CH); 8 stages and more (register budget) or mixed with
    // one-sweep stages: one step.  tools/sim_sor_chain.py ring_prefetch / ring_hazards carry the same rule and check it against the barrier lockstep.
    static constexpr bool MULTI = FA >= 2 && (NB_ == 0 || FB >= 2) && NW < 8, LONE = FA == 1 && NB_ == 0 && NW < 8;
    // INFILL (the one-sweep shapes: a lone solve): a FILL wave (wave NW + 2) copies the WHOLE operand row of every step -- the KG - 1 entries above the band and the
    // band's own 64 -- from memory into the ring by LDS-DMA, four rows per interval, and the first stage reads the ring like every other stage.  With its operand
    // loads, its two 16-byte ring writes per step and the refill arithmetic the first stage needed 1 080 cycles per chunk where the other stages need 810-840
    // (tools/chain_timing.py, each wave alone on its SIMD), and everybody waited for it at the barrier.  (The same rows fetched by the IN wave through registers:
    // correct, and 8 % slower -- that wave is busy 700 of an interval's 1 270 cycles as it is.)
#ifndef SFA_CHAIN_INFILL
#define SFA_CHAIN_INFILL 1
#endif
    static constexpr bool INFILL = SFA_CHAIN_INFILL && LONE && SFA_CHAIN_PF1 == 4;
    static constexpr int NWAVES = NW + 2 + (INFILL ? 1 : 0);
    static constexpr int PF0 = MULTI ? SFA_CHAIN_PF : LONE ? SFA_CHAIN_PF1 : 1;       // first stage (one-sweep stages: ring-fed only with INFILL)
    static constexpr int PFL = MULTI ? SFA_CHAIN_PFL : LONE ? SFA_CHAIN_PF1 : 1;      // stages 1 ..
    // Which stage a wave runs (wave 0 = IN, wave NW + 1 = OUT).  Waves i, i + 4, i + 8 of a workgroup share a SIMD.  Seven stages of 3,2,2,2,2,2,2 sweeps: nine
    // waves -- SIMD 0 takes the two I/O waves and the 3-sweep first stage (wave 4), the other three SIMDs two 2-sweep stages each: at most 4 sweeps on a SIMD where
    // the six-stage shape (3,3,3,2,2,2 on eight waves) puts 5.
    static constexpr bool PERM9 = NW == 7 && NA == 1;
    // seven stages of 2,2,2,2,2,2,3 sweeps on nine waves: the LAST stage (3 ring-fed sweeps, no loads) shares SIMD 0 with the two I/O waves (wave 4); the first stage (2 sweeps +
    // the operand loads + the ring fill) and the other five pair up on SIMDs 1-3: (st0, st3) (st1, st4) (st2, st5) -- at most 4 sweeps on a SIMD
    static constexpr bool PERM9L = NW == 7 && NA == 6 && NB_ == 1;
#ifndef SFA_CHAIN_PERM6
#define SFA_CHAIN_PERM6 0
#endif
    // six stages on eight waves: which stage shares a SIMD with which (pairs of waves (0,4) (1,5) (2,6) (3,7); 0 = IN, 7 = OUT).  0: stage = wave - 1 -- (IN, st3) (st0, st4)
    // (st1, st5) (st2, OUT); 1: the first stage (the only one that loads from memory and fills the operand ring) beside the OUT wave instead of a compute wave --
    // (IN, st3) (st2, st4) (st1, st5) (st0, OUT); 2: beside the IN wave -- (IN, st0) (st3, st4) (st1, st5) (st2, OUT)
    __host__ __device__ static constexpr int stage_of_wave6(int wave) {
        return SFA_CHAIN_PERM6 == 1 ? (wave == 1 ? 2 : wave == 3 ? 0 : wave - 1) : SFA_CHAIN_PERM6 == 2 ? (wave == 4 ? 0 : wave == 1 ? 3 : wave - 1) : wave - 1;
    }
    // WHAT-IF ONLY (-DSFA_WHATIF_SHAPE20 -DSFA_X_OPR_CUT=18, tools/whatif_third_wave.sh; wrong results by construction, never shipped): TEN compute waves -- five
    // stages of 2 sweeps, then five of 1 -- with the two I/O waves twelve waves, three per SIMD: what VERDICT r5 #2 asks to spend a smaller operand ring on.  The ring
    // this shape needs (62 rows) does not fit by far; the what-if cuts it to the 44 rows the LDS has room for (a slot is then rewritten before its last read: the
    // values are wrong, the instruction stream and the LDS traffic are the real ones).  Waves 0 (IN), 4, 8 / 1, 5, 9 / 2, 6, 10 / 3, 7, 11 (OUT) share a SIMD:
    // IN + stages 1, 3 (4 sweeps) / stages 0, 5, 6 (4) / stages 2, 7, 8 (4) / stages 4, 9 + OUT (3)
    static constexpr bool PERM12 = NW == 10 && NA == 5 && NB_ == 5 && FA == 2 && FB == 1;
    __host__ __device__ static constexpr int stage_of_wave12(int wave) {
        constexpr int m[11] = {-1, 0, 2, 4, 1, 5, 7, 9, 3, 6, 8};
        return m[wave];
    }
    __host__ __device__ static constexpr int stage_of_wave(int wave) {
        return PERM12 ? stage_of_wave12(wave) : PERM9 ? (wave == 4 ? 0 : wave < 4 ? wave : wave - 1) : PERM9L ? (wave == 4 ? 6 : wave < 4 ? wave - 1 : wave - 2) : NW == 6 ? stage_of_wave6(wave) : wave - 1;
    }
};

// LDS map of a workgroup (bytes).  rings: [NW+1][2][CH][64] u64 (ring w = input of compute wave w; ring NW = output of the last one);
// tv: per wave [2][CH][F+1] u64, the band above's lane-63 values (fi = 0: "right" of sweep 0, fi = f+1: "top" of sweep f);
// es: per wave [2][F-1][CH] u64, lane 63's iterates of the sweeps that do not go through a ring; dummy: where lanes 0..62 write instead
template <class S, int CH>
struct ChainLds {
    static constexpr int RING = 2 * CH * 64 * 8;
    static constexpr int ring0 = 0;
    static constexpr int tv0 = ring0 + (S::NW + 1) * RING;
    static constexpr int TVW = 2 * CH * (S::FMAX + 1) * 8;                    // per wave
    static constexpr int es0 = tv0 + S::NW * TVW;
    static constexpr int ESW = 2 * (S::FMAX > 1 ? S::FMAX - 1 : 1) * CH * 8;   // per wave
    static constexpr int dummy0 = es0 + S::NW * ESW;
    static constexpr int DUMMY = 64 * 8 + ESW;
    // operand ring (see chain_compute): the rows of SA / SB the group's FIRST sweep reads, kept for the KG - 1 sweeps that read the same rows later --
    // sweep kappa of stage w reads, at its local step s, the row of step s + w - 2 kappa, kappa lanes down.  OPW entries per row: KG - 1 entries of the
    // band above (IN wave), then the 64 of this band (first stage); OPR rows: a row is last read 3 (NW - 1) + 2 (KG - 1) + 3 steps after the first
    // stage used it, and its slot is rewritten (IN wave: the entries of the band above) up to 5 steps before the first stage gets there again
    // Depth: with barrier lockstep (stage w runs chunk I - LEAD - w in interval I, the IN wave writes the band-above entries of chunk c at interval c + AH = LEAD - 1 + c)
    // row r is last read -- stage NW - 1, sweep KG - 1, for its step r - NW + 2 KG - 2 -- in interval LEAD + NW - 1 + floor((r + 2 KG - 2 - NW) / 4)
    // = LEAD - 1 + floor((r + 3 NW + 2 KG - 2) / 4), and its slot is first rewritten (IN wave, row r + OPR) in interval LEAD - 1 + floor((r + OPR) / 4): strictly later for
    // every r iff OPR >= 3 NW + 2 KG + 2.  OPRMIN = 3 (NW - 1) + 2 (KG - 1) + 9 is that bound + 2: two rows of margin, three where they fit (OPR10); the six-stage shapes
    // of 15 sweeps (3,3,3,2,2,2) fit the 160 KB only with two.  tools/sim_sor_chain.py ring_hazards walks every (stage, sweep, step) against these intervals: the bound is
    // tight at one step of read-ahead, and reading further ahead (chain_compute PF) only moves the last read earlier (tests/test_sor_chain_model.py).
    static constexpr int OPW = 64 + S::KG - 1, OPROWB = OPW * 16, OPRMIN = 3 * (S::NW - 1) + 2 * (S::KG - 1) + 9;
    static constexpr int ops0 = (dummy0 + DUMMY + 15) & ~15;
    static constexpr bool OPR10 = ops0 + 2 * (OPRMIN + 1) * OPROWB + 32 <= 160 * 1024, OPR00 = ops0 + 2 * OPRMIN * OPROWB + 32 <= 160 * 1024;
    // a shape that does not fit with the margin takes the bound itself, lowered by what its later stages read ahead beyond one step (the last read of a row -- last
    // stage, last sweep -- comes PFL - 1 steps earlier): 3,2,2,2,2,2,2 needs 51 rows and 51 fit
    static constexpr int OPRTIGHT = 3 * S::NW + 2 * S::KG + 2 - (S::PFL < CH ? S::PFL - 1 : 0);
#ifndef SFA_X_OPR_CUT      // timing experiment only (rows BELOW the derived minimum: a slot may be rewritten before its last read)
#define SFA_X_OPR_CUT 0
#endif
    // INFILL: a group of four rows (4 c .. 4 c + 3) is in flight from the start of interval c to the end of interval c + 1 (chain_fill), where the IN wave wrote the
    // entries above the band in interval c + AH: the slot has to be free AH intervals = 4 AH rows earlier; a multiple of four rows, so that a group never wraps
    static constexpr int OPRFILL = (OPRMIN + 1 + 4 * 2 + 3) / 4 * 4;
    static constexpr int OPR = S::INFILL ? OPRFILL : (OPR10 ? OPRMIN + 1 : OPR00 ? OPRMIN : OPRTIGHT) - SFA_X_OPR_CUT, OPPLANE = OPR * OPROWB;
    static constexpr bool OPRING = SFA_CHAIN_OPRING && S::KG <= 16 && ops0 + 2 * OPPLANE + 32 <= 160 * 1024;
    //  the depth a shape ends up with is never below the write-after-read bound of ITS read-ahead, whatever -DSFA_CHAIN_PF / PFL / PF1 experiment or later
    // edit of the shape table is compiled: a ring that is too shallow races silently (a slot rewritten before its last read), and until now only the Python model
    // (tools/sim_sor_chain.py ring_hazards) stood between such a build and the GPU.  A shape at the tight depth leans on the later stages' read-ahead (MULTI: PFL steps).
    static_assert(SFA_X_OPR_CUT != 0 || !OPRING || S::INFILL || OPR >= 3 * S::NW + 2 * S::KG + 2 - (S::PFL < CH ? S::PFL - 1 : 0), "operand ring below its write-after-read bound");
    static_assert(SFA_X_OPR_CUT != 0 || !OPRING || S::INFILL || OPR10 || OPR00 || (S::MULTI && S::PFL >= 2 && S::PFL < CH), "the tight ring depth is derived for multi-sweep stages that read ahead");
    static_assert(!OPRING || !S::INFILL || (OPR % 4 == 0 && OPR >= OPRMIN + 1 + 8), "FILL wave: groups of four rows never wrap, and a group is in flight for two intervals");
    static constexpr int ticket = ops0 + (OPRING ? 2 * OPPLANE : 0);
    static constexpr int total = ticket + 16;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// compute wave: stage with F fused sweeps
// ---------------------------------------------------------------------------------------------------------------------------------
// a * b with a = the HIGH half of the pair `hi2` broadcast (vt = SA.w, vp = SB.w).  The compiler folds low-half broadcasts into op_sel but
// copies a high half into a fresh register first -- and schedules that copy right behind the load of the operand, which turns the
// P-step prefetch into a stall.  One instruction, no copy:
__device__ __forceinline__ v2f pk_mul_hi(v2f hi2, v2f b) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(hi2), "v"(b));
    return r;
}
// The SOR point update (solver.c:337-343), the same operations in the same order as sor_device.h's sor_point: one rounding per
// operation, no FMA.  hlz = the operand pair (hp, vp) of the PREVIOUS column: its low half is this point's left weight.
template <bool SCALAR_T2 = true>
__device__ __forceinline__ v2f sor_point2(v2f self, v2f right, v2f top, v2f bottom, v2f left, v2f hlz, const float4 &SA, const float4 &SB, float omega) {
    const v2f SAxy = {SA.x, SA.y}, SAzw = {SA.z, SA.w}, SBxy = {SB.x, SB.y}, SBzw = {SB.z, SB.w};
#ifdef SFA_X_NOARITH      // timing experiment only: one dependent operation instead of fifteen
    return self + right * SAxy + top * SBxy + bottom * SAzw + left * hlz;
#endif
    v2f s = __builtin_shufflevector(SBzw, SBzw, 0, 0) * right;       // hp * x_right
    s = s + pk_mul_hi(SAzw, top);                                    // + vt * x_top
    s = s + pk_mul_hi(SBzw, bottom);                                 // + vp * x_bottom
    s = s + SBxy;                                                    // + b
    const v2f B = __builtin_shufflevector(hlz, hlz, 0, 0) * left + s;
    // (a12 B2, a22 B2) as two SCALAR products: written as `{SA.y * B.y, SA.z * B.y}` the vectoriser packs them into one v_pk_mul_f32 whose first operand, the
    // pair (a12, a22), straddles two register pairs of the 16-byte operand and has to be composed by two v_mov_b32 first -- 8 issue cycles instead of 4 per point.
    // Stages of ONE sweep (the lone solve's shape) keep the packed form: their step is a single dependent chain, the moves are off it, and one packed product
    // on the chain is shorter than two scalar ones (one window 263 against 275 us)
    v2f t2;
    if (SCALAR_T2) {
        float t2x, t2y;
        asm("v_mul_f32 %0, %1, %2" : "=v"(t2x) : "v"(SA.y), "v"(B.y));
        asm("v_mul_f32 %0, %1, %2" : "=v"(t2y) : "v"(SA.z), "v"(B.y));
        t2 = (v2f){t2x, t2y};
    } else
        t2 = (v2f){SA.y * B.y, SA.z * B.y};
    const v2f t = SAxy * __builtin_shufflevector(B, B, 0, 0) + t2;
    return self + omega * (t - self);
}

// P0 = PD * CH steps of operand prefetch in registers for sweep 0 (first touch of the rows: Infinity Cache / HBM), P1 for the trailing
// sweeps (the rows this CU fetched 2 f steps earlier: L2).  PD even: the ring parity of a chunk is its position in the unrolled body.
// A slot is refilled one step AFTER its use: the next column's left weight (hp of this column) is read from it in place.
// ROLE 0: every sweep loads its operands from memory (shapes without an operand ring).  ROLE 1 (the group's first stage): sweep 0 loads from memory and
// leaves every row in the LDS ring when it uses it; ROLE 2 and the trailing sweeps of ROLE 1 read the ring, one step ahead of their use (kap0 = sweeps of
// the group in front of this stage, w = the stage).  Measured first (a what-if build that read arbitrary LDS rows): the operand loads of the trailing sweeps,
// not arithmetic or the hand-over between workgroups, were what a launch of several windows waited for -- 16 windows 942 -> 642 us.
// PF: steps by which the ring-fed sweeps read their operand rows ahead of their use (register slots NSL).  One step ahead the 16-byte LDS reads are consumed
// ~10 instructions after their issue and the wave stalls on them in every step -- with two waves per SIMD nobody covers that; two steps ahead they have a whole
// step to land.  PF = 2 is safe where every stage has >= 2 sweeps (tools/sim_sor_chain.py ring_hazards: a 1-sweep stage would read a row before it is written).
template <int F, int CH, int PD, bool SHORT = false, int ROLE = 0, int OPS0 = 0, int OPR = 1, int OPROWB = 0, int KG = 1, int PF_ = 1>
__device__ __forceinline__ void chain_compute(const ChainArgs &a, unsigned char *lds, int ring_in, int ring_out, int tvb, int esb, int dummy, int job, int b,
                                              int k0, int s_start, int lead, int lane, int kap0 = 0, int w = 0) {
    constexpr int OPPLANE = OPR * OPROWB;
    constexpr int F0 = ROLE == 2 ? 0 : 1;                            // sweeps of this wave that load from memory (ROLE 0: all of them, below)
    static_assert(PD % 2 == 0, "the chunk parity must be a compile-time constant of the body position");
    // ring lengths divide the body (PD * CH steps): wide stages (F >= 4: a whole band's sweeps in one workgroup, large batches, slow steps) keep one chunk
    constexpr int P0 = (F >= 4 || SHORT) ? CH : PD * CH, P1 = (F <= 2 && !SHORT) ? P0 : CH;
    const int RP = a.RP;
    const float omega = a.omega;
    const int r0 = 64 * b - k0;
    const long FOFF = 2L * RP + 1;
    const long E0 = (long)(r0 + a.G) * RP + (r0 + a.G) + (long)s_start * RP - (long)(F - 1) * FOFF;   // entry of (first step, sweep F-1, lane 0)
#ifdef SFA_X_NOLOAD       // timing experiment only: zero records, the range check drops every operand load
    const __amdgpu_buffer_rsrc_t rA = plane_rsrc(a.sa + (size_t)job * a.ent + E0, 0);
    const __amdgpu_buffer_rsrc_t rB = plane_rsrc(a.sb + (size_t)job * a.ent + E0, 0);
#else
    const __amdgpu_buffer_rsrc_t rA = plane_rsrc(a.sa + (size_t)job * a.ent + E0, 0xffffff00ull);
    const __amdgpu_buffer_rsrc_t rB = plane_rsrc(a.sb + (size_t)job * a.ent + E0, 0xffffff00ull);
#endif
    unsigned vo[F];
#pragma unroll
    for (int f = 0; f < F; f++) vo[f] = (unsigned)((lane + (F - 1 - f) * FOFF) * 16);
    const unsigned st16 = (unsigned)RP * 16u;
    unsigned so = 0;                                                  // byte offset of the step about to be computed

    float4 sa0[ROLE == 2 ? 1 : P0], sb0[ROLE == 2 ? 1 : P0], sa1[(F > 1 && ROLE == 0) ? F - 1 : 1][ROLE == 0 ? P1 : 1], sb1[(F > 1 && ROLE == 0) ? F - 1 : 1][ROLE == 0 ? P1 : 1];
    if (ROLE != 2) {
#pragma unroll
        for (int j = 0; j < P0; j++) { sa0[j] = bload16(rA, vo[0], j * st16); sb0[j] = bload16(rB, vo[0], j * st16); }
    }
    if (ROLE == 0) {
#pragma unroll
        for (int f = 1; f < F; f++)
#pragma unroll
            for (int j = 0; j < P1; j++) { sa1[f - 1][j] = bload16(rA, vo[f], j * st16); sb1[f - 1][j] = bload16(rB, vo[f], j * st16); }
    }
    // operand ring: byte offset of the row the first stage writes next / each ring-fed sweep reads next (uniform; rows wrap at OPPLANE), the lane's place in
    // a row, and the operands of the current and the next step
    unsigned wr_off = 0, rd_off[F];
    unsigned rd_lane[F];
    // PF == CH ("chunk-ahead", the one-sweep stages of the lone-solve shapes): at the top of chunk c -- right behind the barrier -- a ring-fed stage reads the rows of
    // its steps 4 c + 1 .. 4 c + 4 in one go: all of them are in the ring by then (row i + w - 2 kappa, written by the first stage in its own chunk, at least one
    // interval earlier for every stage w >= 1 with kappa >= w: ring_hazards, mode "chunk"), step 4 c runs on what the previous chunk's top fetched, and no step waits
    // for an LDS round trip on its critical path any more.  One step ahead the two 16-byte reads were issued behind the step's result and consumed ~2 instructions
    // later: every step of a lone solve -- a single dependent chain per wave -- carried a full LDS latency.
    constexpr bool CHUNK_AHEAD = PF_ == CH && ROLE == 2;
    static_assert(!(PF_ == CH && ROLE == 1 && F > 1), "the first stage's trailing sweeps read rows the wave itself wrote at most 2 kappa steps ago");
    constexpr int PF = (PF_ == CH && !CHUNK_AHEAD) ? 1 : PF_;         // (the first stage of a chunk-ahead shape has no ring-fed sweep at all)
    constexpr int NSL = CHUNK_AHEAD ? 2 * CH : PF == 1 ? 2 : 4;       // register slots per ring-fed sweep: a power of two that divides the unrolled body
    static_assert(PF >= 1 && PF < NSL && (PD * CH) % NSL == 0 && (CHUNK_AHEAD || CH % NSL == 0), "slot of a step = its position in the unrolled body");
    constexpr int PRE = CHUNK_AHEAD ? 1 : PF;                         // rows read before the first barrier: those of steps 0 .. PRE - 1
    float4 la[F][NSL], lb[F][NSL];
#pragma unroll
    for (int f = 0; f < F; f++) {
        const int kap = kap0 + f;
        // the ring's base is part of the lane's address (and opaque: the compiler otherwise keeps it in the instruction's offset field, where base + plane do not
        // fit 16 bits, and pays a second v_add_u32 per read pair)
        rd_lane[f] = (unsigned)(OPS0 + (KG - 1 + lane - kap) * 16);
        if (OPPLANE < 65536) asm volatile("" : "+v"(rd_lane[f]));     // (then the second plane is the 16-bit offset of the same address)
        rd_off[f] = (unsigned)(((w - 2 * kap) % OPR + OPR) % OPR * OPROWB);      // the row of step 0: step w - 2 kappa of the first stage (not written yet: zeros)
#pragma unroll
        for (int q = 0; q < NSL; q++) la[f][q] = lb[f][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ROLE != 0 && f >= F0) {
#pragma unroll
            for (int q = 0; q < PRE; q++) {                                 // the rows of steps 0 .. PRE - 1
                la[f][q] = *reinterpret_cast<const float4 *>(lds + rd_lane[f] + rd_off[f]);
                lb[f][q] = *reinterpret_cast<const float4 *>(lds + rd_lane[f] + rd_off[f] + OPPLANE);
                rd_off[f] = rd_off[f] + OPROWB == (unsigned)OPPLANE ? 0u : rd_off[f] + OPROWB;
            }
        }
    }
    const unsigned wr_lane = (unsigned)(OPS0 + (KG - 1 + lane) * 16);
    float2 res[F], selfv[F];
    v2f hlz[F];                                                       // (hp, vp) of the previous column; the first step sits in the zero guards
#pragma unroll
    for (int f = 0; f < F; f++) { res[f] = make_float2(0.f, 0.f); selfv[f] = make_float2(0.f, 0.f); hlz[f] = (v2f){0.f, 0.f}; }

    const unsigned long long *rin = reinterpret_cast<const unsigned long long *>(lds + ring_in) + lane;
    unsigned long long *rout = reinterpret_cast<unsigned long long *>(lds + ring_out) + lane;
    const unsigned long long *tvp = reinterpret_cast<const unsigned long long *>(lds + tvb);
    // lane 63's iterates of sweeps 0 .. F-2 go to the edge staging; the other lanes write into a dummy region (no exec masking)
    unsigned long long *esp = reinterpret_cast<unsigned long long *>(lds + (lane == 63 ? esb : dummy + lane * 8));

    int I = 0;
#ifdef SFA_CHAIN_TIMING
    unsigned long long t_begin = __builtin_readcyclecounter(), t_lead = 0, t_bar = 0, t_first = 0, t_last = 0, t_ldsrd = 0, t_wr = 0;
#endif
    // wave priority of the compute waves against the I/O waves of their SIMD (0 = none; the I/O waves stay at 0)
#ifndef SFA_CHAIN_CPRIO
#define SFA_CHAIN_CPRIO 0
#endif
    if (SFA_CHAIN_CPRIO) __builtin_amdgcn_s_setprio(SFA_CHAIN_CPRIO);
    for (; I < lead; I++) SFA_CHAIN_BARRIER_T(t_lead);
#ifdef SFA_CHAIN_TIMING
    t_first = __builtin_readcyclecounter();
#endif
    const int nbody = a.nch / PD;
    for (int body = 0; body < nbody; body++) {
#pragma unroll
        for (int q = 0; q < PD; q++) {
            SFA_CHAIN_BARRIER_T2(t_bar, t_wr);
            const int par = q & 1;
            // narrow stages read the chunk's LDS inputs at its top (one exposed LDS latency per chunk); wide ones (F >= 4: several waves per SIMD hide
            // it) step by step, which keeps (CH - 1) * (F + 2) register pairs free
            constexpr bool JIT = F >= 4;
            unsigned long long bot[CH], fl[CH][F + 1];
            if (!JIT) {
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    bot[j] = rin[(par * CH + j) * 64];
#pragma unroll
                    for (int fi = 0; fi <= F; fi++) fl[j][fi] = tvp[(par * CH + j) * (F + 1) + fi];
                }
            }
            if (CHUNK_AHEAD) {                                       // the rows of the chunk's steps 1 .. CH - 1 and of the next chunk's first step
#pragma unroll
                for (int f = 0; f < F; f++)
#pragma unroll
                    for (int j = 1; j <= CH; j++) {
                        la[f][(q * CH + j) % NSL] = *reinterpret_cast<const float4 *>(lds + rd_lane[f] + rd_off[f]);
                        lb[f][(q * CH + j) % NSL] = *reinterpret_cast<const float4 *>(lds + rd_lane[f] + rd_off[f] + OPPLANE);
                        rd_off[f] = rd_off[f] + OPROWB == (unsigned)OPPLANE ? 0u : rd_off[f] + OPROWB;
                    }
            }
#ifdef SFA_CHAIN_TIMING
            {   // how long the chunk's LDS reads take (diagnosis only: the wait is forced here)
                unsigned long long ta = __builtin_readcyclecounter();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                t_ldsrd += __builtin_readcyclecounter() - ta;
            }
#endif
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const int j0 = (q * CH + j) % P0, j1 = (q * CH + j) % P1;   // ring slots of this step
                const int p0 = (j0 + P0 - 1) % P0, p1 = (j1 + P1 - 1) % P1;   // ... and of the previous one
                if (JIT) {
                    bot[j] = rin[(par * CH + j) * 64];
#pragma unroll
                    for (int fi = 0; fi <= F; fi++) fl[j][fi] = tvp[(par * CH + j) * (F + 1) + fi];
                    asm volatile("" ::: "memory");             // the reads of the next step stay behind this step's arithmetic
                }
                const float2 bottom0 = u2f(bot[j]);
                float2 sh[F], right0;
                { const float2 t = u2f(fl[j][0]); right0.x = lane_shr1(bottom0.x, t.x); right0.y = lane_shr1(bottom0.y, t.y); }
#pragma unroll
                for (int f = 0; f < F; f++) { const float2 t = u2f(fl[j][f + 1]); sh[f].x = lane_shr1(res[f].x, t.x); sh[f].y = lane_shr1(res[f].y, t.y); }
                float2 nres[F];
#pragma unroll
                for (int f = 0; f < F; f++) {
                    const float2 right = f == 0 ? right0 : sh[f > 0 ? f - 1 : 0];
                    const float2 bottom = f == 0 ? bottom0 : res[f > 0 ? f - 1 : 0];
                    const bool ringfed = ROLE != 0 && f >= F0;
                    const float4 &SA = ringfed ? la[f][(CHUNK_AHEAD ? q * CH + j : j) % NSL] : f == 0 ? sa0[ROLE == 2 ? 0 : j0] : sa1[(f > 0 && ROLE == 0) ? f - 1 : 0][ROLE == 0 ? j1 : 0];
                    const float4 &SB = ringfed ? lb[f][(CHUNK_AHEAD ? q * CH + j : j) % NSL] : f == 0 ? sb0[ROLE == 2 ? 0 : j0] : sb1[(f > 0 && ROLE == 0) ? f - 1 : 0][ROLE == 0 ? j1 : 0];
                    const v2f xn = sor_point2<(F >= 2)>(f2v(selfv[f]), f2v(right), f2v(sh[f]), f2v(bottom), f2v(res[f]), hlz[f], SA, SB, omega);
                    nres[f] = make_float2(xn.x, xn.y);
                    selfv[f] = right;
                }
#pragma unroll
                for (int f = 0; f < F; f++) res[f] = nres[f];
#pragma unroll
                for (int f = 0; f + 1 < F; f++) esp[(par * (F - 1) + f) * CH + j] = f2u(res[f].x, res[f].y);
                rout[(par * CH + j) * 64] = f2u(res[F - 1].x, res[F - 1].y);
#ifdef SFA_X_NORINGWR     // timing experiment only: the first stage does not fill the ring
                if (false) {
#else
                if (ROLE == 1) {                                     // the row this step used goes to the ring (the later sweeps of the group read it there)
#endif
                    *reinterpret_cast<float4 *>(lds + wr_lane + wr_off) = sa0[j0];
                    *reinterpret_cast<float4 *>(lds + wr_lane + wr_off + OPPLANE) = sb0[j0];
                    wr_off = wr_off + OPROWB == (unsigned)OPPLANE ? 0u : wr_off + OPROWB;
                }
                // the PREVIOUS step's slots are refilled now (beyond the last step: zero guards); this step's (hp, vp) stay readable in place
                if (ROLE != 2) { sa0[p0] = bload16(rA, vo[0], so + (P0 - 1) * st16); sb0[p0] = bload16(rB, vo[0], so + (P0 - 1) * st16); }
                if (ROLE == 0) {
#pragma unroll
                    for (int f = 1; f < F; f++) { sa1[f - 1][p1] = bload16(rA, vo[f], so + (P1 - 1) * st16); sb1[f - 1][p1] = bload16(rB, vo[f], so + (P1 - 1) * st16); }
                }
#pragma unroll
                for (int f = 0; f < F; f++) {
                    const bool ringfed = ROLE != 0 && f >= F0;
                    if (ringfed) {
                        hlz[f] = (v2f){lb[f][(CHUNK_AHEAD ? q * CH + j : j) % NSL].z, lb[f][(CHUNK_AHEAD ? q * CH + j : j) % NSL].w};
#ifdef SFA_X_NORINGRD     // timing experiment only: the ring-fed sweeps keep the operands they have (what do the 16-byte LDS reads cost?)
                        if (false) {
#else
                        if (!CHUNK_AHEAD) {
#endif
                            la[f][(j + PF) % NSL] = *reinterpret_cast<const float4 *>(lds + rd_lane[f] + rd_off[f]);              // the row of step + PF
                            lb[f][(j + PF) % NSL] = *reinterpret_cast<const float4 *>(lds + rd_lane[f] + rd_off[f] + OPPLANE);
#ifndef SFA_X_NOWRAP      // timing experiment only: every step reads the same row (no row arithmetic at all: the upper bound of what a cheaper ring addressing could win)
                            rd_off[f] = rd_off[f] + OPROWB == (unsigned)OPPLANE ? 0u : rd_off[f] + OPROWB;
#endif
                        }
                    } else if (f == 0) hlz[0] = (v2f){sb0[ROLE == 2 ? 0 : j0].z, sb0[ROLE == 2 ? 0 : j0].w};
                    else hlz[f] = (v2f){sb1[(f > 0 && ROLE == 0) ? f - 1 : 0][ROLE == 0 ? j1 : 0].z, sb1[(f > 0 && ROLE == 0) ? f - 1 : 0][ROLE == 0 ? j1 : 0].w};
                }
                so += st16;
            }
        }
    }
    I += a.nch;
#ifdef SFA_CHAIN_TIMING
    t_last = __builtin_readcyclecounter();
    if (job == 0 && lane == 0 && (int)blockIdx.x < 64) {
        unsigned long long *o = g_chain_timing + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16;
        o[0] = t_begin; o[1] = t_first; o[2] = t_last; o[3] = t_lead; o[4] = t_bar; o[5] = (unsigned long long)b; o[6] = (unsigned long long)k0; o[7] = (unsigned long long)a.nch;
        o[8] = t_ldsrd; o[9] = t_wr; o[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[14] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    for (; I < a.NI; I++) SFA_CHAIN_BARRIER();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// I/O wave
// ---------------------------------------------------------------------------------------------------------------------------------
// bounded wait for one progress word through the flags descriptor; `dead` (uniform): a wait has given up, never wait again
__device__ __forceinline__ unsigned chain_wait(__amdgpu_buffer_rsrc_t rF, unsigned voff, unsigned soff, unsigned target, unsigned *err, bool &dead) {
    unsigned spins = 0, limit = kSpinLimit;
    for (;;) {
        const unsigned v = __builtin_amdgcn_readfirstlane(bload4_sc1(rF, voff, soff));
        if (v >= target || dead) return v;
        if (spins == 0) {                                   // slow path only: err[1] != 0 shortens the bound (test hook sfa_ctx_set_wait_bound)
            const unsigned o = ld_flag(err + 1);
            if (o) limit = o;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0 || spins > limit) {
            const unsigned e = ld_flag(err);
            if (e || spins > limit) {
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                return v;
            }
        }
    }
}

// What the two I/O waves share: where the workgroup's stages sit
template <class S, int CH>
struct ChainGeo {
    int RP, K, W, H, st0, k0g;
    long FOFF;
    int s_start0, s_startl, r0l;
    long U00, U0l;
    __device__ __forceinline__ ChainGeo(const ChainArgs &a, int b, int g, int c_first) {
        constexpr int NW = S::NW, Fl = S::Fw(NW - 1);
        RP = a.RP; K = a.K; W = a.W; H = a.H; FOFF = 2L * RP + 1;
        st0 = g * NW; k0g = g * S::KG;
        const int O0 = k0g - st0 + 1, r00 = 64 * b - k0g;
        s_start0 = c_first * CH - O0;
        U00 = (long)(r00 + a.G) * RP + (r00 + a.G);
        const int k0l = k0g + S::kw(NW - 1), Ol = k0l - (st0 + NW - 1) + 1;
        s_startl = c_first * CH - Ol; r0l = 64 * b - k0l;
        U0l = (long)(r0l + a.G) * RP + (r0l + a.G);
        (void)Fl;
    }
};

// ---- IN wave: progress words of the producers, the band above's lane-63 values and the previous group's iterate -> LDS ------------------
// AH: intervals between a load and its use (register FIFO); PL: intervals between a poll and the look at its result;
// PUBD: intervals between a store and the progress word that covers it (OUT wave) -- the consumer's thresholds are in published counts,
// so only the producer knows PUBD.
template <class S, int CH, int AH, int PL>
__device__ __forceinline__ void chain_in(const ChainArgs &a, unsigned char *lds, int job, int b, int g, int c_first, int c_first_prev, int lane) {
    using L = ChainLds<S, CH>;
    static_assert(AH % 2 == 0 && PL <= AH, "the LDS parities must be compile-time constants of the body position");
    constexpr int NW = S::NW, LEAD = AH + 1;
#ifndef SFA_CHAIN_HYST
#define SFA_CHAIN_HYST 2
#endif
    constexpr unsigned HYST = SFA_CHAIN_HYST;
    constexpr int NTV = [] { int n = 0; for (int w = 0; w < NW; w++) n += (S::Fw(w) + 1) * CH; return n; }();      // values fetched from the band above per interval
    constexpr int NLT = (NTV + 63) / 64;
    const ChainGeo<S, CH> G(a, b, g, c_first);
    const int RP = G.RP, K = G.K;
    const bool has_up = b > 0, has_prev = g > 0;
    const __amdgpu_buffer_rsrc_t rE = plane_rsrc(a.edge, a.edge_bytes);
    const __amdgpu_buffer_rsrc_t rF = plane_rsrc(a.gflags, a.flag_bytes);
    // x plane as the first stage reads it: base = entry of (first step, sweep 0, lane 0) of stage st0
    const __amdgpu_buffer_rsrc_t rXin = plane_rsrc(a.x + (size_t)job * a.ent + G.U00 + (long)G.s_start0 * RP, 0xffffff00ull);

    // values of the band above, fetched at interval I for the chunk c_first + I - w of stage w
    unsigned tv_voff[NLT], tv_lds[NLT][2];
    unsigned tvx_voff = kOobOffset;                       // stage 0 of the whole solve: "right" of sweep 0 for lane 0 is the INITIAL x(c + 1, r0)
    bool tvx_lane[NLT];
#pragma unroll
    for (int n = 0; n < NLT; n++) {
        const int i = n * 64 + lane;
        int w = 0, base = 0;
        while (w < NW - 1 && i >= base + (S::Fw(w) + 1) * CH) { base += (S::Fw(w) + 1) * CH; w++; }
        const int rel = i - base, fi = rel / CH, j = rel % CH;
        const bool on = i < NTV;
        const int k0w = G.k0g + S::kw(w), Ow = k0w - (G.st0 + w) + 1;
        const int s0 = (c_first - w) * CH - Ow;                                    // first local step of the chunk fetched at interval 0
        const int row = k0w - 1 + fi, col = s0 + j + (fi == 0 ? 1 : -(fi - 1));
        tvx_lane[n] = on && G.st0 + w == 0 && fi == 0;
        const long off = (long)job * a.edge_job + ((long)(b - 1) * K + row) * a.Wp + a.EP + col;
        tv_voff[n] = (on && has_up && !tvx_lane[n]) ? (unsigned)(off * 8) : kOobOffset;
        if (tvx_lane[n]) tvx_voff = (unsigned)(((long)(s0 - G.s_start0 + j + 1) * RP) * 8);   // entry U00 + (s + 1) * RP relative to rXin's base
#pragma unroll
        for (int p = 0; p < 2; p++)                                                // p = parity of the WRITE interval; chunk parity = (p + w) & 1
            tv_lds[n][p] = on ? (unsigned)(L::tv0 + w * L::TVW + ((((p + w) & 1) * CH + j) * (S::Fw(w) + 1) + fi) * 8) : (unsigned)(L::dummy0 + lane * 8);
    }
#ifdef SFA_X_NOXLOAD      // timing experiment only
    const unsigned vXin = kOobOffset;
#else
    const unsigned vXin = (unsigned)(RP + lane + 1) * 8u;                          // (c, r + 1) of sweep 0: one diagonal further, one row down
#endif
    const unsigned st8 = (unsigned)RP * 8u;

    // progress words: published-interval counts of the producers (one word per workgroup)
    const unsigned vF0 = lane == 0 ? 0u : kOobOffset;
    const unsigned so_mine = (unsigned)((((size_t)job * a.NB + b) * a.NG + g) * 4 * kFlagStride);
    const unsigned so_up = so_mine - (unsigned)a.NG * 4u * kFlagStride, so_up2 = so_up - 4u * kFlagStride, so_prev = so_mine - 4u * kFlagStride;
    const unsigned vF_up = has_up ? vF0 : kOobOffset, vF_up2 = (has_up && has_prev) ? vF0 : kOobOffset, vF_prev = has_prev ? vF0 : kOobOffset;
    // what the loads issued at interval I need (see the header; a producer's stage w computes chunk c in its interval c - c_first + LEAD + w,
    // the OUT wave stores it one interval later, and a published count v says: the stores of all intervals < v are complete)
    const int D63 = 63 / CH;
    const int need_up0 = LEAD + 3 + D63;                                            // + I
    const int dcf = c_first - c_first_prev;
    const int need_up20 = 1 + D63 + dcf + LEAD + NW - 1 + 2;                        // + I   ((b-1, g-1): its last stage, chunk c_first + I + 1 + D63)
    const int need_prev0 = dcf + LEAD + NW - 1 + 2;                                 // + I   ((b, g-1): its last stage, chunk c_first + I)
    const unsigned cap = (unsigned)(a.nch + LEAD + NW);                              // at this count a producer has published every real chunk
    unsigned known_up = 0, known_up2 = 0, known_prev = 0, pend_up[AH], pend_up2[AH], pend_prev[AH];
    bool dead = false;
    unsigned long long tvr[AH][NLT], tvxr[AH], xr[AH][CH];
#pragma unroll
    for (int q = 0; q < AH; q++) {
        tvxr[q] = 0; pend_up[q] = 0; pend_up2[q] = 0; pend_prev[q] = 0;
#pragma unroll
        for (int n = 0; n < NLT; n++) tvr[q][n] = 0;
#pragma unroll
        for (int j = 0; j < CH; j++) xr[q][j] = 0;
    }
    unsigned so_tv = 0, so_xin = 0;                                  // advance by one chunk per interval
    unsigned char *const ldsb = lds;
    // operand ring (ChainLds): the KG - 1 entries above the band of the rows the first stage reads -- chunk I of its local steps is fetched at interval I
    // (lane = step of the chunk x entry) and written AH intervals later, one interval before the first stage uses (and writes) the rows themselves
    constexpr bool OPX = L::OPRING && !S::INFILL;                    // (INFILL: the FILL wave brings whole rows)
    const int xj = lane >> 4, xe = lane & 15;
    const bool ox_on = OPX && xj < CH && xe < S::KG - 1;
    const long ox_e = G.U00 + (long)G.s_start0 * RP - (S::KG - 1);
    const __amdgpu_buffer_rsrc_t rOA = plane_rsrc(a.sa + (size_t)job * a.ent + ox_e, 0xffffff00ull);
    const __amdgpu_buffer_rsrc_t rOB = plane_rsrc(a.sb + (size_t)job * a.ent + ox_e, 0xffffff00ull);
    const unsigned ox_voff = ox_on ? (unsigned)((xj * RP + xe) * 16) : kOobOffset;
    float4 oxa[AH], oxb[AH];
#pragma unroll
    for (int q = 0; q < AH; q++) oxa[q] = oxb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned ox_so = 0;
    int ox_row4 = 0;                                                 // ring row of the first step of the chunk written next
#ifdef SFA_CHAIN_TIMING
    unsigned long long t_begin = __builtin_readcyclecounter(), t_bar = 0, t_slow = 0, n_slow = 0, t0, t1;
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();      // 100 MHz, the same clock on every XCD
#endif
    for (int I0 = 0; I0 < a.NI; I0 += AH) {
#pragma unroll
        for (int q = 0; q < AH; q++) {
            const int I = I0 + q;
            const int p = q & 1;
            SFA_CHAIN_BARRIER_T(t_bar);
            // ---- what was fetched AH intervals ago goes to LDS: the compute waves use it in the NEXT interval --------------------------
#pragma unroll
            for (int n = 0; n < NLT; n++) *reinterpret_cast<unsigned long long *>(ldsb + tv_lds[n][p]) = tvx_lane[n] ? tvxr[q] : tvr[q][n];
#pragma unroll
            for (int j = 0; j < CH; j++) reinterpret_cast<unsigned long long *>(ldsb + L::ring0 + ((p * CH + j) * 64) * 8)[lane] = xr[q][j];
            if (OPX && I >= AH) {
                int row = ox_row4 + xj;
                if (row >= L::OPR) row -= L::OPR;
                if (ox_on) {
                    *reinterpret_cast<float4 *>(ldsb + L::ops0 + row * L::OPROWB + xe * 16) = oxa[q];
                    *reinterpret_cast<float4 *>(ldsb + L::ops0 + L::OPPLANE + row * L::OPROWB + xe * 16) = oxb[q];
                }
                ox_row4 = ox_row4 + CH >= L::OPR ? ox_row4 + CH - L::OPR : ox_row4 + CH;
            }
            // ---- the workgroups this one depends on: polls issued PL intervals ago -------------------------------------------------------
            known_up = max(known_up, (unsigned)__builtin_amdgcn_readfirstlane(pend_up[(q + AH - PL) % AH]));
            known_up2 = max(known_up2, (unsigned)__builtin_amdgcn_readfirstlane(pend_up2[(q + AH - PL) % AH]));
            known_prev = max(known_prev, (unsigned)__builtin_amdgcn_readfirstlane(pend_prev[(q + AH - PL) % AH]));
            if (!dead) {
                const unsigned n_up = min((unsigned)(need_up0 + I), cap), n_up2 = min((unsigned)(need_up20 + I), cap), n_prev = min((unsigned)(need_prev0 + I), cap);
                const bool w1 = has_up && known_up < n_up, w2 = has_up && has_prev && known_up2 < n_up2, w3 = has_prev && known_prev < n_prev;
                if (w1 || w2 || w3) {
#ifdef SFA_CHAIN_TIMING
                    SFA_CT_STAMP(t0); n_slow++;
#endif
                    // a blocking wait means this workgroup has caught up with a producer: fall HYST intervals further back, so that the next
                    // intervals are covered by the asynchronous polls again instead of paying a synchronous round trip each
                    if (w1) known_up = chain_wait(rF, vF0, so_up, min(n_up + HYST, cap), a.err, dead);
                    if (w2) known_up2 = chain_wait(rF, vF0, so_up2, min(n_up2 + HYST, cap), a.err, dead);
                    if (w3) known_prev = chain_wait(rF, vF0, so_prev, min(n_prev + HYST, cap), a.err, dead);
#ifdef SFA_CHAIN_TIMING
                    SFA_CT_STAMP(t1); t_slow += t1 - t0;
#endif
                }
            }
#ifndef SFA_X_IN_NOLOADS   // (timing experiment: no loads at all)
            pend_up[q] = bload4_sc1(rF, vF_up, so_up);
            pend_up2[q] = bload4_sc1(rF, vF_up2, so_up2);
            pend_prev[q] = bload4_sc1(rF, vF_prev, so_prev);
            // ---- fetch for the chunk the compute waves run in interval I + AH + 1 ---------------------------------------------------------
#pragma unroll
            for (int n = 0; n < NLT; n++) tvr[q][n] = bload8_sc1(rE, tv_voff[n], so_tv);
            tvxr[q] = bload8_sc1(rXin, tvx_voff, so_xin);
#pragma unroll
            for (int j = 0; j < CH; j++) xr[q][j] = bload8_sc1(rXin, vXin, so_xin + (unsigned)j * st8);
#endif
            if (OPX) {
                oxa[q] = bload16(rOA, ox_voff, ox_so);
                oxb[q] = bload16(rOB, ox_voff, ox_so);
                ox_so += (unsigned)(CH * RP) * 16u;
            }
            so_tv += CH * 8u; so_xin += CH * st8;
        }
    }
#ifdef SFA_CHAIN_TIMING
    if (job == 0 && lane == 0 && (int)blockIdx.x < 64) {
        unsigned long long *o = g_chain_timing + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16;
        o[0] = t_begin; o[1] = __builtin_readcyclecounter(); o[2] = t_bar; o[6] = t_slow; o[8] = n_slow; o[3] = rt_begin; o[4] = __builtin_amdgcn_s_memrealtime();
        o[9] = (unsigned long long)b; o[10] = (unsigned long long)g; o[11] = (unsigned long long)a.NI; o[12] = 0x10ull;
        o[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[14] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
}

// ---- FILL wave (ChainShape::INFILL): the operand rows of the first stage's steps, memory -> ring by LDS-DMA ----------------------------------------------
// Row r of the ring = the first stage's local step r = OPW consecutive entries of a diagonal of SA / SB (the KG - 1 rows above the band, then the band's 64): both
// in memory and in the ring a row is one contiguous piece, and the four rows of a group (4 c .. 4 c + 3) follow each other in the ring (OPR is a multiple of 4).
// Interval c: the group's 4 OPW quads per plane are issued (lane n of the k-th instruction: quad 64 k + n, i.e. row (64 k + n) / OPW, entry (64 k + n) % OPW);
// before the next barrier everything but this interval's instructions has landed (counted vmcnt: nothing but these DMAs is in this wave's queue).  Group c is
// therefore complete before interval c + 2 starts; the first stage reads rows 4 c + 1 .. 4 c + 4 at the top of its chunk c, in interval c + LEAD = c + AH + 1.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void chain_dma16(const void *sbase, unsigned voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte) : "memory", "m0");
}
#pragma clang diagnostic pop
template <class S, int CH, int AH>
__device__ __forceinline__ void chain_fill(const ChainArgs &a, unsigned char *lds, int job, int b, int g, int c_first, int lane) {
    using L = ChainLds<S, CH>;
    static_assert(AH == 2 && CH == 4 && L::OPR % CH == 0, "group c is read from interval c + AH + 1 on and complete before interval c + 2");
    constexpr int NQ = CH * L::OPW, ND = (NQ + 63) / 64;
    static_assert(2 * ND <= 63, "vmcnt is a 6-bit counter");
    const ChainGeo<S, CH> G(a, b, g, c_first);
    const long e0 = (long)job * a.ent + G.U00 + (long)G.s_start0 * G.RP - (S::KG - 1);       // entry 0 of row 0
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)(lds + L::ops0);
    unsigned voff[ND];
    bool on[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        const int n = 64 * k + lane, j = n / L::OPW, e = n % L::OPW;
        on[k] = n < NQ;
        voff[k] = (unsigned)((j * G.RP + e) * 16);
    }
    const float4 *pa = a.sa + e0, *pb = a.sb + e0;
    const long step = (long)CH * G.RP;                               // entries per group
    int row = 0;                                                     // ring row of the group issued next
    for (int I = 0; I < a.NI; I++) {
        SFA_CHAIN_BARRIER();
        const unsigned dst = lds0 + (unsigned)row * (unsigned)L::OPROWB;
#pragma unroll
        for (int k = 0; k < ND; k++) {
            if (on[k]) {
                chain_dma16(pa, voff[k], dst + 1024u * k);
                chain_dma16(pb, voff[k], dst + (unsigned)L::OPPLANE + 1024u * k);
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ND) : "memory");
        pa += step; pb += step;
        row = row + CH >= L::OPR ? 0 : row + CH;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- OUT wave: lane 63's iterates and the group's last iterate leave (sc1 write-through); the workgroup's progress word follows PUBD
// intervals later behind a counted wait.  Nothing but stores is ever in this wave's memory queue, so the count is exact and short.
template <class S, int CH, int AH, int PUBD>
__device__ __forceinline__ void chain_out(const ChainArgs &a, unsigned char *lds, int job, int b, int g, int c_first, int lane) {
    using L = ChainLds<S, CH>;
    constexpr int NW = S::NW, LEAD = AH + 1, Fl = S::Fw(NW - 1);
    constexpr int NES = [] { int n = 0; for (int w = 0; w < NW; w++) n += S::Fw(w) * CH; return n; }();            // lane-63 values stored per interval
    constexpr int NSE = (NES + 63) / 64;
    constexpr int T = NSE + CH + 1;                                  // VMEM instructions of one interval
    static_assert(PUBD * T <= 63, "vmcnt is a 6-bit counter");
    const ChainGeo<S, CH> G(a, b, g, c_first);
    const int RP = G.RP, K = G.K, W = G.W, H = G.H;
    const RsrcWords rE = rsrc_words(a.edge, a.edge_bytes);
    const RsrcWords rF = rsrc_words(a.gflags, a.flag_bytes);
    // x plane as the last stage writes it: base = entry of (first step, sweep Fl-1, lane 0)
    const RsrcWords rXout = rsrc_words(a.x + (size_t)job * a.ent + G.U0l + (long)G.s_startl * RP - (long)(Fl - 1) * G.FOFF, 0xffffff00ull);
    // lane 63's iterates, stored at interval I for the chunk c_first + I - 1 - LEAD - w of stage w
    unsigned es_voff[NSE], es_lds[NSE][2];
    int es_lo[NSE];
#pragma unroll
    for (int n = 0; n < NSE; n++) {
        const int i = n * 64 + lane;
        int w = 0, base = 0;
        while (w < NW - 1 && i >= base + S::Fw(w) * CH) { base += S::Fw(w) * CH; w++; }
        const int Fw = S::Fw(w), rel = i - base, f = rel / CH, j = rel % CH;
        const bool on = i < NES;
        const int k0w = G.k0g + S::kw(w), Ow = k0w - (G.st0 + w) + 1;
        const int s0 = (c_first - 1 - LEAD - w) * CH - Ow;                         // chunk stored at interval 0 (not a real one)
        const long off = (long)job * a.edge_job + ((long)b * K + k0w + f) * a.Wp + a.EP + (s0 + j - 63 - f);
        es_voff[n] = on ? (unsigned)(off * 8) : kOobOffset;
        es_lo[n] = on ? LEAD + w + 1 : 0x7fffffff;                                 // first interval with a real chunk
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int cp = (p + w) & 1;
            es_lds[n][p] = !on ? (unsigned)(L::dummy0 + lane * 8)
                         : f < Fw - 1 ? (unsigned)(L::es0 + w * L::ESW + ((cp * (Fw - 1) + f) * CH + j) * 8)
                                      : (unsigned)(L::ring0 + (w + 1) * L::RING + ((cp * CH + j) * 64 + 63) * 8);
        }
    }
    const int rl = G.r0l + lane - (Fl - 1);
#ifdef SFA_X_NOXSTORE     // timing experiment only
    const unsigned vXout = kOobOffset; (void)rl;
#else
    const unsigned vXout = (rl >= 0 && rl < H) ? (unsigned)lane * 8u : kOobOffset;
#endif
    const int lanef = lane + (Fl - 1);
    const unsigned st8 = (unsigned)RP * 8u;
    const unsigned vF0 = lane == 0 ? 0u : kOobOffset;
    const unsigned so_mine = (unsigned)((((size_t)job * a.NB + b) * a.NG + g) * 4 * kFlagStride);
    unsigned so_es = 0, so_xout = 0;
    int s_out = G.s_startl - (LEAD + NW) * CH;                       // local step (last stage) of row 0 of the chunk stored at interval I
    unsigned char *const ldsb = lds;
#ifdef SFA_CHAIN_TIMING
    unsigned long long t_begin = __builtin_readcyclecounter(), t_bar = 0, t_pub = 0, t0, t1;
#endif
    for (int I0 = 0; I0 < a.NI; I0 += 2) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int I = I0 + q;
            const int p = q & 1;
            SFA_CHAIN_BARRIER_T(t_bar);
            // ---- what the compute waves produced in the PREVIOUS interval leaves ------------------------------------------------------------
#pragma unroll
            for (int n = 0; n < NSE; n++) {
#ifdef SFA_X_OUT_NOLDS    // timing experiment only
                const unsigned long long v = (unsigned long long)I + es_lds[n][p];
#else
                const unsigned long long v = *reinterpret_cast<const unsigned long long *>(ldsb + es_lds[n][p]);
#endif
                const bool act = I >= es_lo[n] && I < es_lo[n] + a.nch;
                batomic_swap8(rE, act ? es_voff[n] : kOobOffset, so_es, v);
            }
            {
                const bool act = I >= LEAD + NW && I < LEAD + NW + a.nch;         // uniform
                const int po = (p + NW - 1) & 1;
                const unsigned long long *rp = reinterpret_cast<const unsigned long long *>(ldsb + L::ring0 + NW * L::RING + (po * CH * 64) * 8) + lane;
                // straight-line on purpose (no fast path for the steady state): the counted wait below relies on exactly T memory instructions per
                // interval, and tools/check_publish_vmcnt.py counts them on the ISA
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    const bool ok = act && (unsigned)(s_out + j - lanef) < (unsigned)W;
                    batomic_swap8(rXout, ok ? vXout : kOobOffset, so_xout + (unsigned)j * st8, rp[j * 64]);
                }
            }
            // ---- publish: the stores of interval I - PUBD are older than the PUBD * T instructions issued since ------------------------------
#ifdef SFA_CHAIN_TIMING
            SFA_CT_STAMP(t0);
#endif
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PUBD * T - 1) : "memory");
#ifdef SFA_CHAIN_TIMING
            SFA_CT_STAMP(t1); t_pub += t1 - t0;
#endif
            batomic_umax4(rF, vF0, so_mine, (unsigned)max(I - PUBD + 1, 0));
            so_es += CH * 8u; s_out += CH;
            if (I >= LEAD + NW) so_xout += CH * st8;
        }
    }
#ifdef SFA_CHAIN_TIMING
    if (job == 0 && lane == 0 && (int)blockIdx.x < 64) {
        unsigned long long *o = g_chain_timing + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16;
        o[0] = t_begin; o[1] = __builtin_readcyclecounter(); o[2] = t_bar; o[4] = t_pub;
        o[9] = (unsigned long long)b; o[10] = (unsigned long long)g; o[11] = (unsigned long long)a.NI; o[12] = 0x20ull;
        o[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[14] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    batomic_umax4(rF, vF0, so_mine, 0x7fffffffu);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------------------------------------------
// waves of a workgroup: 0 = IN, 1 .. NW = the stages, NW + 1 = OUT (with NW = 3 the two I/O waves share SIMD 0, the stages have a SIMD each)
template <int FA, int NA, int FB, int NB_, int CH, int PD, int AH, int PL, int PUBD>
#ifdef SFA_CHAIN_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(SFA_CHAIN_WAVES_PER_EU, SFA_CHAIN_WAVES_PER_EU)))
#endif
__global__ void __launch_bounds__((ChainShape<FA, NA, FB, NB_>::NWAVES * 64)) k_sor_chain(ChainArgs a) {
    using S = ChainShape<FA, NA, FB, NB_>;
    using L = ChainLds<S, CH>;
    constexpr int NW = S::NW, LEAD = AH + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned *s_ticket = reinterpret_cast<unsigned *>(smem + L::ticket);
    if (threadIdx.x == 0) *s_ticket = atomicAdd(a.gflags + (size_t)a.nb * a.NB * a.NG * kFlagStride, 1u);
    // the rings and staging areas are read before they are first written (start-up intervals): zeros, not garbage
    for (int i = threadIdx.x; i < L::ticket / 8; i += S::NWAVES * 64) reinterpret_cast<unsigned long long *>(smem)[i] = 0ull;
    __syncthreads();
    const unsigned t = __builtin_amdgcn_readfirstlane(*s_ticket);
    if (t >= (unsigned)(a.nb * a.NB * a.NG)) return;
    const int job = t % a.nb;
    if (!elem_active(a.active, a.amask, job)) return;        // a passenger: none of its workgroups runs, so none of them waits
    const int2 bg = a.order[t / a.nb];
    const int b = __builtin_amdgcn_readfirstlane(bg.x), g = __builtin_amdgcn_readfirstlane(bg.y);
    // first chunk of the group (even): every stage of the group starts at a local step <= -1 -- with an operand ring SHIFT chunks earlier still, so
    // that the rows in front of the first stage's first step (which nobody writes to the ring) are guard zeros up to KG - 1 entries above the band
    constexpr int SHIFT = L::OPRING ? chain_start_shift(S::KG) : 0;
    const int c_first = (((g * (S::KG - NW)) / CH) & ~1) - SHIFT;
    const int c_first_prev = g > 0 ? ((((g - 1) * (S::KG - NW)) / CH) & ~1) - SHIFT : 0;
#ifdef SFA_X_NOIO         // timing experiment only: the I/O waves just walk through the barriers (nothing crosses workgroups: wrong results, compute speed)
#ifdef SFA_X_NOIO_POLL    // ... except that wave 0 issues ONE real bypassing load (or store) per interval: does a slow access of another wave delay this CU's operand stream?
    if (wave == 0) {
        const __amdgpu_buffer_rsrc_t rF = plane_rsrc(a.gflags, a.flag_bytes);
        const unsigned so = (unsigned)((((size_t)job * a.NB + b) * a.NG + g) * 4);
        unsigned acc = 0, pend[4] = {0, 0, 0, 0};
        for (int I = 0; I < a.NI; I += 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                SFA_CHAIN_BARRIER();
                acc += pend[(q + 1) % 4];                    // a load issued three intervals ago
#if SFA_X_NOIO_POLL == 1
                pend[q] = bload4_sc1(rF, lane == 0 ? 0u : kOobOffset, so);
#elif SFA_X_NOIO_POLL == 2
                pend[q] = __builtin_amdgcn_raw_buffer_load_b32(rF, lane == 0 ? 0u : kOobOffset, so, 0);      // plain (L1/L2 hit)
#elif SFA_X_NOIO_POLL == 3
                bstore4_sc1(rF, lane == 0 ? 0u : kOobOffset, so, (unsigned)I);
#elif SFA_X_NOIO_POLL == 4
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)I, rF, lane == 0 ? 0u : kOobOffset, so, 0);       // plain
#elif SFA_X_NOIO_POLL == 5
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)I, rF, lane == 0 ? 0u : kOobOffset, so, 2);       // nt
#elif SFA_X_NOIO_POLL == 6
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)I, rF, lane == 0 ? 0u : kOobOffset, so, 17);      // sc0 sc1
#elif SFA_X_NOIO_POLL == 7
                if (lane == 0) __hip_atomic_fetch_add(a.gflags + so / 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no-return atomic
#elif SFA_X_NOIO_POLL == 8
                { const __amdgpu_buffer_rsrc_t rX8 = plane_rsrc(a.x + (size_t)job * a.ent, 0xffffff00ull); bstore8_sc1(rX8, lane * 8u, (unsigned)(I & 7) * 512u, 0ull); }   // a whole row, write-through
#elif SFA_X_NOIO_POLL == 10
                {   // four rows as no-return 8-byte atomic swaps + one atomic add
                    unsigned long long *xp = a.x + (size_t)job * a.ent + (size_t)blockIdx.x * 4096 + lane;
#pragma unroll
                    for (int j = 0; j < 4; j++) __hip_atomic_exchange(xp + j * 64, (unsigned long long)I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0) __hip_atomic_fetch_add(a.gflags + so / 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#elif SFA_X_NOIO_POLL == 11 || SFA_X_NOIO_POLL == 12
                {   // four plain rows, then an agent-scope release (L2 write-back) every interval (11) / every fourth (12), then the flag
                    const __amdgpu_buffer_rsrc_t rX8 = plane_rsrc(a.x + (size_t)job * a.ent + (size_t)blockIdx.x * 4096, 0xffffff00ull); v2u z = {(unsigned)I, 0u};
#pragma unroll
                    for (int j = 0; j < 4; j++) __builtin_amdgcn_raw_buffer_store_b64(z, rX8, lane * 8u, (unsigned)(j * 512), 0);
                    if (SFA_X_NOIO_POLL == 11 || q == 3) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_fetch_add(a.gflags + so / 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#elif SFA_X_NOIO_POLL == 13
                {   // two 1-KB write-through stores
                    const __amdgpu_buffer_rsrc_t rX8 = plane_rsrc(a.x + (size_t)job * a.ent + (size_t)blockIdx.x * 4096, 0xffffff00ull); v4u z = {(unsigned)I, 0u, 0u, 0u};
                    __builtin_amdgcn_raw_buffer_store_b128(z, rX8, lane * 16u, 0, 16); __builtin_amdgcn_raw_buffer_store_b128(z, rX8, lane * 16u, 1024, 16);
                }
#elif SFA_X_NOIO_POLL == 14
                {   // four rows as RETURNING swaps (result consumed three intervals later) + one atomic add
                    unsigned long long *xp = a.x + (size_t)job * a.ent + (size_t)blockIdx.x * 4096 + lane;
                    unsigned long long r = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) r ^= __hip_atomic_exchange(xp + j * 64, (unsigned long long)I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pend[q] = (unsigned)r;
                }
#elif SFA_X_NOIO_POLL == 15
                {   // four rows of 8-byte bypassing LOADS
                    const __amdgpu_buffer_rsrc_t rX8 = plane_rsrc(a.x + (size_t)job * a.ent + (size_t)blockIdx.x * 4096, 0xffffff00ull);
                    unsigned long long r = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) r ^= bload8_sc1(rX8, lane * 8u, (unsigned)(j * 512 + (I & 15) * 2048));
                    pend[q] = (unsigned)r;
                }
#elif SFA_X_NOIO_POLL == 16
                {   // four rows as 8-byte atomic swaps + the edge values (12 lanes) as swaps + an atomic max as flag: the OUT wave's traffic
                    unsigned long long *xp = a.x + (size_t)job * a.ent + (size_t)blockIdx.x * 4096 + lane;
#pragma unroll
                    for (int j = 0; j < 4; j++) __hip_atomic_exchange(xp + j * 64, (unsigned long long)I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane < 12) __hip_atomic_exchange(a.edge + (size_t)blockIdx.x * 64 + lane, (unsigned long long)I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_max(a.gflags + so / 4, (unsigned)I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#elif SFA_X_NOIO_POLL == 9
                { const __amdgpu_buffer_rsrc_t rX8 = plane_rsrc(a.x + (size_t)job * a.ent, 0xffffff00ull); v2u z = {0u, 0u}; __builtin_amdgcn_raw_buffer_store_b64(z, rX8, lane * 8u, (unsigned)(I & 7) * 512u, 0); }   // a whole row, plain
#endif
            }
        }
        if (acc == 0x12345678u) bstore4_sc1(rF, 0u, so, acc);
        return;
    }
#endif
    if (wave == 0 || wave == NW + 1) { for (int I = 0; I < a.NI; I++) SFA_CHAIN_BARRIER(); return; }
#endif
#ifdef SFA_X_IN_IDLE
    if (wave == 0) { for (int I = 0; I < a.NI; I++) SFA_CHAIN_BARRIER(); return; }
#endif
#ifdef SFA_X_OUT_IDLE
    if (wave == NW + 1) { for (int I = 0; I < a.NI; I++) SFA_CHAIN_BARRIER(); return; }
#endif
    if (wave == 0) { chain_in<S, CH, AH, PL>(a, smem, job, b, g, c_first, c_first_prev, lane); return; }
    if (wave == NW + 1) { chain_out<S, CH, AH, PUBD>(a, smem, job, b, g, c_first, lane); return; }
    if (S::INFILL && wave == NW + 2) { if constexpr (S::INFILL) chain_fill<S, CH, AH>(a, smem, job, b, g, c_first, lane); return; }
    const int w = S::stage_of_wave(wave);
    const int st = g * NW + w, k0 = g * S::KG + S::kw(w), O = k0 - st + 1;
    const int s_start = c_first * CH - O;
    const int ring_in = L::ring0 + w * L::RING, ring_out = ring_in + L::RING;
    const int tvb = L::tv0 + w * L::TVW, esb = L::es0 + w * L::ESW;
    constexpr bool SHORT = NW >= 8;                  // many waves per workgroup: fewer registers each
    constexpr int PF0 = S::PF0, PFL = S::PFL;              // read-ahead of the operand ring (ChainShape)
    if (L::OPRING) {
        if (w == 0)      chain_compute<FA, CH, PD, SHORT, S::INFILL ? 2 : 1, L::ops0, L::OPR, L::OPROWB, S::KG, PF0>(a, smem, ring_in, ring_out, tvb, esb, L::dummy0, job, b, k0, s_start, LEAD + w, lane, 0, 0);
        else if (w < NA) chain_compute<FA, CH, PD, SHORT, 2, L::ops0, L::OPR, L::OPROWB, S::KG, PFL>(a, smem, ring_in, ring_out, tvb, esb, L::dummy0, job, b, k0, s_start, LEAD + w, lane, S::kw(w), w);
        else             chain_compute<FB, CH, PD, SHORT, 2, L::ops0, L::OPR, L::OPROWB, S::KG, PFL>(a, smem, ring_in, ring_out, tvb, esb, L::dummy0, job, b, k0, s_start, LEAD + w, lane, S::kw(w), w);
    } else if (w < NA) chain_compute<FA, CH, PD, SHORT>(a, smem, ring_in, ring_out, tvb, esb, L::dummy0, job, b, k0, s_start, LEAD + w, lane);
    else               chain_compute<FB, CH, PD, SHORT>(a, smem, ring_in, ring_out, tvb, esb, L::dummy0, job, b, k0, s_start, LEAD + w, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
#ifndef SFA_CHAIN_AH
#define SFA_CHAIN_AH 2
#endif
#ifndef SFA_CHAIN_PL
#define SFA_CHAIN_PL 2
#endif
#ifndef SFA_CHAIN_PUBD
#define SFA_CHAIN_PUBD 2
#endif
constexpr int kChainCH = 4, kChainAH = SFA_CHAIN_AH, kChainPL = SFA_CHAIN_PL, kChainPUBD = SFA_CHAIN_PUBD;

// PL: intervals between a poll of the producers' progress words and the look at its result; PUBD: intervals between a store and the progress word that covers it.
// Both are correct at any value >= 1 (a poll that has not returned is waited for; the publication sits behind a counted vmcnt wait over PUBD * T instructions) and
// trade latency against stalls: 2 / 2 everywhere until round 4; 1 / 1 for the lone-solve shape (one window 276 -> 265 us) and the six-stage shape (64 windows 1398 -> 1379 us,
// 16: 529 -> 514), NOT for 1 x 5 at 8-12 windows (+2 .. +12 %): id 16 is 1 x 5 with 1 / 1 and serves launches of up to 32 bands.
struct ChainShapeInfo { int id, FA, NA, FB, NB_, PD, PL, PUBD; };
// SFA_RELEASE (sfa_internal.h): only the shapes the library picks by default are compiled -- 14 from 73 bands per launch on, 6 / 16 below (sor.hip chain_choice; sweep
// counts that 5 does not divide take the task kernel)
static const ChainShapeInfo kChainShapes[] = {
    SFA_FULL({1, 1, 3, 1, 0, 4, 2, 2},)      // 3 stages of 1 sweep            KG = 3
    SFA_FULL({2, 2, 3, 2, 0, 2, 2, 2},)      // 3 stages of 2                  KG = 6
    SFA_FULL({3, 3, 5, 3, 0, 2, 2, 2},)      // 5 stages of 3                  KG = 15
    SFA_FULL({5, 2, 5, 2, 0, 2, 2, 2},)      // 5 stages of 2                  KG = 10
    {6, 1, 5, 1, 0, 4, 2, 2},      // 5 stages of 1                  KG = 5
    SFA_FULL({8, 3, 2, 3, 0, 2, 2, 2},)      // 2 stages of 3                  KG = 6
    SFA_FULL({9, 5, 6, 5, 0, 2, 2, 2},)      // 6 stages of 5                  KG = 30: a whole band per workgroup (large batches)
    SFA_FULL({10, 3, 10, 3, 0, 2, 2, 2},)    // 10 stages of 3                 KG = 30
    SFA_FULL({11, 3, 3, 2, 3, 2, 1, 1},)     // 3 stages of 3 + 3 of 2         KG = 15: with the I/O waves 8 waves, two per SIMD, at most 5 sweeps on a SIMD (5 x 3: 6)
    SFA_FULL({12, 2, 3, 3, 3, 2, 2, 2},)     // 3 stages of 2 + 3 of 3         KG = 15
    SFA_FULL({13, 3, 1, 2, 6, 2, 1, 1},)     // 1 stage of 3 + 6 of 2          KG = 15: nine waves, at most 4 sweeps on a SIMD (ChainShape::stage_of_wave)
    {16, 1, 5, 1, 0, 4, 1, 1},     // 5 stages of 1, one-interval poll / publication lags: the lone solve
    SFA_FULL({17, 1, 3, 1, 0, 4, 1, 1},)     // 3 stages of 1 with the same lags: every compute wave alone on its SIMD, ten groups per band
    SFA_FULL({19, 1, 6, 1, 0, 4, 1, 1},)     // 6 stages of 1: five groups per band
    {14, 2, 6, 3, 1, 2, 1, 1},     // 6 stages of 2 + 1 of 3         KG = 15: nine waves, the last stage beside the I/O waves (ChainShape::PERM9L)
#ifdef SFA_WHATIF_SHAPE20
    {20, 2, 5, 1, 5, 2, 1, 1},     // what-if only: 5 stages of 2 + 5 of 1: twelve waves, three per SIMD (ChainShape::PERM12)
#endif
};

bool chain_shape(int id, int K, int *KG, int *NW, int *FMAX) {
    for (const ChainShapeInfo &s : kChainShapes)
        if (s.id == id) {
            const int kg = s.NA * s.FA + s.NB_ * s.FB;
            if (K % kg != 0) return false;
            *KG = kg; *NW = s.NA + s.NB_; *FMAX = std::max(s.FA, s.NB_ ? s.FB : s.FA);
            return true;
        }
    return false;
}
int chain_ch() { return kChainCH; }
// the kernel's name as the profiler prints it (bench.py matches counter files against it)
void chain_kernel_name(int id, int NG, char *buf, size_t n) {
    for (const ChainShapeInfo &s : kChainShapes)
        if (s.id == id) {
            if (s.NB_ && s.FB != s.FA)
                snprintf(buf, n, "k_sor_chain<%d,%d,%d,%d,%d,%d,%d,%d,%d> (%d stages of %d + %d of %d sweeps, %d groups per band)", s.FA, s.NA, s.FB, s.NB_, kChainCH, s.PD, kChainAH,
                         s.PL, s.PUBD, s.NA, s.FA, s.NB_, s.FB, NG);
            else
                snprintf(buf, n, "k_sor_chain<%d,%d,%d,%d,%d,%d,%d,%d,%d> (%d stages of %d sweeps, %d groups per band)", s.FA, s.NA, s.FB, s.NB_, kChainCH, s.PD, kChainAH, s.PL, s.PUBD,
                         s.NA + s.NB_, s.FA, NG);
            return;
        }
    snprintf(buf, n, "k_sor_chain shape %d", id);
}
// extra chunks per stage a shape needs because its groups start early (operand ring)
template <int FA, int NA, int FB, int NB_> static int shape_shift() { using S = ChainShape<FA, NA, FB, NB_>; return ChainLds<S, kChainCH>::OPRING ? chain_start_shift(S::KG) : 0; }
int chain_shift(int id) {
    switch (id) {
        SFA_FULL(case 1: return shape_shift<1, 3, 1, 0>();)
        SFA_FULL(case 2: return shape_shift<2, 3, 2, 0>();)
        SFA_FULL(case 3: return shape_shift<3, 5, 3, 0>();)
        SFA_FULL(case 5: return shape_shift<2, 5, 2, 0>();)
        case 6: return shape_shift<1, 5, 1, 0>();
        SFA_FULL(case 8: return shape_shift<3, 2, 3, 0>();)
        SFA_FULL(case 9: return shape_shift<5, 6, 5, 0>();)
        SFA_FULL(case 10: return shape_shift<3, 10, 3, 0>();)
        SFA_FULL(case 11: return shape_shift<3, 3, 2, 3>();)
        case 16: return shape_shift<1, 5, 1, 0>();
        SFA_FULL(case 12: return shape_shift<2, 3, 3, 3>();)
        SFA_FULL(case 13: return shape_shift<3, 1, 2, 6>();)
        SFA_FULL(case 17: return shape_shift<1, 3, 1, 0>();)
        SFA_FULL(case 19: return shape_shift<1, 6, 1, 0>();)
        case 14: return shape_shift<2, 6, 3, 1>();
#ifdef SFA_WHATIF_SHAPE20
        case 20: static_assert(ChainLds<ChainShape<2, 5, 1, 5>, kChainCH>::OPRING, "the what-if needs -DSFA_X_OPR_CUT=18: its ring does not fit"); return shape_shift<2, 5, 1, 5>();
#endif
    }
    return 0;
}
int chain_flag_stride() { return kFlagStride; }
int chain_ah() { return kChainAH; }

template <int FA, int NA, int FB, int NB_, int PD, int PL = kChainPL, int PUBD = kChainPUBD>
static int chain_launch_shape(sfa_ctx *c, const ChainArgs &a, int nwg) {
    using S = ChainShape<FA, NA, FB, NB_>;
    using L = ChainLds<S, kChainCH>;
    size_t lds = L::total;
    lds = std::max(lds, (size_t)sw_int(Switches::CHAIN_LDS, 0));      // experiment: a larger request limits the workgroups per CU
    // more than 64 KB of dynamic LDS has to be allowed per function AND per device (the driver refines on several GPUs from one process, one thread each)
    // the bit of a device is set only once the call has succeeded there; devices beyond the 64 bits are asked every time (the call is cheap and idempotent)
    static std::atomic<unsigned long long> attr_set{0};
    const bool tracked = c->device >= 0 && c->device < 64;
    const unsigned long long bit = tracked ? 1ull << c->device : 0ull;
    if (lds > 64 * 1024 && !(attr_set.load(std::memory_order_relaxed) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sor_chain<FA, NA, FB, NB_, kChainCH, PD, kChainAH, PL, PUBD>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return set_error(c, SFA_ERR_HIP, "k_sor_chain<%d,%d,%d,%d>: %zu bytes of LDS per workgroup refused on device %d: %s", FA, NA, FB, NB_, lds, c->device,
                             hipGetErrorString(e));
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((k_sor_chain<FA, NA, FB, NB_, kChainCH, PD, kChainAH, PL, PUBD>), dim3(nwg), dim3(S::NWAVES * 64), lds, c->stream, a);
    return SFA_OK;
}

int sor_chain_launch(sfa_ctx *c, SorWorkspace &ws, const Geo &g, int K, float omega) {
    ChainArgs a;
    a.sa = (const float4 *)ws.sa.p; a.sb = (const float4 *)ws.sb.p; a.x = (unsigned long long *)ws.x.p; a.edge = (unsigned long long *)ws.edge.p;
    a.gflags = (unsigned *)ws.flags.p; a.order = (const int2 *)ws.order.p; a.err = c->d_err;
    a.ent = ws.ent; a.edge_job = ws.edge_job; a.edge_bytes = (unsigned long long)g.nb * ws.edge_job * 8ull;
    a.flag_bytes = ((unsigned long long)g.nb * ws.nwords + 16) * 4ull;
    a.W = g.w; a.H = g.h; a.K = K; a.NB = ws.NB; a.NG = ws.NG; a.RP = ws.RP; a.G = ws.G; a.nb = g.nb; a.Wp = ws.Wp; a.EP = ws.EP;
    a.nch = ws.NCH; a.NI = ws.NS; a.omega = omega; a.active = g.active; a.amask = g.amask;
    const int nwg = g.nb * ws.NB * ws.NG;
    switch (ws.chain) {
        SFA_FULL(case 1: return chain_launch_shape<1, 3, 1, 0, 4>(c, a, nwg);)
        SFA_FULL(case 2: return chain_launch_shape<2, 3, 2, 0, 2>(c, a, nwg);)
        SFA_FULL(case 3: return chain_launch_shape<3, 5, 3, 0, 2>(c, a, nwg);)
        SFA_FULL(case 5: return chain_launch_shape<2, 5, 2, 0, 2>(c, a, nwg);)
        case 6: return chain_launch_shape<1, 5, 1, 0, 4>(c, a, nwg);
        SFA_FULL(case 8: return chain_launch_shape<3, 2, 3, 0, 2>(c, a, nwg);)
        SFA_FULL(case 9: return chain_launch_shape<5, 6, 5, 0, 2>(c, a, nwg);)
        SFA_FULL(case 10: return chain_launch_shape<3, 10, 3, 0, 2>(c, a, nwg);)
        SFA_FULL(case 11: return chain_launch_shape<3, 3, 2, 3, 2, 1, 1>(c, a, nwg);)
        case 16: return chain_launch_shape<1, 5, 1, 0, 4, 1, 1>(c, a, nwg);
        SFA_FULL(case 12: return chain_launch_shape<2, 3, 3, 3, 2>(c, a, nwg);)
        SFA_FULL(case 13: return chain_launch_shape<3, 1, 2, 6, 2, 1, 1>(c, a, nwg);)
        SFA_FULL(case 17: return chain_launch_shape<1, 3, 1, 0, 4, 1, 1>(c, a, nwg);)
        SFA_FULL(case 19: return chain_launch_shape<1, 6, 1, 0, 4, 1, 1>(c, a, nwg);)
        case 14: return chain_launch_shape<2, 6, 3, 1, 2, 1, 1>(c, a, nwg);
#ifdef SFA_WHATIF_SHAPE20
        case 20: return chain_launch_shape<2, 5, 1, 5, 2, 1, 1>(c, a, nwg);
#endif
        default: return set_error(c, SFA_ERR_ARG, "sor_chain_launch: unknown shape %d", ws.chain);
    }
}

}  // namespace sfa

#ifdef SFA_CHAIN_TIMING
extern "C" int sfa_debug_chain_timing(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sfa::g_chain_timing), sizeof(unsigned long long) * 64 * 16 * 16); }
#endif
