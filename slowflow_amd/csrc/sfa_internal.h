// sfa_internal.h -- internal declarations of libslowflow_amd (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/slowflow_amd.h"

#pragma clang fp contract(off)   // the reference is strict fp32 without FMA contraction (CMakeLists.txt:6)

namespace sfa {

// Windows a lockstep job can hold.  Which of them take part in a launch is a bit set of kMaskWords 64-bit words (WMask): what the host knows (Geo::active) AND what the
// device knows (the windows that have not met the outer threshold yet: sfa_ctx::d_amask, Geo::amask).  Round 5 widened the set from one word to two.
constexpr int kMaskWords = 2;
constexpr int kMaxBatch = 64 * kMaskWords;
struct WMask {
    unsigned long long w[kMaskWords];
    __host__ __device__ bool test(int b) const { return (w[b >> 6] >> (b & 63)) & 1ull; }
    __host__ __device__ void clear(int b) { w[b >> 6] &= ~(1ull << (b & 63)); }
    __host__ __device__ void set(int b) { w[b >> 6] |= 1ull << (b & 63); }
    __host__ __device__ bool any() const { unsigned long long o = 0; for (int i = 0; i < kMaskWords; i++) o |= w[i]; return o != 0; }
    __host__ __device__ bool operator==(const WMask &o) const { bool e = true; for (int i = 0; i < kMaskWords; i++) e = e && w[i] == o.w[i]; return e; }
    __host__ __device__ bool operator!=(const WMask &o) const { return !(*this == o); }
    __host__ __device__ WMask operator&(const WMask &o) const { WMask r; for (int i = 0; i < kMaskWords; i++) r.w[i] = w[i] & o.w[i]; return r; }
    __host__ __device__ WMask andnot(const WMask &o) const { WMask r; for (int i = 0; i < kMaskWords; i++) r.w[i] = w[i] & ~o.w[i]; return r; }
    int count() const { int n = 0; for (int i = 0; i < kMaskWords; i++) n += __builtin_popcountll(w[i]); return n; }
    static WMask none() { WMask r; for (int i = 0; i < kMaskWords; i++) r.w[i] = 0; return r; }
    static WMask first(int nb) {                                   // windows 0 .. nb - 1
        WMask r;
        for (int i = 0; i < kMaskWords; i++) r.w[i] = nb >= 64 * (i + 1) ? ~0ull : nb <= 64 * i ? 0ull : (1ull << (nb - 64 * i)) - 1;
        return r;
    }
    static WMask one(int b) { WMask r = none(); r.w[b >> 6] = 1ull << (b & 63); return r; }
};
constexpr int kMaxTerms = 4 * SFA_MAX_REF;
// What ctx->d_last holds (ONE definition for api.hip and kernels.hip -- ADVICE r5: the layout was repeated by hand in four places): the norms of every window's
// last outer iteration, the windows' finished-block counters of k_update_outer_x, the windows k_outer_threshold could not decide, the exact norms of
// launch_exact_norms.  The pinned mirror ctx->h_red is at least this large (the exact norms are read back into ITS `exact` member, never into pageable memory).
struct LastBlock {
    double last[2 * kMaxBatch];
    unsigned done[kMaxBatch];
    unsigned long long unsure[kMaskWords];
    float exact[2 * kMaxBatch];
};
// How close to a break threshold (relative) the fp64 tree sum of a change norm may come before the decision is taken on the reference's own fp32 running sum
// (k_exact_break / launch_exact_norms).  The reference adds ceil(w / 4) * h block sums to ONE fp32 accumulator (variational_mt.cpp:412-425); each addition rounds
// by at most 2^-24 of the running sum, and the errors need not cancel -- block sums below half an ulp of the accumulator (a converged background behind a few
// moving pixels) are dropped one after the other, always downwards -- so the two sums can differ by up to blocks * 2^-24 of their value: 6.7e-3 at 1024 x 436,
// 6.3e-2 at 2048 x 2048.  Outside that band both decide alike by construction; inside it the reference's arithmetic is repeated.  (Rounds 4-5 used a constant
// 1e-3, the size of the typical difference, not of the bound: ADVICE r5.)
__host__ __device__ inline double break_band(int w, int h) {
    const double bound = (double)((w + 3) / 4) * (double)h * (1.0 / 16777216.0);
    return bound > 1e-3 ? bound : 1e-3;
}

// device-side outer break (api.hip run_level): the host reads the mask of kMaskLag iterations ago from a ring of kMaskRing pinned words
constexpr int kMaskLag = 2, kMaskRing = 4;
static_assert(kMaskLag < kMaskRing, "a ring slot is rewritten kMaskRing iterations after it was read kMaskLag iterations late");

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
struct DevBlock { void *p; size_t bytes; };

}  // namespace sfa

struct sfa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scratch for reductions / flags
    double *d_red = nullptr;      // device, kRedDoubles doubles
    double *h_red = nullptr;      // pinned host mirror
    // thresholds on the device (variational_mt.cpp:436): the windows still iterating, the norms of each window's last iteration, and a ring of pinned
    // copies of the mask (one per outer iteration in flight) with the events that say a copy has landed
    unsigned long long *d_amask = nullptr;         // kMaskWords words
    sfa::LastBlock *d_last = nullptr;              // norms of each window's last iteration + the small device-side state of the break decisions (sfa::LastBlock)
    sfa::WMask *h_amask = nullptr;                      // kMaskRing pinned masks
    hipEvent_t ev_mask[sfa::kMaskRing] = {};
    unsigned *d_err = nullptr;    // device error/timeout word
    // profiling of the SOR solve kernel
    bool profile = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t ev_used = 0;
    double sor_bytes = 0;         // algorithmic bytes of the bracketed launches
    std::vector<hipEvent_t> ev2;  // pairs around the data-term assembly kernel (k_assemble_images)
    size_t ev2_used = 0;
    double asm_pixel_terms = 0;   // pixels x data terms of the bracketed assembly launches
    char sor_kernel[160] = {0};   // the solver kernel (shape) of the last solve launched
    void *rb_tmp = nullptr;       // labelled red-black mode: scratch (du, dv) pair the tile visits ping-pong with, grown on demand
    size_t rb_tmp_bytes = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // the reference's per-iteration "avg change" lines (variational_mt.cpp:404-405, 431-432): sfa_ctx_set_verbose; costs a host round trip per iteration
    bool verbose_changes = false;
    // default-ctx bookkeeping
    int cu_count = 256;
};

namespace sfa {

constexpr int kRedDoubles = 1 << 20;   // 2*kMaxBatch result words + per-block partials: 2 x windows x ceil(w / 64) x 16 row blocks (sfa_job_create checks the size)

int set_error(sfa_ctx *ctx, int code, const char *fmt, ...);
extern thread_local std::string g_thread_err;

// ---------------------------------------------------------------------------------------------------
// Cross-check and what-if paths of the library (the unfused pipeline, other solver shapes, the cut's knobs ...).  ONE record per process, filled once: from the
// environment ONLY when SFA_DEBUG=1 is set at the first sfa_ctx_create -- without it no variable of a caller's environment can change which kernels the drop-in
// library runs --, afterwards only through the test hook sfa_debug_set() (include/slowflow_amd.h).  api.hip holds the table of names.
// ---------------------------------------------------------------------------------------------------
struct Switches {
    enum Id {
        SOR_CHAIN, SOR_BAND, SOR_F, SOR_CH, SOR_LEAD, CHAIN_LDS, RB_TILE, WARP_ALLJ, NO_WARP_SMOOTH, ASSEMBLE_GENERIC, EXACT_DIV, ASM_XCD, NO_DIRECT_OPERANDS,
        NO_UV_ALIAS, DEBUG_ACTIVE, UNFUSED, SHARE_SOR, PYRAMID_UNFUSED, CUT_DISCHARGE, CUT_INNER, CUT_SUPER, CUT_TAIL_INNER, CUT_PER, CUT_TAIL_PER, CUT_TAIL_SUPER,
        CUT_DEBUG, CUT_NO_TAIL, CUT_TAIL, NO_EXACT_BREAK, N
    };
    bool given[N] = {};
    int value[N] = {};
};
// code of the full build only
#ifdef SFA_RELEASE
#define SFA_FULL(...)
#else
#define SFA_FULL(...) __VA_ARGS__
#endif
#ifdef SFA_RELEASE
// RELEASE BUILD (make -C slowflow_amd/csrc release -> build_release/libslowflow_amd.so): no switch exists -- every cross-check / what-if branch
// is dead code the compiler removes, the environment is never read (not even behind SFA_DEBUG=1), sfa_debug_set refuses by name, and only the solver shapes the
// library picks by default are compiled (sor_chain.hip: 2,2,2,2,2,2,3 / 1 x 5 with either lag pair; sor.hip: the task kernel as the one fallback for sweep counts
// no chain shape divides).  Same C-ABI, same results as the full build's defaults; the tests run the full build.
constexpr bool sw_given(Switches::Id) { return false; }
constexpr int sw_int(Switches::Id, int dflt) { return dflt; }
#else
extern Switches g_switches;
inline bool sw_given(Switches::Id i) { return g_switches.given[i]; }                                   // what `getenv(name) != nullptr` used to say
inline int sw_int(Switches::Id i, int dflt) { return g_switches.given[i] ? g_switches.value[i] : dflt; }   // `getenv(name) ? atoi(getenv(name)) : dflt`
#endif

#define SFA_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return sfa::set_error((ctx), SFA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define SFA_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc != SFA_OK) return _rc; \
    } while (0)

inline int round_up(int a, int m) { return (a + m - 1) / m * m; }
inline int host_stride(int w) { return ((w + 3) / 4) * 4; }     // image.c:25
#ifndef SFA_PITCH_ODD
#define SFA_PITCH_ODD 0
#endif
// device rows are 256-B aligned.  SFA_PITCH_ODD (what-if, round 6): an ODD number of 256-byte units per row, so that the rows of a tile and the planes of an image set
// (a whole number of rows apart) do not all fall on addresses that agree modulo 4 KB (1024-column frames: rows 4 096 bytes apart)
inline int dev_pitch(int w) { const int p = round_up(w, 64); return (SFA_PITCH_ODD && ((p / 64) & 1) == 0) ? p + 64 : p; }

// RAII device allocation
struct DevMem {
    void *p = nullptr;
    size_t bytes = 0;
    DevMem() = default;
    DevMem(const DevMem &) = delete;
    DevMem &operator=(const DevMem &) = delete;
    ~DevMem() { release(); }
    int alloc(sfa_ctx *ctx, size_t n);
    void release();
    float *f() const { return static_cast<float *>(p); }
};

// host <-> device plane copies (valid columns only unless full_stride)
int upload_plane(sfa_ctx *ctx, float *dev, int pitch, const float *host, int stride, int w, int h);
int download_plane(sfa_ctx *ctx, float *host, int stride, const float *dev, int pitch, int w, int h);

// ---------------------------------------------------------------------------------------------------
// kernel launchers (kernels.hip).  All planes are device pointers of batch element 0; element b lives
// `es` floats further (es = 0 for a single element).  `nb` = batch size, `active` = bit mask of the
// elements that still iterate (threshold breaks).
// ---------------------------------------------------------------------------------------------------
// pl = pitch*h.  A window b takes part in a launch when bit b is set in `active` (what the host knows) AND in *amask (what the device knows: the windows
// that have not met the outer threshold yet, sfa_ctx::d_amask; null = no device-side mask)
struct Geo { int w, h, pitch; long pl; long es; int nb; WMask active; const unsigned long long *amask; };

struct PenaltyDev { int id; float eps, trunc; };

// several warps (factor != 0) of one flow field in one pass; offsets are arena offsets of element 0 like Term::i1_off, mask_off < 0: no mask
struct WarpJob { long src_off, dst_off, mask_off; int factor; };
struct WarpJobs { WarpJob job[4 * SFA_MAX_REF]; int n; };
void launch_warp_jobs(sfa_ctx *c, const Geo &g, const WarpJobs &J, float *base, const float *wx, const float *wy);
bool launch_warp_smooth(sfa_ctx *c, const Geo &g, const WarpJobs &J, float *base, const float *wx, const float *wy, int method, float *sh, float *sv,
                        const float *dpsis, float alpha, PenaltyDev reg);   // the warps + compute_smoothness of the same flow in one pass; false: not this combination
void launch_warp(sfa_ctx *c, const Geo &g, float *dst3, float *mask, const float *src3, const float *wx, const float *wy, int factor,
                 long src_es /* batch stride of src3 (frames) */);
void launch_deriv_stack(sfa_ctx *c, const Geo &g, float *out24, const float *I1, const float *I2, long es1, long es2);
void launch_convolve(sfa_ctx *c, const Geo &g, float *dst, const float *src, int order, int horiz, int nplanes);
void launch_dpsis(sfa_ctx *c, const Geo &g, float *dst, const float *im3, long im_es, float coef, const float avg[3], const float stdv[3], int hbit);
void launch_smoothness(sfa_ctx *c, const Geo &g, int method, float *sh, float *sv, const float *uu, const float *vv, const float *dpsis,
                       float alpha, PenaltyDev reg);
void launch_sub_laplacian(sfa_ctx *c, const Geo &g, float *dst, const float *src, const float *wh, const float *wv);
void launch_mask_weight(sfa_ctx *c, const Geo &g, float *masks, const float *occ, float data_norm, int ref, int one_direction);
void launch_fill(sfa_ctx *c, float *p, size_t n, float v);
void launch_division_chain(sfa_ctx *c, const float *a, const float *b, float *q_chain, float *q_exact, unsigned char *admitted, size_t n);
void launch_fill_planes(sfa_ctx *c, const Geo &g, float *p, int nplanes, float v);
void launch_zero_planes(sfa_ctx *c, const Geo &g, float *p, int nplanes);
// updates *c->d_amask and c->d_last.  dfa, dfb (or null): the per-pixel |differences| of the two norms as the update left them -- windows whose norm lies within
// 1e-3 of the threshold are then decided by the reference's own fp32 running sums (k_exact_break)
void launch_outer_threshold(sfa_ctx *c, const Geo &g, const double *red, float thres, const float *dfa = nullptr, const float *dfb = nullptr);
void launch_set_mask(sfa_ctx *c, const WMask &v);

// Where a kernel other than k_sor_prepare leaves the solver's operands (diagonal-major planes of a SorWorkspace)
struct SorOperandOut {
    float4 *sa = nullptr, *sb = nullptr; unsigned long long *x = nullptr; unsigned *flags = nullptr;
    long ent = 0; int RP = 0, G = 0, ntasks = 0, nb = 0;
};
struct Term {
    long stack_off; long mask_off; float hd, hg, s; int is_ref;
    // fused form (launch_assemble_images): the image pair the derivative stack is taken of, arena offsets like mask_off;
    // backward: the mask is weighted with the backward (slot < ref) or forward occlusion factor
    long i1_off, i2_off; int backward;
};
struct AssembleArgs {
    Term t[kMaxTerms];
    int n;
    int dt_norm;
    PenaltyDev color, grad;
    // channel weights: NULL => ones.  chw planes live OUTSIDE the element arena stride logic: chw_es batch stride
    const float *chw; long chw_pl; long chw_es; int chw_pitch; int chw_stride0; int lstride;
    int accumulate;      // 1: add to existing a11.. (stage API), 0: start from zero
    int do_laplacian;    // apply sub_laplacian(b1,uu), (b2,vv) at the end
    // fused form only: occlusion / direction weighting of the masks (variational_mt.cpp:293-320) applied on the fly
    float data_norm; int one_direction;
    // fused form only: if op.sa is set, the 2x2 blocks are inverted and the operands written diagonal-major for the solver
    // (what k_sor_prepare does from the planes); a11 .. b2 are then not written at all
    SorOperandOut op;
    int zero_duv;        // fused form: du = dv = 0 (first inner iteration, variational_mt.cpp:323-324): the planes are not read
    int chain_ok = 0;    // fused form, set by the launcher: the scalars admit the shared-reciprocal divisions (kernels.hip: recip_of / div_by)
};
void launch_assemble(sfa_ctx *c, const Geo &g, const AssembleArgs &a, const float *base /*element arena of batch 0*/, float *a11, float *a12,
                     float *a22, float *b1, float *b2, const float *du, const float *dv, const float *uu, const float *vv, const float *sh, const float *sv);

// The same system from the warped image pairs directly: derivative stacks formed in LDS tiles, never stored
// (get_derivatives' 8 filters + :293-320 mask weights + :336-365 in one pass).  occ: occlusion plane.
int launch_assemble_images(sfa_ctx *c, const Geo &g, const AssembleArgs &a, const float *base, float *a11, float *a12, float *a22, float *b1, float *b2,
                            const float *du, const float *dv, const float *uu, const float *vv, const float *sh, const float *sv, const float *occ);

// two-frame refinement (variational.c / variational_aux.c)
void launch_smoothness_2f(sfa_ctx *c, const Geo &g, float *sh, float *sv, const float *uu, const float *vv, const float *dpsis, float half_alpha);
void launch_data_2f(sfa_ctx *c, const Geo &g, const float *D, const float *mask, const float *du, const float *dv, float *a11, float *a12, float *a22, float *b1,
                    float *b2, const float *wx, const float *wy, const float *sh, const float *sv, float hd, float hg);

// ---- occlusion.hip: optimizeOcc (variational_aux_mt.cpp:758-887) ----------------------------------------------------
struct OccSlot {
    long s1_off, s2_off;     // image pair of the slot's successive-frames stack (arena offsets, like Term::i1_off)
    long r1_off, r2_off;     // image pair of its reference-frame stack (variational_mt.cpp:139-144)
    long mask_off;           // raw warp mask
    float rho, omega;        // rho[idx], omega[idx], idx = max(ref-s-1, s-ref)  (:814)
    int label;               // 0: slot in the future (s >= ref), 1: in the past  (:829-837)
};
struct OccArgs { OccSlot slot[2 * SFA_MAX_REF]; int nslots; float hd, hg, penalty; PenaltyDev color, grad; };
void launch_occ_costs(sfa_ctx *c, const Geo &g, const OccArgs &a, const float *base, float *d0, float *d1, long d_es /* window stride of d0,d1 */);
constexpr int kCutWorkPlanes = 13;
// exact two-label cut of sum D_l(p) + alpha * [l_p != l_q] over 4-neighbours; occ = 2*l - 1.  d0, d1, work: planes packed [nb][pl]
int run_grid_cut(sfa_ctx *c, const Geo &g, float *occ, long occ_es, const float *d0, const float *d1, float *work, float alpha);

// du/dv -> uu,vv, zero padding, L1 change norms (variational_mt.cpp:371-402); red = per-element 2 doubles (sum|old_du-du|, sum|old_dv-dv|)
void launch_update_inner(sfa_ctx *c, const Geo &g, float *uu, float *vv, const float *wx, const float *wy, const float *du, const float *dv,
                         const float *old_du, const float *old_dv, double *red /* [nb][2] */, float *dfa = nullptr, float *dfb = nullptr);   // dfa, dfb: the per-pixel terms (launch_exact_norms)
// the same with du,dv taken from a solver workspace's x plane (diagonal-major); old_du == nullptr: zeros; du_out == nullptr:
// du,dv are not stored
void launch_update_inner_x(sfa_ctx *c, const Geo &g, float *uu, float *vv, const float *wx, const float *wy, const SorOperandOut &x, const float *old_du,
                           const float *old_dv, float *du_out, float *dv_out, double *red, float *dfa = nullptr, float *dfb = nullptr);
// the norms of the windows `which` as the reference forms them (fp32 running sums over blocks of four pixels in raster order, the fp32 division) from the per-pixel terms an
// update left in dfa, dfb: out[2 b], out[2 b + 1] (device)
void launch_exact_norms(sfa_ctx *c, const Geo &g, const float *dfa, const float *dfb, const WMask &which, float *out);
// flow update of the LAST inner iteration straight from the x plane, fused with the outer update below (uu, vv, wx, wy all written)
void launch_update_outer_x(sfa_ctx *c, const Geo &g, float *uu, float *vv, float *wx, float *wy, const SorOperandOut &x, double *red, float *dfa = nullptr,
                           float *dfb = nullptr);   // dfa, dfb: where to leave the per-pixel |differences| (launch_outer_threshold)
// sum|uu-wx|, sum|vv-wy|; wx<-uu, wy<-vv (variational_mt.cpp:412-429)
void launch_update_outer(sfa_ctx *c, const Geo &g, float *wx, float *wy, const float *uu, const float *vv, double *red, float *dfa = nullptr, float *dfb = nullptr);
void launch_copy_planes(sfa_ctx *c, const Geo &g, float *dst, const float *src, int nplanes, long dst_es, long src_es);
void launch_scale_plane(sfa_ctx *c, const Geo &g, float *p, float s);

// pyramid (kernels.hip)
void launch_gauss_blur(sfa_ctx *c, const Geo &g, float *dst, float *tmp, const float *src, int nplanes, const float *taps, int radius);
void launch_resize(sfa_ctx *c, float *dst, int dw, int dh, int dpitch, long dpl, long des, const float *src, int sw, int sh, int spitch, long spl, long ses,
                   int nplanes, int nb, float post_scale);
void launch_resize_flow(sfa_ctx *c, float *dstx, float *dsty, int dw, int dh, int dpitch, long des, const float *srcx, const float *srcy, int sw, int sh, int spitch,
                        long ses, int nb, float post_x, float post_y);   // both planes of a flow field, k_resize's arithmetic
void launch_resize_scaled(sfa_ctx *c, float *dst, int dw, int dh, int dpitch, long dpl, long des, const float *src, int sw, int sh, int spitch, long spl,
                          long ses, int nplanes, int nb, float post_scale, double scale_x, double scale_y);
bool launch_pyr_down(sfa_ctx *c, float *dst, int dw, int dh, int dpitch, long dpl, long des, const float *src, int sw, int sh, int spitch, long spl, long ses,
                     int nplanes, int nb, const float *taps, int radius);   // blur + resize fused; false: footprint too large, use the two kernels
void launch_presmooth(sfa_ctx *c, const Geo &g, float *dst, float *tmp, const float *src, int nplanes, float sigma);
void launch_normalize_sums(sfa_ctx *c, const Geo &g, const float *frames3, double *red /* [3][2] */);
void launch_normalize_apply(sfa_ctx *c, const Geo &g, float *frames3, const double avg[3], const double stdv[3]);

// ---------------------------------------------------------------------------------------------------
// SOR (sor.hip)
// ---------------------------------------------------------------------------------------------------
struct SorWorkspace {
    sfa_ctx *ctx = nullptr;
    int w = 0, h = 0, K = 0, nb = 0;
    int NB = 0, NG = 0, RP = 0, ND = 0, G = 0, NS = 0, NCH = 0, ntasks = 0, F = 0, CHK = 0;
    int nwords = 0;               // progress words per window (one per task, each on a 128-byte line of its own)
    long ent = 0;                 // entries per element (ND*RP)
    int band = 0, Wp = 0, EP = 0;   // band kernel: fused sweeps per wave (0 = task kernel), edge row pitch / left pad
    int chain = 0;                  // sor_chain.hip shape id (0 = the band / task kernels); then NG = groups per band, NCH = chunks per stage, NS = barrier intervals
    long edge_job = 0;
    DevMem sa, sb, x, flags, order, edge;
    int configure(sfa_ctx *ctx, int w, int h, int K, int nb);   // (re)allocates for this shape
};
// planes: row-major device planes of element 0 (+es).  inv_out: write the inverted blocks back to a11/a12/a22
// the same in two halves for a producer that writes the operands itself (launch_assemble_images with a.op set)
int sor_operand_target(sfa_ctx *c, SorWorkspace &ws, const Geo &g, int K, SorOperandOut *out);
// du == nullptr: leave the result in the workspace's x plane (read it with launch_update_inner_x)
int sor_run_prepared(sfa_ctx *c, SorWorkspace &ws, const Geo &g, float *du, float *dv, int K, float omega);
// labelled mode slow_flow_sor_order red_black (a different algorithm; sor.hip): K two-colour sweeps on row-major planes, blocks inverted in place
int sor_rb_run(sfa_ctx *c, const Geo &g, float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2, const float *sh,
               const float *sv, int K, float omega);
int sor_run(sfa_ctx *c, SorWorkspace &ws, const Geo &g, float *du, float *dv, float *a11, float *a12, float *a22, const float *b1, const float *b2,
            const float *sh, const float *sv, int K, float omega, bool inv_out);

}  // namespace sfa
