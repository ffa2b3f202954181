// api.hip -- the C-ABI of include/slowflow_amd.h: context, host<->HBM staging, the level / pyramid
// orchestration (Variational_MT::variational and compute_one_level, variational_mt.cpp:169-493, 526-784)
// and the resident batch objects.  All compute is in kernels.hip / sor.hip; there is no CPU path.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>

#include "sfa_internal.h"

#pragma clang fp contract(off)

namespace sfa {

thread_local std::string g_thread_err;

// ---- the process-wide switch record (sfa_internal.h) ----------------------------------------------------------------
#ifndef SFA_RELEASE
Switches g_switches;
#endif
#ifndef SFA_RELEASE
static const char *const kSwitchNames[Switches::N] = {
    "SFA_SOR_CHAIN", "SFA_SOR_BAND", "SFA_SOR_F", "SFA_SOR_CH", "SFA_SOR_LEAD", "SFA_CHAIN_LDS", "SFA_RB_TILE", "SFA_WARP_ALLJ", "SFA_NO_WARP_SMOOTH",
    "SFA_ASSEMBLE_GENERIC", "SFA_EXACT_DIV", "SFA_ASM_XCD", "SFA_NO_DIRECT_OPERANDS", "SFA_NO_UV_ALIAS", "SFA_DEBUG_ACTIVE", "SFA_UNFUSED", "SFA_SHARE_SOR",
    "SFA_PYRAMID_UNFUSED", "SFA_CUT_DISCHARGE", "SFA_CUT_INNER", "SFA_CUT_SUPER", "SFA_CUT_TAIL_INNER", "SFA_CUT_PER", "SFA_CUT_TAIL_PER", "SFA_CUT_TAIL_SUPER",
    "SFA_CUT_DEBUG", "SFA_CUT_NO_TAIL", "SFA_CUT_TAIL", "SFA_NO_EXACT_BREAK"};
static int set_switch(const char *name, const char *value) {
    for (int i = 0; i < Switches::N; i++)
        if (!strcmp(name, kSwitchNames[i])) {
            g_switches.given[i] = value != nullptr;
            g_switches.value[i] = value ? atoi(value) : 0;
            return SFA_OK;
        }
    return SFA_ERR_ARG;
}
// the environment is looked at ONCE per process and only behind SFA_DEBUG=1 (tools/ and the A/B scripts set it)
static void switches_from_environment() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char *d = getenv("SFA_DEBUG");
        if (!d || atoi(d) == 0) return;
        for (int i = 0; i < Switches::N; i++)
            if (const char *e = getenv(kSwitchNames[i])) (void)set_switch(kSwitchNames[i], e);
    });
}
#else
static void switches_from_environment() {}      // (release build: there are no switches, sfa_internal.h)
#endif

int set_error(sfa_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_thread_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

int DevMem::alloc(sfa_ctx *ctx, size_t n) {
    if (n <= bytes && p) return SFA_OK;
    release();
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) { p = nullptr; return set_error(ctx, SFA_ERR_HIP, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
    bytes = n;
    return SFA_OK;
}
void DevMem::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

int upload_plane(sfa_ctx *ctx, float *dev, int pitch, const float *host, int stride, int w, int h) {
    SFA_HIP(ctx, hipMemcpy2DAsync(dev, (size_t)pitch * 4, host, (size_t)stride * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, ctx->stream));
    return SFA_OK;
}
int download_plane(sfa_ctx *ctx, float *host, int stride, const float *dev, int pitch, int w, int h) {
    SFA_HIP(ctx, hipMemcpy2DAsync(host, (size_t)stride * 4, dev, (size_t)pitch * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost, ctx->stream));
    return SFA_OK;
}

static int check_device_error(sfa_ctx *c) {
    unsigned e = 0;
    SFA_HIP(c, hipMemcpyAsync(&e, c->d_err, sizeof e, hipMemcpyDeviceToHost, c->stream));
    SFA_HIP(c, hipStreamSynchronize(c->stream));
    if (e) {
        (void)hipMemsetAsync(c->d_err, 0, sizeof(unsigned), c->stream);
        return set_error(c, SFA_ERR_TIMEOUT, "SOR pipeline: a bounded in-kernel wait gave up (code %u)", e);
    }
    return SFA_OK;
}

static PenaltyDev pen(const sfa_penalty &p) { return PenaltyDev{p.id, p.eps, p.trunc}; }

// ---------------------------------------------------------------------------------------------------
// Level: device-resident state of `nb` frame windows at one pyramid level.
// Element arena (floats, per window; the same element stride for every plane of every level), PL = pitch*h of the level:
//   PERSISTENT part, one per level (the pyramid is built before the coarse-to-fine loop and the flow travels from level to level):
//     wx wy (2 PL), frames F x 3 PL
//   TRANSIENT part, ONE for all levels (only one level is refined at a time), sized for the finest level:
//     planes : uu vv du dv odu odv sh sv a11 a12 a22 b1 b2 occ dpsis   (15 PL)
//     masks  : 2*ref PL
//     warped : [slot][w_s|w_sp1] (3 PL each); a factor-0 warp is the frame itself and is not materialised
//     stacks : [slot][succ|toref][24 PL] -- only in the unfused form (SFA_UNFUSED=1, kept to cross-check the fused kernel)
//     tmp    : two colour images for the pyramid / presmoothing: they are dead before the first warp, so they LIE ON the warped images (6 PL <= 12 ref PL)
// Offsets handed to kernels (Term, WarpJob, OccSlot) are relative to `base` (the transient part) and may point into the persistent part.
// Round 3, arena + solver workspaces: 0.43 -> 0.30 GB per 1024x436 window (S = 2, 5 levels), 4 -> 1.63 GB per 2048x2048 window (6 levels).
// ---------------------------------------------------------------------------------------------------
enum { P_WX = 0, P_WY, P_UU, P_VV, P_DU, P_DV, P_ODU, P_ODV, P_SH, P_SV, P_A11, P_A12, P_A22, P_B1, P_B2, P_OCC, P_DPSIS, P_COUNT };

struct Level {
    int w = 0, h = 0, pitch = 0, lstride = 0, ref = 0, F = 0, nb = 0;
    bool fused = true;
    long pl = 0, es = 0;
    long off_masks = 0, off_warp = 0, off_stacks = 0, off_tmp = 0;
    float *base = nullptr;    // transient part, element 0
    float *pbase = nullptr;   // this level's persistent part, element 0
    float *plane(int i) const { return i < 2 ? pbase + (long)i * pl : base + (long)(i - 2) * pl; }
    float *mask(int s) const { return base + off_masks + (long)s * pl; }
    float *stack(int s, int toref) const { return base + off_stacks + ((long)s * 2 + toref) * 24 * pl; }
    // frame s + sp1 warped by (s + sp1 - ref) flow steps.  Slot s's second image IS slot s + 1's first one (the same frame, the same number of steps): the reference
    // warps it once per slot (variational_mt.cpp:100,109); here the two slots read one buffer (S = 3: four warps per get_derivatives instead of six)
    float *warp(int s, int sp1) const { return base + off_warp + (long)(s + sp1) * 3 * pl; }
    float *frame(int f) const { return pbase + 2 * pl + (long)f * 3 * pl; }
    float *tmp() const { return base + off_tmp; }
    Geo geo() const { return Geo{w, h, pitch, pl, es, nb, WMask::first(nb), nullptr}; }
    Geo geo(const WMask &active) const { return Geo{w, h, pitch, pl, es, nb, active, nullptr}; }
    static long persistent_floats(int pitch, int h, int ref) { return (long)pitch * h * (2 + (2L * ref + 1) * 3); }
    static long transient_floats(int pitch, int h, int ref, bool fused) {
        return (long)pitch * h * ((P_COUNT - 2) + 2 * ref + (2L * ref + 1) * 3 + (fused ? 0 : 2L * ref * 2 * 24));     // masks, one warped image per frame (the reference frame's slot stays empty), stacks
    }
    void layout(float *transient, float *persistent, int w_, int h_, int lstride_, int ref_, int nb_, long es_, bool fused_) {
        base = transient; pbase = persistent; w = w_; h = h_; pitch = dev_pitch(w_); lstride = lstride_; ref = ref_; F = 2 * ref_ + 1; nb = nb_; es = es_; fused = fused_;
        pl = (long)pitch * h;
        off_masks = (long)(P_COUNT - 2) * pl;
        off_warp = off_masks + 2L * ref * pl;
        off_stacks = off_warp + (2L * ref + 1) * 3 * pl;
        off_tmp = off_warp;                                  // 6 PL inside the (2 ref + 1) * 3 >= 9 PL of the warped images (ref >= 1)
    }
};

struct ChannelWeights { const float *dev = nullptr; long pl = 0, es = 0; int pitch = 0, stride0 = 0; };

// the image pair of slot s (variational_mt.cpp:98-110): frames s, s+1 warped by (s-ref), (s-ref+1) flow steps
static const float *pair_image(const Level &L, int s, int sp1) { return (s + sp1 - L.ref == 0) ? L.frame(s + sp1) : L.warp(s, sp1); }

// get_derivatives (variational_mt.cpp:87-166).  Fused form: only the warps; the filters run inside the assembly kernel.
// with_smoothness: compute_smoothness (:333) of the same flow field leaves in the same pass (the caller's next step, fused form with one inner iteration); returns
// whether it did
static bool get_derivatives(sfa_ctx *c, const Level &L, const sfa_params &p, const Geo &g, const bool need_toref[2 * SFA_MAX_REF], bool with_smoothness = false) {
    const int ref = L.ref;
    WarpJobs J;
    J.n = 0;
    // one warp per frame f != ref, by f - ref steps: it is w_s of slot f (:100) and w_sp1 of slot f - 1 (:109) -- the same frame by the same steps -- and yields the mask
    // of the slot on its own side of the reference frame: slot f backwards (f < ref), slot f - 1 forwards (f > ref); a zero-step warp is a copy (:723-728): never made
    for (int f = p.one_direction ? ref + 1 : 0; f <= 2 * ref; f++) {
        if (f == ref) continue;
        const int s = f < ref ? f : f - 1;                       // the slot whose mask this warp yields; L.warp(f, 0) == L.warp(f - 1, 1)
        J.job[J.n++] = WarpJob{L.frame(f) - L.base, L.warp(f < ref ? f : f - 1, f < ref ? 0 : 1) - L.base, L.mask(s) - L.base, f - ref};
    }
    const bool smoothed = with_smoothness && L.fused &&
                          launch_warp_smooth(c, g, J, L.base, L.plane(P_WX), L.plane(P_WY), p.smoothing, L.plane(P_SH), L.plane(P_SV), L.plane(P_DPSIS), p.alpha, pen(p.robust_reg));
    if (!smoothed) launch_warp_jobs(c, g, J, L.base, L.plane(P_WX), L.plane(P_WY));
    for (int s = p.one_direction ? ref : 0; s < 2 * ref; s++) {
        if (L.fused) continue;
        const float *w_s = pair_image(L, s, 0), *w_sp1 = pair_image(L, s, 1);
        launch_deriv_stack(c, g, L.stack(s, 0), w_s, w_sp1, L.es, L.es);                                        // :113-133
        // the to-reference stack (:136-161) only feeds add_data_and_match_ref (omega > 0) and optimizeOcc
        if (need_toref[s]) {
            if (s < ref) launch_deriv_stack(c, g, L.stack(s, 1), w_s, L.frame(ref), L.es, L.es);                 // :139-141
            else         launch_deriv_stack(c, g, L.stack(s, 1), L.frame(ref), w_sp1, L.es, L.es);               // :143-144
        }
    }
    return smoothed;
}

// compute_one_level (variational_mt.cpp:169-493) for all batch elements in lockstep.
// change: nb x 2 floats (host).  Thresholds <= 0 never break, so no host round trip is needed.
// optimizeOcc (variational_aux_mt.cpp:758-887) for all windows: data costs from the warped pairs, exact two-label cut
static int optimize_occlusions(sfa_ctx *c, const Level &L, const sfa_params &p, const Geo &g, DevMem &scratch) {
    const int ref = L.ref;
    const size_t n = (size_t)L.nb * L.pl;
    SFA_TRY(scratch.alloc(c, (2 + kCutWorkPlanes) * n * sizeof(float)));
    float *d0 = scratch.f(), *d1 = d0 + n, *work = d1 + n;
    OccArgs oa;
    memset(&oa, 0, sizeof oa);
    oa.nslots = 2 * ref; oa.hd = p.delta / 3.0f; oa.hg = p.gamma / 3.0f; oa.penalty = p.occlusion_penalty;
    oa.color = pen(p.robust_color); oa.grad = pen(p.robust_grad);
    for (int s = 0; s < 2 * ref; s++) {
        const float *i1 = pair_image(L, s, 0), *i2 = pair_image(L, s, 1);
        const float *r1 = s < ref ? i1 : L.frame(ref), *r2 = s < ref ? L.frame(ref) : i2;                  // variational_mt.cpp:139-144
        const int idx = std::max(ref - s - 1, s - ref);
        oa.slot[s] = OccSlot{i1 - L.base, i2 - L.base, r1 - L.base, r2 - L.base, L.off_masks + (long)s * L.pl, p.rho[idx], p.omega[idx], s >= ref ? 0 : 1};
    }
    launch_occ_costs(c, g, oa, L.base, d0, d1, L.pl);
    return run_grid_cut(c, g, L.plane(P_OCC), L.es, d0, d1, work, p.occlusion_alpha);
}

// occ_log (level 0 only, or null): [nb][niter_alter][pl] floats, the labels after the discrete step of alternation a >= 1 -- what the
// reference writes as <slow_flow_occlusions_output><a>.png at every level, the finest level's file surviving (:275-285)
static int run_level(sfa_ctx *c, const Level &L, const sfa_params &p, const ChannelWeights &cw, SorWorkspace &sorws, DevMem &cut_scratch, float *change,
                     float *occ_log) {
    const int ref = L.ref;
    const float gamma_over3 = p.gamma / 3.0f, delta_over3 = p.delta / 3.0f;                                   // :548-549
    WMask active = WMask::first(L.nb);
    const WMask all = active;
    Geo g = L.geo(active);
    // The reference's per-iteration lines (variational_mt.cpp:404-405, 431-432: "inner it i avg change a,b" / "outer it i avg change a,b" under verbosity(VER_CMD)).
    // Printing them needs the norms on the host after every iteration -- a synchronisation per iteration --, so it is off unless SFA_VERBOSE_CHANGES is set (the C++
    // class and the driver set it when the cfg's `verbose` asks for it).  Batches print one line per window that still iterates, in window order.
    const bool verbose = c->verbose_changes;
    auto print_changes = [&](const char *what, int it, const WMask &who) {
        if (hipMemcpyAsync(c->h_red, c->d_red, 2 * L.nb * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return;
        const double n = (double)L.w * L.h;
        for (int b = 0; b < L.nb; b++)
            if (who.test(b)) {
                if (L.nb > 1) printf("[window %d] ", b);
                printf("%s %d\tavg change %g,%g\n", what, it, (double)(float)(c->h_red[2 * b] / n), (double)(float)(c->h_red[2 * b + 1] / n));
            }
        fflush(stdout);
    };

    // occlusions: 0, or -1 with one_direction / occlusion reasoning (:216-220)
    {
        const float occ0 = (p.one_direction || p.occlusion_reasoning) ? -1.0f : 0.0f;
        launch_fill_planes(c, g, L.plane(P_OCC), 1, occ0);
    }
    float data_norm = 0;                                                                                     // :223-226
    for (int s = 0; s < ref; s++) data_norm += p.rho[s] + p.omega[s];

    launch_dpsis(c, g, L.plane(P_DPSIS), L.frame(ref), L.es, 5.0f, p.norm_avg, p.norm_std, p.hbit);           // :257
    // uu = wx + du, vv = wy + dv (:396-397), and wx <- uu, wy <- vv at the end of every outer iteration (:428-429).  With ONE inner iteration and the fused update
    // (k_update_outer_x) the two pairs of planes always hold the same values when anybody reads them: smoothness and assembly then read wx, wy, and the update
    // writes 16 instead of 32 bytes per pixel.
    const bool uv_alias = L.fused && p.sor_order != 1 && !sw_given(Switches::NO_DIRECT_OPERANDS) && p.niter_inner == 1 && !sw_given(Switches::NO_UV_ALIAS) && !verbose;
    float *const UU = uv_alias ? L.plane(P_WX) : L.plane(P_UU), *const VV = uv_alias ? L.plane(P_WY) : L.plane(P_VV);
    if (!uv_alias) launch_copy_planes(c, g, L.plane(P_UU), L.plane(P_WX), 2, L.es, L.es);                     // :260-261 (wx,wy and uu,vv adjacent)

    // which terms are active (:343-361), in the reference's call order
    AssembleArgs aa;
    memset(&aa, 0, sizeof aa);
    bool need_toref[2 * SFA_MAX_REF] = {false};
    for (int s = 0; s < ref; s++) {
        if (!p.one_direction) {
            if (p.rho[ref - 1 - s] > 0)
                aa.t[aa.n++] = Term{L.off_stacks + ((long)s * 2 + 0) * 24 * L.pl, L.off_masks + (long)s * L.pl, p.rho[ref - 1 - s] * delta_over3,
                                    p.rho[ref - 1 - s] * gamma_over3, (float)(s - ref), 0};
            if (p.omega[ref - 1 - s] > 0) {
                aa.t[aa.n++] = Term{L.off_stacks + ((long)s * 2 + 1) * 24 * L.pl, L.off_masks + (long)s * L.pl, p.omega[ref - 1 - s] * delta_over3,
                                    p.omega[ref - 1 - s] * gamma_over3, (float)(s - ref), 1};
                need_toref[s] = true;
            }
        }
        if (p.rho[s] > 0)
            aa.t[aa.n++] = Term{L.off_stacks + ((long)(ref + s) * 2 + 0) * 24 * L.pl, L.off_masks + (long)(ref + s) * L.pl, p.rho[s] * delta_over3,
                                p.rho[s] * gamma_over3, (float)s, 0};
        if (p.omega[s] > 0) {
            aa.t[aa.n++] = Term{L.off_stacks + ((long)(ref + s) * 2 + 1) * 24 * L.pl, L.off_masks + (long)(ref + s) * L.pl, p.omega[s] * delta_over3,
                                p.omega[s] * gamma_over3, (float)(s + 1), 1};
            need_toref[ref + s] = true;
        }
    }
    for (int t = 0; t < aa.n; t++) {                                     // slot and image pair of each term
        Term &T = aa.t[t];
        const int slot = (int)((T.mask_off - L.off_masks) / L.pl);
        const float *i1 = pair_image(L, slot, 0), *i2 = pair_image(L, slot, 1);
        if (T.is_ref) { if (slot < ref) i2 = L.frame(ref); else i1 = L.frame(ref); }                         // :139-144
        T.i1_off = i1 - L.base; T.i2_off = i2 - L.base; T.backward = slot < ref;
    }
    aa.data_norm = data_norm; aa.one_direction = p.one_direction;
    for (int t = 0; t < aa.n; t++)
        if (aa.t[t].is_ref && aa.t[t].s == 0) return set_error(c, SFA_ERR_REF_FRAME, "Frame compared to reference frame is the reference frame itself!");
    aa.dt_norm = p.dataterm_norm;
    aa.color = pen(p.robust_color); aa.grad = pen(p.robust_grad);
    aa.chw = cw.dev; aa.chw_pl = cw.pl; aa.chw_es = cw.es; aa.chw_pitch = cw.pitch; aa.chw_stride0 = cw.stride0; aa.lstride = L.lstride;
    aa.accumulate = 0; aa.do_laplacian = 1;

    const bool use_thres_in = p.thres_inner > 0, use_thres_out = p.thres_outer > 0;
    double *red = c->d_red;
    const double npx = (double)L.h * L.w;
    // The outer break (:431-436) is taken ON THE DEVICE: k_outer_threshold clears a window's bit in *d_amask when its norms meet the threshold, and every
    // kernel (the solver included) leaves the windows without a bit alone.  The host never waits for the iteration it has just queued: it reads the mask
    // of kLag iterations ago (a superset -- windows only ever leave) to stop queueing once nothing iterates any more, so the GPU always has work queued
    // and at most kLag iterations of empty launches follow the last window's break.  One blocking read per level (the norms), not one per iteration.
    constexpr int kLag = kMaskLag, kRing = kMaskRing;
    const bool dbg = sw_given(Switches::DEBUG_ACTIVE);
    g.amask = c->d_amask;
    // With an outer threshold the update leaves the per-pixel terms of the norms in the a11 / a12 planes (dead by then: the direct form never writes them, the other
    // forms' solver has read them), so that a window whose fp64 norm lies within break_band() of the threshold can be decided by the reference's own fp32 running sums
    float *const dfa = use_thres_out && !sw_given(Switches::NO_EXACT_BREAK) ? L.plane(P_A11) : nullptr, *const dfb = dfa ? L.plane(P_A12) : nullptr;
    float *const ifa = use_thres_in && !sw_given(Switches::NO_EXACT_BREAK) ? L.plane(P_A11) : nullptr, *const ifb = ifa ? L.plane(P_A12) : nullptr;   // the inner break's
    SFA_HIP(c, hipMemsetAsync(c->d_last, 0, sizeof(LastBlock), c->stream));   // (the norms, the windows' finished-block counters of k_update_outer_x, ...)
    launch_set_mask(c, all);

    for (int alter = 0; alter < p.niter_alter; alter++) {
        active = all;
        g.active = active;
        if (use_thres_out && alter > 0) launch_set_mask(c, all);
        bool smoothed = get_derivatives(c, L, p, g, need_toref, uv_alias);                                  // :266 (+ :333 of the first outer iteration)
        if (alter > 0 && p.occlusion_reasoning && !p.one_direction) SFA_TRY(optimize_occlusions(c, L, p, g, cut_scratch));   // :269-272
        if (alter > 0 && p.occlusion_reasoning && occ_log)                                                  // :275-285
            launch_copy_planes(c, g, occ_log + (long)alter * L.pl, L.plane(P_OCC), 1, (long)p.niter_alter * L.pl, L.es);
        for (int outer = 0; outer < p.niter_outer; outer++) {
            if (use_thres_out && outer >= kLag) {
                const int slot = (outer - kLag) % kRing;
                SFA_HIP(c, hipEventSynchronize(c->ev_mask[slot]));
                for (int i = 0; i < kMaskWords; i++) active.w[i] = *(volatile unsigned long long *)&c->h_amask[slot].w[i];
                if (!active.any()) break;                                                                // :436, every window
            }
            if (dbg) fprintf(stderr, "level %dx%d alter %d outer %d known active %d\n", L.w, L.h, alter, outer, active.count());
            g.active = active;
            if (outer > 0) smoothed = get_derivatives(c, L, p, g, need_toref, uv_alias);                    // :289-290 (+ :333)
            if (!L.fused) launch_mask_weight(c, g, L.mask(0), L.plane(P_OCC), data_norm, ref, p.one_direction);   // :293-320
            // in the direct form the first inner iteration never touches du / dv / old du / old dv: they are zeros by construction.
            // Windows that already met a threshold stay in the lockstep launches as passengers: every kernel skips them (Geo::active and
            // Geo::amask; the solver's workgroups of a passenger return as soon as they have drawn their ticket).
            const bool red_black = p.sor_order == 1;           // labelled mode: works on the row-major planes, never on the diagonal-major operands
            const bool direct_outer = L.fused && !red_black && !sw_given(Switches::NO_DIRECT_OPERANDS);
            if (!direct_outer) launch_zero_planes(c, g, L.plane(P_DU), 2);                                   // :323-324 (du, dv adjacent)
            WMask in_active = active, outer_done = WMask::none();
            for (int inner = 0; inner < p.niter_inner; inner++) {
                Geo gi = g;
                gi.active = in_active;
                const bool direct = direct_outer;
                // an inner break is decided behind this iteration: the update leaves the per-pixel terms of its norms (a11 / a12 planes: dead, see dfa), so that a
                // window whose fp64 norm lies within break_band() of the threshold can be decided by the reference's own fp32 sums
                const bool decide_in = use_thres_in && inner + 1 < p.niter_inner && ifa;
                const bool first_zero = direct && inner == 0;        // du = dv = 0 known, planes possibly stale
                if (!first_zero) {
                    launch_copy_planes(c, gi, L.plane(P_ODU), L.plane(P_DU), 2, L.es, L.es);                // :329-330
                }
                if (!(smoothed && inner == 0))                      // (uv_alias: one inner iteration, UU / VV are wx / wy -- what the fused pass read)
                    launch_smoothness(c, gi, p.smoothing, L.plane(P_SH), L.plane(P_SV), UU, VV, L.plane(P_DPSIS), p.alpha,
                                      pen(p.robust_reg));                                                   // :333
                // the fused assembly can leave the solver's operands directly (no a11 .. b2 planes, no prepare pass) when the
                // whole batch is solved in one launch
                aa.op = SorOperandOut();
                aa.zero_duv = first_zero ? 1 : 0;
                if (direct) SFA_TRY(sor_operand_target(c, sorws, gi, p.niter_solver, &aa.op));
                if (L.fused)
                    SFA_TRY(launch_assemble_images(c, gi, aa, L.base, L.plane(P_A11), L.plane(P_A12), L.plane(P_A22), L.plane(P_B1), L.plane(P_B2), L.plane(P_DU),
                                                   L.plane(P_DV), UU, VV, L.plane(P_SH), L.plane(P_SV), L.plane(P_OCC)));   // :293-365
                else
                    launch_assemble(c, gi, aa, L.base, L.plane(P_A11), L.plane(P_A12), L.plane(P_A22), L.plane(P_B1), L.plane(P_B2), L.plane(P_DU),
                                    L.plane(P_DV), L.plane(P_UU), L.plane(P_VV), L.plane(P_SH), L.plane(P_SV));   // :336-365
                if (direct) {
                    SFA_TRY(sor_run_prepared(c, sorws, gi, nullptr, nullptr, p.niter_solver, p.sor_omega));             // :368
                } else if (red_black) {
                    SFA_TRY(sor_rb_run(c, gi, L.plane(P_DU), L.plane(P_DV), L.plane(P_A11), L.plane(P_A12), L.plane(P_A22), L.plane(P_B1), L.plane(P_B2),
                                       L.plane(P_SH), L.plane(P_SV), p.niter_solver, p.sor_omega));
                } else {
                    SFA_TRY(sor_run(c, sorws, gi, L.plane(P_DU), L.plane(P_DV), L.plane(P_A11), L.plane(P_A12), L.plane(P_A22), L.plane(P_B1),
                                    L.plane(P_B2), L.plane(P_SH), L.plane(P_SV), p.niter_solver, p.sor_omega, false));   // :368
                }
                if (direct && inner + 1 == p.niter_inner && !verbose) {
                    // last inner iteration: nothing reads its inner norms or du/dv; the flow update and the outer update run as one pass (with the per-iteration
                    // lines on somebody does read the inner norms, :404-405: the two passes below)
                    launch_update_outer_x(c, gi, uv_alias ? nullptr : L.plane(P_UU), uv_alias ? nullptr : L.plane(P_VV), L.plane(P_WX), L.plane(P_WY), aa.op, red, dfa, dfb);   // :396-397 + :412-429
                    outer_done = in_active;
                } else if (direct) {
                    const bool keep = inner + 1 < p.niter_inner;      // du, dv are read again only by a further inner iteration
                    launch_update_inner_x(c, gi, L.plane(P_UU), L.plane(P_VV), L.plane(P_WX), L.plane(P_WY), aa.op, first_zero ? nullptr : L.plane(P_ODU),
                                          first_zero ? nullptr : L.plane(P_ODV), keep ? L.plane(P_DU) : nullptr, keep ? L.plane(P_DV) : nullptr, red, decide_in ? ifa : nullptr,
                                          decide_in ? ifb : nullptr);                                       // :371-402
                } else
                    launch_update_inner(c, gi, L.plane(P_UU), L.plane(P_VV), L.plane(P_WX), L.plane(P_WY), L.plane(P_DU), L.plane(P_DV), L.plane(P_ODU),
                                        L.plane(P_ODV), red, decide_in ? ifa : nullptr, decide_in ? ifb : nullptr);   // :371-402
                if (verbose) print_changes("\tinner it", inner, in_active);                                  // :404-405
                if (use_thres_in && inner + 1 < p.niter_inner) {
                    SFA_HIP(c, hipMemcpyAsync(c->h_red, red, 2 * L.nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                    SFA_HIP(c, hipStreamSynchronize(c->stream));
                    WMask close = WMask::none();
                    for (int b = 0; b < L.nb; b++)
                        if (in_active.test(b)) {
                            const double ad = c->h_red[2 * b] / npx, dd = c->h_red[2 * b + 1] / npx, dm = (ad < dd) ? dd : ad;
                            if (decide_in && fabs(dm - (double)p.thres_inner) <= break_band(L.w, L.h) * (double)p.thres_inner) { close.set(b); continue; }
                            const float a = (float)ad, d = (float)dd;
                            if (std::max(a, d) < p.thres_inner) in_active.clear(b);                         // :407
                        }
                    if (close.any()) {
                        float *const dx = c->d_last->exact;
                        const float *const hx = reinterpret_cast<const LastBlock *>(c->h_red)->exact;        // pinned (a pageable target would be a staged, blocking copy)
                        launch_exact_norms(c, gi, ifa, ifb, close, dx);
                        SFA_HIP(c, hipMemcpyAsync(const_cast<float *>(hx), dx, 2 * L.nb * sizeof(float), hipMemcpyDeviceToHost, c->stream));
                        SFA_HIP(c, hipStreamSynchronize(c->stream));
                        for (int b = 0; b < L.nb; b++)
                            if (close.test(b) && std::max(hx[2 * b], hx[2 * b + 1]) < p.thres_inner) in_active.clear(b);   // :407 on the reference's own sums
                    }
                    if (!in_active.any()) break;
                }
            }
            if (outer_done != active) {
                // windows that left the inner loop early (or the unfused form): their outer update.  The reductions only write the result
                // words of the windows of their Geo::active, so the norms of the windows updated above stay in `red`.
                Geo go = g;
                go.active = active.andnot(outer_done);
                launch_update_outer(c, go, L.plane(P_WX), L.plane(P_WY), L.plane(P_UU), L.plane(P_VV), red, dfa, dfb);            // :412-429
            }
            if (verbose) print_changes("outer it", outer, active);                                            // :431-432
            const bool last_iter = (alter == p.niter_alter - 1 && outer == p.niter_outer - 1);
            if (use_thres_out || last_iter) launch_outer_threshold(c, g, red, use_thres_out ? p.thres_outer : 0.0f, dfa, dfb);   // :431-436
            if (use_thres_out) {
                const int slot = outer % kRing;
                SFA_HIP(c, hipMemcpyAsync(&c->h_amask[slot], c->d_amask, sizeof(WMask), hipMemcpyDeviceToHost, c->stream));
                SFA_HIP(c, hipEventRecord(c->ev_mask[slot], c->stream));
            }
        }
    }
    // the norms of every window's last outer iteration -- where the caller wants them (the finest level: the coarser levels' are overwritten, variational_mt.cpp:761), and
    // only there does the host wait for the level: a blocking read per level left the GPU idle for ~30 us five times per run (a lone window: 2 % of its time)
    (void)npx;
    if (change) {
        SFA_HIP(c, hipMemcpyAsync(c->h_red, c->d_last->last, 2 * L.nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        SFA_HIP(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < 2 * L.nb; i++) change[i] = (float)c->h_red[i];
    }
    return SFA_OK;
}

// ---------------------------------------------------------------------------------------------------
// pyramid geometry (variational_mt.cpp:576-652)
// ---------------------------------------------------------------------------------------------------
static int gaussian_filter_order(float sigma) {      // image.c:320-322
    int order = (int)floor(3 * sigma) + 1;
    if (order == 0) order = 1;
    return order;
}
static int pyramid_sizes(int w, int h, int layers, float p_scale, int *ws, int *hs) {
    const float sigma = 1 / sqrtf(2 * p_scale);      // :578
    const int order = gaussian_filter_order(sigma);
    int L = layers;
    for (int l = 0; l < layers; l++) {
        if (l == 0) { ws[0] = w; hs[0] = h; }
        else {
            ws[l] = (int)(float)floor(ws[l - 1] * p_scale);   // :609-611 (product rounded to fp32 before the floor)
            hs[l] = (int)(float)floor(hs[l - 1] * p_scale);
        }
        if (floor(ws[l] * p_scale) <= order + 1 || floor(hs[l] * p_scale) <= order + 1) { L = l; break; }   // :647-651
    }
    return L;
}
// cv::getGaussianKernel(ksize, sigma, CV_32F) with ksize = cvRound(sigma*8+1)|1 (cv::GaussianBlur, Size(0,0), CV_32F)
static int cv_gauss_taps(float sigma, float *k) {
    const int ksize = ((int)lrint((double)sigma * 4 * 2 + 1)) | 1;
    const double scale2X = -0.5 / ((double)sigma * sigma);
    double sum = 0;
    for (int i = 0; i < ksize; i++) {
        const double x = i - (ksize - 1) * 0.5;
        k[i] = (float)exp(scale2X * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < ksize; i++) k[i] = (float)(k[i] * sum);
    return ksize / 2;
}

}  // namespace sfa

using namespace sfa;

// ---------------------------------------------------------------------------------------------------
// the resident job (variational over a batch)
// ---------------------------------------------------------------------------------------------------
struct sfa_job {
    sfa_ctx *ctx = nullptr;
    sfa_params p;
    int w = 0, h = 0, nb = 0, ref = 0, F = 0, L = 0;
    int ws[64], hs[64];
    long es = 0;                       // floats per element (all levels)
    std::vector<long> level_off;       // offset of each level's PERSISTENT part inside the element
    long trans_off = 0;                // offset of the transient part (shared by all levels)
    bool share_sor = false;            // large frames: one solver workspace for all levels (re-shaped per level), not one per level
    DevMem arena;                      // nb * es floats
    DevMem init_flow;                  // nb x 2 planes at level-0 pitch: the uploaded initial flow
    DevMem chw;                        // nb x 3 planes (level-0 pitch) or empty
    DevMem cut_scratch;                // occlusion step: 2 cost planes + the cut's work planes, [nb][pl] each, grown on demand
    bool has_chw = false;
    int chw_stride0 = 0;
    std::vector<std::unique_ptr<SorWorkspace>> sor;   // one per level: no re-allocation between runs
    std::vector<float> change;         // nb x 2
    double mpix_iters = 0;
    Level level(int l) const {
        Level Lv;
        Lv.layout(arena.f() + trans_off, arena.f() + level_off[l], ws[l], hs[l], l == 0 ? host_stride0 : host_stride(ws[l]), ref, nb, es, fused);
        return Lv;
    }
    int host_stride0 = 0;
    bool fused = true;                 // false: SFA_UNFUSED=1 at creation (stack planes materialised; cross-check only)
    WMask presmoothed = WMask::none();    // windows whose level-0 frames already hold the presmoothed images (cfg sigma > 0): smoothing is applied once per upload
    bool keep_alt_occ = false;         // record the occlusion labels of every alternation (slow_flow_occlusions_output, variational_mt.cpp:275-285)
    DevMem occ_log;                    // [nb][niter_alter][pl(level 0)]
};

extern "C" {

int sfa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sfa_ctx_create(int device, sfa_ctx **out) {
    if (!out) return set_error(nullptr, SFA_ERR_ARG, "sfa_ctx_create: out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error(nullptr, SFA_ERR_NO_DEVICE, "no HIP device available: slowflow_amd has no CPU fallback");
    if (device < 0 || device >= n) return set_error(nullptr, SFA_ERR_ARG, "device %d out of range (%d devices)", device, n);
    switches_from_environment();
    std::unique_ptr<sfa_ctx> c(new sfa_ctx());
    c->device = device;
    SFA_HIP(c.get(), hipSetDevice(device));
    SFA_HIP(c.get(), hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SFA_HIP(c.get(), hipMalloc((void **)&c->d_red, kRedDoubles * sizeof(double)));
    SFA_HIP(c.get(), hipHostMalloc((void **)&c->h_red, sizeof(LastBlock) + 64, hipHostMallocDefault));
    SFA_HIP(c.get(), hipMalloc((void **)&c->d_amask, 64));
    SFA_HIP(c.get(), hipMalloc((void **)&c->d_last, sizeof(LastBlock)));
    SFA_HIP(c.get(), hipMemset(c->d_last, 0, sizeof(LastBlock)));
    SFA_HIP(c.get(), hipHostMalloc((void **)&c->h_amask, kMaskRing * sizeof(WMask), hipHostMallocDefault));
    for (auto &e : c->ev_mask) SFA_HIP(c.get(), hipEventCreateWithFlags(&e, hipEventDisableTiming));
    SFA_HIP(c.get(), hipMalloc((void **)&c->d_err, 64));
    SFA_HIP(c.get(), hipMemset(c->d_err, 0, 64));
    SFA_HIP(c.get(), hipEventCreate(&c->t0));
    SFA_HIP(c.get(), hipEventCreate(&c->t1));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    *out = c.release();
    return SFA_OK;
}

void sfa_ctx_destroy(sfa_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto e : c->ev) (void)hipEventDestroy(e);
    for (auto e : c->ev2) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->h_red) (void)hipHostFree(c->h_red);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_amask) (void)hipFree(c->d_amask);
    if (c->d_last) (void)hipFree(c->d_last);
    if (c->h_amask) (void)hipHostFree(c->h_amask);
    for (auto e : c->ev_mask) if (e) (void)hipEventDestroy(e);
    if (c->rb_tmp) (void)hipFree(c->rb_tmp);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *sfa_last_error(const sfa_ctx *c) { return c ? c->err.c_str() : g_thread_err.c_str(); }

int sfa_ctx_sync(sfa_ctx *c) {
    if (!c) return set_error(nullptr, SFA_ERR_ARG, "null context");
    SFA_HIP(c, hipSetDevice(c->device));
    return check_device_error(c);
}

void sfa_params_default(sfa_params *p) {           // slow_flow.cpp:64-128
    memset(p, 0, sizeof *p);
    p->S = 2; p->one_direction = 0; p->smoothing = 1; p->dataterm_norm = 1;
    p->niter_alter = 10; p->niter_outer = 10; p->niter_inner = 1; p->niter_solver = 30;
    p->thres_outer = 1e-5f; p->thres_inner = 1e-5f; p->sor_omega = 1.9f;
    p->alpha = 4.0f; p->gamma = 6.0f; p->delta = 1.0f;
    p->robust_color = sfa_penalty{1, 0.001f, 0.5f};
    p->robust_grad = p->robust_color; p->robust_reg = p->robust_color;
    for (int a = 0; a < SFA_MAX_REF; a++) { p->rho[a] = 1; p->omega[a] = 1; }       // variational_mt.cpp:561-568: "1.0" where the cfg names nothing ...
    p->omega[0] = 0; p->omega[1] = 2;                                               // ... and slow_flow.cpp:96-99 for the first two
    p->hbit = 1;
    for (int k = 0; k < 3; k++) { p->norm_avg[k] = 0; p->norm_std[k] = 1; }
    p->occlusion_reasoning = 1; p->layers = 1; p->p_scale = 0.9f; p->presmooth_sigma = 0;
    p->sor_order = 0;                                                                  // the reference's raster order
    p->occlusion_penalty = 0.1f; p->occlusion_alpha = 0.1f; p->niter_graphc = 10;     // slow_flow.cpp:117-118 (the class itself falls back to 1.0 / 0.5, variational_mt.cpp:189-190)
}

int sfa_pyramid_sizes(int w, int h, int layers, float p_scale, int *ws, int *hs) {
    if (layers < 1 || layers > 64 || !ws || !hs) return 0;
    return pyramid_sizes(w, h, layers, p_scale, ws, hs);
}

// ---- profiling / timing -----------------------------------------------------------------------------
int sfa_profile_enable(sfa_ctx *c, int on) {
    if (!c) return SFA_ERR_ARG;
    SFA_HIP(c, hipSetDevice(c->device));
    c->profile = on != 0;
    c->ev_used = 0;
    c->sor_bytes = 0;
    c->ev2_used = 0;
    c->asm_pixel_terms = 0;
    if (on && c->ev.empty()) {
        c->ev.resize(4096);
        for (auto &e : c->ev) SFA_HIP(c, hipEventCreate(&e));
        c->ev2.resize(4096);
        for (auto &e : c->ev2) SFA_HIP(c, hipEventCreate(&e));
    }
    return SFA_OK;
}
int sfa_profile_read_kernels(sfa_ctx *c, int *n_asm, double *asm_ms_total, double *asm_pixel_terms, char *sor_kernel, int sor_kernel_len) {
    if (!c) return SFA_ERR_ARG;
    SFA_HIP(c, hipStreamSynchronize(c->stream));
    double tot = 0;
    for (size_t i = 0; i + 1 < c->ev2_used; i += 2) {
        float ms = 0;
        SFA_HIP(c, hipEventElapsedTime(&ms, c->ev2[i], c->ev2[i + 1]));
        tot += ms;
    }
    if (n_asm) *n_asm = (int)(c->ev2_used / 2);
    if (asm_ms_total) *asm_ms_total = tot;
    if (asm_pixel_terms) *asm_pixel_terms = c->asm_pixel_terms;
    if (sor_kernel && sor_kernel_len > 0) snprintf(sor_kernel, (size_t)sor_kernel_len, "%s", c->sor_kernel);
    c->ev2_used = 0;
    c->asm_pixel_terms = 0;
    return SFA_OK;
}
int sfa_profile_read(sfa_ctx *c, int *n, double *ms_total, double *bytes_total) {
    if (!c) return SFA_ERR_ARG;
    SFA_HIP(c, hipStreamSynchronize(c->stream));
    double tot = 0;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0;
        SFA_HIP(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        tot += ms;
    }
    if (n) *n = (int)(c->ev_used / 2);
    if (ms_total) *ms_total = tot;
    if (bytes_total) *bytes_total = c->sor_bytes;
    c->ev_used = 0;
    c->sor_bytes = 0;
    return SFA_OK;
}
int sfa_debug_set(const char *name, const char *value) {
    if (!name) return set_error(nullptr, SFA_ERR_ARG, "sfa_debug_set: null name");
#ifdef SFA_RELEASE
    (void)value;
    return set_error(nullptr, SFA_ERR_ARG, "sfa_debug_set('%s'): this is the release build of the library -- it has no cross-check / what-if paths (build the full library: make -C slowflow_amd/csrc)", name);
#else
    switches_from_environment();                    // so that a later first sfa_ctx_create cannot overwrite what is set here
    if (set_switch(name, value) != SFA_OK) return set_error(nullptr, SFA_ERR_ARG, "sfa_debug_set: unknown switch '%s'", name);
    return SFA_OK;
#endif
}
int sfa_ctx_set_verbose(sfa_ctx *c, int on) {
    if (!c) return SFA_ERR_ARG;
    c->verbose_changes = on != 0;
    return SFA_OK;
}
int sfa_ctx_set_wait_bound(sfa_ctx *c, unsigned spins) {
    if (!c) return SFA_ERR_ARG;
    SFA_HIP(c, hipSetDevice(c->device));
    SFA_HIP(c, hipMemcpyAsync(c->d_err + 1, &spins, sizeof spins, hipMemcpyHostToDevice, c->stream));
    SFA_HIP(c, hipStreamSynchronize(c->stream));
    return SFA_OK;
}
int sfa_timer_start(sfa_ctx *c) {
    if (!c) return SFA_ERR_ARG;
    SFA_HIP(c, hipEventRecord(c->t0, c->stream));
    return SFA_OK;
}
int sfa_timer_stop(sfa_ctx *c, float *ms) {
    if (!c) return SFA_ERR_ARG;
    SFA_HIP(c, hipEventRecord(c->t1, c->stream));
    SFA_HIP(c, hipEventSynchronize(c->t1));
    float v = 0;
    SFA_HIP(c, hipEventElapsedTime(&v, c->t0, c->t1));
    if (ms) *ms = v;
    return SFA_OK;
}

// ---- stage entry points on host planes ------------------------------------------------------------------
struct Staging {
    sfa_ctx *c;
    DevMem mem;
    int w, h, pitch; long pl;
    float *plane(int i) { return mem.f() + (long)i * pl; }
    int init(sfa_ctx *ctx, int w_, int h_, int nplanes) {
        c = ctx; w = w_; h = h_; pitch = dev_pitch(w_); pl = (long)pitch * h_;
        SFA_HIP(c, hipSetDevice(c->device));
        SFA_TRY(mem.alloc(c, (size_t)nplanes * pl * sizeof(float)));
        SFA_HIP(c, hipMemsetAsync(mem.p, 0, (size_t)nplanes * pl * sizeof(float), c->stream));
        return SFA_OK;
    }
    Geo geo() const { return Geo{w, h, pitch, pl, 0, 1, WMask::first(1), nullptr}; }
    int up(int i, const float *host, int stride, int n = 1) {
        for (int k = 0; k < n; k++) SFA_TRY(upload_plane(c, plane(i + k), pitch, host + (size_t)k * stride * h, stride, w, h));
        return SFA_OK;
    }
    int down(float *host, int stride, int i, int n = 1) {
        for (int k = 0; k < n; k++) SFA_TRY(download_plane(c, host + (size_t)k * stride * h, stride, plane(i + k), pitch, w, h));
        return SFA_OK;
    }
};
#define CHECK_ARGS(cond, msg) do { if (!(cond)) return set_error(ctx, SFA_ERR_ARG, "%s: %s", __func__, msg); } while (0)

int sfa_image_warp(sfa_ctx *ctx, float *dst3, float *mask, const float *src3, const float *wx, const float *wy, int w, int h, int stride, int factor) {
    CHECK_ARGS(ctx && dst3 && src3 && wx && wy && w > 0 && h > 0 && stride >= w, "bad arguments");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 9));
    SFA_TRY(s.up(0, src3, stride, 3)); SFA_TRY(s.up(3, wx, stride)); SFA_TRY(s.up(4, wy, stride));
    if (mask) SFA_TRY(s.up(8, mask, stride));
    launch_warp(ctx, s.geo(), s.plane(5), mask ? s.plane(8) : nullptr, s.plane(0), s.plane(3), s.plane(4), factor, 0);
    SFA_TRY(s.down(dst3, stride, 5, 3));
    if (mask) SFA_TRY(s.down(mask, stride, 8));
    return sfa_ctx_sync(ctx);
}

int sfa_derivative_stack(sfa_ctx *ctx, float *out8x3, const float *I1, const float *I2, int w, int h, int stride) {
    CHECK_ARGS(ctx && out8x3 && I1 && I2 && w > 0 && h >= 4 && stride >= w, "bad arguments (h >= 4 needed by the 5-tap vertical filter, image.c:443)");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 30));
    SFA_TRY(s.up(24, I1, stride, 3)); SFA_TRY(s.up(27, I2, stride, 3));
    launch_deriv_stack(ctx, s.geo(), s.plane(0), s.plane(24), s.plane(27), 0, 0);
    SFA_TRY(s.down(out8x3, stride, 0, 24));
    return sfa_ctx_sync(ctx);
}

int sfa_convolve(sfa_ctx *ctx, float *dst, const float *src, int w, int h, int stride, int order, int horizontal) {
    CHECK_ARGS(ctx && dst && src && w > 0 && h > 0 && stride >= w && (order == 1 || order == 2), "bad arguments");
    CHECK_ARGS(horizontal || h >= (order == 2 ? 4 : 2), "image too small for the vertical fast path");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 2));
    SFA_TRY(s.up(0, src, stride));
    launch_convolve(ctx, s.geo(), s.plane(1), s.plane(0), order, horizontal, 1);
    SFA_TRY(s.down(dst, stride, 1));
    return sfa_ctx_sync(ctx);
}

int sfa_dpsis_weight(sfa_ctx *ctx, float *dst, const float *im3, int w, int h, int stride, float coef, const float avg[3], const float std_dev[3], int hbit) {
    CHECK_ARGS(ctx && dst && im3 && avg && std_dev && w > 0 && h >= 4 && stride >= w, "bad arguments");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 4));
    SFA_TRY(s.up(0, im3, stride, 3));
    launch_dpsis(ctx, s.geo(), s.plane(3), s.plane(0), 0, coef, avg, std_dev, hbit);
    SFA_TRY(s.down(dst, stride, 3));
    return sfa_ctx_sync(ctx);
}

int sfa_smoothness(sfa_ctx *ctx, int method, float *dst_horiz, float *dst_vert, const float *uu, const float *vv, const float *dpsis, int w, int h,
                   int stride, float alpha, const sfa_penalty *reg) {
    CHECK_ARGS(ctx && dst_horiz && dst_vert && uu && vv && dpsis && reg && w > 0 && h >= 2 && stride >= w, "bad arguments");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 5));
    SFA_TRY(s.up(0, uu, stride)); SFA_TRY(s.up(1, vv, stride)); SFA_TRY(s.up(2, dpsis, stride));
    launch_smoothness(ctx, s.geo(), method, s.plane(3), s.plane(4), s.plane(0), s.plane(1), s.plane(2), alpha, pen(*reg));
    SFA_TRY(s.down(dst_horiz, stride, 3)); SFA_TRY(s.down(dst_vert, stride, 4));
    return sfa_ctx_sync(ctx);
}

int sfa_sub_laplacian(sfa_ctx *ctx, float *dst, const float *src, const float *wh, const float *wv, int w, int h, int stride) {
    CHECK_ARGS(ctx && dst && src && wh && wv && w > 0 && h > 0 && stride >= w, "bad arguments");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 4));
    SFA_TRY(s.up(0, dst, stride)); SFA_TRY(s.up(1, src, stride)); SFA_TRY(s.up(2, wh, stride)); SFA_TRY(s.up(3, wv, stride));
    launch_sub_laplacian(ctx, s.geo(), s.plane(0), s.plane(1), s.plane(2), s.plane(3));
    SFA_TRY(s.down(dst, stride, 0));
    return sfa_ctx_sync(ctx);
}

int sfa_division_chain(sfa_ctx *ctx, const float *a, const float *b, float *q_chain, float *q_exact, unsigned char *admitted, size_t n) {
    CHECK_ARGS(ctx && a && b && q_chain && q_exact && admitted && n > 0, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    DevMem da, db, dq, de, dm;
    SFA_TRY(da.alloc(ctx, n * 4)); SFA_TRY(db.alloc(ctx, n * 4)); SFA_TRY(dq.alloc(ctx, n * 4)); SFA_TRY(de.alloc(ctx, n * 4)); SFA_TRY(dm.alloc(ctx, n));
    SFA_HIP(ctx, hipMemcpyAsync(da.p, a, n * 4, hipMemcpyHostToDevice, ctx->stream));
    SFA_HIP(ctx, hipMemcpyAsync(db.p, b, n * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_division_chain(ctx, da.f(), db.f(), dq.f(), de.f(), (unsigned char *)dm.p, n);
    SFA_HIP(ctx, hipMemcpyAsync(q_chain, dq.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SFA_HIP(ctx, hipMemcpyAsync(q_exact, de.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SFA_HIP(ctx, hipMemcpyAsync(admitted, dm.p, n, hipMemcpyDeviceToHost, ctx->stream));
    return sfa_ctx_sync(ctx);
}

int sfa_occlusion_costs(sfa_ctx *ctx, const sfa_params *p, float *d0, float *d1, const float *const *masks, const float *const *succ1,
                        const float *const *succ2, const float *const *ref1, const float *const *ref2, int w, int h, int stride) {
    CHECK_ARGS(ctx && p && d0 && d1 && masks && succ1 && succ2 && ref1 && ref2 && w > 0 && h >= 4 && stride >= w, "bad arguments");
    CHECK_ARGS(p->S >= 2 && p->S - 1 <= SFA_MAX_REF, "unsupported slow_flow_S");
    const int ref = p->S - 1, ns = 2 * ref;
    Staging s;
    // planes: 0,1 costs; 2 .. 2+ns masks; then per slot 4 colour images
    SFA_TRY(s.init(ctx, w, h, 2 + ns + ns * 12));
    OccArgs oa;
    memset(&oa, 0, sizeof oa);
    oa.nslots = ns; oa.hd = p->delta / 3.0f; oa.hg = p->gamma / 3.0f; oa.penalty = p->occlusion_penalty;
    oa.color = pen(p->robust_color); oa.grad = pen(p->robust_grad);
    for (int k = 0; k < ns; k++) {
        const int i0 = 2 + ns + k * 12;
        SFA_TRY(s.up(2 + k, masks[k], stride));
        SFA_TRY(s.up(i0, succ1[k], stride, 3)); SFA_TRY(s.up(i0 + 3, succ2[k], stride, 3));
        SFA_TRY(s.up(i0 + 6, ref1[k], stride, 3)); SFA_TRY(s.up(i0 + 9, ref2[k], stride, 3));
        const int idx = std::max(ref - k - 1, k - ref);
        oa.slot[k] = OccSlot{i0 * s.pl, (i0 + 3) * s.pl, (i0 + 6) * s.pl, (i0 + 9) * s.pl, (2 + k) * s.pl, p->rho[idx], p->omega[idx], k >= ref ? 0 : 1};
    }
    launch_occ_costs(ctx, s.geo(), oa, s.plane(0), s.plane(0), s.plane(1), 0);
    SFA_TRY(s.down(d0, stride, 0)); SFA_TRY(s.down(d1, stride, 1));
    return sfa_ctx_sync(ctx);
}

int sfa_grid_cut(sfa_ctx *ctx, float *occ, const float *d0, const float *d1, int w, int h, int stride, float alpha) {
    CHECK_ARGS(ctx && occ && d0 && d1 && w > 0 && h > 0 && stride >= w && alpha >= 0, "bad arguments");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 3 + kCutWorkPlanes));
    SFA_TRY(s.up(1, d0, stride)); SFA_TRY(s.up(2, d1, stride));
    SFA_TRY(run_grid_cut(ctx, s.geo(), s.plane(0), 0, s.plane(1), s.plane(2), s.plane(3), alpha));
    SFA_TRY(s.down(occ, stride, 0));
    return sfa_ctx_sync(ctx);
}

int sfa_add_data_and_match(sfa_ctx *ctx, float *a11, float *a12, float *a22, float *b1, float *b2, const float *mask, const float *du, const float *dv,
                           const float *D8x3, const float *const chw[3], int w, int h, int stride, float delta_over3, float gamma_over3, float sfac,
                           int ref_term, int dt_norm, const sfa_penalty *color, const sfa_penalty *grad) {
    CHECK_ARGS(ctx && a11 && a12 && a22 && b1 && b2 && mask && du && dv && D8x3 && color && grad && w > 0 && h > 0 && stride >= w, "bad arguments");
    if (ref_term && sfac == 0) return set_error(ctx, SFA_ERR_REF_FRAME, "Frame compared to reference frame is the reference frame itself!");
    Staging s;
    // planes: 0-4 system, 5 mask, 6 du, 7 dv, 8..31 stack, 32..34 chw
    SFA_TRY(s.init(ctx, w, h, 35));
    float *sysm[5] = {a11, a12, a22, b1, b2};
    for (int i = 0; i < 5; i++) SFA_TRY(s.up(i, sysm[i], stride));
    SFA_TRY(s.up(5, mask, stride)); SFA_TRY(s.up(6, du, stride)); SFA_TRY(s.up(7, dv, stride));
    SFA_TRY(s.up(8, D8x3, stride, 24));
    AssembleArgs aa;
    memset(&aa, 0, sizeof aa);
    aa.n = 1;
    aa.t[0] = Term{8 * s.pl, 5 * s.pl, delta_over3, gamma_over3, sfac, ref_term};
    aa.dt_norm = dt_norm; aa.color = pen(*color); aa.grad = pen(*grad);
    if (chw) {
        for (int k = 0; k < 3; k++) SFA_TRY(upload_plane(ctx, s.plane(32 + k), s.pitch, chw[k], stride, w, h));
        aa.chw = s.plane(32); aa.chw_pl = s.pl; aa.chw_es = 0; aa.chw_pitch = s.pitch; aa.chw_stride0 = stride; aa.lstride = stride;
    }
    aa.accumulate = 1; aa.do_laplacian = 0;
    launch_assemble(ctx, s.geo(), aa, s.plane(0), s.plane(0), s.plane(1), s.plane(2), s.plane(3), s.plane(4), s.plane(6), s.plane(7), nullptr, nullptr,
                    nullptr, nullptr);
    for (int i = 0; i < 5; i++) SFA_TRY(s.down(sysm[i], stride, i));
    return sfa_ctx_sync(ctx);
}

int sfa_gaussian_blur(sfa_ctx *ctx, float *dst, const float *src, int w, int h, int stride, float sigma) {
    CHECK_ARGS(ctx && dst && src && w > 0 && h > 0 && stride >= w && sigma > 0, "bad arguments");
    float taps[64];
    CHECK_ARGS(sigma * 8 + 1 < 33, "sigma too large");
    const int r = cv_gauss_taps(sigma, taps);
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 3));
    SFA_TRY(s.up(0, src, stride));
    launch_gauss_blur(ctx, s.geo(), s.plane(1), s.plane(2), s.plane(0), 1, taps, r);
    SFA_TRY(s.down(dst, stride, 1));
    return sfa_ctx_sync(ctx);
}

int sfa_resize_linear(sfa_ctx *ctx, float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride) {
    CHECK_ARGS(ctx && dst && src && dw > 0 && dh > 0 && sw > 0 && sh > 0 && dstride >= dw && sstride >= sw, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    DevMem a, b;
    const int sp = dev_pitch(sw), dp = dev_pitch(dw);
    SFA_TRY(a.alloc(ctx, (size_t)sp * sh * 4)); SFA_TRY(b.alloc(ctx, (size_t)dp * dh * 4));
    SFA_TRY(upload_plane(ctx, a.f(), sp, src, sstride, sw, sh));
    launch_resize(ctx, b.f(), dw, dh, dp, (long)dp * dh, 0, a.f(), sw, sh, sp, (long)sp * sh, 0, 1, 1, 1.0f);
    SFA_TRY(download_plane(ctx, dst, dstride, b.f(), dp, dw, dh));
    return sfa_ctx_sync(ctx);
}

int sfa_gaussian_presmooth(sfa_ctx *ctx, float *dst, const float *src, int w, int h, int stride, float sigma) {
    CHECK_ARGS(ctx && dst && src && w > 0 && h > 0 && stride >= w && sigma > 0, "bad arguments");
    const int order = std::max(1, (int)floor(3 * sigma) + 1);
    CHECK_ARGS(order <= 16 && w > 2 * order && h > 2 * order, "image smaller than the filter (image.c:545-574 assumes width > 2*order)");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 3));
    SFA_TRY(s.up(0, src, stride));
    launch_presmooth(ctx, s.geo(), s.plane(1), s.plane(2), s.plane(0), 1, sigma);
    SFA_TRY(s.down(dst, stride, 1));
    return sfa_ctx_sync(ctx);
}

int sfa_resize_linear_fx(sfa_ctx *ctx, float *dst, int dw, int dh, int dstride, const float *src, int sw, int sh, int sstride, double fx, double fy) {
    CHECK_ARGS(ctx && dst && src && dw > 0 && dh > 0 && sw > 0 && sh > 0 && dstride >= dw && sstride >= sw && fx > 0 && fy > 0, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    DevMem a, b;
    const int sp = dev_pitch(sw), dp = dev_pitch(dw);
    SFA_TRY(a.alloc(ctx, (size_t)sp * sh * 4)); SFA_TRY(b.alloc(ctx, (size_t)dp * dh * 4));
    SFA_TRY(upload_plane(ctx, a.f(), sp, src, sstride, sw, sh));
    launch_resize_scaled(ctx, b.f(), dw, dh, dp, (long)dp * dh, 0, a.f(), sw, sh, sp, (long)sp * sh, 0, 1, 1, 1.0f, 1.0 / fx, 1.0 / fy);
    SFA_TRY(download_plane(ctx, dst, dstride, b.f(), dp, dw, dh));
    return sfa_ctx_sync(ctx);
}

// ---- SOR ----------------------------------------------------------------------------------------------------
struct sfa_sor_batch {
    sfa_ctx *ctx = nullptr;
    int w = 0, h = 0, nb = 0, pitch = 0;
    long pl = 0, es = 0;
    DevMem mem;          // nb x 9 planes: du dv a11 a12 a22 b1 b2 sh sv
    SorWorkspace ws;
    float *plane(int b, int i) const { return mem.f() + b * es + (long)i * pl; }
};

int sfa_sor_batch_create(sfa_ctx *ctx, int w, int h, int batch, sfa_sor_batch **out) {
    CHECK_ARGS(ctx && out && w > 0 && h > 0 && batch > 0 && batch <= kMaxBatch, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<sfa_sor_batch> sb(new sfa_sor_batch());
    sb->ctx = ctx; sb->w = w; sb->h = h; sb->nb = batch; sb->pitch = dev_pitch(w);
    sb->pl = (long)sb->pitch * h; sb->es = 9 * sb->pl;
    SFA_TRY(sb->mem.alloc(ctx, (size_t)batch * sb->es * sizeof(float)));
    SFA_HIP(ctx, hipMemsetAsync(sb->mem.p, 0, (size_t)batch * sb->es * sizeof(float), ctx->stream));
    *out = sb.release();
    return SFA_OK;
}
void sfa_sor_batch_destroy(sfa_sor_batch *sb) {
    if (!sb) return;
    (void)hipSetDevice(sb->ctx->device);
    (void)hipStreamSynchronize(sb->ctx->stream);
    delete sb;
}
int sfa_sor_batch_upload(sfa_sor_batch *sb, int b, const float *du, const float *dv, const float *a11, const float *a12, const float *a22, const float *b1,
                         const float *b2, const float *sh, const float *sv, int stride) {
    sfa_ctx *ctx = sb ? sb->ctx : nullptr;
    CHECK_ARGS(sb && b >= 0 && b < sb->nb && du && dv && a11 && a12 && a22 && b1 && b2 && sh && sv && stride >= sb->w, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    const float *src[9] = {du, dv, a11, a12, a22, b1, b2, sh, sv};
    for (int i = 0; i < 9; i++) SFA_TRY(upload_plane(ctx, sb->plane(b, i), sb->pitch, src[i], stride, sb->w, sb->h));
    SFA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFA_OK;
}
int sfa_sor_batch_run(sfa_sor_batch *sb, int iterations, float omega) {
    sfa_ctx *ctx = sb ? sb->ctx : nullptr;
    CHECK_ARGS(sb, "null batch");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    Geo g{sb->w, sb->h, sb->pitch, sb->pl, sb->es, sb->nb, WMask::first(sb->nb), nullptr};
    return sor_run(ctx, sb->ws, g, sb->plane(0, 0), sb->plane(0, 1), sb->plane(0, 2), sb->plane(0, 3), sb->plane(0, 4), sb->plane(0, 5), sb->plane(0, 6),
                   sb->plane(0, 7), sb->plane(0, 8), iterations, omega, true);
}
int sfa_sor_batch_download(sfa_sor_batch *sb, int b, float *du, float *dv, int stride) {
    sfa_ctx *ctx = sb ? sb->ctx : nullptr;
    CHECK_ARGS(sb && b >= 0 && b < sb->nb && du && dv && stride >= sb->w, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    SFA_TRY(download_plane(ctx, du, stride, sb->plane(b, 0), sb->pitch, sb->w, sb->h));
    SFA_TRY(download_plane(ctx, dv, stride, sb->plane(b, 1), sb->pitch, sb->w, sb->h));
    return sfa_ctx_sync(ctx);
}

int sfa_sor_coupled(sfa_ctx *ctx, sfa_image *du, sfa_image *dv, sfa_image *a11, sfa_image *a12, sfa_image *a22, sfa_image *b1, sfa_image *b2,
                    sfa_image *dpsis_horiz, sfa_image *dpsis_vert, int iterations, float omega) {
    CHECK_ARGS(ctx && du && dv && a11 && a12 && a22 && b1 && b2 && dpsis_horiz && dpsis_vert, "null image");
    const int w = du->width, h = du->height, stride = du->stride;
    CHECK_ARGS(w > 0 && h > 0 && stride >= w, "bad image geometry");
    sfa_image *im[9] = {du, dv, a11, a12, a22, b1, b2, dpsis_horiz, dpsis_vert};
    for (auto *i : im) CHECK_ARGS(i->data && i->width == w && i->height == h && i->stride == stride, "images must share one geometry");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 9));
    for (int i = 0; i < 9; i++) SFA_TRY(s.up(i, im[i]->data, stride));
    SorWorkspace ws;
    SFA_TRY(sor_run(ctx, ws, s.geo(), s.plane(0), s.plane(1), s.plane(2), s.plane(3), s.plane(4), s.plane(5), s.plane(6), s.plane(7), s.plane(8), iterations,
                    omega, true));
    SFA_TRY(s.down(du->data, stride, 0)); SFA_TRY(s.down(dv->data, stride, 1));
    if (!(w < 2 || h < 2 || iterations < 1))                                          // the fast path inverts the blocks in place (solver.c:104-106)
        for (int i = 2; i < 5; i++) SFA_TRY(s.down(im[i]->data, stride, i));
    return sfa_ctx_sync(ctx);
}

int sfa_sor_red_black(sfa_ctx *ctx, sfa_image *du, sfa_image *dv, sfa_image *a11, sfa_image *a12, sfa_image *a22, sfa_image *b1, sfa_image *b2,
                      sfa_image *dpsis_horiz, sfa_image *dpsis_vert, int iterations, float omega) {
    CHECK_ARGS(ctx && du && dv && a11 && a12 && a22 && b1 && b2 && dpsis_horiz && dpsis_vert, "null image");
    const int w = du->width, h = du->height, stride = du->stride;
    CHECK_ARGS(w > 0 && h > 0 && stride >= w, "bad image geometry");
    sfa_image *im[9] = {du, dv, a11, a12, a22, b1, b2, dpsis_horiz, dpsis_vert};
    for (auto *i : im) CHECK_ARGS(i->data && i->width == w && i->height == h && i->stride == stride, "images must share one geometry");
    Staging s;
    SFA_TRY(s.init(ctx, w, h, 9));
    for (int i = 0; i < 9; i++) SFA_TRY(s.up(i, im[i]->data, stride));
    SFA_TRY(sor_rb_run(ctx, s.geo(), s.plane(0), s.plane(1), s.plane(2), s.plane(3), s.plane(4), s.plane(5), s.plane(6), s.plane(7), s.plane(8), iterations, omega));
    for (int i = 0; i < 5; i++) SFA_TRY(s.down(im[i]->data, stride, i));
    return sfa_ctx_sync(ctx);
}

void sor_coupled(sfa_image *du, sfa_image *dv, sfa_image *a11, sfa_image *a12, sfa_image *a22, sfa_image *b1, sfa_image *b2, sfa_image *dpsis_horiz,
                 sfa_image *dpsis_vert, const int iterations, const float omega) {
    static std::mutex mu;
    static sfa_ctx *def = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!def && sfa_ctx_create(0, &def) != SFA_OK) {
        fprintf(stderr, "error in sor_coupled(): %s\n", sfa_last_error(nullptr));
        exit(1);
    }
    if (sfa_sor_coupled(def, du, dv, a11, a12, a22, b1, b2, dpsis_horiz, dpsis_vert, iterations, omega) != SFA_OK) {
        fprintf(stderr, "error in sor_coupled(): %s\n", sfa_last_error(def));
        exit(1);
    }
}

// ---- the original two-frame refinement (variational.c:19-143) --------------------------------------------------
void sfa_params_2frame_default(sfa_params_2frame *p) {                                   // variational.c:86-98
    if (!p) return;
    p->alpha = 1.0f; p->gamma = 0.71f; p->delta = 0.0f; p->sigma = 1.00f;
    p->niter_outer = 5; p->niter_inner = 1; p->niter_solver = 30; p->sor_omega = 1.9f;
}

int sfa_variational_2frame(sfa_ctx *ctx, float *wx, float *wy, int w, int h, int stride, const float *im1, const float *im2, const sfa_params_2frame *pp) {
    CHECK_ARGS(ctx && wx && wy && im1 && im2 && w >= 2 && h >= 5 && stride >= w, "bad arguments (h >= 5, w >= 2)");
    sfa_params_2frame p;
    if (pp) p = *pp; else sfa_params_2frame_default(&p);
    const float half_alpha = 0.5f * p.alpha, hg = p.gamma * 0.5f / 3.0f, hd = p.delta * 0.5f / 3.0f;   // :113-115
    enum { WX, WY, UU, VV, DU, DV, SH, SV, A11, A12, A22, B1, B2, MASK, DPS, IM1, IM2 = IM1 + 3, WIM2 = IM2 + 3, STACK = WIM2 + 3, NPL = STACK + 24 };
    Staging s;
    SFA_TRY(s.init(ctx, w, h, NPL));
    SFA_TRY(s.up(WX, wx, stride)); SFA_TRY(s.up(WY, wy, stride));
    SFA_TRY(s.up(IM1, im1, stride, 3)); SFA_TRY(s.up(IM2, im2, stride, 3));
    const Geo g = s.geo();
    const float zero3[3] = {0, 0, 0}, one3[3] = {1, 1, 1};
    launch_dpsis(ctx, g, s.plane(DPS), s.plane(IM1), 0, 5.0f, zero3, one3, 0);                           // :35
    SorWorkspace ws;
    for (int outer = 0; outer < p.niter_outer; outer++) {
        launch_warp(ctx, g, s.plane(WIM2), s.plane(MASK), s.plane(IM2), s.plane(WX), s.plane(WY), 1, 0);   // :41
        launch_deriv_stack(ctx, g, s.plane(STACK), s.plane(WIM2), s.plane(IM1), 0, 0);                   // :43 (mean of both, dt = im2 - im1)
        launch_zero_planes(ctx, g, s.plane(DU), 2);                                                      // :45-46
        launch_copy_planes(ctx, g, s.plane(UU), s.plane(WX), 2, 0, 0);                                   // :48-49
        for (int inner = 0; inner < p.niter_inner; inner++) {
            launch_smoothness_2f(ctx, g, s.plane(SH), s.plane(SV), s.plane(UU), s.plane(VV), s.plane(DPS), half_alpha);   // :54
            launch_data_2f(ctx, g, s.plane(STACK), s.plane(MASK), s.plane(DU), s.plane(DV), s.plane(A11), s.plane(A12), s.plane(A22), s.plane(B1), s.plane(B2),
                           s.plane(WX), s.plane(WY), s.plane(SH), s.plane(SV), hd, hg);                  // :55-57
            SFA_TRY(sor_run(ctx, ws, g, s.plane(DU), s.plane(DV), s.plane(A11), s.plane(A12), s.plane(A22), s.plane(B1), s.plane(B2), s.plane(SH), s.plane(SV),
                            p.niter_solver, p.sor_omega, false));                                       // :59
            // uu = wx + du, vv = wy + dv (:62-67); the change norms of the shared kernel are not used here
            launch_update_inner(ctx, g, s.plane(UU), s.plane(VV), s.plane(WX), s.plane(WY), s.plane(DU), s.plane(DV), s.plane(DU), s.plane(DV), ctx->d_red);
        }
        launch_copy_planes(ctx, g, s.plane(WX), s.plane(UU), 2, 0, 0);                                   // :70-71
    }
    SFA_TRY(check_device_error(ctx));
    SFA_TRY(s.down(wx, stride, WX)); SFA_TRY(s.down(wy, stride, WY));
    return sfa_ctx_sync(ctx);
}

void variational(sfa_image *wx, sfa_image *wy, const sfa_color_image *im1, const sfa_color_image *im2, sfa_params_2frame *params) {
    static std::mutex mu;
    static sfa_ctx *def = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!def && sfa_ctx_create(0, &def) != SFA_OK) {
        fprintf(stderr, "error in variational(): %s\n", sfa_last_error(nullptr));
        exit(1);
    }
    const bool ok = wx && wy && im1 && im2 && wx->data && wy->data && im1->c1 && im2->c1 && wy->width == wx->width && wy->height == wx->height &&
                    wy->stride == wx->stride && im1->width == wx->width && im1->height == wx->height && im1->stride == wx->stride &&
                    im2->width == wx->width && im2->height == wx->height && im2->stride == wx->stride &&
                    im1->c2 == im1->c1 + (size_t)im1->stride * im1->height && im2->c2 == im2->c1 + (size_t)im2->stride * im2->height;
    if (!ok || sfa_variational_2frame(def, wx->data, wy->data, wx->width, wx->height, wx->stride, im1->c1, im2->c1, params) != SFA_OK) {
        fprintf(stderr, "error in variational(): %s\n", ok ? sfa_last_error(def) : "images must share one geometry (color_image_new layout)");
        exit(1);
    }
}

// ---- frames resident in HBM, normalize (variational_mt.cpp:17-85) on them ------------------------------------------------------------
struct sfa_sequence {
    sfa_ctx *ctx = nullptr;
    int w = 0, h = 0, n = 0, pitch = 0;
    long pl = 0;
    DevMem mem;                        // n x 3 planes at the device pitch
    DevMem sums;                       // n x 6 doubles: per frame and channel sum(I), sum(I*I)
    float *frame(int f) const { return mem.f() + (long)f * 3 * pl; }
    Geo geo() const { return Geo{w, h, pitch, pl, 0, 1, WMask::first(1), nullptr}; }
};

int sfa_sequence_create(sfa_ctx *ctx, int w, int h, int n_frames, sfa_sequence **out) {
    CHECK_ARGS(ctx && out && w > 0 && h > 0 && n_frames > 0, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<sfa_sequence> q(new sfa_sequence());
    q->ctx = ctx; q->w = w; q->h = h; q->n = n_frames; q->pitch = dev_pitch(w); q->pl = (long)q->pitch * h;
    SFA_TRY(q->mem.alloc(ctx, (size_t)n_frames * 3 * q->pl * sizeof(float)));
    SFA_HIP(ctx, hipMemsetAsync(q->mem.p, 0, (size_t)n_frames * 3 * q->pl * sizeof(float), ctx->stream));
    SFA_TRY(q->sums.alloc(ctx, (size_t)n_frames * 6 * sizeof(double)));
    *out = q.release();
    return SFA_OK;
}
void sfa_sequence_destroy(sfa_sequence *q) {
    if (!q) return;
    (void)hipSetDevice(q->ctx->device);
    (void)hipStreamSynchronize(q->ctx->stream);
    delete q;
}
int sfa_sequence_upload(sfa_sequence *q, int f, const float *frame3, int stride) {
    sfa_ctx *ctx = q ? q->ctx : nullptr;
    CHECK_ARGS(q && f >= 0 && f < q->n && frame3 && stride >= q->w, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < 3; k++) SFA_TRY(upload_plane(ctx, q->frame(f) + k * q->pl, q->pitch, frame3 + (size_t)k * stride * q->h, stride, q->w, q->h));
    // the copies read pageable host memory of the caller: wait for them, so that the buffer may be freed or reused on return (like sfa_job_upload;
    // the driver calls this from its decode threads, where the wait hides behind the decoding of the next frame)
    SFA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFA_OK;
}
int sfa_sequence_download(sfa_sequence *q, int f, float *frame3, int stride) {
    sfa_ctx *ctx = q ? q->ctx : nullptr;
    CHECK_ARGS(q && f >= 0 && f < q->n && frame3 && stride >= q->w, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < 3; k++) SFA_TRY(download_plane(ctx, frame3 + (size_t)k * stride * q->h, stride, q->frame(f) + k * q->pl, q->pitch, q->w, q->h));
    return sfa_ctx_sync(ctx);
}
// normalize() over frames [f0, f0 + n): statistics over exactly these frames (the reference's `-jet k` mode normalises over that jet's frames only,
// slow_flow.cpp:418-424,673), every frame's sums by the same kernels in the same order whatever the number of frames
// normalize() (variational_mt.cpp:17-85) in its three parts, so that a sequence spread over several GPUs is normalised with ONE set of statistics: (1) the six
// fp64 sums of every frame -- a deterministic kernel: the same bits on whichever GPU holds the frame --, (2) the statistics from the per-frame sums in frame
// order, plain host arithmetic, (3) I <- (I - avg) / std on the resident frames.  sfa_sequence_normalize is the three in a row.
int sfa_sequence_frame_sums(sfa_sequence *q, int f0, int n, double *sums) {
    sfa_ctx *ctx = q ? q->ctx : nullptr;
    CHECK_ARGS(q && sums && f0 >= 0 && n > 0 && f0 + n <= q->n, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    double *dsum = reinterpret_cast<double *>(q->sums.p);
    for (int f = f0; f < f0 + n; f++) launch_normalize_sums(ctx, q->geo(), q->frame(f), dsum + 6 * f);
    SFA_HIP(ctx, hipMemcpyAsync(sums, dsum + 6 * f0, (size_t)6 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SFA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFA_OK;
}
int sfa_normalize_statistics(const double *sums, int n, int w, int h, double avg[3], double std_dev[3]) {
    if (!sums || n <= 0 || w <= 0 || h <= 0 || !avg || !std_dev) return set_error(nullptr, SFA_ERR_ARG, "sfa_normalize_statistics: bad arguments");
    for (int k = 0; k < 3; k++) { avg[k] = 0; std_dev[k] = 0; }
    for (int f = 0; f < n; f++)
        for (int k = 0; k < 3; k++) {
            avg[k] += sums[6 * f + 2 * k] / (h * w);                                     // variational_mt.cpp:41-47
            std_dev[k] += sums[6 * f + 2 * k + 1] / (h * w);
        }
    for (int k = 0; k < 3; k++) {
        avg[k] /= n;
        std_dev[k] = sqrt((std_dev[k] / n) - avg[k] * avg[k]) / 255.0f;                 // :52
    }
    return SFA_OK;
}
int sfa_sequence_apply_normalization(sfa_sequence *q, int f0, int n, const double avg[3], const double std_dev[3]) {
    sfa_ctx *ctx = q ? q->ctx : nullptr;
    CHECK_ARGS(q && avg && std_dev && f0 >= 0 && n > 0 && f0 + n <= q->n, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    for (int f = f0; f < f0 + n; f++) launch_normalize_apply(ctx, q->geo(), q->frame(f), avg, std_dev);
    return sfa_ctx_sync(ctx);
}
int sfa_sequence_normalize(sfa_sequence *q, int f0, int n, double avg[3], double std_dev[3]) {
    sfa_ctx *ctx = q ? q->ctx : nullptr;
    CHECK_ARGS(q && avg && std_dev && f0 >= 0 && n > 0 && f0 + n <= q->n, "bad arguments");
    std::vector<double> hs((size_t)6 * n);
    SFA_TRY(sfa_sequence_frame_sums(q, f0, n, hs.data()));
    SFA_TRY(sfa_normalize_statistics(hs.data(), n, q->w, q->h, avg, std_dev));
    return sfa_sequence_apply_normalization(q, f0, n, avg, std_dev);
}

int sfa_normalize(sfa_ctx *ctx, float *const *frames, int F, int w, int h, int stride, double avg[3], double std_dev[3]) {
    CHECK_ARGS(ctx && frames && F > 0 && w > 0 && h > 0 && stride >= w && avg && std_dev, "bad arguments");
    for (int f = 0; f < F; f++) CHECK_ARGS(frames[f], "null frame");
    sfa_sequence *q = nullptr;
    SFA_TRY(sfa_sequence_create(ctx, w, h, F, &q));
    std::unique_ptr<sfa_sequence, void (*)(sfa_sequence *)> guard(q, sfa_sequence_destroy);
    for (int f = 0; f < F; f++) SFA_TRY(sfa_sequence_upload(q, f, frames[f], stride));
    SFA_TRY(sfa_sequence_normalize(q, 0, F, avg, std_dev));
    for (int f = 0; f < F; f++) SFA_TRY(sfa_sequence_download(q, f, frames[f], stride));
    return SFA_OK;
}

// ---- job ------------------------------------------------------------------------------------------------------
int sfa_job_create(sfa_ctx *ctx, const sfa_params *p, int w, int h, int batch, sfa_job **out) {
    CHECK_ARGS(ctx && p && out && w >= 2 && h >= 5 && batch > 0 && batch <= kMaxBatch, "bad arguments (h >= 5, w >= 2)");
    CHECK_ARGS(p->S >= 2 && p->S - 1 <= SFA_MAX_REF && p->layers >= 1 && p->layers <= 64, "unsupported slow_flow_S / slow_flow_layers");
    CHECK_ARGS(2L * kMaxBatch + 2L * batch * ((w + 63) / 64) * 16 <= kRedDoubles, "batch x width beyond the change norms' scratch (sfa_internal.h: kRedDoubles)");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<sfa_job> j(new sfa_job());
    j->ctx = ctx; j->p = *p; j->w = w; j->h = h; j->nb = batch; j->ref = p->S - 1; j->F = 2 * j->ref + 1;
    j->L = pyramid_sizes(w, h, p->layers, p->p_scale, j->ws, j->hs);
    CHECK_ARGS(j->L >= 1, "image too small for even one pyramid level");
    j->fused = sw_int(Switches::UNFUSED, 0) == 0;
    j->level_off.resize(j->L);
    long off = 0;
    for (int l = 0; l < j->L; l++) {
        j->level_off[l] = off;
        off += Level::persistent_floats(dev_pitch(j->ws[l]), j->hs[l], j->ref);
    }
    j->trans_off = off;
    off += Level::transient_floats(dev_pitch(j->ws[0]), j->hs[0], j->ref, j->fused);     // level 0 is the largest
    j->es = off;
    // the solver workspaces (40 bytes per entry of the diagonal-major planes) of all levels together are 3.5 x the finest one's: from 2 Mpx on, one
    // workspace is re-shaped level by level (a few memsets per level against hundreds of ms of refinement); below that every level keeps its own
    j->share_sor = (double)w * h >= 2.0e6 || sw_given(Switches::SHARE_SOR);
    j->host_stride0 = host_stride(w);
    SFA_TRY(j->arena.alloc(ctx, (size_t)batch * j->es * sizeof(float)));
    SFA_HIP(ctx, hipMemsetAsync(j->arena.p, 0, (size_t)batch * j->es * sizeof(float), ctx->stream));
    SFA_TRY(j->init_flow.alloc(ctx, (size_t)batch * 2 * dev_pitch(w) * h * sizeof(float)));
    SFA_HIP(ctx, hipMemsetAsync(j->init_flow.p, 0, (size_t)batch * 2 * dev_pitch(w) * h * sizeof(float), ctx->stream));
    j->change.assign(2 * batch, 0.f);
    for (int l = 0; l < j->L; l++) j->sor.emplace_back(new SorWorkspace());
    // sum over levels of outer x inner solves (thresholds may end earlier; this is the scheduled amount)
    double px = 0;
    for (int l = 0; l < j->L; l++) px += (double)j->ws[l] * j->hs[l];
    j->mpix_iters = px * p->niter_alter * p->niter_outer * p->niter_inner * p->niter_solver * batch / 1e6;
    *out = j.release();
    return SFA_OK;
}
void sfa_job_destroy(sfa_job *j) {
    if (!j) return;
    (void)hipSetDevice(j->ctx->device);
    (void)hipStreamSynchronize(j->ctx->stream);
    delete j;
}
double sfa_job_mpix_iters(const sfa_job *j) { return j ? j->mpix_iters : 0; }
double sfa_job_device_bytes(const sfa_job *j) {
    if (!j) return 0;
    double b = (double)j->arena.bytes + j->init_flow.bytes + j->chw.bytes + j->cut_scratch.bytes + j->occ_log.bytes;
    for (const auto &w : j->sor) b += (double)w->sa.bytes + w->sb.bytes + w->x.bytes + w->flags.bytes + w->order.bytes + w->edge.bytes;
    return b;
}

static int job_set_channel_weights(sfa_job *j, int b, int stride, const float *const chw[3]) {
    sfa_ctx *ctx = j->ctx;
    if (chw) {
        // weights keep the level-0 host geometry, padding lanes included (the reference indexes them linearly)
        CHECK_ARGS((long)stride * j->h < (1L << 31), "channel weights: the linear pixel index must fit 31 bits");
        if (!j->has_chw) {
            SFA_TRY(j->chw.alloc(ctx, (size_t)j->nb * 3 * dev_pitch(stride) * j->h * sizeof(float)));
            launch_fill(ctx, j->chw.f(), (size_t)j->nb * 3 * dev_pitch(stride) * j->h, 1.0f);
            j->has_chw = true;
            j->chw_stride0 = stride;
        }
        CHECK_ARGS(j->chw_stride0 == stride, "channel weights of all elements must share one stride");
        const int cp = dev_pitch(stride);
        for (int k = 0; k < 3; k++) {
            CHECK_ARGS(chw[k], "null weight plane");
            SFA_TRY(upload_plane(ctx, j->chw.f() + ((long)b * 3 + k) * cp * j->h, cp, chw[k], stride, stride, j->h));
        }
    } else if (j->has_chw) {
        // a slot that held weighted channels before (jobs are reused for batch after batch) goes back to all ones
        launch_fill(ctx, j->chw.f() + (long)b * 3 * dev_pitch(j->chw_stride0) * j->h, (size_t)3 * dev_pitch(j->chw_stride0) * j->h, 1.0f);
    }
    return SFA_OK;
}

int sfa_job_upload(sfa_job *j, int b, const float *const *frames, int n_frames, const float *wx, const float *wy, int stride, const float *const chw[3]) {
    sfa_ctx *ctx = j ? j->ctx : nullptr;
    CHECK_ARGS(j && b >= 0 && b < j->nb && frames && n_frames == j->F && stride >= j->w, "bad arguments (n_frames must be 2*(S-1)+1)");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    Level L0 = j->level(0);
    j->host_stride0 = stride;
    j->presmoothed.clear(b);
    for (int f = 0; f < j->F; f++) {
        CHECK_ARGS(frames[f], "null frame");
        for (int k = 0; k < 3; k++)
            SFA_TRY(upload_plane(ctx, L0.frame(f) + b * j->es + k * L0.pl, L0.pitch, frames[f] + (size_t)k * stride * j->h, stride, j->w, j->h));
    }
    float *f0 = j->init_flow.f() + (long)b * 2 * L0.pl;
    if (wx) SFA_TRY(upload_plane(ctx, f0, L0.pitch, wx, stride, j->w, j->h));
    else SFA_HIP(ctx, hipMemsetAsync(f0, 0, L0.pl * sizeof(float), ctx->stream));
    if (wy) SFA_TRY(upload_plane(ctx, f0 + L0.pl, L0.pitch, wy, stride, j->w, j->h));
    else SFA_HIP(ctx, hipMemsetAsync(f0 + L0.pl, 0, L0.pl * sizeof(float), ctx->stream));
    SFA_TRY(job_set_channel_weights(j, b, stride, chw));
    SFA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFA_OK;
}

// the same as sfa_job_upload with the frames taken from a sequence resident on the job's GPU (device-to-device copies on the job's stream): a frame is
// sent over PCIe once however many windows it is part of (S=3: five windows, both directions) and normalize never brings it back to the host
int sfa_job_upload_resident(sfa_job *j, int b, const sfa_sequence *q, const int *frame_index, int n_frames, const float *wx, const float *wy, int stride,
                            const float *const chw[3]) {
    sfa_ctx *ctx = j ? j->ctx : nullptr;
    CHECK_ARGS(j && q && frame_index && b >= 0 && b < j->nb && n_frames == j->F && stride >= j->w, "bad arguments (n_frames must be 2*(S-1)+1)");
    CHECK_ARGS(q->w == j->w && q->h == j->h && q->ctx->device == ctx->device, "the sequence must have the job's frame size and live on the job's GPU");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    Level L0 = j->level(0);
    j->host_stride0 = stride;
    j->presmoothed.clear(b);
    for (int f = 0; f < j->F; f++) {
        CHECK_ARGS(frame_index[f] >= 0 && frame_index[f] < q->n, "frame index out of range");
        SFA_HIP(ctx, hipMemcpyAsync(L0.frame(f) + b * j->es, q->frame(frame_index[f]), (size_t)3 * L0.pl * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    float *f0 = j->init_flow.f() + (long)b * 2 * L0.pl;
    if (wx) SFA_TRY(upload_plane(ctx, f0, L0.pitch, wx, stride, j->w, j->h));
    else SFA_HIP(ctx, hipMemsetAsync(f0, 0, L0.pl * sizeof(float), ctx->stream));
    if (wy) SFA_TRY(upload_plane(ctx, f0 + L0.pl, L0.pitch, wy, stride, j->w, j->h));
    else SFA_HIP(ctx, hipMemsetAsync(f0 + L0.pl, 0, L0.pl * sizeof(float), ctx->stream));
    SFA_TRY(job_set_channel_weights(j, b, stride, chw));
    if (wx || wy || chw) SFA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host buffers may go away after the call
    return SFA_OK;
}

int sfa_job_reset_flow(sfa_job *j) { return j ? SFA_OK : SFA_ERR_ARG; }   // the initial flow is re-read by every run

int sfa_job_run(sfa_job *j) {
    sfa_ctx *ctx = j ? j->ctx : nullptr;
    CHECK_ARGS(j, "null job");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    const sfa_params &p = j->p;
    const int nb = j->nb, F = j->F, L = j->L;
    const WMask all = WMask::first(nb);
    // ---- pyramid (variational_mt.cpp:583-652) -----------------------------------------------------------
    const float sigma = 1 / sqrtf(2 * p.p_scale);
    float taps[64];
    const int radius = cv_gauss_taps(sigma, taps);
    Level L0 = j->level(0);
    if (p.presmooth_sigma > 0 && (j->presmoothed & all) != all) {                     // :590-597
        // the smoothed frames replace the uploaded ones, once per upload: a job may be run again (warm-up + timed runs, a second pass
        // over the same windows) and must then start from the same images
        float *tmp = L0.tmp();
        for (int f = 0; f < F; f++) {
            launch_presmooth(ctx, L0.geo(all), tmp + 3 * L0.pl, tmp, L0.frame(f), 3, p.presmooth_sigma);
            launch_copy_planes(ctx, L0.geo(all.andnot(j->presmoothed)), L0.frame(f), tmp + 3 * L0.pl, 3, L0.es, L0.es);
        }
        j->presmoothed = all;
    }
    for (int l = 1; l < L; l++) {
        Level Lp = j->level(l - 1), Lc = j->level(l);
        float *tmp = Lp.tmp();
        // all F frames (3 F consecutive planes per window) in one pass: :607 + :611 fused
        if (!sw_given(Switches::PYRAMID_UNFUSED) &&
            launch_pyr_down(ctx, Lc.frame(0), Lc.w, Lc.h, Lc.pitch, Lc.pl, Lc.es, Lp.frame(0), Lp.w, Lp.h, Lp.pitch, Lp.pl, Lp.es, 3 * F, nb, taps, radius))
            continue;
        for (int f = 0; f < F; f++) {
            launch_gauss_blur(ctx, Lp.geo(all), tmp + 3 * Lp.pl, tmp, Lp.frame(f), 3, taps, radius);            // :607
            launch_resize(ctx, Lc.frame(f), Lc.w, Lc.h, Lc.pitch, Lc.pl, Lc.es, tmp + 3 * Lp.pl, Lp.w, Lp.h, Lp.pitch, Lp.pl, Lp.es, 3, nb, 1.0f);   // :611
        }
    }
    // ---- initial flow at the coarsest level (:662-681) ---------------------------------------------------
    Level Lt = j->level(L - 1);
    if (L > 1) {
        const float fx = (1.0f * Lt.w) / j->w, fy = (1.0f * Lt.h) / j->h;
        launch_resize_flow(ctx, Lt.plane(P_WX), Lt.plane(P_WY), Lt.w, Lt.h, Lt.pitch, Lt.es, j->init_flow.f(), j->init_flow.f() + L0.pl, j->w, j->h, L0.pitch, 2 * L0.pl, nb, fx, fy);
    } else {
        launch_copy_planes(ctx, L0.geo(all), L0.plane(P_WX), j->init_flow.f(), 2, L0.es, 2 * L0.pl);
    }
    ChannelWeights cw;
    if (j->has_chw) {
        cw.dev = j->chw.f(); cw.pitch = dev_pitch(j->chw_stride0); cw.pl = (long)cw.pitch * j->h; cw.es = 3 * cw.pl; cw.stride0 = j->chw_stride0;
    }
    // ---- coarse to fine (:684-762) -----------------------------------------------------------------------------
    for (int l = L - 1; l >= 0; l--) {
        Level Lc = j->level(l);
        if (l < L - 1) {
            Level Ln = j->level(l + 1);
            const float fx = (1.0f * Lc.w) / Ln.w, fy = (1.0f * Lc.h) / Ln.h;                                   // :703-704
            launch_resize_flow(ctx, Lc.plane(P_WX), Lc.plane(P_WY), Lc.w, Lc.h, Lc.pitch, Lc.es, Ln.plane(P_WX), Ln.plane(P_WY), Ln.w, Ln.h, Ln.pitch, Ln.es, nb, fx, fy);   // :711,716
        }
        SFA_TRY(run_level(ctx, Lc, p, cw, *j->sor[j->share_sor ? 0 : l], j->cut_scratch, l == 0 ? j->change.data() : nullptr, l == 0 && j->keep_alt_occ ? j->occ_log.f() : nullptr));   // :761
    }
    SFA_HIP(ctx, hipGetLastError());
    return SFA_OK;
}

int sfa_job_download(sfa_job *j, int b, float *wx, float *wy, int stride, float change[2]) {
    sfa_ctx *ctx = j ? j->ctx : nullptr;
    CHECK_ARGS(j && b >= 0 && b < j->nb && wx && wy && stride >= j->w, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    Level L0 = j->level(0);
    SFA_TRY(download_plane(ctx, wx, stride, L0.plane(P_WX) + b * j->es, L0.pitch, j->w, j->h));
    SFA_TRY(download_plane(ctx, wy, stride, L0.plane(P_WY) + b * j->es, L0.pitch, j->w, j->h));
    if (change) { change[0] = j->change[2 * b]; change[1] = j->change[2 * b + 1]; }
    return sfa_ctx_sync(ctx);
}

int sfa_job_keep_alternation_occlusions(sfa_job *j, int on) {
    sfa_ctx *ctx = j ? j->ctx : nullptr;
    CHECK_ARGS(j, "null job");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    j->keep_alt_occ = on != 0;
    if (on) {
        const size_t n = (size_t)j->nb * std::max(1, j->p.niter_alter) * dev_pitch(j->w) * j->h * sizeof(float);
        SFA_TRY(j->occ_log.alloc(ctx, n));
        SFA_HIP(ctx, hipMemsetAsync(j->occ_log.p, 0, n, ctx->stream));
    }
    return SFA_OK;
}

int sfa_job_download_alternation_occlusions(sfa_job *j, int b, int alter, float *occ, int stride) {
    sfa_ctx *ctx = j ? j->ctx : nullptr;
    CHECK_ARGS(j && b >= 0 && b < j->nb && occ && stride >= j->w, "bad arguments");
    CHECK_ARGS(j->keep_alt_occ && j->occ_log.p, "sfa_job_keep_alternation_occlusions was not enabled before the run");
    CHECK_ARGS(alter >= 1 && alter < j->p.niter_alter, "alternation out of range: labels exist for 1 <= alter < niter_alter (variational_mt.cpp:269)");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    const int pitch = dev_pitch(j->w);
    const long pl = (long)pitch * j->h;
    SFA_TRY(download_plane(ctx, occ, stride, j->occ_log.f() + ((long)b * j->p.niter_alter + alter) * pl, pitch, j->w, j->h));
    return sfa_ctx_sync(ctx);
}

int sfa_job_download_occlusions(sfa_job *j, int b, float *occ, int stride) {
    sfa_ctx *ctx = j ? j->ctx : nullptr;
    CHECK_ARGS(j && b >= 0 && b < j->nb && occ && stride >= j->w, "bad arguments");
    SFA_HIP(ctx, hipSetDevice(ctx->device));
    Level L0 = j->level(0);
    SFA_TRY(download_plane(ctx, occ, stride, L0.plane(P_OCC) + b * j->es, L0.pitch, j->w, j->h));
    return sfa_ctx_sync(ctx);
}

// ---- host-plane convenience entry points ------------------------------------------------------------------------
static int variational_host(sfa_ctx *ctx, const sfa_params *p, float *wx, float *wy, int w, int h, int stride, const float *const *frames, int n_frames,
                            const float *const chw[3], float *occlusions_out, float change[2]) {
    sfa_job *j = nullptr;
    SFA_TRY(sfa_job_create(ctx, p, w, h, 1, &j));
    std::unique_ptr<sfa_job, void (*)(sfa_job *)> guard(j, sfa_job_destroy);
    SFA_TRY(sfa_job_upload(j, 0, frames, n_frames, wx, wy, stride, chw));
    SFA_TRY(sfa_job_run(j));
    SFA_TRY(sfa_job_download(j, 0, wx, wy, stride, change));
    if (occlusions_out) {
        Level L0 = j->level(0);
        SFA_TRY(download_plane(ctx, occlusions_out, stride, L0.plane(P_OCC), L0.pitch, w, h));
        SFA_TRY(sfa_ctx_sync(ctx));
    }
    return SFA_OK;
}

int sfa_variational(sfa_ctx *ctx, const sfa_params *p, float *wx, float *wy, int w, int h, int stride, const float *const *frames, int n_frames,
                    const float *const chw[3], float *occlusions_out, float change[2]) {
    CHECK_ARGS(ctx && p && wx && wy && frames, "null argument");
    return variational_host(ctx, p, wx, wy, w, h, stride, frames, n_frames, chw, occlusions_out, change);
}

int sfa_compute_one_level(sfa_ctx *ctx, const sfa_params *p, float *wx, float *wy, int w, int h, int stride, const float *const *frames, int n_frames,
                          const float *const chw[3], float *occlusions_out, float change[2]) {
    CHECK_ARGS(ctx && p && wx && wy && frames, "null argument");
    sfa_params q = *p;
    q.layers = 1;
    q.presmooth_sigma = 0;
    return variational_host(ctx, &q, wx, wy, w, h, stride, frames, n_frames, chw, occlusions_out, change);
}

}  // extern "C"
