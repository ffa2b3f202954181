// occlusion.hip -- the discrete step between alternations: Variational_AUX_MT::optimizeOcc
// (variational_aux_mt.cpp:758-887), gfx950.
//
//   1. k_occ_costs: the two data costs per pixel.  psi (not psi') of the colour / gradient constancy residuals of
//      every slot against its successor and against the reference frame, past slots -> label "occluded in the
//      future" (1), future slots -> "occluded in the past" (0), each normalised by its mask-weighted slot weights,
//      x dt_scale_graphc, + penalty for label 1.  Iz, Ixz, Iyz are formed in an LDS tile from the image pairs (the
//      derivative stacks are never stored); same operations as the stack rows, bit for bit.
//   2. The reference hands these costs to GCO's alpha-expansion with a Potts term alpha on the 4-connected grid.  With
//      two labels one expansion is an exact s-t minimum cut, which is what runs here: a synchronous (deterministic)
//      push-relabel on the grid, all frame windows of a batch in the same launches, with breadth-first global
//      relabelling.  The scarcer terminal is made the active one (with the default penalty almost every pixel
//      prefers label 0, so pushing from the few label-1 pixels converges in a handful of rounds).
//      GCO v3.0 is not vendored: parity unpinned; the oracle is an exact fp64 Dinic cut and the tests compare energies.
#include "sfa_device.h"

#include <utility>

#pragma clang fp contract(off)

namespace sfa {

// ---------------------------------------------------------------------------------------------------
// psi(x^2), v4sf overloads (modified_l1_norm.h:24-26, quadratic_function.h:18-20, lorentzian.h:24-32,
// trunc_modified_l1_norm.h:27-36, geman_mcclure.h:24-26)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float psi_apply_vec(const PenaltyDev &p, float xsq) {
    const float e2 = p.eps * p.eps;
    switch (p.id) {
    case 0: return xsq;
    case 2: return (float)log(1 + 0.5 * (double)xsq / (double)e2);                    // epsilon_sq is a double member there
    case 3: {
        float out = sqrt_rn(xsq + e2);
        if (sqrt_rn(xsq) > p.trunc) out = sqrt_rn(p.trunc + e2);                      // sic: truncation, not its square
        return out;
    }
    case 4: return __fdiv_rn(xsq, (xsq + 1.0f) * (xsq + 1.0f));
    default: return sqrt_rn(xsq + e2);
    }
}

#define DT_SCALE_GRAPHC 0.01f   // variational_aux_mt.h:24

constexpr int OC_X = 64, OC_Y = 8, OC_W = OC_X + 4, OC_R = OC_Y + 4, OC_NT = 512;   // tile, halo 2
struct OccTileAcc {
    const float *t; int x0, y0;      // global coordinates of the tile's LDS origin
    __device__ __forceinline__ float operator()(int x, int y) const { return t[(y - y0) * OC_W + (x - x0)]; }
};

// one block = 64x8 pixels of one window, one pixel per thread
__global__ void __launch_bounds__(OC_NT) k_occ_costs(OccArgs a, const float *__restrict__ base, float *__restrict__ d0, float *__restrict__ d1, long des, Geo g) {
    __shared__ float sZ[3][OC_R * OC_W];
    const int b = blockIdx.z;
    if (!elem_active(g.active, b)) return;
    const long eb = b * g.es;
    const int x0 = blockIdx.x * OC_X - 2, y0 = blockIdx.y * OC_Y - 2;
    const int tx = threadIdx.x & 63;
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = x0 + 2 + tx, y = y0 + 2 + ty;
    const bool ok = x < g.w && y < g.h;
    const bool y_in = y >= 2 && y + 2 < g.h;
    const int c = (ty + 2) * OC_W + (tx + 2);
    float e[2] = {0.0f, 0.0f}, n[2] = {0.0f, 0.0f};
    for (int s = 0; s < a.nslots; s++) {
        const OccSlot &S = a.slot[s];
        const float m = ok ? base[eb + S.mask_off + (size_t)y * g.pitch + x] : 0.0f;
        float term = 0.0f;
        for (int kind = 0; kind < 2; kind++) {                   // successive pair (:817-819), reference-frame pair (:822-827)
            const float wgt = kind ? S.omega : S.rho;
            // a zero weight contributes rho*hd*m*psi = +0 to a non-negative sum: skipped (the first product must still start the sum)
            if (wgt == 0.0f && kind == 1) continue;
            const float *pa = base + eb + (kind ? S.r1_off : S.s1_off), *pb = base + eb + (kind ? S.r2_off : S.s2_off);
            __syncthreads();
            for (int i = threadIdx.x; i < 3 * OC_R * OC_W; i += OC_NT) {
                const int ch = i / (OC_R * OC_W), r = (i / OC_W) % OC_R, q = i % OC_W;
                const int gy = y0 + r, gx = clampi(x0 + q, 0, g.w - 1);              // replicated columns: fixed-offset taps (image.c:501-516)
                if (gy < 0 || gy >= g.h) continue;
                const size_t o = ch * g.pl + (size_t)gy * g.pitch + gx;
                sZ[ch][r * OC_W + q] = pa[o] - pb[o];                                 // Iz, variational_mt.cpp:122 / :141,144
            }
            __syncthreads();
            if (!ok) continue;
            float iz[3], ixz[3], iyz[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float *Z = sZ[ch];
                iz[ch] = Z[c];
                ixz[ch] = tap5(Z[c - 2], Z[c - 1], Z[c], Z[c + 1], Z[c + 2]);        // :132
                if (y_in) iyz[ch] = tap5(Z[c - 2 * OC_W], Z[c - OC_W], Z[c], Z[c + OC_W], Z[c + 2 * OC_W]);   // :133
                else      iyz[ch] = d5y(OccTileAcc{Z, x0, y0}, x, y, g.h);
            }
            const float pc = psi_apply_vec(a.color, iz[0] * iz[0] + iz[1] * iz[1] + iz[2] * iz[2]);
            const float pg = psi_apply_vec(a.grad, ixz[0] * ixz[0] + ixz[1] * ixz[1] + ixz[2] * ixz[2] + iyz[0] * iyz[0] + iyz[1] * iyz[1] + iyz[2] * iyz[2]);
            if (kind == 0) term = wgt * a.hd * m * pc;                                // :817
            else           term += wgt * a.hd * m * pc;                               // :822
            term += wgt * a.hg * m * pg;                                              // :818, :823
        }
        const int l = S.label;                                                        // :829-837
        e[l] += term;
        n[l] += m * (S.rho + S.rho + S.omega + S.omega);
    }
    if (!ok) return;
    const size_t o = b * des + (size_t)y * g.pitch + x;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (n[l] == 0) n[l] = 1;                                                      // :846-849
        const float cst = __fdiv_rn(DT_SCALE_GRAPHC * e[l], n[l]) + a.penalty * l;     // :851
        (l ? d1 : d0)[o] = cst;
    }
}
void launch_occ_costs(sfa_ctx *c, const Geo &g, const OccArgs &a, const float *base, float *d0, float *d1, long des) {
    hipLaunchKernelGGL(k_occ_costs, dim3((g.w + OC_X - 1) / OC_X, (g.h + OC_Y - 1) / OC_Y, g.nb), dim3(OC_NT), 0, c->stream, a, base, d0, d1, des, g);
}

// ---------------------------------------------------------------------------------------------------
// two-label cut: synchronous push-relabel on the grid
//   node p: excess e(p) (arc from the active terminal, saturated up front), tc(p) = residual arc to the other terminal,
//   c[d](p) = residual arc to neighbour d (0:+x 1:-x 2:+y 3:-y), height hgt(p).  Orientation per window: flip = 0
//   pushes from the pixels that prefer label 0 (D1 > D0) towards label-1 pixels, flip = 1 the other way round.
// ---------------------------------------------------------------------------------------------------
constexpr int kCutInf = 1 << 30;
struct CutPlanes { float *e, *tc, *c[4], *f[4]; int *hgt, *hgt_next; };   // collect reads hgt, writes hgt_next (deterministic rounds)
__device__ __forceinline__ long cut_nb(int d, int x, int y, int w, int h, int pitch) {   // offset of neighbour d or 0 if outside
    switch (d) {
    case 0: return x + 1 < w ? 1 : 0;
    case 1: return x > 0 ? -1 : 0;
    case 2: return y + 1 < h ? pitch : 0;
    default: return y > 0 ? -pitch : 0;
    }
}

// counts[b] = {#(D1 > D0), #(D1 < D0)}
__global__ void k_cut_count(const float *__restrict__ d0, const float *__restrict__ d1, unsigned *__restrict__ counts, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    int pos = 0, neg = 0;
    if (x < g.w && y < g.h) {
        const size_t o = b * g.pl + (size_t)y * g.pitch + x;
        const float u = d1[o] - d0[o];
        pos = u > 0; neg = u < 0;
    }
    const unsigned long long mp = __ballot(pos), mn = __ballot(neg);
    if ((threadIdx.x & 63) == 0) {
        if (mp) atomicAdd(&counts[2 * b], (unsigned)__popcll(mp));
        if (mn) atomicAdd(&counts[2 * b + 1], (unsigned)__popcll(mn));
    }
}
__global__ void k_cut_init(CutPlanes P, const float *__restrict__ d0, const float *__restrict__ d1, const unsigned *__restrict__ counts, float alpha, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = b * g.pl + (size_t)y * g.pitch + x;
    const bool flip = counts[2 * b + 1] < counts[2 * b];            // fewer label-1 pixels: push from them
    float u = d1[o] - d0[o];
    if (flip) u = -u;
    P.e[o] = u > 0 ? u : 0.0f;
    P.tc[o] = u < 0 ? -u : 0.0f;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        P.c[d][o] = cut_nb(d, x, y, g.w, g.h, g.pitch) ? alpha : 0.0f;
        P.f[d][o] = 0.0f;
    }
    P.hgt[o] = 0;
}
// global relabel: exact distance to the passive terminal in the residual graph (Bellman-Ford sweeps, monotone)
__global__ void k_cut_bfs_init(CutPlanes P, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = b * g.pl + (size_t)y * g.pitch + x;
    P.hgt[o] = P.tc[o] > 0 ? 1 : kCutInf;
}
__global__ void k_cut_bfs_sweep(CutPlanes P, unsigned *__restrict__ changed, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    bool ch = false;
    if (x < g.w && y < g.h) {
        const size_t o = b * g.pl + (size_t)y * g.pitch + x;
        int hb = P.hgt[o];
        if (hb > 1) {
            int best = hb;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const long off = cut_nb(d, x, y, g.w, g.h, g.pitch);
                if (off && P.c[d][o] > 0) {
                    const int hn = P.hgt[o + off];
                    if (hn < kCutInf && hn + 1 < best) best = hn + 1;
                }
            }
            if (best < hb) { P.hgt[o] = best; ch = true; }       // neighbours read this sweep's or the last sweep's value: both valid lower bounds of a monotone relaxation
        }
    }
    if (__ballot(ch) && (threadIdx.x & 63) == 0) atomicOr(changed, 1u);
}
// push: decisions from the heights of the previous phase only; flows go to per-direction buffers (no atomics)
__global__ void k_cut_push(CutPlanes P, Geo g, int hmax) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = b * g.pl + (size_t)y * g.pitch + x;
    float e = P.e[o];
    const int hp = P.hgt[o];
    float f[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (e > 0 && hp < hmax) {
        float tc = P.tc[o];
        if (tc > 0) {
            const float dlt = fminf(e, tc);
            e -= dlt; tc -= dlt;
            P.tc[o] = tc;
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const long off = cut_nb(d, x, y, g.w, g.h, g.pitch);
            if (!off || !(e > 0)) continue;
            const float cd = P.c[d][o];
            if (cd > 0 && P.hgt[o + off] == hp - 1) {
                const float dlt = fminf(e, cd);
                e -= dlt;
                P.c[d][o] = cd - dlt;
                f[d] = dlt;
            }
        }
        P.e[o] = e;
    }
#pragma unroll
    for (int d = 0; d < 4; d++) P.f[d][o] = f[d];
}
// collect the flows the neighbours sent, then relabel an active node that has no admissible arc left
__global__ void k_cut_collect(CutPlanes P, unsigned *__restrict__ active, Geo g, int hmax) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    bool act = false;
    if (x < g.w && y < g.h) {
        const size_t o = b * g.pl + (size_t)y * g.pitch + x;
        float e = P.e[o];
        float c[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            c[d] = P.c[d][o];
            const long off = cut_nb(d, x, y, g.w, g.h, g.pitch);
            if (off) {
                const float in = P.f[d ^ 1][o + off];             // what neighbour d pushed towards this node
                if (in > 0) { e += in; c[d] += in; P.c[d][o] = c[d]; }
            }
        }
        P.e[o] = e;
        const int hp = P.hgt[o];
        if (e > 0 && hp < hmax) {
            int best = kCutInf;
            bool admissible = P.tc[o] > 0;
            if (admissible) best = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const long off = cut_nb(d, x, y, g.w, g.h, g.pitch);
                if (off && c[d] > 0) {
                    const int hn = P.hgt[o + off];
                    if (hn == hp - 1) admissible = true;
                    if (hn < best) best = hn;
                }
            }
            int hnew = hp;
            if (!admissible) hnew = best >= hmax ? hmax : best + 1;
            P.hgt_next[o] = hnew;
            act = hnew < hmax;
        } else
            P.hgt_next[o] = hp;
    }
    if (__ballot(act) && (threadIdx.x & 63) == 0) atomicOr(active, 1u);
}
// after the final breadth-first pass: a node that still reaches the passive terminal lies on its side
__global__ void k_cut_labels(float *__restrict__ occ, long occ_es, CutPlanes P, const unsigned *__restrict__ counts, Geo g) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    const size_t o = b * g.pl + (size_t)y * g.pitch + x;
    const bool flip = counts[2 * b + 1] < counts[2 * b];
    const bool passive_side = P.hgt[o] < kCutInf;
    // flip = 0: active terminal = label 0, passive = label 1;  flip = 1: the reverse
    const int l = flip ? (passive_side ? 0 : 1) : (passive_side ? 1 : 0);
    occ[b * occ_es + (size_t)y * g.pitch + x] = (float)(2 * l - 1);                  // variational_aux_mt.cpp:876
}

// Runs the cut for every window of the batch.  d0, d1: cost planes [nb][pl]; work: kCutWorkPlanes planes [nb][pl] of scratch;
// occ: output planes of element 0 (+occ_es per window).  Bounded: gives up (SFA_ERR_TIMEOUT) after max_rounds.
int run_grid_cut(sfa_ctx *c, const Geo &g, float *occ, long occ_es, const float *d0, const float *d1, float *work, float alpha) {
    CutPlanes P;
    const size_t n = (size_t)g.nb * g.pl;
    P.e = work; P.tc = work + n;
    for (int d = 0; d < 4; d++) { P.c[d] = work + (2 + d) * n; P.f[d] = work + (6 + d) * n; }
    P.hgt = reinterpret_cast<int *>(work + 10 * n);
    P.hgt_next = reinterpret_cast<int *>(work + 11 * n);
    unsigned *flags = reinterpret_cast<unsigned *>(c->d_red);     // [0]: changed / active, [2 ..]: per-window counts
    unsigned *counts = flags + 2;
    unsigned *h_flag = reinterpret_cast<unsigned *>(c->h_red);
    Geo gc = g;
    gc.es = g.pl;                                                 // the cut planes are packed [nb][pl]
    const dim3 grid = grid2d(gc), block = block2d();
    const int hmax = g.w * g.h;
    SFA_HIP(c, hipMemsetAsync(flags, 0, (2 + 2 * g.nb) * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_cut_count, grid, block, 0, c->stream, d0, d1, counts, gc);
    hipLaunchKernelGGL(k_cut_init, grid, block, 0, c->stream, P, d0, d1, counts, alpha, gc);
    auto global_relabel = [&]() -> int {
        hipLaunchKernelGGL(k_cut_bfs_init, grid, block, 0, c->stream, P, gc);
        for (long guard = 0; guard < (long)g.w * g.h + 8; guard += 8) {
            SFA_HIP(c, hipMemsetAsync(flags, 0, sizeof(unsigned), c->stream));
            for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k_cut_bfs_sweep, grid, block, 0, c->stream, P, flags, gc);
            SFA_HIP(c, hipMemcpyAsync(h_flag, flags, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            SFA_HIP(c, hipStreamSynchronize(c->stream));
            if (!*h_flag) return SFA_OK;
        }
        return set_error(c, SFA_ERR_TIMEOUT, "grid cut: breadth-first relabelling did not settle");   // unreachable: distances are < w*h
    };
    SFA_TRY(global_relabel());
    const int max_rounds = 64 * (g.w + g.h);
    bool done = false;
    for (int round = 0; round < max_rounds && !done; round += 8) {
        SFA_HIP(c, hipMemsetAsync(flags, 0, sizeof(unsigned), c->stream));
        for (int i = 0; i < 8; i++) {
            hipLaunchKernelGGL(k_cut_push, grid, block, 0, c->stream, P, gc, hmax);
            if (i == 7) SFA_HIP(c, hipMemsetAsync(flags, 0, sizeof(unsigned), c->stream));   // only the last collect's verdict counts
            hipLaunchKernelGGL(k_cut_collect, grid, block, 0, c->stream, P, flags, gc, hmax);
            std::swap(P.hgt, P.hgt_next);
        }
        SFA_HIP(c, hipMemcpyAsync(h_flag, flags, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        SFA_HIP(c, hipStreamSynchronize(c->stream));
        done = !*h_flag;
        if (!done && (round / 8) % 4 == 3) SFA_TRY(global_relabel());   // strands excess that can no longer reach the terminal
    }
    if (!done) return set_error(c, SFA_ERR_TIMEOUT, "grid cut: push-relabel did not settle in %d rounds", max_rounds);
    SFA_TRY(global_relabel());
    hipLaunchKernelGGL(k_cut_labels, grid, block, 0, c->stream, occ, occ_es, P, counts, gc);
    SFA_HIP(c, hipGetLastError());
    return SFA_OK;
}

}  // namespace sfa
