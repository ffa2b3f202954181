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
#include <vector>

#include <cstdlib>
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
    if (!elem_active(g, b)) return;
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
            // the halo-2 tile starts two columns left of a 64-aligned one: the aligned quads x0-2 .. x0+69 cover it, one 16-byte load per image and item
            // (the scalar form issued ten loads per thread and stage); columns outside the image are replicated (fixed-offset taps, image.c:501-516)
            constexpr int OC_Q = (OC_W + 2 + 3) / 4;
            for (int i = threadIdx.x; i < 3 * OC_R * OC_Q; i += OC_NT) {
                const int ch = i / (OC_R * OC_Q), r = (i / OC_Q) % OC_R, qa = i % OC_Q;
                const int gy = y0 + r, gx = x0 - 2 + 4 * qa;
                if (gy < 0 || gy >= g.h) continue;
                const size_t ro = ch * g.pl + (size_t)gy * g.pitch;
                float z[4];
                if (gx >= 0 && gx + 3 < g.w) {
                    const float4 va = *reinterpret_cast<const float4 *>(pa + ro + gx), vb = *reinterpret_cast<const float4 *>(pb + ro + gx);
                    z[0] = va.x - vb.x; z[1] = va.y - vb.y; z[2] = va.z - vb.z; z[3] = va.w - vb.w;      // Iz, variational_mt.cpp:122 / :141,144
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) { const int cx = clampi(gx + e, 0, g.w - 1); z[e] = pa[ro + cx] - pb[ro + cx]; }
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int q = 4 * qa - 2 + e;
                    if (q >= 0 && q < OC_W) sZ[ch][r * OC_W + q] = z[e];
                }
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
// raise a grid-wide flag: read first, thousands of blocks raising the same word would serialise in the L2 atomic unit
__device__ __forceinline__ void cut_raise(unsigned *flag) {
    if (!__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(flag, 1u);
}
// Per-tile (= per-block footprint) bookkeeping, so that a round costs what the active region costs and not the whole image:
//   idle[t]: executed rounds in a row that left tile t unchanged (0 = has active nodes, saturates at 2).  A tile with idle >= 2 has
//   zero flow buffers and identical height buffers, so its push and collect are no-ops unless a 4-neighbour tile is active
//   (idle == 0: it may push flow across the border) -- those blocks return after reading the flags.  The arithmetic of the
//   blocks that run is unchanged, so the rounds are the same rounds.
//   dirty[t]: tile t changed in the previous breadth-first sweep; a tile is relaxed again only if it or a neighbour is dirty.
struct CutTiles { int *idle, *idle_next, *dirty, *dirty_next; };
__device__ __forceinline__ int cut_tile(int b) { return (b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; }
template <class Pred>
__device__ __forceinline__ bool cut_tile_nb_any(const int *flag, int t, Pred pred) {      // any 4-neighbour tile of the same window satisfying pred
    return (blockIdx.x > 0 && pred(flag[t - 1])) || (blockIdx.x + 1 < gridDim.x && pred(flag[t + 1])) ||
           (blockIdx.y > 0 && pred(flag[t - (int)gridDim.x])) || (blockIdx.y + 1 < gridDim.y && pred(flag[t + (int)gridDim.x]));
}
__device__ __forceinline__ long cut_nb(int d, int x, int y, int w, int h, int pitch) {   // offset of neighbour d or 0 if outside
    switch (d) {
    case 0: return x + 1 < w ? 1 : 0;
    case 1: return x > 0 ? -1 : 0;
    case 2: return y + 1 < h ? pitch : 0;
    default: return y > 0 ? -pitch : 0;
    }
}

// counts[b] = {#(D1 > D0), #(D1 < D0)}.  gridDim.y row-strided blocks per window, one atomic pair per block
constexpr int kCountRows = 8;
__global__ void __launch_bounds__(BX *BY) k_cut_count(const float *__restrict__ d0, const float *__restrict__ d1, unsigned *__restrict__ counts, Geo g) {
    __shared__ unsigned s_cnt[2];
    const int b = blockIdx.z;
    const int x = blockIdx.x * BX + threadIdx.x;
    if (threadIdx.x == 0 && threadIdx.y == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    unsigned pos = 0, neg = 0;
    if (x < g.w)
        for (int y = blockIdx.y * BY + threadIdx.y; y < g.h; y += gridDim.y * BY) {
            const size_t o = b * g.pl + (size_t)y * g.pitch + x;
            const float u = d1[o] - d0[o];
            pos += u > 0; neg += u < 0;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { pos += __shfl_down(pos, off); neg += __shfl_down(neg, off); }
    if ((threadIdx.x & 63) == 0) { if (pos) atomicAdd(&s_cnt[0], pos); if (neg) atomicAdd(&s_cnt[1], neg); }
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        if (s_cnt[0]) atomicAdd(&counts[2 * b], s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&counts[2 * b + 1], s_cnt[1]);
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
// the same start on the sweep's own tiling (64 x kBfsRows), which also learns which tiles have anything to relax: a tile whose nodes all sit next to the
// terminal (distance 1, the minimum) never changes -- `fixed`, never run -- every other tile starts dirty
constexpr int kBfsRowsDecl = 16;
__global__ void __launch_bounds__(BX *BY) k_cut_bfs_init_tiles(CutPlanes P, int *__restrict__ dirty, int *__restrict__ fixed, Geo g) {
    const int b = blockIdx.z, t = cut_tile(b);
    const int x = blockIdx.x * BX + threadIdx.x;
    bool far = false;
#pragma unroll
    for (int k = 0; k < kBfsRowsDecl / BY; k++) {
        const int y = blockIdx.y * kBfsRowsDecl + threadIdx.y + BY * k;
        if (x < g.w && y < g.h) {
            const size_t o = b * g.pl + (size_t)y * g.pitch + x;
            const bool near = P.tc[o] > 0;
            P.hgt[o] = near ? 1 : kCutInf;
            far |= !near;
        }
    }
    const int any = __syncthreads_or(far);
    if (threadIdx.x == 0 && threadIdx.y == 0) { dirty[t] = any; fixed[t] = !any; }
}
// One launch relaxes every tile that is dirty or has a dirty 4-neighbour tile: the tile (64 x kBfsRows pixels, 4 rows per thread) and
// a one-pixel halo of heights go to LDS, the residual-arc masks to registers, and the block iterates to the tile's fixpoint with
// the halo frozen (any value read there is a valid upper bound of a monotone relaxation, so the order of the tiles does not
// matter; the global fixpoint -- no tile changes -- is the exact distance).  Distances travel a tile per launch, not a pixel.
constexpr int kBfsRows = kBfsRowsDecl, kBfsPer = kBfsRows / BY, kBfsIters = 96;
__global__ void __launch_bounds__(BX *BY) k_cut_bfs_sweep(CutPlanes P, CutTiles T, const int *__restrict__ fixed, unsigned *__restrict__ changed, Geo g) {
    __shared__ int sh[kBfsRows + 2][BX + 2];
    const int b = blockIdx.z;
    const int t = cut_tile(b);
    const bool first = threadIdx.x == 0 && threadIdx.y == 0;
    // nothing it reads has changed since it last ran -- or nothing in it can change (k_cut_bfs_init_tiles: every node next to the terminal)
    if ((fixed && fixed[t]) || (!T.dirty[t] && !cut_tile_nb_any(T.dirty, t, [](int v) { return v != 0; }))) {
        if (first) T.dirty_next[t] = 0;
        return;
    }
    const int x0 = blockIdx.x * BX, y0 = blockIdx.y * kBfsRows, tid = threadIdx.y * BX + threadIdx.x;
    const size_t wb = (size_t)b * g.pl;
    for (int i = tid; i < (kBfsRows + 2) * (BX + 2); i += BX * BY) {                          // tile + halo (outside the image: unreachable)
        const int ly = i / (BX + 2), lx = i % (BX + 2), x = x0 + lx - 1, y = y0 + ly - 1;
        sh[ly][lx] = (x >= 0 && x < g.w && y >= 0 && y < g.h) ? P.hgt[wb + (size_t)y * g.pitch + x] : kCutInf;
    }
    const int x = x0 + threadIdx.x;
    unsigned mask = 0;                                                                        // 4 bits per pixel: residual arc to neighbour d
    int h0[kBfsPer];
#pragma unroll
    for (int k = 0; k < kBfsPer; k++) {
        const int y = y0 + threadIdx.y + BY * k;
        if (x < g.w && y < g.h) {
            const size_t o = wb + (size_t)y * g.pitch + x;
#pragma unroll
            for (int d = 0; d < 4; d++)
                if (cut_nb(d, x, y, g.w, g.h, g.pitch) && P.c[d][o] > 0) mask |= 1u << (4 * k + d);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBfsPer; k++) h0[k] = sh[threadIdx.y + BY * k + 1][threadIdx.x + 1];
    for (int it = 0; it < kBfsIters; it++) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < kBfsPer; k++) {
            const int ly = threadIdx.y + BY * k + 1, lx = threadIdx.x + 1;
            const int hb = sh[ly][lx];
            if (hb > 1 && ((mask >> (4 * k)) & 15u)) {
                int best = hb;
                const unsigned m = mask >> (4 * k);
                if (m & 1u) best = min(best, sh[ly][lx + 1] + 1);                              // kCutInf + 1 does not overflow and never wins
                if (m & 2u) best = min(best, sh[ly][lx - 1] + 1);
                if (m & 4u) best = min(best, sh[ly + 1][lx] + 1);
                if (m & 8u) best = min(best, sh[ly - 1][lx] + 1);
                if (best < hb) { sh[ly][lx] = best; ch = true; }
            }
        }
        if (!__syncthreads_or(ch)) break;
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < kBfsPer; k++) {
        const int y = y0 + threadIdx.y + BY * k;
        const int hn = sh[threadIdx.y + BY * k + 1][threadIdx.x + 1];
        if (x < g.w && y < g.h && hn < h0[k]) { P.hgt[wb + (size_t)y * g.pitch + x] = hn; any = true; }
    }
    const int tile_changed = __syncthreads_or(any);
    if (first) { T.dirty_next[t] = tile_changed; if (tile_changed) cut_raise(changed); }
}
// push: decisions from the heights of the previous phase only; flows go to per-direction buffers (no atomics)
__device__ __forceinline__ void cut_push_node(const CutPlanes &P, const int *__restrict__ hgt, size_t o, int x, int y, const Geo &g, int hmax) {
    float e = P.e[o];
    const int hp = hgt[o];
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
            if (cd > 0 && hgt[o + off] == hp - 1) {
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
// collect the flows the neighbours sent, then relabel an active node that has no admissible arc left; true if the node stays active
__device__ __forceinline__ bool cut_collect_node(const CutPlanes &P, const int *__restrict__ hgt, int *__restrict__ hgt_next, size_t o, int x, int y,
                                                 const Geo &g, int hmax) {
    float e = P.e[o];
    float c[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        c[d] = P.c[d][o];
        const long off = cut_nb(d, x, y, g.w, g.h, g.pitch);
        if (off) {
            const float in = P.f[d ^ 1][o + off];                 // what neighbour d pushed towards this node
            if (in > 0) { e += in; c[d] += in; P.c[d][o] = c[d]; }
        }
    }
    P.e[o] = e;
    const int hp = hgt[o];
    if (e > 0 && hp < hmax) {
        int best = kCutInf;
        bool admissible = P.tc[o] > 0;
        if (admissible) best = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const long off = cut_nb(d, x, y, g.w, g.h, g.pitch);
            if (off && c[d] > 0) {
                const int hn = hgt[o + off];
                if (hn == hp - 1) admissible = true;
                if (hn < best) best = hn;
            }
        }
        int hnew = hp;
        if (!admissible) hnew = best >= hmax ? hmax : best + 1;
        hgt_next[o] = hnew;
        return hnew < hmax;
    }
    hgt_next[o] = hp;
    return false;
}
__global__ void __launch_bounds__(BX *BY) k_cut_push(CutPlanes P, CutTiles T, Geo g, int hmax) {
    const int b = blockIdx.z;
    if (T.idle[cut_tile(b)] >= 2) return;                        // no excess to push and the flow buffers are already zero
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= g.w || y >= g.h) return;
    cut_push_node(P, P.hgt, b * g.pl + (size_t)y * g.pitch + x, x, y, g, hmax);
}
__global__ void __launch_bounds__(BX *BY) k_cut_collect(CutPlanes P, CutTiles T, unsigned *__restrict__ active, Geo g, int hmax) {
    __shared__ int s_act;
    const int b = blockIdx.z;
    const int t = cut_tile(b);
    const bool first = threadIdx.x == 0 && threadIdx.y == 0;
    const int self = T.idle[t];
    if (self >= 2 && !cut_tile_nb_any(T.idle, t, [](int v) { return v == 0; })) {          // settled, and no neighbour tile pushed this round
        if (first) T.idle_next[t] = 2;
        return;
    }
    if (first) s_act = 0;
    __syncthreads();
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    bool act = false;
    if (x < g.w && y < g.h) act = cut_collect_node(P, P.hgt, P.hgt_next, b * g.pl + (size_t)y * g.pitch + x, x, y, g, hmax);
    if (__ballot(act) && (threadIdx.x & 63) == 0) s_act = 1;
    __syncthreads();
    if (first) { T.idle_next[t] = s_act ? 0 : min(self + 1, 2); if (s_act) cut_raise(active); }
}
// after a global relabelling: both height buffers carry the new heights, and the tile bookkeeping is rebuilt from the nodes --
// a tile is active if it holds an active node, cooling (one more round, to clear its flow buffers) if it took part before, else settled
__global__ void __launch_bounds__(BX *BY) k_cut_mark(CutPlanes P, CutTiles T, unsigned *__restrict__ n_active, unsigned cap, Geo g, int hmax) {
    __shared__ int s_act;
    const int b = blockIdx.z;
    const int t = cut_tile(b);
    const bool first = threadIdx.x == 0 && threadIdx.y == 0;
    if (first) s_act = 0;
    __syncthreads();
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    bool act = false;
    if (x < g.w && y < g.h) {
        const size_t o = b * g.pl + (size_t)y * g.pitch + x;
        const int hp = P.hgt[o];
        P.hgt_next[o] = hp;
        act = P.e[o] > 0 && hp < hmax;
    }
    if (__ballot(act) && (threadIdx.x & 63) == 0) s_act = 1;
    __syncthreads();
    if (first) {
        T.idle[t] = s_act ? 0 : (T.idle[t] < 2 ? 1 : 2);
        // active tiles, counted up to `cap` (beyond it only "many" matters, and thousands of atomics on one word would cost more than the pass)
        if (s_act && __hip_atomic_load(n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= cap) atomicAdd(n_active, 1u);
    }
}
// The sparse phase.  Once the global relabelling has stranded what cannot reach the terminal, a handful of tiles per window stay
// active for dozens of rounds, and a round of grid launches costs its launch overhead.  Here ONE workgroup per window runs the same
// rounds (same node functions, same tile bookkeeping, __syncthreads as the round barrier -- windows are independent problems) on
// the listed tiles only, `rounds` (even: the height buffers end where the host expects them) at most, stopping early when the
// window has no active tile left.  flags[0] |= 1 if a window still has active tiles at the end.
constexpr int kTailThreads = 1024;
__global__ void __launch_bounds__(kTailThreads) k_cut_tail(CutPlanes P, CutTiles T, unsigned *__restrict__ flags, Geo g, int hmax, int tiles_x, int tiles_y, int rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    const int NT = tiles_x * tiles_y, b = blockIdx.x, tid = threadIdx.x;
    unsigned char *idle = tail_smem, *idle_next = tail_smem + NT;                             // [NT] each
    int *list = reinterpret_cast<int *>(tail_smem + ((2 * NT + 15) & ~15));                  // [NT]: tile | push-flag << 30
    __shared__ int s_cnt, s_active;
    int *gidle = T.idle + (size_t)b * NT;
    for (int t = tid; t < NT; t += kTailThreads) idle[t] = (unsigned char)min(gidle[t], 2);
    const int *hgt = P.hgt;
    int *hgt_next = P.hgt_next;
    bool still_active = true;
    for (int r = 0; r < rounds; r++) {
        if (tid == 0) { s_cnt = 0; s_active = 0; }
        __syncthreads();
        for (int t = tid; t < NT; t += kTailThreads) {                                          // who takes part in this round
            const int self = idle[t], bx = t % tiles_x, by = t / tiles_x;
            const bool nb0 = (bx > 0 && idle[t - 1] == 0) || (bx + 1 < tiles_x && idle[t + 1] == 0) || (by > 0 && idle[t - tiles_x] == 0) ||
                             (by + 1 < tiles_y && idle[t + tiles_x] == 0);
            if (self == 0) s_active = 1;
            if (self < 2 || nb0) {
                list[atomicAdd(&s_cnt, 1)] = t | ((self < 2) << 30);
                idle_next[t] = (unsigned char)min(self + 1, 2);                                  // unless a node stays active (below)
            } else
                idle_next[t] = 2;
        }
        __syncthreads();
        still_active = s_active != 0;
        if (!still_active && (r & 1) == 0) break;                                             // settled, and an even number of rounds done
        const int cnt = s_cnt;
        for (int i = tid / (BX * BY); i < cnt; i += kTailThreads / (BX * BY)) {                 // push
            const int e = list[i];
            if (!(e >> 30)) continue;
            const int t = e & 0x3fffffff, x = (t % tiles_x) * BX + (tid % BX), y = (t / tiles_x) * BY + (tid / BX) % BY;
            if (x < g.w && y < g.h) cut_push_node(P, hgt, b * g.pl + (size_t)y * g.pitch + x, x, y, g, hmax);
        }
        __syncthreads();
        for (int i = tid / (BX * BY); i < cnt; i += kTailThreads / (BX * BY)) {                 // collect + relabel
            const int t = list[i] & 0x3fffffff, x = (t % tiles_x) * BX + (tid % BX), y = (t / tiles_x) * BY + (tid / BX) % BY;
            bool act = false;
            if (x < g.w && y < g.h) act = cut_collect_node(P, hgt, hgt_next, b * g.pl + (size_t)y * g.pitch + x, x, y, g, hmax);
            if (__ballot(act) && (tid & 63) == 0) idle_next[t] = 0;
        }
        __syncthreads();
        { const int *h = hgt; hgt = hgt_next; hgt_next = const_cast<int *>(h); }
        { unsigned char *q = idle; idle = idle_next; idle_next = q; }
    }
    __syncthreads();
    bool any = false;
    for (int t = tid; t < NT; t += kTailThreads) { gidle[t] = idle[t]; any |= idle[t] == 0; }
    if (__ballot(any) && (tid & 63) == 0) cut_raise(&flags[0]);
}
// ---------------------------------------------------------------------------------------------------
// Tile discharge (round 3): the same synchronous push / collect-and-relabel rounds, but run INSIDE a 64 x 16 tile for as many rounds as the tile
// stays active: the node state (excess, terminal arc, four neighbour arcs) sits in registers, heights and the flows of the round in LDS, the one-pixel
// ring of heights around the tile is read once and stays frozen.  Flow that leaves the tile is summed per border node and direction and handed to
// the neighbour tile at the end (its excess and its reverse arc).  Tiles run in four colours (2 x 2 pattern), one colour per launch: no two tiles of a
// launch touch (not even at a corner), so the ring a tile reads is what its neighbours last wrote, a border node is written by one block only, and
// every step is an ordinary push or relabel of a sequential execution -- labels stay valid (h(p) <= h(q) + 1 on residual arcs) and the result is a
// maximum preflow like before; the labelling (k_cut_labels: who still reaches the passive terminal) does not depend on which maximum flow was found.
// One launch moves excess across a whole tile where a grid round moved it one pixel.
// act[t] (64 x 16 tiles, the breadth-first sweep's tiling): the tile holds an active node, or has been handed flow since it last ran.
// ---------------------------------------------------------------------------------------------------
constexpr int kDisRows = kBfsRows;
// PER rows per thread: the block is 64 x (kDisRows / PER) threads.  One row per thread (16 waves) halves a round's latency twice over against four rows
// per thread, which is what the last batches of a cut consist of (a handful of tiles, each running its rounds alone on a CU)
template <int PER>
__global__ void __launch_bounds__(BX * (kDisRows / PER)) k_cut_discharge(CutPlanes P, int *__restrict__ act, unsigned *__restrict__ flags, Geo g, int hmax, int colour,
                                                          int tiles_x, int tiles_y, int max_rounds) {
    constexpr int kDisPer = PER, TYN = kDisRows / PER;             // rows per thread, threads along y
    __shared__ int sh[2][kDisRows + 2][BX + 2];
    __shared__ float sf[4][kDisRows][BX];
    const int b = blockIdx.z;
    const int bx = 2 * (int)blockIdx.x + (colour & 1), by = 2 * (int)blockIdx.y + (colour >> 1);
    if (bx >= tiles_x || by >= tiles_y) return;
    const int t = (b * tiles_y + by) * tiles_x + bx;
    if (!act[t]) return;
    const bool first = threadIdx.x == 0 && threadIdx.y == 0;
    const int x0 = bx * BX, y0 = by * kDisRows, tid = threadIdx.y * BX + threadIdx.x;
    const size_t wb = (size_t)b * g.pl;
    for (int i = tid; i < (kDisRows + 2) * (BX + 2); i += BX * TYN) {                          // tile + ring (outside the image: unreachable, and no arc leads there)
        const int ly = i / (BX + 2), lx = i % (BX + 2), x = x0 + lx - 1, y = y0 + ly - 1;
        const int v = (x >= 0 && x < g.w && y >= 0 && y < g.h) ? P.hgt[wb + (size_t)y * g.pitch + x] : kCutInf;
        sh[0][ly][lx] = v; sh[1][ly][lx] = v;
    }
    const int x = x0 + threadIdx.x, lx = threadIdx.x + 1;
    float e[kDisPer], tc[kDisPer], c[kDisPer][4], out[kDisPer][4];
    bool in[kDisPer];
#pragma unroll
    for (int k = 0; k < kDisPer; k++) {
        const int y = y0 + threadIdx.y + TYN * k;
        in[k] = x < g.w && y < g.h;
        e[k] = 0.0f; tc[k] = 0.0f;
#pragma unroll
        for (int d = 0; d < 4; d++) { c[k][d] = 0.0f; out[k][d] = 0.0f; }
        if (in[k]) {
            const size_t o = wb + (size_t)y * g.pitch + x;
            e[k] = P.e[o]; tc[k] = P.tc[o];
#pragma unroll
            for (int d = 0; d < 4; d++) c[k][d] = P.c[d][o];                                    // 0 towards the outside of the image (k_cut_init)
        }
    }
    __syncthreads();
    int cur = 0;
    bool still = true;
    for (int r = 0; r < max_rounds && still; r++) {
        // push: decisions from the heights of the previous round only
#pragma unroll
        for (int k = 0; k < kDisPer; k++) {
            const int ly = threadIdx.y + TYN * k + 1;
            float f[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int hp = sh[cur][ly][lx];
            if (e[k] > 0 && hp < hmax) {
                if (tc[k] > 0) { const float dlt = fminf(e[k], tc[k]); e[k] -= dlt; tc[k] -= dlt; }
                const int hn[4] = {sh[cur][ly][lx + 1], sh[cur][ly][lx - 1], sh[cur][ly + 1][lx], sh[cur][ly - 1][lx]};
#pragma unroll
                for (int d = 0; d < 4; d++)
                    if (e[k] > 0 && c[k][d] > 0 && hn[d] == hp - 1) {
                        const float dlt = fminf(e[k], c[k][d]);
                        e[k] -= dlt; c[k][d] -= dlt; f[d] = dlt;
                    }
            }
#pragma unroll
            for (int d = 0; d < 4; d++) sf[d][ly - 1][threadIdx.x] = f[d];
            // what leaves the tile waits in registers
            if (threadIdx.x == BX - 1) out[k][0] += f[0];
            if (threadIdx.x == 0) out[k][1] += f[1];
            if (ly == kDisRows) out[k][2] += f[2];
            if (ly == 1) out[k][3] += f[3];
        }
        __syncthreads();
        // collect what the neighbours inside the tile sent, then relabel an active node that has no admissible arc left
        bool a = false;
#pragma unroll
        for (int k = 0; k < kDisPer; k++) {
            const int ly = threadIdx.y + TYN * k + 1, row = ly - 1;
            float inc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (threadIdx.x + 1 < BX) inc[0] = sf[1][row][threadIdx.x + 1];
            if (threadIdx.x > 0) inc[1] = sf[0][row][threadIdx.x - 1];
            if (row + 1 < kDisRows) inc[2] = sf[3][row + 1][threadIdx.x];
            if (row > 0) inc[3] = sf[2][row - 1][threadIdx.x];
#pragma unroll
            for (int d = 0; d < 4; d++)
                if (inc[d] > 0) { e[k] += inc[d]; c[k][d] += inc[d]; }
            const int hp = sh[cur][ly][lx];
            int hnew = hp;
            if (e[k] > 0 && hp < hmax) {
                int best = kCutInf;
                bool admissible = tc[k] > 0;
                if (admissible) best = 0;
                const int hn[4] = {sh[cur][ly][lx + 1], sh[cur][ly][lx - 1], sh[cur][ly + 1][lx], sh[cur][ly - 1][lx]};
#pragma unroll
                for (int d = 0; d < 4; d++)
                    if (c[k][d] > 0) {
                        if (hn[d] == hp - 1) admissible = true;
                        if (hn[d] < best) best = hn[d];
                    }
                if (!admissible) hnew = best >= hmax ? hmax : best + 1;
                a |= hnew < hmax;
            }
            sh[cur ^ 1][ly][lx] = hnew;
        }
        still = __syncthreads_or(a);
        cur ^= 1;
    }
    // the tile goes home; what left it goes to the neighbours (none of them runs in this launch)
    bool sent[4] = {false, false, false, false};
#pragma unroll
    for (int k = 0; k < kDisPer; k++) {
        if (!in[k]) continue;
        const int y = y0 + threadIdx.y + TYN * k, ly = threadIdx.y + TYN * k + 1;
        const size_t o = wb + (size_t)y * g.pitch + x;
        P.e[o] = e[k]; P.tc[o] = tc[k];
#pragma unroll
        for (int d = 0; d < 4; d++) P.c[d][o] = c[k][d];
        P.hgt[o] = sh[cur][ly][lx];
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (out[k][d] > 0) {
                const size_t q = d == 0 ? o + 1 : d == 1 ? o - 1 : d == 2 ? o + g.pitch : o - g.pitch;
                P.e[q] += out[k][d];
                P.c[d ^ 1][q] += out[k][d];
                sent[d] = true;
            }
    }
    bool any_sent = false;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const int s = __syncthreads_or(sent[d]);
        if (first && s) act[d == 0 ? t + 1 : d == 1 ? t - 1 : d == 2 ? t + tiles_x : t - tiles_x] = 1;
        any_sent |= s != 0;
    }
    if (first) {
        act[t] = still ? 1 : 0;
        if (still || any_sent) cut_raise(&flags[0]);
    }
}
// after a global relabelling: which tiles hold an active node (flags[1]: their number)
constexpr int kDisPer = kDisRows / BY;
__global__ void __launch_bounds__(BX *BY) k_cut_mark_tiles(CutPlanes P, int *__restrict__ act, unsigned *__restrict__ n_active, unsigned cap, Geo g, int hmax, int tiles_x, int tiles_y) {
    const int b = blockIdx.z, t = (b * tiles_y + blockIdx.y) * tiles_x + blockIdx.x;
    const int x = blockIdx.x * BX + threadIdx.x;
    bool a = false;
#pragma unroll
    for (int k = 0; k < kDisPer; k++) {
        const int y = blockIdx.y * kDisRows + threadIdx.y + BY * k;
        if (x < g.w && y < g.h) {
            const size_t o = b * g.pl + (size_t)y * g.pitch + x;
            a |= P.e[o] > 0 && P.hgt[o] < hmax;
        }
    }
    const int any = __syncthreads_or(a);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        act[t] = any;
        // counted up to `cap`: beyond it only "many" matters, and thousands of atomics on one word would cost more than the pass
        if (any && __hip_atomic_load(n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= cap) atomicAdd(n_active, 1u);
    }
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

// Runs the cut for every window of the batch.  d0, d1: cost planes [nb][pl]; work: kCutWorkPlanes planes [nb][pl] of scratch (the last one: tile flags);
// occ: output planes of element 0 (+occ_es per window).  Bounded: gives up (SFA_ERR_TIMEOUT) after max_rounds.
int run_grid_cut(sfa_ctx *c, const Geo &g, float *occ, long occ_es, const float *d0, const float *d1, float *work, float alpha) {
    CutPlanes P;
    const size_t n = (size_t)g.nb * g.pl;
    P.e = work; P.tc = work + n;
    for (int d = 0; d < 4; d++) { P.c[d] = work + (2 + d) * n; P.f[d] = work + (6 + d) * n; }
    P.hgt = reinterpret_cast<int *>(work + 10 * n);
    P.hgt_next = reinterpret_cast<int *>(work + 11 * n);
    unsigned *flags = reinterpret_cast<unsigned *>(c->d_red);     // [0]: changed / active, [1]: active tiles, [2 .. 5]: changed, per sweep of a batch, [8 ..]: per-window counts
    unsigned *counts = flags + 8;
    unsigned *h_flag = reinterpret_cast<unsigned *>(c->h_red);
    Geo gc = g;
    gc.es = g.pl;                                                 // the cut planes are packed [nb][pl]
    const dim3 grid = grid2d(gc), block = block2d();
    const int hmax = g.w * g.h;
    const size_t ntiles = (size_t)grid.x * grid.y * grid.z;       // far fewer than the n floats of work plane 12 (one tile = BX * BY pixels)
    CutTiles T;
    T.idle = reinterpret_cast<int *>(work + 12 * n);
    T.idle_next = T.idle + ntiles;
    T.dirty = T.idle + 2 * ntiles;
    T.dirty_next = T.idle + 3 * ntiles;
    SFA_HIP(c, hipMemsetAsync(flags, 0, (8 + 2 * g.nb) * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_cut_count, dim3(grid.x, std::min((int)grid.y, kCountRows), grid.z), block, 0, c->stream, d0, d1, counts, gc);
    hipLaunchKernelGGL(k_cut_init, grid, block, 0, c->stream, P, d0, d1, counts, alpha, gc);
    SFA_HIP(c, hipMemsetAsync(T.idle, 0, ntiles * sizeof(int), c->stream));                  // no tile is settled yet (k_cut_mark refines this)
    const dim3 bfs_grid(grid.x, (g.h + kBfsRows - 1) / kBfsRows, grid.z);          // its own, taller tiles; the dirty flags are indexed by this grid
    constexpr unsigned kTailMaxTiles = 48;                        // active tiles per window up to which one workgroup per window is the faster way
    unsigned n_active = ~0u;                                      // tiles with active nodes (counted up to that bound), as of the last global relabelling
    auto global_relabel = [&]() -> int {
        hipLaunchKernelGGL(k_cut_bfs_init, grid, block, 0, c->stream, P, gc);
        SFA_HIP(c, hipMemsetAsync(T.dirty, 1, ntiles * sizeof(int), c->stream));            // every tile starts dirty (any non-zero value)
        constexpr int kSweeps = 4;                                 // per convergence check (even: the dirty buffers end where they started)
        for (long guard = 0; guard < (long)g.w * g.h + 8; guard += kSweeps) {
            SFA_HIP(c, hipMemsetAsync(flags, 0, 2 * sizeof(unsigned), c->stream));
            for (int i = 0; i < kSweeps; i++) {
                hipLaunchKernelGGL(k_cut_bfs_sweep, bfs_grid, block, 0, c->stream, P, T, (const int *)nullptr, flags, gc);
                std::swap(T.dirty, T.dirty_next);
            }
            // new heights everywhere: resynchronise the height buffers and the tile states, and count the active tiles -- valid if this
            // batch of sweeps changed nothing (then the heights were final before it); otherwise repeated after the next batch
            hipLaunchKernelGGL(k_cut_mark, grid, block, 0, c->stream, P, T, flags + 1, kTailMaxTiles * (unsigned)g.nb, gc, hmax);
            SFA_HIP(c, hipMemcpyAsync(h_flag, flags, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            SFA_HIP(c, hipStreamSynchronize(c->stream));
            if (!h_flag[0]) { n_active = h_flag[1]; return SFA_OK; }
        }
        return set_error(c, SFA_ERR_TIMEOUT, "grid cut: breadth-first relabelling did not settle");   // unreachable: distances are < w*h
    };
    // ---- tile discharge (the default; SFA_CUT_DISCHARGE=0: the grid rounds below, kept as the cross-check) ----
    const bool discharge = sw_int(Switches::CUT_DISCHARGE, 1) != 0;
    if (discharge) {
        const int tiles_x = (int)bfs_grid.x, tiles_y = (int)bfs_grid.y;
        int *act = T.idle, *fixed = T.idle_next;                  // the grid rounds' bookkeeping is not in use: [nb][tiles_y][tiles_x] flags each
        const unsigned few_tiles = std::max(8u, (unsigned)(tiles_x * tiles_y * g.nb) / 64);     // active tiles up to which a batch is a tail batch
        const dim3 dgrid((tiles_x + 1) / 2, (tiles_y + 1) / 2, g.nb);
        const int kInner = sw_int(Switches::CUT_INNER, 16);      // rounds a tile runs per local relabelling, at most
        const int kSuper = sw_int(Switches::CUT_SUPER, 2);       // visits of every colour between two relabellings
        const int kTailInner = sw_int(Switches::CUT_TAIL_INNER, 32);   // the same once few tiles are active
        const int kPer = sw_int(Switches::CUT_PER, 2), kTailPer = sw_int(Switches::CUT_TAIL_PER, 1);   // rows per thread
        const int kTailSuper = sw_int(Switches::CUT_TAIL_SUPER, 4);
        // exact distances, then which tiles hold active nodes; flags[1]: any at all
        auto relabel_and_mark = [&]() -> int {
            hipLaunchKernelGGL(k_cut_bfs_init_tiles, bfs_grid, block, 0, c->stream, P, T.dirty, fixed, gc);
            // a sweep that changes nothing leaves the exact distances (so do all after it): one flag per sweep of a batch, the last one decides
            constexpr int kSweeps = 4;
            for (long guard = 0; guard < (long)g.w * g.h + 8; guard += kSweeps) {
                SFA_HIP(c, hipMemsetAsync(flags, 0, 6 * sizeof(unsigned), c->stream));
                for (int i = 0; i < kSweeps; i++) {
                    hipLaunchKernelGGL(k_cut_bfs_sweep, bfs_grid, block, 0, c->stream, P, T, fixed, flags + 2 + i, gc);
                    std::swap(T.dirty, T.dirty_next);
                }
                hipLaunchKernelGGL(k_cut_mark_tiles, bfs_grid, block, 0, c->stream, P, act, flags + 1, few_tiles, gc, hmax, tiles_x, tiles_y);
                SFA_HIP(c, hipMemcpyAsync(h_flag, flags, 6 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
                SFA_HIP(c, hipStreamSynchronize(c->stream));
                if (!h_flag[2 + kSweeps - 1]) {
                    if (sw_given(Switches::CUT_DEBUG)) {
                        std::vector<int> ha((size_t)tiles_x * tiles_y * g.nb);
                        (void)hipMemcpy(ha.data(), act, ha.size() * sizeof(int), hipMemcpyDeviceToHost);
                        long na = 0;
                        for (int v : ha) na += v != 0;
                        fprintf(stderr, "cut %dx%d: relabel settled after %ld sweeps, %ld of %zu tiles active\n", g.w, g.h, guard + kSweeps, na, ha.size());
                    }
                    return SFA_OK;
                }
            }
            return set_error(c, SFA_ERR_TIMEOUT, "grid cut: breadth-first relabelling did not settle");
        };
        SFA_TRY(relabel_and_mark());
        const int max_batches = 64 * (g.w + g.h);
        int batch = 0;
        // a maximum preflow is reached when, with exact distances, no node with excess reaches the passive terminal any more
        while (h_flag[1]) {
            if (++batch > max_batches) return set_error(c, SFA_ERR_TIMEOUT, "grid cut: tile discharge did not settle in %d batches", max_batches);
            // between two relabellings (which strand the excess that can no longer reach the terminal -- left to itself it would climb to hmax one
            // height per round) every colour is visited kSuper times
            // few tiles left: a visit costs next to nothing and the relabelling as much as ever -- more of the former per latter
            const bool tail = h_flag[1] <= few_tiles;
            const int supers = tail ? kTailSuper : kSuper, rounds = tail ? kTailInner : kInner;
            const int per = tail ? kTailPer : kPer;
            for (int s = 0; s < supers; s++)
                for (int colour = 0; colour < 4; colour++)
                    if (per == 1)      hipLaunchKernelGGL(k_cut_discharge<1>, dgrid, dim3(BX, kDisRows), 0, c->stream, P, act, flags, gc, hmax, colour, tiles_x, tiles_y, rounds);
                    else if (per == 2) hipLaunchKernelGGL(k_cut_discharge<2>, dgrid, dim3(BX, kDisRows / 2), 0, c->stream, P, act, flags, gc, hmax, colour, tiles_x, tiles_y, rounds);
                    else               hipLaunchKernelGGL(k_cut_discharge<4>, dgrid, dim3(BX, kDisRows / 4), 0, c->stream, P, act, flags, gc, hmax, colour, tiles_x, tiles_y, rounds);
            SFA_TRY(relabel_and_mark());
        }
        hipLaunchKernelGGL(k_cut_labels, grid, block, 0, c->stream, occ, occ_es, P, counts, gc);
        SFA_HIP(c, hipGetLastError());
        return SFA_OK;
    }
    SFA_TRY(global_relabel());
    const int max_rounds = 64 * (g.w + g.h);
    // Rounds come in batches between two global relabellings (which strand the excess that can no longer reach the terminal):
    // grid launches (kGridRounds rounds of push + collect over all tiles) while many tiles are active, the one-workgroup-per-window
    // kernel (kTailRounds rounds at most) once few are.  flags[0]: some node still active.
    constexpr int kGridRounds = 8, kTailRounds = 32;           // both even: the height / tile-state buffers end where they started
    const int tiles_w = (int)(grid.x * grid.y);
    const size_t tail_lds = (((size_t)2 * tiles_w + 15) & ~(size_t)15) + (size_t)4 * tiles_w;
    // worth it only when a grid launch is mostly block scheduling (many windows); a single window's grid is a few thousand blocks.
    // SFA_CUT_NO_TAIL / SFA_CUT_TAIL force one way (cross-check in the tests)
    const bool tail_fits = tail_lds <= 150 * 1024 && !sw_given(Switches::CUT_NO_TAIL) && (ntiles >= 6000 || sw_given(Switches::CUT_TAIL));
    bool done = false;
    int batches = 0;
    for (int round = 0; round < max_rounds && !done;) {
        SFA_HIP(c, hipMemsetAsync(flags, 0, 2 * sizeof(unsigned), c->stream));
        if (tail_fits && n_active <= kTailMaxTiles * (unsigned)g.nb) {
            hipLaunchKernelGGL(k_cut_tail, dim3(g.nb), dim3(kTailThreads), tail_lds, c->stream, P, T, flags, gc, hmax, (int)grid.x, (int)grid.y, kTailRounds);
            round += kTailRounds;
        } else {
            for (int i = 0; i < kGridRounds; i++) {
                hipLaunchKernelGGL(k_cut_push, grid, block, 0, c->stream, P, T, gc, hmax);
                if (i == kGridRounds - 1) SFA_HIP(c, hipMemsetAsync(flags, 0, 2 * sizeof(unsigned), c->stream));   // only the last collect's verdict counts
                hipLaunchKernelGGL(k_cut_collect, grid, block, 0, c->stream, P, T, flags, gc, hmax);
                std::swap(P.hgt, P.hgt_next);
                std::swap(T.idle, T.idle_next);
            }
            round += kGridRounds;
        }
        SFA_HIP(c, hipMemcpyAsync(h_flag, flags, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        SFA_HIP(c, hipStreamSynchronize(c->stream));
        done = !h_flag[0];
        // relabel after every batch while few tiles are active (it strands what only climbs) and after the first batch; with many
        // active tiles the flow is still moving and a relabelling per 8 rounds would cost more than the rounds: every 4th batch
        constexpr int dense_every = 4;
        batches++;
        const bool dense = n_active > kTailMaxTiles * (unsigned)g.nb;
        if (!done && (!dense || batches == 1 || batches % dense_every == 0)) SFA_TRY(global_relabel());
    }
    if (!done) return set_error(c, SFA_ERR_TIMEOUT, "grid cut: push-relabel did not settle in %d rounds", max_rounds);
    SFA_TRY(global_relabel());
    hipLaunchKernelGGL(k_cut_labels, grid, block, 0, c->stream, occ, occ_es, P, counts, gc);
    SFA_HIP(c, hipGetLastError());
    return SFA_OK;
}

}  // namespace sfa
