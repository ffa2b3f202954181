// epic.cpp -- see epic.h.  Written from the algorithm of epic_flow_extended/epic.cpp + epic_aux.cpp (cited per function); where the order of
// floating-point operations or of tie-breaking decides the result (distance sweeps, graph search, kernel sums) it follows the reference so that the
// Nadaraya-Watson route is reproduced bit for bit; the affine fits are solved by an own least-squares routine (the reference calls LAPACK's sgels).
#include "epic.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>

void epic_params_default(epic_params_t *p) {                                // epic.cpp:127-136
    strcpy(p->method, "LA");
    p->saliency_th = 0.045f; p->pref_nn = 25; p->pref_th = 5.0f; p->nn = 100; p->coef_kernel = 0.8f; p->euc = 0.001f; p->verbose = 0;
}

bool read_matches(const char *filename, epic_matches &out) {                // io.c:23-47
    FILE *f = fopen(filename, "r");
    if (!f) return false;
    out.m.clear();
    float x1, y1, x2, y2;
    while (!feof(f) && fscanf(f, "%f %f %f %f%*[^\n]", &x1, &y1, &x2, &y2) == 4) {
        out.m.push_back(x1); out.m.push_back(y1); out.m.push_back(x2); out.m.push_back(y2);
    }
    fclose(f);
    return true;
}

bool read_edges(const char *filename, int width, int height, epic_edges &out) {   // io.c:14-20
    FILE *f = fopen(filename, "rb");
    if (!f) return false;
    out.width = width; out.height = height;
    out.cost.assign((size_t)width * height, 0.0f);
    const bool ok = fread(out.cost.data(), sizeof(float), out.cost.size(), f) == out.cost.size();
    fclose(f);
    return ok;
}

color_image_t *rgb_to_lab(const color_image_t *im) {                        // image.c:694-726
    color_image_t *res = color_image_new(im->width, im->height);
    const int npix = im->stride * im->height;
    const float T = 0.008856;
    const float color_attenuation = 1.5f;
    for (int i = 0; i < npix; i++) {
        const float r = im->c1[i] / 255.f, g = im->c2[i] / 255.f, b = im->c3[i] / 255.f;
        float X = 0.412453 * r + 0.357580 * g + 0.180423 * b;
        float Y = 0.212671 * r + 0.715160 * g + 0.072169 * b;
        float Z = 0.019334 * r + 0.119193 * g + 0.950227 * b;
        X /= 0.950456;
        Z /= 1.088754;
        const float Y3 = pow(Y, 1. / 3);
        const float fX = X > T ? pow(X, 1. / 3) : 7.787 * X + 16 / 116.;
        const float fY = Y > T ? Y3 : 7.787 * Y + 16 / 116.;
        const float fZ = Z > T ? pow(Z, 1. / 3) : 7.787 * Z + 16 / 116.;
        const float L = Y > T ? 116 * Y3 - 16.0 : 903.3 * Y;
        const float A = 500 * (fX - fY), B = 200 * (fY - fZ);
        const float l2 = (L / 100) * (L / 100);
        const float q = l2 - 0.6;                                            // pow2(float(pow2(L/100) - 0.6)): the argument of pow2 is a float
        const float correct_lab = exp(-color_attenuation * (q * q));         // dark or light areas have less reliable colours
        res->c1[i] = L; res->c2[i] = A * correct_lab; res->c3[i] = B * correct_lab;
    }
    return res;
}

image_t *saliency(sfa_ctx *ctx, const color_image_t *im, float sigma_image, float sigma_matrix) {   // image.c:729-791
    const int w = im->width, h = im->height, st = im->stride;
    const size_t pl = (size_t)st * h;
    std::vector<float> sim(3 * pl, 0.f), ix(3 * pl, 0.f), iy(3 * pl, 0.f);
    const float *src[3] = {im->c1, im->c2, im->c3};
    for (int c = 0; c < 3; c++) {                                            // smooth, then the 3-tap derivatives of the smoothed image
        if (sfa_gaussian_presmooth(ctx, &sim[c * pl], src[c], w, h, st, sigma_image) != SFA_OK) return nullptr;
        if (sfa_convolve(ctx, &ix[c * pl], &sim[c * pl], w, h, st, 1, 1) != SFA_OK) return nullptr;
        if (sfa_convolve(ctx, &iy[c * pl], &sim[c * pl], w, h, st, 1, 0) != SFA_OK) return nullptr;
    }
    std::vector<float> xx(pl, 0.f), xy(pl, 0.f), yy(pl, 0.f), sm(pl, 0.f);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {                                        // autocorrelation matrix, summed over the channels left to right
            const size_t o = (size_t)y * st + x;
            xx[o] = ix[o] * ix[o] + ix[pl + o] * ix[pl + o] + ix[2 * pl + o] * ix[2 * pl + o];
            xy[o] = ix[o] * iy[o] + ix[pl + o] * iy[pl + o] + ix[2 * pl + o] * iy[2 * pl + o];
            yy[o] = iy[o] * iy[o] + iy[pl + o] * iy[pl + o] + iy[2 * pl + o] * iy[2 * pl + o];
        }
    std::vector<float> *mats[3] = {&xx, &xy, &yy};
    for (auto *m : mats) {                                                   // integrate it
        if (sfa_gaussian_presmooth(ctx, sm.data(), m->data(), w, h, st, sigma_matrix) != SFA_OK) return nullptr;
        *m = sm;
    }
    image_t *res = image_new(w, h);
    image_erase(res);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {                                        // smallest eigenvalue
            const size_t o = (size_t)y * st + x;
            const float t = 0.5f * (xx[o] + yy[o]);
            const float d = std::max(0.0f, t * t + xy[o] * xy[o] - xx[o] * yy[o]);
            res->data[o] = sqrtf(std::max(0.0f, t - sqrtf(d)));
        }
    return res;
}

// ---- geodesic nearest neighbours (epic_aux.cpp:44-380) ------------------------------------------------------------------------------------
static float huge_float() {                                                 // memset(.., 0x7F, ..) of the reference: 0x7F7F7F7F
    float v;
    const unsigned bits = 0x7F7F7F7Fu;
    memcpy(&v, &bits, 4);
    return v;
}

// one raster sweep of the distance transform in direction (dx, dy) with label propagation (epic_aux.cpp:92-146)
static float arg_sweep(const std::vector<float> &cost, std::vector<float> &A, std::vector<int> &L, int tx, int ty, int dx, int dy) {
    const float INF = std::numeric_limits<float>::infinity();
    const int bx = dx > 0 ? 0 : tx - 1, by = dy > 0 ? 0 : ty - 1, ex = dx > 0 ? tx : -1, ey = dy > 0 ? ty : -1;
    float max_diff = 0.0f;
    for (int j = by; j != ey; j += dy)
        for (int i = bx; i != ex; i += dx) {
            float t1, t2;
            int l1, l2;
            if (j == by) { t1 = INF; l1 = -1; } else { t1 = A[i + (size_t)(j - dy) * tx]; l1 = L[i + (size_t)(j - dy) * tx]; }
            if (i == bx) { t2 = INF; l2 = -1; } else { t2 = A[i - dx + (size_t)j * tx]; l2 = L[i - dx + (size_t)j * tx]; }
            const float dt12 = fabsf(t1 - t2);
            const float C = cost[i + (size_t)j * tx];
            float t0;
            int l0;
            if (dt12 > C) {                                                  // degenerate case: the front comes from one side only
                if (t1 < t2) { t0 = t1 + C; l0 = l1; } else { t0 = t2 + C; l0 = l2; }
            } else {
                t0 = 0.5 * (t1 + t2 + sqrtf(2 * C * C - dt12 * dt12));
                l0 = (t1 < t2) ? l1 : l2;
            }
            float &a = A[i + (size_t)j * tx];
            if (t0 < a) {
                max_diff = std::max(max_diff, a - t0);
                a = t0;
                L[i + (size_t)j * tx] = l0;
            }
        }
    return max_diff;
}

struct node_dist { int node; float dis; };
struct smallest_on_top { bool operator()(const node_dist &a, const node_dist &b) const { return a.dis > b.dis; } };

void epic_nearest_seeds(const std::vector<int> &seeds, const epic_edges &cost, int nn, epic_nn &out, std::vector<int> &labels) {
    const int tx = cost.width, ty = cost.height, ns = (int)(seeds.size() / 2);
    const float HUGE_F = huge_float();
    std::vector<float> dmap((size_t)tx * ty, HUGE_F);
    labels.assign((size_t)tx * ty, -1);
    for (int i = 0; i < ns; i++) {                                           // epic_aux.cpp:305-309: a later seed on the same pixel wins
        const size_t p = seeds[2 * i] + (size_t)seeds[2 * i + 1] * tx;
        labels[p] = i;
        dmap[p] = cost.cost[p];
    }
    {                                                                        // weighted_distance_transform (:160-180): sweeps in four directions until stable
        const int sx[4] = {-1, 1, 1, -1}, sy[4] = {1, 1, -1, -1};
        const int max_iter = 40;
        const float min_change = 1;
        int i = 0, end_iter = 4;
        while (++i <= end_iter) {
            const float change = arg_sweep(cost.cost, dmap, labels, tx, ty, sx[i % 4], sy[i % 4]);
            if (change > min_change) end_iter = std::min(max_iter, i + 3);   // finish the turn
        }
    }
    // neighbourhood graph of the label regions (:226-283): an edge between two seeds whose regions touch, its length the smallest sum of the two
    // distances across the border; stored as adjacency lists with ascending neighbour index (the order of the reference's sorted CSR rows)
    struct Edge { long key; float len; };
    std::vector<Edge> found;
    {
        std::vector<Edge> all;
        for (int j = 1; j < ty; j++)
            for (int i = 1; i < tx; i++) {
                const size_t o = i + (size_t)j * tx;
                const int l0 = labels[o], l1 = labels[o - 1], l2 = labels[o - tx];
                if (l0 != l1) all.push_back(Edge{(long)std::min(l0, l1) + ((long)std::max(l0, l1) << 32), dmap[o] + dmap[o - 1]});
                if (l0 != l2) all.push_back(Edge{(long)std::min(l0, l2) + ((long)std::max(l0, l2) << 32), dmap[o] + dmap[o - tx]});
            }
        std::sort(all.begin(), all.end(), [](const Edge &a, const Edge &b) { return a.key < b.key || (a.key == b.key && a.len < b.len); });
        for (size_t k = 0; k < all.size(); k++)
            if (k == 0 || all[k].key != all[k - 1].key) found.push_back(all[k]);            // the minimum of every border
    }
    std::vector<std::vector<std::pair<int, float>>> adj(ns);
    for (const Edge &e : found) {
        const int a = (int)(e.key & 0xffffffffL), b = (int)(e.key >> 32);
        if (a < 0 || b < 0 || a >= ns || b >= ns) continue;
        adj[a].push_back({b, e.len});
        adj[b].push_back({a, e.len});
    }
    for (auto &v : adj) std::sort(v.begin(), v.end(), [](const std::pair<int, float> &x, const std::pair<int, float> &y) { return x.first < y.first; });
    // k nearest seeds of every seed: Dijkstra on the graph (:44-87)
    std::vector<int> nnf((size_t)ns * nn, -1);
    std::vector<float> dis((size_t)ns * nn, HUGE_F);
    std::vector<float> done(ns);
    for (int seed = 0; seed < ns; seed++) {
        std::fill(done.begin(), done.end(), HUGE_F);
        std::priority_queue<node_dist, std::vector<node_dist>, smallest_on_top> heap;
        heap.push(node_dist{seed, 0.f});
        done[seed] = 0;
        int n = 0;
        while (!heap.empty()) {
            const node_dist cur = heap.top();
            heap.pop();
            if (cur.dis > done[cur.node]) continue;
            nnf[(size_t)seed * nn + n] = cur.node;
            dis[(size_t)seed * nn + n] = cur.dis;
            if (++n >= nn) break;
            for (const auto &e : adj[cur.node]) {
                const float newd = cur.dis + e.second;
                if (newd >= done[e.first]) continue;
                heap.push(node_dist{e.first, newd});
                done[e.first] = newd;
            }
        }
    }
    // every query point (the seeds themselves) takes the list of the seed whose region it lies in, plus its own distance to it (:366-374)
    out.nn = nn;
    out.index.assign((size_t)ns * nn, -1);
    out.dist.assign((size_t)ns * nn, HUGE_F);
    for (int q = 0; q < ns; q++) {
        const size_t p = seeds[2 * q] + (size_t)seeds[2 * q + 1] * tx;
        const int s = labels[p];
        const float d = dmap[p];
        for (int j = 0; j < nn; j++) {
            out.index[(size_t)q * nn + j] = nnf[(size_t)s * nn + j];
            out.dist[(size_t)q * nn + j] = d + dis[(size_t)s * nn + j];
        }
    }
}

// ---- models (epic_aux.cpp:386-497) ------------------------------------------------------------------------------------------------------------
static void fit_nadarayawatson(std::vector<float> &res, const epic_nn &k, const std::vector<float> &vects) {
    const int ns = (int)(vects.size() / 2), nn = k.nn;
    res.assign((size_t)ns * 2, 0.f);
    for (int i = 0; i < ns; i++) {
        float u = 0.0f, v = 0.0f, s = 0.0f;
        for (int j = i * nn; j < (i + 1) * nn; j++) {
            const float d = k.dist[j];
            const int jj = k.index[j];
            if (jj < 0) continue;                                            // fewer than nn reachable seeds (the reference reads out of bounds here)
            u += d * vects[2 * jj];
            v += d * vects[2 * jj + 1];
            s += d;
        }
        res[2 * i] = u / s;
        res[2 * i + 1] = v / s;
    }
}

// weighted least squares  min sum_k c_k^2 (a x_k + b y_k + t - target_k)^2  for (a, b, t): 3x3 normal equations in double, Gaussian elimination with pivoting
static void solve3(const double M[3][3], const double r[3], double out[3]) {
    double A[3][4];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = M[i][j]; A[i][3] = r[i]; }
    for (int c = 0; c < 3; c++) {
        int piv = c;
        for (int i = c + 1; i < 3; i++) if (fabs(A[i][c]) > fabs(A[piv][c])) piv = i;
        if (piv != c) for (int j = 0; j < 4; j++) std::swap(A[c][j], A[piv][j]);
        if (A[c][c] == 0) { out[0] = out[1] = out[2] = 0; return; }
        for (int i = c + 1; i < 3; i++) {
            const double f = A[i][c] / A[c][c];
            for (int j = c; j < 4; j++) A[i][j] -= f * A[c][j];
        }
    }
    for (int i = 2; i >= 0; i--) {
        double s = A[i][3];
        for (int j = i + 1; j < 3; j++) s -= A[i][j] * out[j];
        out[i] = s / A[i][i];
    }
}

void epic_fit_localaffine(std::vector<float> &affine, const epic_nn &k, const std::vector<int> &seeds, const std::vector<float> &vects) {
    const int ns = (int)(vects.size() / 2), nn = k.nn;
    affine.assign((size_t)ns * 6, 0.f);
    for (int i = 0; i < ns; i++) {
        // the design rows of epic_aux.cpp:438-456: every neighbour weighted by its kernel value (the seed itself by 0.96 of it), plus four points 0.1 px around the
        // seed with 1 % of its weight to rule out collinear configurations.  The x- and y-equations share the matrix and decouple.
        double M[3][3] = {{0}}, rx[3] = {0}, ry[3] = {0};
        auto add = [&](float x, float y, float wx, float wy, float c) {
            const double X = (double)(x * c), Y = (double)(y * c), C = (double)c, tx = (double)((x + wx) * c), ty = (double)((y + wy) * c);
            const double row[3] = {X, Y, C};
            for (int a = 0; a < 3; a++) {
                for (int b = 0; b < 3; b++) M[a][b] += row[a] * row[b];
                rx[a] += row[a] * tx;
                ry[a] += row[a] * ty;
            }
        };
        float coefi = 0.0f;
        for (int j = 0; j < nn; j++) {
            const int s = k.index[(size_t)i * nn + j];
            if (s < 0) continue;                                             // fewer than nn reachable seeds (the reference reads out of bounds here)
            float coef = k.dist[(size_t)i * nn + j];
            if (s == i) { coefi = 0.01f * coef; coef *= 0.96f; }
            add((float)seeds[2 * s], (float)seeds[2 * s + 1], vects[2 * s], vects[2 * s + 1], coef);
        }
        const float xi = (float)seeds[2 * i], yi = (float)seeds[2 * i + 1], ui = vects[2 * i], vi = vects[2 * i + 1];
        add(xi + 0.1f, yi, ui, vi, coefi);
        add(xi, yi + 0.1f, ui, vi, coefi);
        add(xi - 0.1f, yi, ui, vi, coefi);
        add(xi, yi - 0.1f, ui, vi, coefi);
        double px[3], py[3];
        solve3(M, rx, px);
        solve3(M, ry, py);
        float *aff = &affine[(size_t)i * 6];
        aff[0] = (float)px[0]; aff[1] = (float)px[1]; aff[2] = (float)px[2];
        aff[3] = (float)py[0]; aff[4] = (float)py[1]; aff[5] = (float)py[2];
    }
}

// ---- epic() (epic.cpp:147-235) --------------------------------------------------------------------------------------------------------------------
static void to_seeds_and_vects(const std::vector<float> &m, std::vector<int> &seeds, std::vector<float> &vects) {   // epic.cpp:31-56
    const int n = (int)(m.size() / 4);
    seeds.resize((size_t)2 * n);
    vects.resize((size_t)2 * n);
    for (int i = 0; i < n; i++) {
        seeds[2 * i] = (int)m[4 * i]; seeds[2 * i + 1] = (int)m[4 * i + 1];
        vects[2 * i] = m[4 * i + 2] - m[4 * i]; vects[2 * i + 1] = m[4 * i + 3] - m[4 * i + 1];
    }
}

static void kernel(epic_nn &k, float coef) {                                 // epic.cpp:196-198: exp(-coef * d) + 1e-08, the sum formed in double
    for (float &d : k.dist) d = expf(-coef * d) + 1e-08;
}

int epic(sfa_ctx *ctx, image_t *flowx, image_t *flowy, const color_image_t *im, const epic_matches &input, epic_edges &edges, const epic_params_t *params) {
    const int w = im->width, h = im->height;
    if (edges.width != w || edges.height != h) return 2;
    std::vector<float> m(input.m);
    for (size_t i = 0; i + 3 < m.size(); i += 4) {                           // epic.cpp:15-28
        m[i] = std::max(0.f, std::min(m[i], (float)(w - 1)));         m[i + 1] = std::max(0.f, std::min(m[i + 1], (float)(h - 1)));
        m[i + 2] = std::max(0.f, std::min(m[i + 2], (float)(w - 1))); m[i + 3] = std::max(0.f, std::min(m[i + 3], (float)(h - 1)));
    }
    if (params->verbose) printf("%d input matches\n", (int)(m.size() / 4));
    if (params->euc) for (float &c : edges.cost) c += params->euc;           // :155-163
    if (params->saliency_th) {                                               // :59-74
        if (!ctx) return -1;
        image_t *s = saliency(ctx, im, 0.8f, 1.0f);
        if (!s) return -1;
        std::vector<float> keep;
        for (size_t i = 0; i + 3 < m.size(); i += 4)
            if (s->data[(int)(m[i + 1] * s->stride + m[i])] >= params->saliency_th) keep.insert(keep.end(), m.begin() + i, m.begin() + i + 4);
        image_delete(s);
        m.swap(keep);
        if (params->verbose) printf("Saliency filtering, remaining %d matches\n", (int)(m.size() / 4));
    }
    std::vector<int> seeds, labels;
    std::vector<float> vects;
    if (params->pref_nn && !m.empty()) {                                     // :77-123: drop matches that disagree with the estimate from their neighbours
        const float th2 = params->pref_th * params->pref_th;
        const int nns = std::min(params->pref_nn + 1, (int)(m.size() / 4));
        if (nns != params->pref_nn + 1) fprintf(stderr, "Warning: not enough matches for prefiltering\n");
        to_seeds_and_vects(m, seeds, vects);
        epic_nn k;
        epic_nearest_seeds(seeds, edges, nns, k, labels);
        kernel(k, params->coef_kernel);
        std::vector<float> est;
        fit_nadarayawatson(est, k, vects);
        std::vector<float> keep;
        for (size_t i = 0; i < m.size() / 4; i++) {
            const float ex = est[2 * i] - vects[2 * i], ey = est[2 * i + 1] - vects[2 * i + 1];
            if (ex * ex + ey * ey < th2) keep.insert(keep.end(), m.begin() + 4 * i, m.begin() + 4 * i + 4);
        }
        m.swap(keep);
        if (params->verbose) printf("Consistenct filter, remaining %d matches\n", (int)(m.size() / 4));
    }
    if (m.empty()) return 1;
    const int nns = std::min(params->nn, (int)(m.size() / 4));
    if (nns < params->nn) fprintf(stderr, "Warning: not enough matches for interpolating\n");
    to_seeds_and_vects(m, seeds, vects);
    epic_nn k;
    epic_nearest_seeds(seeds, edges, nns, k, labels);
    kernel(k, params->coef_kernel);
    if (!strcmp(params->method, "LA")) {                                     // :204-208, apply :479-497
        std::vector<float> aff;
        epic_fit_localaffine(aff, k, seeds, vects);
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const float *a = &aff[(size_t)labels[(size_t)j * w + i] * 6];
                flowx->data[(size_t)j * flowx->stride + i] = a[0] * i + a[1] * j + a[2] - i;
                flowy->data[(size_t)j * flowy->stride + i] = a[3] * i + a[4] * j + a[5] - j;
            }
    } else if (!strcmp(params->method, "NW")) {                              // :209-213, apply :410-424
        std::vector<float> sv;
        fit_nadarayawatson(sv, k, vects);
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const int s = labels[(size_t)j * w + i];
                flowx->data[(size_t)j * flowx->stride + i] = sv[2 * s];
                flowy->data[(size_t)j * flowy->stride + i] = sv[2 * s + 1];
            }
    } else {
        fprintf(stderr, "method %s not recognized\n", params->method);
        return 3;
    }
    return 0;
}
