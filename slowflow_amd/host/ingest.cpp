// ingest.cpp -- see ingest.h.  Restated from the reference's driver utilities; those live in utils.cpp, which needs OpenCV and
// cannot be built here: parity unpinned (tests/test_host.py checks the defining properties).
#include "ingest.h"

#include <cmath>
#include <cstring>
#include <vector>

void rawWeighting(color_image_t *weights, int red_x, int red_y, float weight) {
    weight = fminf(fmaxf(weight, 0.0f), 3.0f);                                        // utils.cpp:1337
    const float other = 0.5f * (3 - weight);
    for (int y = 0; y < weights->height; y++)
        for (int x = 0; x < weights->width; x++) {
            const size_t o = (size_t)y * weights->stride + x;
            float r = other, g = other, b = other;
            if ((y + (1 - red_y)) % 2 == 0) {                                         // blue row (:1342)
                const bool green = (red_y == 1 && (x + (1 - red_x)) % 2 == 0) || (red_y == 0 && (x + red_x) % 2 == 0);
                if (green) g = weight; else b = weight;
            } else {                                                                  // red row (:1357)
                const bool green = (red_y == 0 && (x + (1 - red_x)) % 2 == 0) || (red_y == 1 && (x + red_x) % 2 == 0);
                if (green) g = weight; else r = weight;
            }
            weights->c1[o] = r; weights->c2[o] = g; weights->c3[o] = b;
        }
}

void bayer2rgbGR(const image_t *src, color_image_t *dst, int red_x, int red_y) {
    const int W = src->width, H = src->height;
    auto S = [&](int y, int x) { return src->data[(size_t)y * src->stride + x]; };
    auto G = [&](int y, int x) -> float & { return dst->c2[(size_t)y * dst->stride + x]; };
    auto mirror = [](int v, int n, int d) { const int q = v + d; return (q < 0 || q > n - 1) ? v - d : q; };   // (x>0)?x-1:x+1 etc. (:1244-1247)
    // green first: it is the densest (:1242-1276)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int xm1 = mirror(x, W, -1), xp1 = mirror(x, W, 1), ym1 = mirror(y, H, -1), yp1 = mirror(y, H, 1);
            const bool blue_row = (y + (1 - red_y)) % 2 == 0;
            const bool green = blue_row ? (x + red_x) % 2 == 0 : (x + (1 - red_x)) % 2 == 0;
            G(y, x) = green ? S(y, x) : (float)(0.25 * (S(ym1, x) + S(yp1, x) + S(y, xm1) + S(y, xp1)));
        }
    // red and blue through the green ratio (:1279-1333)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int xm1 = mirror(x, W, -1), xp1 = mirror(x, W, 1), ym1 = mirror(y, H, -1), yp1 = mirror(y, H, 1);
            const size_t o = (size_t)y * dst->stride + x;
            const float g = G(y, x);
            const float vert = (float)(g * 0.5 * (S(ym1, x) / G(ym1, x) + S(yp1, x) / G(yp1, x)));
            const float horz = (float)(g * 0.5 * (S(y, xm1) / G(y, xm1) + S(y, xp1) / G(y, xp1)));
            const float diag = (float)(g * 0.25 * (S(ym1, xm1) / G(ym1, xm1) + S(ym1, xp1) / G(ym1, xp1) + S(yp1, xm1) / G(yp1, xm1) + S(yp1, xp1) / G(yp1, xp1)));
            const bool blue_row = (y + (1 - red_y)) % 2 == 0;
            if (blue_row) {
                if ((x + red_x) % 2 == 0) { dst->c1[o] = vert; dst->c3[o] = horz; }    // green pixel
                else                      { dst->c1[o] = diag; dst->c3[o] = S(y, x); } // blue pixel
            } else {
                if ((x + (1 - red_x)) % 2 == 0) { dst->c1[o] = horz; dst->c3[o] = vert; }     // green pixel
                else                            { dst->c1[o] = S(y, x); dst->c3[o] = diag; }  // red pixel
            }
        }
}

void bayer2rgb_cv8u(const image_t *src, color_image_t *dst, int red_x, int red_y) {
    const int W = src->width, H = src->height;
    std::vector<int> m((size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const long v = lrintf(src->data[(size_t)y * src->stride + x]);            // cvRound under the default rounding mode: half to even
            m[(size_t)y * W + x] = (int)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    auto M = [&](int y, int x) { return m[(size_t)y * W + x]; };
    auto put = [&](int y, int x, int r, int g, int b) {
        const size_t o = (size_t)y * dst->stride + x;
        dst->c1[o] = (float)r; dst->c2[o] = (float)g; dst->c3[o] = (float)b;
    };
    auto copy = [&](int y, int x, int ys, int xs) {
        const size_t o = (size_t)y * dst->stride + x, s = (size_t)ys * dst->stride + xs;
        dst->c1[o] = dst->c1[s]; dst->c2[o] = dst->c2[s]; dst->c3[o] = dst->c3[s];
    };
    if (W < 3 || H < 3) {                                                              // no interior: nothing to interpolate from
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) put(y, x, 0, 0, 0);
        return;
    }
    for (int y = 1; y < H - 1; y++) {
        const bool red_row = ((y - red_y) & 1) == 0;
        for (int x = 1; x < W - 1; x++) {
            const bool red_col = ((x - red_x) & 1) == 0;
            const int c = M(y, x);
            if (red_row == red_col) {                                                  // a red or a blue site
                const int cross = (M(y - 1, x) + M(y + 1, x) + M(y, x - 1) + M(y, x + 1) + 2) >> 2;
                const int diag = (M(y - 1, x - 1) + M(y - 1, x + 1) + M(y + 1, x - 1) + M(y + 1, x + 1) + 2) >> 2;
                if (red_row) put(y, x, c, cross, diag); else put(y, x, diag, cross, c);
            } else {                                                                   // a green site: its row's colour left and right, the other one above and below
                const int horiz = (M(y, x - 1) + M(y, x + 1) + 1) >> 1, vert = (M(y - 1, x) + M(y + 1, x) + 1) >> 1;
                if (red_row) put(y, x, horiz, c, vert); else put(y, x, vert, c, horiz);
            }
        }
        copy(y, 0, y, 1);
        copy(y, W - 1, y, W - 2);
    }
    for (int x = 0; x < W; x++) { copy(0, x, 1, x); copy(H - 1, x, H - 2, x); }
}

color_image_t *color_image_crop(const color_image_t *img, int cx, int cy, int ex, int ey) {
    const int x0 = cx - ex / 2, x1 = cx + ex / 2, y0 = cy - ey / 2, y1 = cy + ey / 2;    // Range(center - extent/2, center + extent/2)
    if (x0 < 0 || y0 < 0 || x1 > img->width || y1 > img->height || x1 <= x0 || y1 <= y0) return nullptr;
    color_image_t *out = color_image_new(x1 - x0, y1 - y0);
    color_image_erase(out);
    const float *sp[3] = {img->c1, img->c2, img->c3};
    float *dp[3] = {out->c1, out->c2, out->c3};
    for (int c = 0; c < 3; c++)
        for (int y = y0; y < y1; y++) memcpy(dp[c] + (size_t)(y - y0) * out->stride, sp[c] + (size_t)y * img->stride + x0, (size_t)(x1 - x0) * sizeof(float));
    return out;
}

color_image_t *color_image_rescale(sfa_ctx *ctx, const color_image_t *img, float scale) {
    const int w = img->width, h = img->height;
    const int dw = (int)lrint((double)w * scale), dh = (int)lrint((double)h * scale);   // saturate_cast<int>(src.cols * fx)
    if (dw < 1 || dh < 1) return nullptr;
    color_image_t *blur = color_image_new(w, h), *out = color_image_new(dw, dh);
    color_image_erase(blur); color_image_erase(out);
    const float sigma = (float)(1 / sqrt(2 * scale));                                   // slow_flow.cpp:551
    const float *sp[3] = {img->c1, img->c2, img->c3};
    float *bp[3] = {blur->c1, blur->c2, blur->c3}, *dp[3] = {out->c1, out->c2, out->c3};
    int rc = SFA_OK;
    for (int c = 0; c < 3 && rc == SFA_OK; c++) {
        rc = sfa_gaussian_blur(ctx, bp[c], sp[c], w, h, img->stride, sigma);
        if (rc == SFA_OK) rc = sfa_resize_linear_fx(ctx, dp[c], dw, dh, out->stride, bp[c], w, h, blur->stride, scale, scale);
    }
    color_image_delete(blur);
    if (rc != SFA_OK) { color_image_delete(out); return nullptr; }
    return out;
}
