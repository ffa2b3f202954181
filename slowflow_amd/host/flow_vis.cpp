// flow_vis.cpp -- see flow_vis.h
#include "flow_vis.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace {

const int RY = 15, YG = 6, GC = 4, CB = 11, BM = 13, MR = 6, NCOLS = RY + YG + GC + CB + BM + MR;   // 55 hues

struct Wheel {
    int c[NCOLS][3];
    Wheel() {
        int k = 0;
        for (int i = 0; i < RY; i++, k++) set(k, 255, 255 * i / RY, 0);
        for (int i = 0; i < YG; i++, k++) set(k, 255 - 255 * i / YG, 255, 0);
        for (int i = 0; i < GC; i++, k++) set(k, 0, 255, 255 * i / GC);
        for (int i = 0; i < CB; i++, k++) set(k, 0, 255 - 255 * i / CB, 255);
        for (int i = 0; i < BM; i++, k++) set(k, 255 * i / BM, 0, 255);
        for (int i = 0; i < MR; i++, k++) set(k, 255, 0, 255 - 255 * i / MR);
    }
    void set(int k, int r, int g, int b) { c[k][0] = r; c[k][1] = g; c[k][2] = b; }
};
const Wheel wheel;

bool known(float v) { return std::fabs(v) <= UNKNOWN_FLOW_THRESH; }

}  // namespace

void computeColor(float fx, float fy, unsigned char pix[3]) {
    const float rad = std::sqrt(fx * fx + fy * fy);
    const float a = std::atan2(-fy, -fx) / (float)M_PI;          // hue from the direction
    const float fk = (a + 1.0f) / 2.0f * (NCOLS - 1);
    const int k0 = (int)fk, k1 = (k0 + 1) % NCOLS;
    const float f = fk - k0;
    for (int b = 0; b < 3; b++) {
        const float col0 = wheel.c[k0][b] / 255.0f, col1 = wheel.c[k1][b] / 255.0f;
        float col = (1 - f) * col0 + f * col1;
        if (rad <= 1) col = 1 - rad * (1 - col);                  // saturation grows with the magnitude
        else col *= .75f;                                         // out of range
        pix[b] = (unsigned char)(int)(255.0f * col);
    }
}

png_image flowColorImg(const image_t *wx, const image_t *wy, int verbose, float maxrad) {
    const int width = wx->width, height = wx->height;
    png_image img;
    img.width = width; img.height = height; img.channels = 3; img.depth = 8;
    img.samples.assign((size_t)width * height * 3, 0);
    if (maxrad <= 0) {                                            // motion range (utils.cpp:1008-1029)
        double maxx = -999, maxy = -999, minx = 999, miny = 999;
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                const double fx = wx->data[(size_t)y * wx->stride + x], fy = wy->data[(size_t)y * wy->stride + x];
                if (std::fabs(fx) > width || std::fabs(fy) > height) continue;
                maxx = std::max(maxx, fx); maxy = std::max(maxy, fy);
                minx = std::min(minx, fx); miny = std::min(miny, fy);
                const float rad = (float)std::sqrt(fx * fx + fy * fy);
                maxrad = std::max(maxrad, rad);
            }
        if (verbose > 0) printf("max motion: %.4f  motion range: u = %.3f .. %.3f;  v = %.3f .. %.3f\n", maxrad, minx, maxx, miny, maxy);
    }
    if (!(maxrad > 0)) maxrad = 1;                                // flow == 0 everywhere (or nothing usable)
    if (verbose > 0) fprintf(stderr, "normalizing by %g\n", maxrad);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const double fx = wx->data[(size_t)y * wx->stride + x], fy = wy->data[(size_t)y * wy->stride + x];
            unsigned char pix[3] = {0, 0, 0};
            if (!(std::isnan(fx) || std::isnan(fy) || std::fabs(fx) > width || std::fabs(fy) > height))
                computeColor((float)(fx / maxrad), (float)(fy / maxrad), pix);
            for (int k = 0; k < 3; k++) img.samples[((size_t)y * width + x) * 3 + k] = pix[k];
        }
    return img;
}

double computeEPE(const image_t *flow_x, const image_t *flow_y, const image_t *gt_x, const image_t *gt_y, const image_t *mask) {
    if (flow_x->height != gt_x->height || flow_x->width != gt_x->width) {
        fprintf(stderr, "Error (computeEPE): Dimension do not fit between ground truth and estimation!\n");
        return -1;
    }
    double epe = 0;
    long counter = 0;
    for (int y = 0; y < gt_x->height; y++)
        for (int x = 0; x < gt_x->width; x++) {
            if (mask && mask->data[(size_t)y * mask->stride + x] == 0) continue;
            const float gx = gt_x->data[(size_t)y * gt_x->stride + x], gy = gt_y->data[(size_t)y * gt_y->stride + x];
            const float ex = flow_x->data[(size_t)y * flow_x->stride + x], ey = flow_y->data[(size_t)y * flow_y->stride + x];
            if (!known(gx) || !known(gy) || !known(ex) || !known(ey)) continue;
            const float t1 = ex - gx, t2 = ey - gy;
            epe += std::sqrt(t1 * t1 + t2 * t2);                  // float sqrt of a float, accumulated in double (:58-63)
            counter++;
        }
    return counter ? epe / counter : epe;
}

double computeAAE(const image_t *flow_x, const image_t *flow_y, const image_t *gt_x, const image_t *gt_y, const image_t *mask) {
    if (flow_x->height != gt_x->height || flow_x->width != gt_x->width) {
        fprintf(stderr, "Error (computeAAE): Dimension do not fit between ground truth and estimation!\n");
        return -1;
    }
    double aae = 0;
    long counter = 0;
    for (int y = 0; y < gt_x->height; y++)
        for (int x = 0; x < gt_x->width; x++) {
            if (mask && mask->data[(size_t)y * mask->stride + x] == 0) continue;
            const float gx = gt_x->data[(size_t)y * gt_x->stride + x], gy = gt_y->data[(size_t)y * gt_y->stride + x];
            const float ex = flow_x->data[(size_t)y * flow_x->stride + x], ey = flow_y->data[(size_t)y * flow_y->stride + x];
            if (!known(gx) || !known(gy) || !known(ex) || !known(ey)) continue;
            const double n1 = std::sqrt(ex * ex + ey * ey + 1.0f * 1.0f), n2 = std::sqrt(gx * gx + gy * gy + 1.0f * 1.0f);   // float sums (:126-127)
            const double t1 = ex * gx, t2 = ey * gy;                                                                         // float products (:129-130)
            aae += std::acos(std::min((t1 + t2 + 1.0) / (n1 * n2), 1.0));
            counter++;
        }
    return counter ? aae / counter : aae;
}

image_t *flow_resize_nearest(const image_t *src, float scale) {
    const int dw = (int)std::lround((double)src->width * scale), dh = (int)std::lround((double)src->height * scale);
    if (dw <= 0 || dh <= 0) return nullptr;
    image_t *dst = image_new(dw, dh);
    image_erase(dst);
    const double ifx = 1.0 / scale, ify = 1.0 / scale;
    for (int y = 0; y < dh; y++) {
        const int sy = std::min((int)std::floor(y * ify), src->height - 1);
        for (int x = 0; x < dw; x++) {
            const int sx = std::min((int)std::floor(x * ifx), src->width - 1);
            dst->data[(size_t)y * dst->stride + x] = src->data[(size_t)sy * src->stride + sx] * scale;
        }
    }
    return dst;
}
