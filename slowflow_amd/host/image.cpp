// image.cpp -- see image.h
#include "image.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

static float *aligned_floats(size_t n) {
    void *p = nullptr;
    if (posix_memalign(&p, 64, (n ? n : 1) * sizeof(float)) != 0) {
        fprintf(stderr, "Error: image allocation of %zu floats failed\n", n);
        exit(1);                                          // the reference's behaviour (image.c:27-30)
    }
    return static_cast<float *>(p);
}

extern "C" {

image_t *image_new(int width, int height) {
    image_t *im = static_cast<image_t *>(malloc(sizeof(image_t)));
    if (!im) { fprintf(stderr, "Error: image_new() - not enough memory !\n"); exit(1); }
    im->width = width; im->height = height; im->stride = ((width + 3) / 4) * 4;
    im->data = aligned_floats((size_t)im->stride * height);
    return im;
}
image_t *image_cpy(const image_t *src) {
    image_t *d = image_new(src->width, src->height);
    memcpy(d->data, src->data, sizeof(float) * src->stride * src->height);
    return d;
}
void image_erase(image_t *im) { memset(im->data, 0, sizeof(float) * im->stride * im->height); }
void image_delete(image_t *im) { if (im) { free(im->data); free(im); } }
void image_mul_scalar(image_t *im, float s) {
    const size_t n = (size_t)im->stride * im->height;
    for (size_t i = 0; i < n; i++) im->data[i] *= s;
}
color_image_t *color_image_new(int width, int height) {
    color_image_t *im = static_cast<color_image_t *>(malloc(sizeof(color_image_t)));
    if (!im) { fprintf(stderr, "Error: color_image_new() - not enough memory !\n"); exit(1); }
    im->width = width; im->height = height; im->stride = ((width + 3) / 4) * 4;
    const size_t plane = (size_t)im->stride * height;
    im->c1 = aligned_floats(3 * plane);
    im->c2 = im->c1 + plane;
    im->c3 = im->c2 + plane;
    return im;
}
color_image_t *color_image_cpy(const color_image_t *src) {
    color_image_t *d = color_image_new(src->width, src->height);
    memcpy(d->c1, src->c1, sizeof(float) * 3 * src->stride * src->height);
    return d;
}
void color_image_erase(color_image_t *im) { memset(im->c1, 0, sizeof(float) * 3 * im->stride * im->height); }
void color_image_delete(color_image_t *im) { if (im) { free(im->c1); free(im); } }

}  // extern "C"
