// image.h -- host-side image containers of the drop-in: the same C structs and the subset of functions that the
// variational path and its driver use in the reference (epic_flow_extended/image.h:17-66).  Pure host memory
// management; all per-pixel work of the path runs on the GPU behind include/slowflow_amd.h.
#ifndef SLOWFLOW_AMD_HOST_IMAGE_H
#define SLOWFLOW_AMD_HOST_IMAGE_H

#ifdef __cplusplus
extern "C" {
#endif

/* 1-channel image: stride = width rounded up to a multiple of 4, data 16-byte aligned (image.h:17-23) */
typedef struct image_s { int width, height, stride; float *data; } image_t;
/* 3 planar channels, c2 = c1 + stride*height, c3 = c2 + stride*height (image.h:26-33) */
typedef struct color_image_s { int width, height, stride; float *c1, *c2, *c3; } color_image_t;

image_t *image_new(int width, int height);                 /* uninitialised, like the reference (image.c:17-33) */
image_t *image_cpy(const image_t *src);
void image_erase(image_t *image);
void image_delete(image_t *image);
void image_mul_scalar(image_t *image, float scalar);       /* image.c:49-57 */
color_image_t *color_image_new(int width, int height);
color_image_t *color_image_cpy(const color_image_t *src);
void color_image_erase(color_image_t *image);
void color_image_delete(color_image_t *image);

#ifdef __cplusplus
}
#endif
#endif
