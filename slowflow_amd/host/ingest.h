// ingest.h -- what the driver does to a frame between the file and the path (slow_flow.cpp:447-600), without OpenCV:
// Bayer demosaicing (the reference's own bilinear / green-ratio routine), raw channel weighting, cropping, and the
// anti-aliased rescaling (cv::GaussianBlur + cv::resize semantics, run through the C-ABI's pyramid operators).
#ifndef SLOWFLOW_AMD_HOST_INGEST_H
#define SLOWFLOW_AMD_HOST_INGEST_H

#include "../../include/slowflow_amd.h"
#include "image.h"

/* utils.cpp:1336-1374: per-pixel data-term weights of the three channels for a Bayer mosaic whose first red pixel sits at
 * (red_x, red_y); the measured colour of a pixel gets `weight` (clamped to [0,3]), the two interpolated ones (3-weight)/2 */
void rawWeighting(color_image_t *weights, int red_x, int red_y, float weight);
/* utils.cpp:1241-1334 (bayer2rgbGR): green by 4-neighbour averaging, red / blue through the local green ratio; mirrored
 * borders.  src: the mosaic (1 plane), dst: R, G, B planes */
void bayer2rgbGR(const image_t *src, color_image_t *dst, int red_x, int red_y);
/* raw_demosaicing 2 (slow_flow.cpp:502-520): the mosaic is converted to 8 bit with saturation (Mat::convertTo(CV_8UC1): round half to even, clamp)
 * and demosaiced as cv::cvtColor(CV_Bayer*2RGB) does on 8-bit data: at a red / blue site green = (4 neighbours + 2) >> 2 and the opposite colour =
 * (4 diagonals + 2) >> 2, at a green site each colour = (its 2 neighbours + 1) >> 1; the outermost rows and columns repeat their inner neighbours.
 * OpenCV is not in this image: the arithmetic is restated from its documented bilinear Bayer conversion, parity unpinned.  dst: R, G, B planes, 0..255 */
void bayer2rgb_cv8u(const image_t *src, color_image_t *dst, int red_x, int red_y);
/* img.rowRange / colRange of slow_flow.cpp:543-546; returns a new image (caller frees) */
color_image_t *color_image_crop(const color_image_t *img, int center_x, int center_y, int extent_x, int extent_y);
/* slow_flow.cpp:550-553: GaussianBlur(sigma = 1/sqrt(2*scale), BORDER_REPLICATE) then resize(Size(0,0), scale, scale,
 * INTER_LINEAR); returns a new image of cvRound(w*scale) x cvRound(h*scale) or NULL (see sfa_last_error) */
color_image_t *color_image_rescale(sfa_ctx *ctx, const color_image_t *img, float scale);

#endif
