// flow_vis.h -- the callers on the output side of the path: flow colour coding and the error measures against ground
// truth (reference: utils/utils.cpp:39-142 computeEPE / computeAAE, :998-1052 flowColorImg; the colour wheel itself is
// the Middlebury devkit's colorcode.cpp, which the reference includes from a path outside its tree -- configuration.h
// -- so it is restated here from the published method of Baker et al., "A Database and Evaluation Methodology for
// Optical Flow": parity unpinned for the colours, pinned by the wheel's known anchor colours in tests/host).
#ifndef SLOWFLOW_AMD_HOST_FLOW_VIS_H
#define SLOWFLOW_AMD_HOST_FLOW_VIS_H

#include "image.h"
#include "png.h"

#define UNKNOWN_FLOW_THRESH 1e9           /* flowIO.h of the devkit: larger magnitudes mean "no ground truth here" */

/* colour of the flow vector (fx, fy), already divided by the normalising radius; pix = {R, G, B} */
void computeColor(float fx, float fy, unsigned char pix[3]);
/* flowColorImg (utils.cpp:998): maxrad <= 0 -> the largest magnitude among vectors with |u| <= w, |v| <= h; 0 -> 1.
   NaN or out-of-range vectors are black. Returns an 8-bit RGB image. */
png_image flowColorImg(const image_t *wx, const image_t *wy, int verbose = 0, float maxrad = -1);
/* mean end-point error over pixels where both flows are known; optional mask (0 = skip); -1 on size mismatch (utils.cpp:39) */
double computeEPE(const image_t *flow_x, const image_t *flow_y, const image_t *gt_x, const image_t *gt_y, const image_t *mask = nullptr);
/* mean angular error (radians) of the space-time vectors (u, v, 1) (utils.cpp:111) */
double computeAAE(const image_t *flow_x, const image_t *flow_y, const image_t *gt_x, const image_t *gt_y, const image_t *mask = nullptr);
/* ground-truth preparation of slow_flow.cpp:634-641: nearest-neighbour resize by `scale` (cv::resize INTER_NEAREST:
   src = min(floor(dst / scale), n-1), size = round(n * scale)) and values times scale */
image_t *flow_resize_nearest(const image_t *src, float scale);

#endif
