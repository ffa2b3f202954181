// io.h -- the file formats either side of the path, without third-party image libraries:
//   .flo  Middlebury flow (float 202021.25, int w, int h, interleaved u,v rows) -- io.c:53-101 of the reference
//   .ppm / .pgm  binary P6 / P5, 8 or 16 bit (big-endian samples), .pfm (Pf / PF, float) and .png (png.h)
#ifndef SLOWFLOW_AMD_HOST_IO_H
#define SLOWFLOW_AMD_HOST_IO_H

#include "image.h"

int writeFlowFile(const char *filename, const image_t *flowx, const image_t *flowy);   /* 0 on success */
image_t **readFlowFile(const char *filename);                                           /* [0]=u, [1]=v; NULL on failure */
/* loads a frame as 3 float planes (grey images are replicated); *maxval = 255 / 65535 / 1 (pfm). NULL on failure */
color_image_t *color_image_load(const char *filename, int *maxval);
/* binary P5, 8 bit: value = clamp(round(scale * (v + offset)), 0, 255); 0 on success */
int writePGM(const char *filename, const image_t *img, float offset, float scale);

#endif
