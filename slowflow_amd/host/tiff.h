// tiff.h -- baseline TIFF in (and a minimal writer for tests and tools): the reference's own configuration names its input as
// `0%06i.tif`, 16-bit raw Bayer frames read through cv::imread(..., CV_LOAD_IMAGE_UNCHANGED) (cfgs/slow_flow.cfg:5,13-16,
// slow_flow.cpp:470).  Reader: classic TIFF (not BigTIFF), both byte orders, first image directory, strips (not tiles),
// chunky planar configuration, 8 or 16 bits per sample, 1 (grey) or 3/4 (RGB, RGBA: alpha dropped) samples per pixel,
// compression none / LZW / Deflate (with or without the horizontal predictor) / PackBits; photometric WhiteIsZero is inverted.
#ifndef SLOWFLOW_AMD_HOST_TIFF_H
#define SLOWFLOW_AMD_HOST_TIFF_H

#include "png.h"

bool tiff_read(const char *filename, png_image &out);            // false on any malformed or unsupported file
bool tiff_write(const char *filename, const png_image &img);     // uncompressed, little-endian, one strip; false on I/O failure

#endif
