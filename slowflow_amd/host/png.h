// png.h -- PNG in and out over zlib alone (the reference goes through OpenCV's imread/imwrite: slow_flow.cpp:466-476,
// :556-566,:913-925). Reader: colour types 0 (grey), 2 (RGB), 3 (palette), 4 (grey+alpha), 6 (RGBA) at 8 or 16 bit
// (and 1/2/4-bit grey/palette), non-interlaced, all five scanline filters; alpha is dropped as imread does without
// IMREAD_UNCHANGED. Writer: 8/16-bit grey or RGB, filter 0, one IDAT.
#ifndef SLOWFLOW_AMD_HOST_PNG_H
#define SLOWFLOW_AMD_HOST_PNG_H

#include <cstdint>
#include <vector>

struct png_image {
    int width = 0, height = 0;
    int channels = 0;                 // 1 (grey) or 3 (RGB, in file order R,G,B)
    int depth = 0;                    // 8 or 16: samples[] holds values in 0..255 or 0..65535
    std::vector<uint16_t> samples;    // interleaved, row-major, width*height*channels
};

bool png_read(const char *filename, png_image &out);            // false on any malformed or unsupported file
bool png_write(const char *filename, const png_image &img);     // false on I/O failure

#endif
