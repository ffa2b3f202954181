// png.cpp -- see png.h. Format per the PNG specification (ISO/IEC 15948): 8-byte signature, chunks of
// {length, type, data, crc32(type+data)}, IHDR first, IDAT data is one zlib stream of filtered scanlines.
#include "png.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

const unsigned char kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool write_chunk(FILE *f, const char type[4], const unsigned char *data, size_t n) {
    unsigned char head[8], tail[4];
    put32(head, (uint32_t)n);
    memcpy(head + 4, type, 4);
    uLong crc = crc32(0L, head + 4, 4);
    if (n) crc = crc32(crc, data, (uInt)n);
    put32(tail, (uint32_t)crc);
    return fwrite(head, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
}

}  // namespace

bool png_read(const char *filename, png_image &out) {
    FILE *f = fopen(filename, "rb");
    if (!f) return false;
    std::vector<unsigned char> file;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    if (file.size() < 8 + 25 || memcmp(file.data(), kSignature, 8) != 0) return false;

    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat, palette;
    bool seen_end = false;
    for (size_t pos = 8; pos + 12 <= file.size() && !seen_end;) {
        const uint32_t len = be32(&file[pos]);
        const unsigned char *type = &file[pos + 4], *data = &file[pos + 8];
        if (len > file.size() - pos - 12) return false;
        if (be32(data + len) != (uint32_t)crc32(crc32(0L, type, 4), data, len)) return false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return false;
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return false;
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!seen_end || ctype < 0 || w == 0 || h == 0 || w > 65535 || h > 65535 || (uint64_t)w * h > (1ull << 28) || interlace != 0) return false;   // a corrupt header must not ask for terabytes
    int comps;
    switch (ctype) {
        case 0: comps = 1; break;
        case 2: comps = 3; break;
        case 3: comps = 1; break;
        case 4: comps = 2; break;
        case 6: comps = 4; break;
        default: return false;
    }
    const bool sub_byte = depth == 1 || depth == 2 || depth == 4;
    if (!(depth == 8 || (depth == 16 && ctype != 3) || (sub_byte && (ctype == 0 || ctype == 3)))) return false;
    if (ctype == 3 && (palette.empty() || palette.size() % 3)) return false;

    const size_t bpp = (size_t)(comps * depth + 7) / 8;                   // filter unit
    const size_t line = ((size_t)w * comps * depth + 7) / 8;
    if ((line + 1) * (size_t)h > idat.size() * (size_t)1100 + 4096) return false;      // deflate expands by at most ~1032 : 1: the data cannot hold this image
    std::vector<unsigned char> raw((line + 1) * (size_t)h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return false;

    std::vector<unsigned char> prev(line, 0), cur(line);
    out.width = (int)w; out.height = (int)h;
    out.channels = (ctype == 0 || ctype == 4) ? 1 : 3;
    out.depth = depth == 16 ? 16 : 8;
    out.samples.assign((size_t)w * h * out.channels, 0);
    for (uint32_t y = 0; y < h; y++) {
        const unsigned char *src = &raw[(line + 1) * y];
        const int filter = src[0];
        if (filter > 4) return false;
        for (size_t i = 0; i < line; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: pred = paeth(a, b, c); break;
            }
            cur[i] = (unsigned char)(src[1 + i] + pred);
        }
        uint16_t *dst = &out.samples[(size_t)y * w * out.channels];
        for (uint32_t x = 0; x < w; x++) {
            if (ctype == 3 || sub_byte) {
                unsigned v;
                if (depth == 8) v = cur[x];
                else {
                    const unsigned per = 8 / depth, shift = (per - 1 - x % per) * depth;
                    v = (cur[x / per] >> shift) & ((1u << depth) - 1);
                }
                if (ctype == 3) {
                    if ((size_t)v * 3 + 2 >= palette.size()) return false;
                    for (int k = 0; k < 3; k++) dst[x * 3 + k] = palette[v * 3 + k];
                } else {
                    dst[x] = (uint16_t)(v * 255u / ((1u << depth) - 1));      // grey levels spread over 0..255
                }
            } else {
                const size_t bytes = depth / 8;
                for (int k = 0; k < out.channels; k++) {
                    const unsigned char *p = &cur[((size_t)x * comps + k) * bytes];
                    dst[(size_t)x * out.channels + k] = depth == 16 ? (uint16_t)((p[0] << 8) | p[1]) : p[0];
                }
            }
        }
        prev.swap(cur);
    }
    return true;
}

bool png_write(const char *filename, const png_image &img) {
    if (img.width <= 0 || img.height <= 0 || (img.channels != 1 && img.channels != 3) || (img.depth != 8 && img.depth != 16)) return false;
    if (img.samples.size() != (size_t)img.width * img.height * img.channels) return false;
    const size_t bytes = img.depth / 8, line = (size_t)img.width * img.channels * bytes;
    std::vector<unsigned char> raw((line + 1) * (size_t)img.height);
    for (int y = 0; y < img.height; y++) {
        unsigned char *d = &raw[(line + 1) * y];
        *d++ = 0;
        const uint16_t *s = &img.samples[(size_t)y * img.width * img.channels];
        for (size_t i = 0; i < (size_t)img.width * img.channels; i++) {
            if (bytes == 2) { *d++ = s[i] >> 8; *d++ = s[i] & 255; }
            else *d++ = (unsigned char)(s[i] > 255 ? 255 : s[i]);
        }
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;

    FILE *f = fopen(filename, "wb");
    if (!f) return false;
    unsigned char ihdr[13];
    put32(ihdr, (uint32_t)img.width);
    put32(ihdr + 4, (uint32_t)img.height);
    ihdr[8] = (unsigned char)img.depth;
    ihdr[9] = img.channels == 3 ? 2 : 0;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    bool ok = fwrite(kSignature, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr, 13) && write_chunk(f, "IDAT", z.data(), zn)
              && write_chunk(f, "IEND", nullptr, 0);
    return (fclose(f) == 0) && ok;
}
