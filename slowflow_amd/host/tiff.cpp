// tiff.cpp -- see tiff.h.  Written against the TIFF 6.0 specification (baseline + the LZW extension); no libtiff in this image.
#include "tiff.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>

#include <zlib.h>

namespace {

struct Reader {
    std::vector<unsigned char> d;
    bool be = false;
    bool ok(size_t off, size_t n) const { return off <= d.size() && n <= d.size() - off; }
    uint16_t u16(size_t o) const { return be ? (uint16_t)((d[o] << 8) | d[o + 1]) : (uint16_t)(d[o] | (d[o + 1] << 8)); }
    uint32_t u32(size_t o) const {
        return be ? ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]
                  : ((uint32_t)d[o + 3] << 24) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 1] << 8) | d[o];
    }
};

// values of one directory entry (types BYTE 1, SHORT 3, LONG 4), wherever they are stored
bool entry_values(const Reader &r, size_t e, std::vector<uint32_t> &v) {
    const uint16_t type = r.u16(e + 2);
    const uint32_t count = r.u32(e + 4);
    const size_t sz = type == 1 ? 1 : type == 3 ? 2 : type == 4 ? 4 : 0;
    if (!sz || count == 0 || count > (1u << 26)) return false;
    size_t off = e + 8;
    if (sz * count > 4) { off = r.u32(e + 8); if (!r.ok(off, sz * count)) return false; }
    v.resize(count);
    for (uint32_t i = 0; i < count; i++) v[i] = sz == 1 ? r.d[off + i] : sz == 2 ? r.u16(off + 2 * i) : r.u32(off + 4 * i);
    return true;
}

bool unpack_bits(const unsigned char *src, size_t n, std::vector<unsigned char> &out, size_t want) {
    size_t i = 0;
    while (i < n && out.size() < want) {
        const int c = (signed char)src[i++];
        if (c >= 0) {
            if (i + c + 1 > n) return false;
            out.insert(out.end(), src + i, src + i + c + 1);
            i += c + 1;
        } else if (c != -128) {
            if (i >= n) return false;
            out.insert(out.end(), (size_t)(1 - c), src[i++]);
        }
    }
    return out.size() >= want;
}

// TIFF flavour of LZW: codes packed most significant bit first, widths 9..12 switching one code early, Clear = 256, EndOfInformation = 257
bool unpack_lzw(const unsigned char *src, size_t n, std::vector<unsigned char> &out, size_t want) {
    std::vector<uint16_t> prefix(4096), length(4096);
    std::vector<unsigned char> suffix(4096), first(4096);
    for (int i = 0; i < 256; i++) { prefix[i] = 0xffff; length[i] = 1; suffix[i] = first[i] = (unsigned char)i; }
    int nbits = 9, next = 258, old = -1;
    uint32_t acc = 0;
    int have = 0;
    size_t pos = 0;
    auto emit = [&](int code) {
        const size_t at = out.size(), len = length[code];
        out.resize(at + len);
        for (size_t k = len; k-- > 0; code = prefix[code]) out[at + k] = suffix[code];
    };
    while (out.size() < want) {
        while (have < nbits) { if (pos >= n) return out.size() >= want; acc = (acc << 8) | src[pos++]; have += 8; }
        const int code = (int)((acc >> (have - nbits)) & ((1u << nbits) - 1));
        have -= nbits;
        if (code == 257) break;
        if (code == 256) { nbits = 9; next = 258; old = -1; continue; }
        if (old < 0) {
            if (code >= 256) return false;
            emit(code);
        } else {
            if (code > next || next >= 4096) return false;
            prefix[next] = (uint16_t)old;
            length[next] = (uint16_t)(length[old] + 1);
            first[next] = first[old];
            suffix[next] = code < next ? first[code] : first[old];
            next++;
            emit(code);
            if (next >= (1 << nbits) - 1 && nbits < 12) nbits++;
        }
        old = code;
    }
    return out.size() >= want;
}

}  // namespace

bool tiff_read(const char *filename, png_image &out) {
    Reader r;
    {
        FILE *f = fopen(filename, "rb");
        if (!f) return false;
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (n < 8) { fclose(f); return false; }
        r.d.resize((size_t)n);
        const bool got = fread(r.d.data(), 1, (size_t)n, f) == (size_t)n;
        fclose(f);
        if (!got) return false;
    }
    if (r.d[0] == 'I' && r.d[1] == 'I') r.be = false;
    else if (r.d[0] == 'M' && r.d[1] == 'M') r.be = true;
    else return false;
    if (r.u16(2) != 42) return false;                              // 43 would be BigTIFF
    const size_t ifd = r.u32(4);
    if (!r.ok(ifd, 2)) return false;
    const int nent = r.u16(ifd);
    if (!r.ok(ifd + 2, (size_t)nent * 12)) return false;
    uint32_t width = 0, height = 0, compression = 1, photometric = 1, spp = 1, rows_per_strip = 0xffffffffu, planar = 1, predictor = 1, fill_order = 1;
    std::vector<uint32_t> bits{1}, offsets, counts, fmt{1}, v;
    bool tiled = false;
    for (int i = 0; i < nent; i++) {
        const size_t e = ifd + 2 + (size_t)i * 12;
        const uint16_t tag = r.u16(e);
        if (tag == 322 || tag == 323 || tag == 324 || tag == 325) { tiled = true; continue; }
        if (!entry_values(r, e, v)) continue;                      // types this reader has no use for (ASCII, RATIONAL ...)
        switch (tag) {
        case 256: width = v[0]; break;
        case 257: height = v[0]; break;
        case 258: bits = v; break;
        case 259: compression = v[0]; break;
        case 262: photometric = v[0]; break;
        case 266: fill_order = v[0]; break;
        case 273: offsets = v; break;
        case 277: spp = v[0]; break;
        case 278: rows_per_strip = v[0]; break;
        case 279: counts = v; break;
        case 284: planar = v[0]; break;
        case 317: predictor = v[0]; break;
        case 339: fmt = v; break;
        default: break;
        }
    }
    if (tiled || width == 0 || height == 0 || width > 65535 || height > 65535 || (uint64_t)width * height > (1ull << 28)) return false;   // 268 Mpx: far beyond any camera, keeps a corrupt header from asking for tens of GB
    if (spp != 1 && spp != 3 && spp != 4) return false;
    if ((spp > 1 && planar != 1) || fill_order != 1 || fmt[0] != 1) return false;
    const uint32_t bps = bits[0];
    for (uint32_t b : bits) if (b != bps) return false;
    if (bps != 8 && bps != 16) return false;
    if (photometric > 2) return false;                             // 0 WhiteIsZero, 1 BlackIsZero, 2 RGB
    // grey images have one sample and a grey interpretation, colour images three (a fourth -- ExtraSamples, e.g. alpha -- is dropped) and RGB: anything
    // else (three samples called grey, one sample called RGB) would decode to a plausible-looking wrong image
    if (!((spp == 1 && photometric <= 1) || (spp >= 3 && photometric == 2))) return false;
    const bool deflate = compression == 8 || compression == 32946;
    if (compression != 1 && compression != 5 && compression != 32773 && !deflate) return false;
    if (predictor != 1 && !(predictor == 2 && (compression == 5 || deflate))) return false;
    if (offsets.empty() || offsets.size() != counts.size()) {
        if (offsets.size() == 1 && counts.empty() && compression == 1) counts.assign(1, (uint32_t)((size_t)width * height * spp * (bps / 8)));   // some writers omit it
        else return false;
    }
    if (rows_per_strip == 0) return false;
    if (rows_per_strip > height) rows_per_strip = height;
    const size_t row_bytes = (size_t)width * spp * (bps / 8);
    const int ch = spp == 1 ? 1 : 3;
    {   // the strips must lie in the file, and an uncompressed image must bring its bytes, before anything is allocated for it
        uint64_t have = 0;
        for (size_t i = 0; i < offsets.size(); i++) { if (!r.ok(offsets[i], counts[i])) return false; have += counts[i]; }
        if ((uint64_t)offsets.size() * rows_per_strip < height) return false;
        if (compression == 1 && have < (uint64_t)row_bytes * height) return false;
        if (have == 0) return false;
    }
    out.width = (int)width; out.height = (int)height; out.channels = ch; out.depth = (int)bps;
    out.samples.assign((size_t)width * height * ch, 0);
    const uint32_t maxv = bps == 16 ? 65535u : 255u;
    std::vector<unsigned char> strip;
    uint32_t y0 = 0;
    for (size_t s = 0; s < offsets.size() && y0 < height; s++, y0 += rows_per_strip) {
        const uint32_t rows = std::min(rows_per_strip, height - y0);
        const size_t want = row_bytes * rows;
        if (!r.ok(offsets[s], counts[s])) return false;
        const unsigned char *src = r.d.data() + offsets[s];
        strip.clear();
        if (compression == 1) { if (counts[s] < want) return false; strip.assign(src, src + want); }
        else if (compression == 5) { if (!unpack_lzw(src, counts[s], strip, want)) return false; }
        else if (deflate) {
            strip.resize(want);
            uLongf got = (uLongf)want;
            const int zrc = uncompress(strip.data(), &got, src, (uLong)counts[s]);
            if ((zrc != Z_OK && zrc != Z_BUF_ERROR) || got < want) return false;
        }
        else if (!unpack_bits(src, counts[s], strip, want)) return false;
        for (uint32_t yy = 0; yy < rows; yy++) {
            const unsigned char *row = strip.data() + (size_t)yy * row_bytes;
            std::vector<uint32_t> prev(spp, 0);
            for (uint32_t x = 0; x < width; x++)
                for (uint32_t k = 0; k < spp; k++) {
                    const size_t i = ((size_t)x * spp + k) * (bps / 8);
                    uint32_t val = bps == 8 ? row[i] : (r.be ? (uint32_t)((row[i] << 8) | row[i + 1]) : (uint32_t)(row[i] | (row[i + 1] << 8)));
                    if (predictor == 2) { val = (val + prev[k]) & maxv; prev[k] = val; }   // horizontal differencing, per sample
                    if (photometric == 0) val = maxv - val;
                    if ((int)k < ch) out.samples[((size_t)(y0 + yy) * width + x) * ch + k] = (uint16_t)val;
                }
        }
    }
    return y0 >= height;
}

bool tiff_write(const char *filename, const png_image &img) {
    if (img.width <= 0 || img.height <= 0 || (img.channels != 1 && img.channels != 3) || (img.depth != 8 && img.depth != 16)) return false;
    const uint64_t nbytes64 = (uint64_t)img.width * img.height * img.channels * (img.depth / 8);
    if (nbytes64 > 0xfff00000ull) return false;                    // classic TIFF has 32-bit offsets: the strip and the directory behind it must fit
    const uint32_t nbytes = (uint32_t)nbytes64;
    std::vector<unsigned char> d;
    auto p16 = [&](uint32_t v) { d.push_back((unsigned char)(v & 255)); d.push_back((unsigned char)(v >> 8)); };
    auto p32 = [&](uint32_t v) { p16(v & 0xffff); p16(v >> 16); };
    d.push_back('I'); d.push_back('I'); p16(42); p32(8 + nbytes + (nbytes & 1));      // the directory follows the pixel data
    for (size_t i = 0; i < (size_t)img.width * img.height * img.channels; i++) {
        if (img.depth == 8) d.push_back((unsigned char)img.samples[i]);
        else p16(img.samples[i]);
    }
    if (nbytes & 1) d.push_back(0);
    const uint32_t ifd = (uint32_t)d.size();
    const int nent = 10;
    const uint32_t bits_off = ifd + 2 + nent * 12 + 4;               // BitsPerSample of an RGB image does not fit the entry
    auto entry = [&](uint16_t tag, uint16_t type, uint32_t count, uint32_t value) { p16(tag); p16(type); p32(count); if (type == 3 && count == 1) { p16(value); p16(0); } else p32(value); };
    p16(nent);
    entry(256, 4, 1, (uint32_t)img.width);
    entry(257, 4, 1, (uint32_t)img.height);
    if (img.channels == 1) entry(258, 3, 1, (uint32_t)img.depth); else entry(258, 3, 3, bits_off);
    entry(259, 3, 1, 1);
    entry(262, 3, 1, img.channels == 1 ? 1 : 2);
    entry(273, 4, 1, 8);
    entry(277, 3, 1, (uint32_t)img.channels);
    entry(278, 4, 1, (uint32_t)img.height);
    entry(279, 4, 1, nbytes);
    entry(284, 3, 1, 1);
    p32(0);
    if (img.channels == 3) { p16((uint32_t)img.depth); p16((uint32_t)img.depth); p16((uint32_t)img.depth); }
    FILE *f = fopen(filename, "wb");
    if (!f) return false;
    const bool ok = fwrite(d.data(), 1, d.size(), f) == d.size();
    return fclose(f) == 0 && ok;
}
