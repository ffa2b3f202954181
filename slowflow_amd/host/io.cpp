// io.cpp -- see io.h
#include "io.h"
#include "png.h"
#include "tiff.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int writeFlowFile(const char *filename, const image_t *fx, const image_t *fy) {
    FILE *f = fopen(filename, "wb");
    if (!f) { fprintf(stderr, "Error while opening %s\n", filename); return -1; }
    const float tag = 202021.25f;
    const int w = fx->width, h = fx->height;
    fwrite(&tag, sizeof(float), 1, f);
    fwrite(&w, sizeof(int), 1, f);
    fwrite(&h, sizeof(int), 1, f);
    std::vector<float> row(2 * (size_t)w);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) { row[2 * x] = fx->data[(size_t)y * fx->stride + x]; row[2 * x + 1] = fy->data[(size_t)y * fy->stride + x]; }
        fwrite(row.data(), sizeof(float), row.size(), f);
    }
    fclose(f);
    return 0;
}

image_t **readFlowFile(const char *filename) {
    FILE *f = fopen(filename, "rb");
    if (!f) { fprintf(stderr, "readFlow() error: could not open file  %s\n", filename); return nullptr; }
    float tag = 0;
    int w = 0, h = 0;
    if (fread(&tag, 4, 1, f) != 1 || fread(&w, 4, 1, f) != 1 || fread(&h, 4, 1, f) != 1 || tag != 202021.25f || w <= 0 || h <= 0 || w > 65535 || h > 65535 ||
        (long long)w * h > (1ll << 28)) { fclose(f); return nullptr; }
    image_t **flow = (image_t **)malloc(2 * sizeof(image_t *));
    flow[0] = image_new(w, h);
    flow[1] = image_new(w, h);
    image_erase(flow[0]);
    image_erase(flow[1]);
    std::vector<float> row(2 * (size_t)w);
    for (int y = 0; y < h; y++) {
        if (fread(row.data(), sizeof(float), row.size(), f) != row.size()) break;
        for (int x = 0; x < w; x++) { flow[0]->data[(size_t)y * flow[0]->stride + x] = row[2 * x]; flow[1]->data[(size_t)y * flow[1]->stride + x] = row[2 * x + 1]; }
    }
    fclose(f);
    return flow;
}

static bool next_token(FILE *f, std::string &tok) {
    tok.clear();
    int c;
    for (;;) {
        c = fgetc(f);
        if (c == EOF) return false;
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (!isspace(c)) break;
    }
    while (c != EOF && !isspace(c)) { tok.push_back((char)c); c = fgetc(f); }
    return true;
}

color_image_t *color_image_load(const char *filename, int *maxval_out) {
    FILE *f = fopen(filename, "rb");
    if (!f) { fprintf(stderr, "could not open %s\n", filename); return nullptr; }
    unsigned char sig[4] = {0, 0, 0, 0};
    const bool got_sig = fread(sig, 1, 4, f) == 4;
    const bool is_png = got_sig && sig[0] == 0x89 && sig[1] == 'P' && sig[2] == 'N' && sig[3] == 'G';       // PNG (what the reference's Sintel sequences are stored as)
    const bool is_tiff = got_sig && ((sig[0] == 'I' && sig[1] == 'I' && sig[2] == 42 && sig[3] == 0) || (sig[0] == 'M' && sig[1] == 'M' && sig[2] == 0 && sig[3] == 42));   // TIFF (cfgs/slow_flow.cfg:5)
    if (is_png || is_tiff) {
        fclose(f);
        png_image png;
        if (is_png && !png_read(filename, png)) { fprintf(stderr, "%s: not a PNG this reader handles (non-interlaced, 8/16 bit)\n", filename); return nullptr; }
        if (is_tiff && !tiff_read(filename, png)) { fprintf(stderr, "%s: not a TIFF this reader handles (strips, 8/16 bit grey or RGB, none / LZW / PackBits)\n", filename); return nullptr; }
        color_image_t *pim = color_image_new(png.width, png.height);
        color_image_erase(pim);
        for (int y = 0; y < png.height; y++)
            for (int x = 0; x < png.width; x++)
                for (int k = 0; k < 3; k++)          // file order R,G,B = the order after the reference's BGR2RGB (slow_flow.cpp:527)
                    (k == 0 ? pim->c1 : k == 1 ? pim->c2 : pim->c3)[(size_t)y * pim->stride + x] =
                        (float)png.samples[((size_t)y * png.width + x) * png.channels + (png.channels == 3 ? k : 0)];
        if (maxval_out) *maxval_out = png.depth == 16 ? 65535 : 255;
        return pim;
    }
    rewind(f);
    std::string magic, t;
    if (!next_token(f, magic)) { fclose(f); return nullptr; }
    color_image_t *im = nullptr;
    if (magic == "P5" || magic == "P6") {
        const int ch = magic == "P6" ? 3 : 1;
        int w, h, maxv;
        if (!next_token(f, t)) { fclose(f); return nullptr; }
        w = atoi(t.c_str());
        if (!next_token(f, t)) { fclose(f); return nullptr; }
        h = atoi(t.c_str());
        if (!next_token(f, t)) { fclose(f); return nullptr; }
        maxv = atoi(t.c_str());                     // next_token consumed the single whitespace after maxval
        if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (long long)w * h > (1ll << 28) || maxv <= 0 || maxv > 65535) { fclose(f); return nullptr; }
        const int bps = maxv > 255 ? 2 : 1;
        std::vector<unsigned char> row((size_t)w * ch * bps);
        im = color_image_new(w, h);
        color_image_erase(im);
        for (int y = 0; y < h; y++) {
            if (fread(row.data(), 1, row.size(), f) != row.size()) { color_image_delete(im); fclose(f); return nullptr; }
            for (int x = 0; x < w; x++)
                for (int k = 0; k < 3; k++) {
                    const size_t i = ((size_t)x * ch + (ch == 3 ? k : 0)) * bps;
                    const float v = bps == 2 ? (float)((row[i] << 8) | row[i + 1]) : (float)row[i];
                    (k == 0 ? im->c1 : k == 1 ? im->c2 : im->c3)[(size_t)y * im->stride + x] = v;
                }
        }
        if (maxval_out) *maxval_out = maxv > 255 ? 65535 : 255;
    } else if (magic == "PF" || magic == "Pf") {
        const int ch = magic == "PF" ? 3 : 1;
        int w, h;
        if (!next_token(f, t)) { fclose(f); return nullptr; }
        w = atoi(t.c_str());
        if (!next_token(f, t)) { fclose(f); return nullptr; }
        h = atoi(t.c_str());
        if (!next_token(f, t)) { fclose(f); return nullptr; }
        const double scale = atof(t.c_str());
        if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (long long)w * h > (1ll << 28) || scale >= 0) { fclose(f); return nullptr; }   // little-endian files only (negative scale)
        std::vector<float> row((size_t)w * ch);
        im = color_image_new(w, h);
        color_image_erase(im);
        for (int y = h - 1; y >= 0; y--) {          // pfm rows are stored bottom to top
            if (fread(row.data(), sizeof(float), row.size(), f) != row.size()) { color_image_delete(im); fclose(f); return nullptr; }
            for (int x = 0; x < w; x++)
                for (int k = 0; k < 3; k++) (k == 0 ? im->c1 : k == 1 ? im->c2 : im->c3)[(size_t)y * im->stride + x] = row[(size_t)x * ch + (ch == 3 ? k : 0)];
        }
        if (maxval_out) *maxval_out = 1;
    }
    fclose(f);
    return im;
}

int writePGM(const char *filename, const image_t *img, float offset, float scale) {
    FILE *f = fopen(filename, "wb");
    if (!f) return -1;
    fprintf(f, "P5\n%d %d\n255\n", img->width, img->height);
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            float v = scale * (img->data[(size_t)y * img->stride + x] + offset);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            fputc((int)(v + 0.5f), f);
        }
    return fclose(f) == 0 ? 0 : -1;
}
