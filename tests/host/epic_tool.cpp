// Drives slowflow_amd/host/epic.cpp on inputs written by tests/test_epic.py: <dir>/epic_rgb.bin (3 planes of h*stride floats, 0..255), epic_matches.txt
// (x1 y1 x2 y2 per line, further columns ignored), epic_edges.bin (w*h floats).  Writes epic_lab.bin, epic_fx.bin, epic_fy.bin and -- with `gpu` -- epic_sal.bin.
//   epic_tool dir w h method saliency_th pref_nn pref_th nn coef_kernel euc [gpu]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "epic.h"

static void wr(const std::string &p, const float *d, size_t n) { std::ofstream f(p.c_str(), std::ios::binary); f.write(reinterpret_cast<const char *>(d), (std::streamsize)(n * sizeof(float))); }

int main(int argc, char **argv) {
    if (argc < 11) { fprintf(stderr, "usage: epic_tool dir w h method saliency_th pref_nn pref_th nn coef_kernel euc [gpu]\n"); return 2; }
    const std::string dir = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    epic_params_t p;
    epic_params_default(&p);
    if (strcmp(p.method, "LA") != 0 || p.nn != 100 || p.pref_nn != 25) { fprintf(stderr, "defaults differ from epic.cpp:127-136\n"); return 1; }
    strncpy(p.method, argv[4], sizeof p.method - 1);
    p.saliency_th = (float)atof(argv[5]); p.pref_nn = atoi(argv[6]); p.pref_th = (float)atof(argv[7]); p.nn = atoi(argv[8]); p.coef_kernel = (float)atof(argv[9]);
    p.euc = (float)atof(argv[10]);
    const bool gpu = argc > 11 && !strcmp(argv[11], "gpu");
    color_image_t *rgb = color_image_new(w, h);
    {
        std::ifstream f((dir + "/epic_rgb.bin").c_str(), std::ios::binary);
        f.read(reinterpret_cast<char *>(rgb->c1), (std::streamsize)((size_t)3 * rgb->stride * h * sizeof(float)));
        if (!f) { fprintf(stderr, "epic_rgb.bin unreadable\n"); return 2; }
    }
    color_image_t *lab = rgb_to_lab(rgb);
    wr(dir + "/epic_lab.bin", lab->c1, (size_t)3 * lab->stride * h);
    epic_matches m;
    epic_edges e;
    if (!read_matches((dir + "/epic_matches.txt").c_str(), m) || !read_edges((dir + "/epic_edges.bin").c_str(), w, h, e)) { fprintf(stderr, "matches / edges unreadable\n"); return 2; }
    sfa_ctx *ctx = nullptr;
    if (gpu && sfa_ctx_create(0, &ctx) != SFA_OK) { fprintf(stderr, "%s\n", sfa_last_error(nullptr)); return 3; }
    if (gpu) {
        image_t *s = saliency(ctx, lab, 0.8f, 1.0f);
        if (!s) { fprintf(stderr, "saliency: %s\n", sfa_last_error(ctx)); return 3; }
        wr(dir + "/epic_sal.bin", s->data, (size_t)s->stride * h);
        image_delete(s);
    }
    image_t *fx = image_new(w, h), *fy = image_new(w, h);
    image_erase(fx); image_erase(fy);
    const int rc = epic(ctx, fx, fy, lab, m, e, &p);
    if (rc != 0) { fprintf(stderr, "epic -> %d\n", rc); return 4; }
    wr(dir + "/epic_fx.bin", fx->data, (size_t)fx->stride * h);
    wr(dir + "/epic_fy.bin", fy->data, (size_t)fy->stride * h);
    printf("epic ok: %d matches\n", m.count());
    if (ctx) sfa_ctx_destroy(ctx);
    return 0;
}
