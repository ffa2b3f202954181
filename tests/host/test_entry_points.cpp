// GPU test of the entry points north_star names, driven the way the reference's own caller drives them (slow_flow.cpp:673, :865-888,
// :1018-1023): frames in color_image_new containers -> normalize(seq, F, params) -> Variational_MT (forward: setChannelWeights; backward:
// none) ::variational(wx, wy, im, params) -> getOcclusions().  Inputs come from files written by tests/test_host.py, results go back as raw
// planes; the Python side compares them with the C-ABI binding's results bit for bit.  The literal C symbols `sor_coupled` (solver.h:11)
// and `variational` (variational.h:34) are called from test_entry_symbols.cpp with the reference's own prototypes.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "image.h"
#include "parameter_list.h"
#include "variational_mt.h"

int run_reference_symbols(const std::string &dir);            // test_entry_symbols.cpp

static bool read_floats(const std::string &path, float *dst, size_t n) {
    std::ifstream f(path.c_str(), std::ios::binary);
    f.read(reinterpret_cast<char *>(dst), (std::streamsize)(n * sizeof(float)));
    return (size_t)f.gcount() == n * sizeof(float);
}
static void write_plane(const std::string &path, const image_t *im) {
    std::ofstream f(path.c_str(), std::ios::binary);
    f.write(reinterpret_cast<const char *>(im->data), (std::streamsize)((size_t)im->stride * im->height * sizeof(float)));
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: test_entry_points dir width height nframes\n"); return 2; }
    const std::string dir = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]), n = atoi(argv[4]);
    ParameterList params(dir + "/ep.cfg");
    const int S = params.parameter<int>("slow_flow_S"), F = 2 * (S - 1) + 1;
    if (n < F) return 2;
    std::vector<color_image_t *> seq(n), seq_back(n);
    for (int f = 0; f < n; f++) {
        seq[f] = color_image_new(w, h);
        if (!read_floats(dir + "/ep_frame_" + std::to_string(f) + ".bin", seq[f]->c1, (size_t)3 * seq[f]->stride * h)) { fprintf(stderr, "frame %d unreadable\n", f); return 2; }
        seq_back[n - 1 - f] = seq[f];                                                  // slow_flow.cpp:590-591
    }
    color_image_t *channel_weights = color_image_new(w, h);
    if (!read_floats(dir + "/ep_chw.bin", channel_weights->c1, (size_t)3 * channel_weights->stride * h)) return 2;

    normalize(&seq[0], (u_int32_t)n, params);                                          // slow_flow.cpp:673
    {
        std::ofstream cfg((dir + "/ep_after_normalize.cfg").c_str());                 // the published statistics (variational_mt.cpp:71-84)
        cfg << params;
    }
    for (int dir_pass = 0; dir_pass < 2; dir_pass++) {
        ParameterList thread_params(params);                                          // slow_flow.cpp:708
        image_t *wx = image_new(w, h), *wy = image_new(w, h);
        image_erase(wx); image_erase(wy);                                             // :865-868
        Variational_MT minimizer;
        if (dir_pass == 0) minimizer.setChannelWeights(channel_weights);              // forward only (:876 vs :1018)
        color_image_t *const *im = dir_pass == 0 ? &seq[0] : &seq_back[n - F];       // the window of jet 0 (:721-724)
        const Point2f chg = minimizer.variational(wx, wy, im, thread_params);         // :888 / :1023
        const std::string tag = dir_pass == 0 ? "fwd" : "bwd";
        write_plane(dir + "/ep_" + tag + "_wx.bin", wx);
        write_plane(dir + "/ep_" + tag + "_wy.bin", wy);
        write_plane(dir + "/ep_" + tag + "_occ.bin", minimizer.getOcclusions());
        std::ofstream c((dir + "/ep_" + tag + "_change.txt").c_str());
        c.precision(9);
        c << chg.x << " " << chg.y << "\n";
        if (thread_params.parameter<int>("final", "-1") != 0) { fprintf(stderr, "`final` was not written back\n"); return 1; }   // variational_mt.cpp:527,764
        image_delete(wx); image_delete(wy);
    }
    for (int f = 0; f < n; f++) color_image_delete(seq[f]);
    color_image_delete(channel_weights);
    const int rc = run_reference_symbols(dir);
    if (rc == 0) printf("entry points OK\n");
    return rc;
}
