// The reference's two C entry points of the path under their own names and prototypes, as a caller relinked against libslowflow_amd.so
// sees them: `sor_coupled` (epic_flow_extended/solver.h:11) and `variational` (epic_flow_extended/variational.h:34).  This TU does NOT
// include include/slowflow_amd.h: the declarations below are the reference headers' (image_t / color_image_t are the host mirror's,
// layout-identical to image.h:17-33).
#include <cstdio>
#include <fstream>
#include <string>

#include "image.h"

extern "C" {
typedef struct variational_params_s {       // variational.h:16-25
    float alpha, gamma, delta, sigma;
    int niter_outer, niter_inner, niter_solver;
    float sor_omega;
} variational_params_t;
void sor_coupled(image_t *du, image_t *dv, image_t *a11, image_t *a12, image_t *a22, image_t *b1, image_t *b2, image_t *dpsis_horiz, image_t *dpsis_vert,
                 const int iterations, const float omega);                                                      // solver.h:11
void variational(image_t *wx, image_t *wy, const color_image_t *im1, const color_image_t *im2, variational_params_t *params);   // variational.h:34
}

static bool rd(const std::string &path, float *dst, size_t n) {
    std::ifstream f(path.c_str(), std::ios::binary);
    f.read(reinterpret_cast<char *>(dst), (std::streamsize)(n * sizeof(float)));
    return (size_t)f.gcount() == n * sizeof(float);
}
static void wr(const std::string &path, const float *src, size_t n) {
    std::ofstream f(path.c_str(), std::ios::binary);
    f.write(reinterpret_cast<const char *>(src), (std::streamsize)(n * sizeof(float)));
}

int run_reference_symbols(const std::string &dir) {
    // ---- sor_coupled on a system written by the Python side: sym_sor.txt = "w h K omega", sym_sor_<name>.bin ----------------------
    {
        int w = 0, h = 0, K = 0;
        float omega = 0;
        std::ifstream m((dir + "/sym_sor.txt").c_str());
        if (!(m >> w >> h >> K >> omega)) { fprintf(stderr, "sym_sor.txt unreadable\n"); return 2; }
        const char *names[9] = {"du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv"};
        image_t *im[9];
        for (int i = 0; i < 9; i++) {
            im[i] = image_new(w, h);
            if (!rd(dir + "/sym_sor_" + names[i] + ".bin", im[i]->data, (size_t)im[i]->stride * h)) { fprintf(stderr, "%s unreadable\n", names[i]); return 2; }
        }
        sor_coupled(im[0], im[1], im[2], im[3], im[4], im[5], im[6], im[7], im[8], K, omega);
        for (int i = 0; i < 5; i++) wr(dir + "/sym_sor_out_" + names[i] + ".bin", im[i]->data, (size_t)im[i]->stride * h);
        for (int i = 0; i < 9; i++) image_delete(im[i]);
    }
    // ---- variational (two frames): sym_var.txt = "w h alpha gamma delta sigma outer inner solver omega" ------------------------------
    {
        int w = 0, h = 0;
        variational_params_t p;
        std::ifstream m((dir + "/sym_var.txt").c_str());
        if (!(m >> w >> h >> p.alpha >> p.gamma >> p.delta >> p.sigma >> p.niter_outer >> p.niter_inner >> p.niter_solver >> p.sor_omega)) { fprintf(stderr, "sym_var.txt unreadable\n"); return 2; }
        color_image_t *im1 = color_image_new(w, h), *im2 = color_image_new(w, h);
        image_t *wx = image_new(w, h), *wy = image_new(w, h);
        if (!rd(dir + "/sym_var_im1.bin", im1->c1, (size_t)3 * im1->stride * h) || !rd(dir + "/sym_var_im2.bin", im2->c1, (size_t)3 * im2->stride * h) ||
            !rd(dir + "/sym_var_wx.bin", wx->data, (size_t)wx->stride * h) || !rd(dir + "/sym_var_wy.bin", wy->data, (size_t)wy->stride * h)) { fprintf(stderr, "sym_var inputs unreadable\n"); return 2; }
        variational(wx, wy, im1, im2, &p);
        wr(dir + "/sym_var_out_wx.bin", wx->data, (size_t)wx->stride * h);
        wr(dir + "/sym_var_out_wy.bin", wy->data, (size_t)wy->stride * h);
        color_image_delete(im1); color_image_delete(im2); image_delete(wx); image_delete(wy);
    }
    return 0;
}
