// ref_bind_epic.cpp -- binding TU of OURS around the reference's own epic() (epic_flow_extended/epic.cpp:147), compiled by oracle/Makefile together with the
// reference's unmodified epic.cpp / epic_aux.cpp / image.c where they lie under /root/reference.  TEST INFRASTRUCTURE ONLY: it exists so that
// slowflow_amd/host/epic.cpp can be compared with the reference's output.  Nothing of the reference is copied: this file only includes its headers and calls it.
#include <string.h>

#include "epic.h"           // the reference's (-I$(REF)/epic_flow_extended)

extern "C" {
// images as plain planes: lab3 = 3 planes of h*stride floats (L, a, b as rgb_to_lab returns them); matches = nmatch rows x1 y1 x2 y2; edges = w*h floats (modified: euc)
int ref_epic(float *flowx, float *flowy, int w, int h, int stride, float *lab3, float *matches, int nmatch, float *edges, const char *method, float saliency_th,
             int pref_nn, float pref_th, int nn, float coef_kernel, float euc) {
    image_t fx = {w, h, stride, flowx}, fy = {w, h, stride, flowy};
    color_image_t im = {w, h, stride, lab3, lab3 + (size_t)stride * h, lab3 + 2 * (size_t)stride * h};
    float_image m = {matches, 4, nmatch};
    float_image e = {edges, w, h};
    epic_params_t p;
    epic_params_default(&p);
    strncpy(p.method, method, sizeof p.method - 1);
    p.saliency_th = saliency_th; p.pref_nn = pref_nn; p.pref_th = pref_th; p.nn = nn; p.coef_kernel = coef_kernel; p.euc = euc; p.verbose = 0;
    epic(&fx, &fy, &im, &m, &e, &p, 1);
    return 0;
}
}
