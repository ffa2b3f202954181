// epic.h -- the caller on the input side of the path: EpicFlow's edge-preserving sparse-to-dense interpolation of matches, which the reference
// uses to initialise (wx, wy) when `deep_matching 1` (slow_flow.cpp:801-863 -> epic(), epic_flow_extended/epic.cpp:147, epic_aux.cpp).
// From-scratch implementation of the same algorithm:
//   matches -> clamp to the image (epic.cpp:15-28) -> saliency filter (:59-74, image.c:729-791) -> consistency filter (:77-123)
//   -> geodesic distance transform of all seeds over the edge-cost map with label propagation (epic_aux.cpp:92-199)
//   -> neighbourhood graph of the label regions (:226-283) -> k nearest seeds of every seed by Dijkstra on that graph (:44-87)
//   -> kernel exp(-coef * distance) -> locally-weighted affine (LA, :430-497) or Nadaraya-Watson (NW, :386-424) model per seed -> dense field.
// The image-side filtering (Gaussian smoothing, 3-tap derivatives of the saliency) runs on the GPU through the path's own, reference-pinned operators
// (sfa_gaussian_presmooth, sfa_convolve); everything else is host code like the reference's.  What is NOT here: producing the matches and the edge map --
// DeepMatching and the SED edge detector are third-party programs the reference starts with system() (slow_flow.cpp:744-790).
#ifndef SLOWFLOW_AMD_HOST_EPIC_H
#define SLOWFLOW_AMD_HOST_EPIC_H

#include <vector>

#include "../../include/slowflow_amd.h"
#include "image.h"

/* epic_params_t (epic.h:6-15) with the reference's defaults (epic.cpp:127-136) */
struct epic_params_t {
    char method[20];      /* "LA" locally-weighted affine, "NW" Nadaraya-Watson */
    float saliency_th;    /* matches from pixels below this saliency are dropped (0: off) */
    int pref_nn;          /* neighbours of the consistency check (0: off) */
    float pref_th;        /* its threshold in pixels */
    int nn;               /* neighbours of the interpolation */
    float coef_kernel;    /* exp(-coef * geodesic distance) */
    float euc;            /* constant added to the edge cost */
    int verbose;
};
void epic_params_default(epic_params_t *params);

/* rows of x1 y1 x2 y2 (read_matches, io.c:23-47: further columns of a line are ignored) */
struct epic_matches { std::vector<float> m; int count() const { return (int)(m.size() / 4); } };
/* width x height floats, row-major (read_edges, io.c:14-20) */
struct epic_edges { std::vector<float> cost; int width = 0, height = 0; };

bool read_matches(const char *filename, epic_matches &out);
bool read_edges(const char *filename, int width, int height, epic_edges &out);

/* image.c:694-726: L*a*b* with the reference's attenuation of unreliable colours; returns a new image */
color_image_t *rgb_to_lab(const color_image_t *im);
/* image.c:729-791: smallest eigenvalue of the smoothed autocorrelation matrix; filters on the GPU of ctx; NULL on a GPU error */
image_t *saliency(sfa_ctx *ctx, const color_image_t *im, float sigma_image, float sigma_matrix);

/* epic() (epic.cpp:147): flowx / flowy (size of im) <- dense interpolation of the matches; `edges` is modified (euc is added) as in the reference.
 * 0 on success, < 0 on a GPU error (sfa_last_error(ctx)), > 0: no usable match. */
int epic(sfa_ctx *ctx, image_t *flowx, image_t *flowy, const color_image_t *im_lab, const epic_matches &matches, epic_edges &edges, const epic_params_t *params);

/* the pieces, exposed for the tests ------------------------------------------------------------------------------------------------------ */
struct epic_nn { std::vector<int> index; std::vector<float> dist; int nn = 0; };       /* per seed: nn neighbours (-1 / huge when fewer exist) */
/* geodesic distance transform + labels + k nearest seeds of every seed; labels: closest seed of every pixel (width*height) */
void epic_nearest_seeds(const std::vector<int> &seeds_xy, const epic_edges &cost, int nn, epic_nn &out, std::vector<int> &labels);
/* least-squares affine model per seed (6 floats: flow-free mapping x' = a0 x + a1 y + a2, y' = a3 x + a4 y + a5) */
void epic_fit_localaffine(std::vector<float> &affine, const epic_nn &nnw, const std::vector<int> &seeds_xy, const std::vector<float> &vects);

#endif
