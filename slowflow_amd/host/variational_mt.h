// variational_mt.h -- drop-in for the reference's Variational_MT (epic_flow_extended/variational_mt.h:21-71): same
// class name, entry point, argument meaning and side effects, with the whole coarse-to-fine refinement executed on
// an MI355X through the C-ABI of include/slowflow_amd.h.  There is no CPU path: without a usable GPU variational()
// throws std::runtime_error.
#ifndef SLOWFLOW_AMD_HOST_VARIATIONAL_MT_H
#define SLOWFLOW_AMD_HOST_VARIATIONAL_MT_H

#include <stdint.h>
#include <sys/types.h>

#include "image.h"
#include "parameter_list.h"
#include "../../include/slowflow_amd.h"

/* standardize an image sequence to zero mean and std 255 per channel, in place; publishes the six statistics as
 * slow_flow_img_norm_{avg,std}_{1,2,3} with 6 significant digits (variational_mt.cpp:17-85).  Runs on the GPU of
 * params["gpu_device"] (default 0). */
void normalize(color_image_t **seq, u_int32_t F, ParameterList &params);

/* what normalize() does with its six statistics: publish them as slow_flow_img_norm_{avg,std}_{1,2,3}, 6 significant digits (variational_mt.cpp:71-84);
 * for callers that normalise frames already resident on a GPU (sfa_sequence_normalize) */
void publish_normalization(ParameterList &params, const double avg[3], const double std_dev[3]);

/* fill the C-ABI parameter block from the cfg keys exactly as Variational_MT::variational / compute_one_level read
 * them (variational_mt.cpp:173-192, 533-568), including defaults and the grad->color fallback */
sfa_params sfa_params_from_cfg(ParameterList &params, bool one_direction);

class Variational_MT {
public:
    Variational_MT();
    ~Variational_MT();
    Variational_MT(const Variational_MT &) = delete;
    Variational_MT &operator=(const Variational_MT &) = delete;

    /* Compute a refinement of the optical flow (wx and wy are modified in place) over the 2*(S-1)+1 frames im[] whose
     * middle one is the reference frame.  Returns the mean absolute flow change of the last outer iteration. */
    Point2f variational(image_t *wx, image_t *wy, color_image_t *const *im, ParameterList &params);

    void setChannelWeights(color_image_t *weights) { channel_w = weights; }
    image_t *getOcclusions() { return occlusions; }

    /* additive: which GPU this object uses (default: params["gpu_device"], else 0) */
    void setDevice(int device) { device_override = device; }

    bool one_direction;

private:
    sfa_ctx *ctx;
    int ctx_device;
    int device_override;
    color_image_t *channel_w;
    image_t *occlusions;
};

#endif
