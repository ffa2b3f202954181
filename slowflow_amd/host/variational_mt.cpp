// variational_mt.cpp -- see variational_mt.h
#include "variational_mt.h"

#include "png.h"

#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <vector>

sfa_params sfa_params_from_cfg(ParameterList &params, bool one_direction) {
    sfa_params p;
    sfa_params_default(&p);
    p.S = params.parameter<int>("slow_flow_S");
    p.one_direction = one_direction ? 1 : 0;
    if (params.exists("slow_flow_method") && params.parameter("slow_flow_method") == "forward") p.one_direction = 1;   // variational_mt.cpp:551
    p.smoothing = params.parameter<int>("slow_flow_smoothing", "0");
    p.dataterm_norm = params.parameter<bool>("slow_flow_dataterm", "1");
    p.niter_alter = params.parameter<int>("slow_flow_niter_alter", "1");
    p.niter_outer = params.parameter<int>("slow_flow_niter_outer");
    p.niter_inner = params.parameter<int>("slow_flow_niter_inner");
    p.niter_solver = params.parameter<int>("slow_flow_niter_solver");
    p.thres_outer = params.parameter<float>("slow_flow_thres_outer");
    p.thres_inner = params.parameter<float>("slow_flow_thres_inner");
    p.sor_omega = params.parameter<float>("slow_flow_sor_omega");
    p.alpha = params.parameter<float>("slow_flow_alpha");
    p.gamma = params.parameter<float>("slow_flow_gamma");
    p.delta = params.parameter<float>("slow_flow_delta");
    p.robust_color.id = params.parameter<int>("slow_flow_robust_color");
    p.robust_color.eps = params.parameter<float>("slow_flow_robust_color_eps");
    p.robust_color.trunc = params.parameter<float>("slow_flow_robust_color_truncation");
    if (params.exists("slow_flow_robust_grad")) {                                    // variational_mt.cpp:556-557
        p.robust_grad.id = params.parameter<int>("slow_flow_robust_grad");
        p.robust_grad.eps = params.parameter<float>("slow_flow_robust_grad_eps");
        p.robust_grad.trunc = params.parameter<float>("slow_flow_robust_grad_truncation");
    } else p.robust_grad = p.robust_color;
    p.robust_reg.id = params.parameter<int>("slow_flow_robust_reg");
    p.robust_reg.eps = params.parameter<float>("slow_flow_robust_reg_eps");
    p.robust_reg.trunc = params.parameter<float>("slow_flow_robust_reg_truncation");
    const int ref = p.S - 1;
    for (int a = 0; a < ref && a < SFA_MAX_REF; a++) {                               // variational_mt.cpp:561-568
        std::stringstream so, sr;
        so << "slow_flow_omega_" << a;
        sr << "slow_flow_rho_" << a;
        p.omega[a] = params.parameter<float>(so.str(), "1.0");
        p.rho[a] = params.parameter<float>(sr.str(), "1.0");
    }
    p.hbit = params.parameter<bool>("16bit", "0");
    const char *avg[3] = {"slow_flow_img_norm_avg_1", "slow_flow_img_norm_avg_2", "slow_flow_img_norm_avg_3"};
    const char *std_[3] = {"slow_flow_img_norm_std_1", "slow_flow_img_norm_std_2", "slow_flow_img_norm_std_3"};
    for (int k = 0; k < 3; k++) {                                                    // variational_mt.cpp:250-251
        p.norm_avg[k] = (float)params.parameter<double>(avg[k], "0");
        p.norm_std[k] = (float)params.parameter<double>(std_[k], "1");
    }
    p.occlusion_reasoning = params.parameter<bool>("slow_flow_occlusion_reasoning", "0");
    p.occlusion_penalty = params.parameter<float>("slow_flow_occlusion_penalty", "1.0");
    p.occlusion_alpha = params.parameter<float>("slow_flow_occlusion_alpha", "0.5");
    p.niter_graphc = params.parameter<int>("slow_flow_niter_graphc", "10");
    // additive key: "red_black" selects the labelled two-colour mode (a different algorithm: does not reproduce the reference); anything else the reference order
    p.sor_order = (params.exists("slow_flow_sor_order") && params.parameter("slow_flow_sor_order") == "red_black") ? 1 : 0;
    p.layers = params.parameter<int>("slow_flow_layers");
    p.p_scale = params.parameter<float>("slow_flow_p_scale");
    p.presmooth_sigma = params.parameter<float>("sigma", "0") > 0 ? params.parameter<float>("slow_flow_sigma") : 0.0f;   // :590-591
    return p;
}

static sfa_ctx *make_ctx(int device) {
    sfa_ctx *c = nullptr;
    if (sfa_ctx_create(device, &c) != SFA_OK) throw std::runtime_error(std::string("slowflow_amd: ") + sfa_last_error(nullptr));
    return c;
}

void normalize(color_image_t **seq, u_int32_t F, ParameterList &params) {
    if (F == 0) return;
    sfa_ctx *c = make_ctx(params.parameter<int>("gpu_device", "0"));
    std::vector<float *> frames(F);
    for (u_int32_t f = 0; f < F; f++) frames[f] = seq[f]->c1;
    double avg[3], sd[3];
    const int rc = sfa_normalize(c, frames.data(), (int)F, seq[0]->width, seq[0]->height, seq[0]->stride, avg, sd);
    const std::string err = rc == SFA_OK ? "" : sfa_last_error(c);
    sfa_ctx_destroy(c);
    if (rc != SFA_OK) throw std::runtime_error("slowflow_amd normalize: " + err);
    publish_normalization(params, avg, sd);
}

void publish_normalization(ParameterList &params, const double avg[3], const double sd[3]) {
    const char *ka[3] = {"slow_flow_img_norm_avg_1", "slow_flow_img_norm_avg_2", "slow_flow_img_norm_avg_3"};
    const char *ks[3] = {"slow_flow_img_norm_std_1", "slow_flow_img_norm_std_2", "slow_flow_img_norm_std_3"};
    if (params.verbosity(VER_CMD))
        for (int k = 0; k < 3; k++) std::cout << "Intensities normalized by (I - " << avg[k] << ") / " << sd[k] << std::endl;
    for (int k = 0; k < 3; k++) {                                                    // stringstream <<: 6 significant digits (:71-84)
        std::stringstream a, s;
        a << avg[k];
        s << sd[k];
        params.insert(ka[k], a.str(), true);
        params.insert(ks[k], s.str(), true);
    }
}

Variational_MT::Variational_MT() : one_direction(false), ctx(nullptr), ctx_device(-1), device_override(-1), channel_w(nullptr), occlusions(nullptr) {}

Variational_MT::~Variational_MT() {
    if (occlusions) image_delete(occlusions);
    if (ctx) sfa_ctx_destroy(ctx);
}

Point2f Variational_MT::variational(image_t *wx, image_t *wy, color_image_t *const *im, ParameterList &params) {
    params.insert("final", "0", true);                                               // variational_mt.cpp:527
    sfa_params p = sfa_params_from_cfg(params, one_direction);
    one_direction = p.one_direction != 0;
    const int F = 2 * (p.S - 1) + 1;
    const int device = device_override >= 0 ? device_override : params.parameter<int>("gpu_device", "0");
    if (!ctx || ctx_device != device) {
        if (ctx) sfa_ctx_destroy(ctx);
        ctx = make_ctx(device);
        ctx_device = device;
    }
    std::vector<const float *> frames(F);
    for (int f = 0; f < F; f++) frames[f] = im[f]->c1;
    const float *chw[3] = {nullptr, nullptr, nullptr};
    if (channel_w) { chw[0] = channel_w->c1; chw[1] = channel_w->c2; chw[2] = channel_w->c3; }
    if (occlusions) image_delete(occlusions);                                        // variational_mt.cpp:203-204
    occlusions = image_new(wx->width, wx->height);
    image_erase(occlusions);
    float change[2] = {0, 0};
    // the resident-job form of sfa_variational: the same calls, plus the labels of every alternation when the caller asks for them
    const bool occ_files = p.occlusion_reasoning && params.exists("slow_flow_occlusions_output");       // variational_mt.cpp:275
    sfa_job *job = nullptr;
    int rc = sfa_job_create(ctx, &p, wx->width, wx->height, 1, &job);
    if (rc == SFA_OK && occ_files) rc = sfa_job_keep_alternation_occlusions(job, 1);
    if (rc == SFA_OK) rc = sfa_job_upload(job, 0, frames.data(), F, wx->data, wy->data, wx->stride, channel_w ? chw : nullptr);
    // the per-iteration "avg change" lines of variational_mt.cpp:404-405, 431-432 are printed by the library when asked to (a host round trip per iteration)
    (void)sfa_ctx_set_verbose(ctx, params.verbosity(VER_CMD) ? 1 : 0);
    if (rc == SFA_OK) rc = sfa_job_run(job);
    if (rc == SFA_OK) rc = sfa_job_download(job, 0, wx->data, wy->data, wx->stride, change);
    if (rc == SFA_OK) rc = sfa_job_download_occlusions(job, 0, occlusions->data, occlusions->stride);
    for (int a = 1; a < p.niter_alter && rc == SFA_OK && occ_files; a++) {                              // :275-285: <prefix><alter>.png
        image_t *o = image_new(wx->width, wx->height);
        image_erase(o);
        rc = sfa_job_download_alternation_occlusions(job, 0, a, o->data, o->stride);
        if (rc == SFA_OK) {
            png_image im;
            im.width = o->width; im.height = o->height; im.channels = 1; im.depth = 8;
            im.samples.resize((size_t)o->width * o->height);
            for (int y = 0; y < o->height; y++)
                for (int x = 0; x < o->width; x++)                                                      // (occ + 1) * 0.5 -> 8 bit, scale 255 (:277-279)
                    im.samples[(size_t)y * o->width + x] = (uint16_t)((o->data[(size_t)y * o->stride + x] + 1) * 0.5f * 255.0f + 0.5f);
            std::stringstream occF;
            occF << params.parameter("slow_flow_occlusions_output") << a << ".png";
            png_write(occF.str().c_str(), im);
        }
        image_delete(o);
    }
    const std::string err = rc == SFA_OK ? "" : sfa_last_error(ctx);
    if (job) sfa_job_destroy(job);
    if (rc == SFA_ERR_REF_FRAME) throw std::logic_error("Frame compared to reference frame is the reference frame itself!");   // aux:419-421
    if (rc != SFA_OK) throw std::runtime_error("slowflow_amd variational: " + err);
    params.setParameter<int>("final", 0);                                            // variational_mt.cpp:764
    return Point2f(change[0], change[1]);
}
