// slow_flow.cpp -- the slow_flow driver of the drop-in: same cfg keys, command line and outputs as the reference's
// slow_flow.cpp (usage :58-62, defaults :64-128, frame indexing :411-465, jet loop :706-1047, .flo outputs :908-911,
// :1027-1030, resume :794,958, config.cfg :684-688) for the path this repository implements:
//   frames -> normalize -> per jet: forward / backward Variational_MT::variational -> flow * steps -> .flo
// The jets of a sequence are independent; they are sharded over the node's GPUs (one host thread per GPU) and each
// GPU refines `gpu_batch` frame windows in lockstep.  Out of scope here (and rejected with a message): the third-party
// demosaicers (raw_demosaicing 1, 2), DeepMatching/EpicFlow initialisation, adaptive frame rates -- they
// live in third-party code (OpenCV, MATLAB SED, DeepMatching) outside the path.
//
// New, additive keys: gpus (default: all visible), gpu_batch (windows refined in lockstep per job, default 32), gpu_streams (default 2), gpu_device (first device, default 0).
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "flow_vis.h"
#include "image.h"
#include "ingest.h"
#include "io.h"
#include "parameter_list.h"
#include "variational_mt.h"

using std::string;

static void usage() {
    printf("usage:\n    ./slow_flow [cfg] -overwrite -resume -deep_settings [settings] -threads -fr [select one specific adaptive frame rate] "
           "-jet [select one specific high speed flow]\n\n");
}

static void setDefault(ParameterList &p) {        // slow_flow.cpp:64-128
    const char *kv[][2] = {
        {"verbose", "0"}, {"threads", "1"}, {"16bit", "1"}, {"raw", "1"}, {"raw_weight", "1"}, {"raw_demosaicing", "1"}, {"raw_red_loc", "1,0"},
        {"Jets", "1"}, {"adaptive", "1"}, {"max_fps", "200"}, {"ref_fps", "20"}, {"scale", "1.0f"}, {"sigma", "0.0f"}, {"deep_matching", "1"},
        {"dm_scale", "1.0f"}, {"slow_flow_method", "symmetric"}, {"slow_flow_S", "2"}, {"slow_flow_dataterm", "1"}, {"slow_flow_smoothing", "1"},
        {"slow_flow_alpha", "4.0f"}, {"slow_flow_gamma", "6.0f"}, {"slow_flow_delta", "1.0f"}, {"slow_flow_rho_0", "1"}, {"slow_flow_rho_1", "1"},
        {"slow_flow_omega_0", "0"}, {"slow_flow_omega_1", "2"}, {"slow_flow_layers", "1"}, {"slow_flow_p_scale", "0.9f"}, {"slow_flow_niter_alter", "10"},
        {"slow_flow_niter_graphc", "10"}, {"slow_flow_niter_outer", "10"}, {"slow_flow_thres_outer", "1e-5"}, {"slow_flow_niter_inner", "1"},
        {"slow_flow_thres_inner", "1e-5"}, {"slow_flow_niter_solver", "30"}, {"slow_flow_sor_omega", "1.9f"}, {"slow_flow_occlusion_reasoning", "1"},
        {"slow_flow_occlusion_penalty", "0.1"}, {"slow_flow_occlusion_alpha", "0.1"}, {"slow_flow_output_occlusions", "1"}, {"slow_flow_robust_color", "1"},
        {"slow_flow_robust_color_eps", "0.001"}, {"slow_flow_robust_color_truncation", "0.5"}, {"slow_flow_robust_reg", "1"},
        {"slow_flow_robust_reg_eps", "0.001"}, {"slow_flow_robust_reg_truncation", "0.5"}};
    for (auto &e : kv) p.insert(e[0], e[1], true);
}

static bool file_exists(const string &f) { return access(f.c_str(), F_OK) != -1; }
static void mkdirs(const string &path) {
    string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty()) mkdir(cur.c_str(), 0777);
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}
static string fmt1(const string &format, int a) { char b[1024]; snprintf(b, sizeof b, format.c_str(), a); return b; }
static string fmt2(const string &format, int a, int c) { char b[1024]; snprintf(b, sizeof b, format.c_str(), a, c); return b; }

struct Window {
    unsigned jet; bool backward; string out; double seconds; int gpu; double epe, aae;   // epe/aae < 0: no ground truth
    Window(unsigned j, bool b, const string &o) : jet(j), backward(b), out(o), seconds(0), gpu(-1), epe(-1), aae(-1) {}
};

int main(int argc, char **argv) {
    if (argc < 2) { usage(); return 1; }
    ParameterList params;
    setDefault(params);
    if (argv[1][0] != '-' && file_exists(argv[1])) params.read(argv[1]);
    else { std::cerr << "Couldn't find " << argv[1] << "!" << std::endl; return -1; }

    bool overwrite_output = false, resume_frame = false;
    int selected_jet = -1;
    for (int i = 1; i < argc; i++) {                                                 // slow_flow.cpp:168-193
        const char *a = argv[i];
        if (a[0] != '-') continue;
        if (!strcmp(a, "-h") || !strcmp(a, "-help")) usage();
        else if (!strcmp(a, "-overwrite")) overwrite_output = true;
        else if (!strcmp(a, "-resume")) resume_frame = true;
        else if (!strcmp(a, "-deep_settings") && i + 1 < argc) i++;
        else if (!strcmp(a, "-threads") && i + 1 < argc) params.insert("threads", argv[++i], true);
        else if (!strcmp(a, "-fr") && i + 1 < argc) i++;
        else if (!strcmp(a, "-jet") && i + 1 < argc) { selected_jet = atoi(argv[++i]); resume_frame = true; }
        else { fprintf(stderr, "unknown argument %s", a); usage(); }
    }
    if (params.parameter<bool>("deep_matching")) { std::cerr << "deep_matching=1 needs the external DeepMatching/SED/EpicFlow stage: not part of this build (set deep_matching 0)" << std::endl; return 2; }
    const bool raw = params.exists("raw") && params.parameter<bool>("raw");
    if (raw && params.parameter<int>("raw_demosaicing", "0") != 0) {
        std::cerr << "raw_demosaicing 1 (Hamilton-Adams, P. Getreuer) and 2 (OpenCV) are third-party code absent from the reference tree: use raw_demosaicing 0 "
                     "(the reference's own bilinear / green-ratio routine) or provide demosaiced frames with raw 0" << std::endl;
        return 2;
    }
    const std::vector<int> red_loc = params.splitParameter<int>("raw_red_loc", "0,0");   // :439
    const float scale = params.parameter<float>("scale", "1.0");

    const int steps = params.parameter<int>("slow_flow_S") - 1, ref = steps;         // :208-209
    const int max_fps = params.parameter<int>("max_fps", "1");
    const int jet_fps = params.exists("jet_fps") ? params.parameter<int>("jet_fps") : max_fps;
    const int skip = (int)((1.0f * max_fps) / jet_fps);                              // :220
    const bool sintel = params.parameter<bool>("sintel", "0"), subframes = params.parameter<bool>("subframes", "0");
    unsigned start = params.sequence_start;
    const size_t cut = params.file.find_last_of('/') + 1;
    string sequence_path = params.file.substr(0, cut), format = params.file.substr(cut);
    if (sequence_path.empty() || params.output.empty()) { std::cerr << "cfg needs 'file' and 'output'" << std::endl; return -1; }
    const string format_flow = format.substr(0, format.find_last_of('.'));
    if (sintel && !subframes) start *= 1000;
    params.sequence_start = start;

    if (!resume_frame && !overwrite_output) {                                        // never overwrite a results folder (:254-265)
        string np = params.output;
        if (np.back() == '/') np.pop_back();
        const string base = np;
        int num = 1;
        while (file_exists(np)) { std::cerr << np << " already exists!" << std::endl; np = base + "_" + std::to_string(num++); }
        params.output = np;
    }
    if (params.output.back() != '/') params.output += "/";
    mkdirs(params.output);

    const int frames = 1 + (params.Jets + 2) * steps;                                // :411
    unsigned start_f = 0, end_f = frames, start_j = 0, end_j = params.Jets;
    if (resume_frame && selected_jet >= 0) {                                         // :418-424
        start_f = selected_jet * steps;
        end_f = std::min(frames, 1 + (selected_jet + 3) * steps);
        start_j = selected_jet;
        end_j = std::min((int)params.Jets, selected_jet + 1);
    }
    if (start_f > end_f) return 0;

    // ---- read the image sequence (:447-592, without OpenCV: PNG, binary PPM / PGM / PFM) ------------------------------
    std::vector<color_image_t *> seq(frames, nullptr), seq_back(frames, nullptr);
    sfa_ctx *ingest_ctx = nullptr;
    for (unsigned f = start_f; f < end_f; f++) {
        string img_file;
        if (!sintel) img_file = fmt1(sequence_path + format, (int)start - ref * skip + (int)f * skip);
        else {
            int sintel_frame = start / 1000, hfr = (int)f * skip - ref * skip + (int)(start % 1000);
            while (hfr < 0) { sintel_frame--; hfr += 42; }
            while (hfr > 41) { sintel_frame++; hfr -= 42; }
            img_file = fmt2(sequence_path + format, sintel_frame, hfr);
        }
        std::cout << "Reading " << img_file << "..." << std::endl;
        int maxval = 255;
        seq[f] = color_image_load(img_file.c_str(), &maxval);
        if (!seq[f]) { std::cerr << "cannot read frame " << img_file << " (PNG or binary PPM/PGM/PFM expected)" << std::endl; return 3; }
        if (raw) {                                                                   // demosaicing (:482-527): the mosaic is the grey image
            image_t mosaic = {seq[f]->width, seq[f]->height, seq[f]->stride, seq[f]->c1};
            color_image_t *rgb = color_image_new(seq[f]->width, seq[f]->height);
            color_image_erase(rgb);
            bayer2rgbGR(&mosaic, rgb, red_loc.size() > 0 ? red_loc[0] : 0, red_loc.size() > 1 ? red_loc[1] : 0);
            color_image_delete(seq[f]);
            seq[f] = rgb;
        }
        if (!params.exists("raw") || params.parameter<float>("raw_weight", "1.0") == 1.0f) {   // :531
            if (params.extent.x > 0 || params.extent.y > 0) {                        // use only a part of the images (:533-536)
                color_image_t *part = color_image_crop(seq[f], params.center.x, params.center.y, params.extent.x, params.extent.y);
                if (!part) { std::cerr << "center / extent do not fit the " << seq[f]->width << "x" << seq[f]->height << " frames" << std::endl; return 3; }
                color_image_delete(seq[f]);
                seq[f] = part;
            }
            if (scale != 1) {                                                        // blur + resize against aliasing (:550-553), on the GPU
                if (!ingest_ctx && sfa_ctx_create(params.parameter<int>("gpu_device", "0"), &ingest_ctx) != SFA_OK) { std::cerr << sfa_last_error(nullptr) << std::endl; return 4; }
                color_image_t *small = color_image_rescale(ingest_ctx, seq[f], scale);
                if (!small) { std::cerr << "rescaling failed: " << sfa_last_error(ingest_ctx) << std::endl; return 4; }
                color_image_delete(seq[f]);
                seq[f] = small;
            }
        }
        seq_back[frames - 1 - f] = seq[f];                                           // :590-591
    }
    const int width = seq[start_f]->width, height = seq[start_f]->height;
    color_image_t *channel_weights = color_image_new(width, height);                 // :597-598 (all ones without raw weighting)
    for (size_t i = 0; i < (size_t)3 * channel_weights->stride * height; i++) channel_weights->c1[i] = 1.0f;
    if (raw) rawWeighting(channel_weights, red_loc.size() > 0 ? red_loc[0] : 0, red_loc.size() > 1 ? red_loc[1] : 0, params.parameter<float>("raw_weight", "1.0"));   // :599-600
    if (ingest_ctx) { sfa_ctx_destroy(ingest_ctx); ingest_ctx = nullptr; }

    // ---- ground truth, if the cfg names it (:603-661): .flo -> crop -> nearest resize * scale -> gt/flow_%05i.{png,flo} --------
    std::vector<image_t **> gt(params.Jets, nullptr);
    if (!params.file_gt.empty()) {
        mkdirs(params.output + "gt/");
        for (unsigned j = start_j; j < end_j; j++) {
            string path;
            if (!sintel) path = fmt1(params.file_gt, (int)start + (int)j * steps);
            else {
                int sintel_frame = start / 1000, hfr = (int)j * steps + (int)(start % 1000);
                while (hfr < 0) { sintel_frame--; hfr += 42; }
                while (hfr > 41) { sintel_frame++; hfr -= 42; }
                path = fmt2(params.file_gt, sintel_frame, hfr);
            }
            std::cout << path << std::endl;
            if (!file_exists(path)) continue;
            image_t **g = readFlowFile(path.c_str());
            if (!g) { std::cout << "No gt flow for frame " << std::endl; continue; }
            if (params.center.x > 0) {                                               // crop like the frames (:634-637)
                for (int c = 0; c < 2; c++) {
                    const int x0 = params.center.x - params.extent.x / 2, y0 = params.center.y - params.extent.y / 2;
                    const int cw = 2 * (params.extent.x / 2), chh = 2 * (params.extent.y / 2);
                    if (x0 < 0 || y0 < 0 || x0 + cw > g[c]->width || y0 + chh > g[c]->height) { std::cerr << "center / extent do not fit the ground truth" << std::endl; return 3; }
                    image_t *part = image_new(cw, chh);
                    image_erase(part);
                    for (int y = 0; y < chh; y++) memcpy(part->data + (size_t)y * part->stride, g[c]->data + (size_t)(y0 + y) * g[c]->stride + x0, sizeof(float) * cw);
                    image_delete(g[c]);
                    g[c] = part;
                }
            }
            for (int c = 0; c < 2; c++) {                                            // nearest, not linear: motion discontinuities (:640-641)
                image_t *r = flow_resize_nearest(g[c], scale);
                image_delete(g[c]);
                g[c] = r;
            }
            gt[j] = g;
            const string base = params.output + "gt/" + fmt1("flow_%05i", (int)params.sequence_start + (int)j * steps);
            png_write((base + ".png").c_str(), flowColorImg(g[0], g[1], params.verbosity(VER_CMD)));
            writeFlowFile((base + ".flo").c_str(), g[0], g[1]);
        }
    }

    normalize(&seq[start_f], end_f - start_f, params);                               // :673
    {
        std::ofstream infos((params.output + "config.cfg").c_str());                 // :684-688
        infos << "# SlowFlow variational estimation\n" << params;
    }

    // ---- the windows to refine: forward and backward of every jet (:706-1047) --------------------------------------
    std::vector<Window> todo;
    for (unsigned j = start_j; j < end_j; j++) {
        const int f = j * steps;
        const string fwd = sintel ? fmt2(params.output + format_flow + ".flo", start + f * skip, 0) : fmt1(params.output + format_flow + ".flo", start + f * skip);
        const string bwd = sintel ? fmt2(params.output + format_flow + "_back.flo", start + f * skip + steps * skip, 0)
                                  : fmt1(params.output + format_flow + "_back.flo", start + f * skip + steps * skip);
        if (!resume_frame || !file_exists(fwd)) todo.push_back(Window(j, false, fwd));
        else std::cout << "Forward flow from frame " << start + f << " to " << start + f * skip + steps * skip << " already exist!" << std::endl;
        if (!resume_frame || !file_exists(bwd)) todo.push_back(Window(j, true, bwd));
        else std::cout << "Backward flow from frame " << start + f * skip << " to " << start + f * skip + steps * skip << " already exist!" << std::endl;
    }

    int ngpu = sfa_device_count();
    if (ngpu <= 0) { std::cerr << "no HIP device: slowflow_amd has no CPU fallback" << std::endl; return 4; }
    if (params.exists("gpus")) ngpu = std::max(1, std::min(ngpu, params.parameter<int>("gpus")));
    const int dev0 = params.parameter<int>("gpu_device", "0");
    const int batch = std::max(1, std::min(64, params.parameter<int>("gpu_batch", "32")));
    const int F = 2 * ref + 1;
    const bool backward_forward_only = params.exists("method") && params.parameter("method") == "forward";   // :1019-1020

    std::mutex io_mu;
    bool failed = false;
    // `gpu_streams` workers per GPU (default 2), each with its own context = HIP stream: two lockstep groups fill each other's
    // ramp-up / drain phases (the reference runs its windows from `threads` OpenMP threads the same way, slow_flow.cpp:706)
    const int streams = std::max(1, std::min(4, params.parameter<int>("gpu_streams", "2")));
    const int nworkers = ngpu * streams;
    auto worker = [&](int wk) {
        const int g = wk / streams;
        // contiguous block of windows per worker: neighbouring jets share frames
        const size_t lo = todo.size() * wk / nworkers, hi = todo.size() * (wk + 1) / nworkers;
        if (lo >= hi) return;
        sfa_ctx *ctx = nullptr;
        if (sfa_ctx_create(dev0 + g, &ctx) != SFA_OK) { std::lock_guard<std::mutex> l(io_mu); std::cerr << sfa_last_error(nullptr) << std::endl; failed = true; return; }
        ParameterList tp(params);                                                    // one copy per thread (:708)
        for (size_t b0 = lo; b0 < hi && !failed; b0 += batch) {
            const size_t nb = std::min((size_t)batch, hi - b0);
            // forward and backward windows share one lockstep job (the channel weights are per window) unless the backward solver runs with
            // different parameters ("method forward", :1019-1020): then one job per direction
            for (int dirpass = 0; dirpass < (backward_forward_only ? 2 : 1); dirpass++) {
                std::vector<size_t> idx;
                for (size_t i = 0; i < nb; i++)
                    if (!backward_forward_only || (int)todo[b0 + i].backward == dirpass) idx.push_back(b0 + i);
                if (idx.empty()) continue;
                sfa_params sp = sfa_params_from_cfg(tp, dirpass == 1 ? backward_forward_only : false);
                sfa_job *job = nullptr;
                const auto t0 = std::chrono::steady_clock::now();
                int rc = sfa_job_create(ctx, &sp, width, height, (int)idx.size(), &job);
                for (size_t e = 0; e < idx.size() && rc == SFA_OK; e++) {
                    const Window &wd = todo[idx[e]];
                    const int f = wd.jet * steps;
                    color_image_t *const *im = wd.backward ? &seq_back[frames - 1 - f - 3 * steps] : &seq[f];   // :721-724
                    std::vector<const float *> fr(F);
                    for (int k = 0; k < F; k++) fr[k] = im[k]->c1;
                    const float *chw[3] = {channel_weights->c1, channel_weights->c2, channel_weights->c3};
                    // only the forward solver gets the channel weights (:876 vs :1018); without raw weighting they are all ones (:597-598), which is
                    // what a NULL pointer means to the library (x * 1.0f is exact: same bits, three planes less to read per pixel)
                    rc = sfa_job_upload(job, (int)e, fr.data(), F, nullptr, nullptr, im[0]->stride, (wd.backward || !raw) ? nullptr : chw);
                }
                if (rc == SFA_OK) rc = sfa_job_run(job);
                for (size_t e = 0; e < idx.size() && rc == SFA_OK; e++) {
                    Window &wd = todo[idx[e]];
                    image_t *wx = image_new(width, height), *wy = image_new(width, height);
                    image_erase(wx); image_erase(wy);
                    float change[2];
                    rc = sfa_job_download(job, (int)e, wx->data, wy->data, wx->stride, change);
                    if (rc == SFA_OK) {
                        image_mul_scalar(wx, (float)steps);                          // :908-909
                        image_mul_scalar(wy, (float)steps);
                        if (writeFlowFile(wd.out.c_str(), wx, wy) != 0) rc = SFA_ERR_ARG;
                    }
                    if (rc == SFA_OK && !wd.backward) {                              // the flow as a colour image next to the .flo (:913-925)
                        const int fnum = (int)start + (int)wd.jet * steps * skip;
                        png_write((params.output + "frame_" + std::to_string(fnum) + ".png").c_str(), flowColorImg(wx, wy, 0));
                        image_t **g = gt[wd.jet];
                        if (g && g[0]->width == width && g[0]->height == height) {   // additive: error against the ground truth, reported in timings.json
                            wd.epe = computeEPE(wx, wy, g[0], g[1]);
                            wd.aae = computeAAE(wx, wy, g[0], g[1]);
                        }
                    }
                    if (rc == SFA_OK && sp.occlusion_reasoning && !wd.backward && tp.parameter<bool>("slow_flow_output_occlusions", "0")) {
                        // the final occlusion labels of the forward window as an image, (occ + 1) / 2 * 255 like :276-279 (there: one PNG per
                        // alternation; here the last one, as PGM)
                        image_t *occ = image_new(width, height);
                        rc = sfa_job_download_occlusions(job, (int)e, occ->data, occ->stride);
                        if (rc == SFA_OK) {
                            mkdirs(params.output + "occlusion/");                    // :676-677
                            const string of = params.output + "occlusion/" + wd.out.substr(wd.out.find_last_of('/') + 1, wd.out.find_last_of('.') - wd.out.find_last_of('/') - 1) + "_occ.pgm";
                            writePGM(of.c_str(), occ, 1.0f, 127.5f);
                        }
                        image_delete(occ);
                    }
                    image_delete(wx); image_delete(wy);
                    wd.gpu = dev0 + g;
                }
                const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                for (size_t e = 0; e < idx.size(); e++) todo[idx[e]].seconds = secs / idx.size();
                if (job) sfa_job_destroy(job);
                std::lock_guard<std::mutex> l(io_mu);
                if (rc != SFA_OK) { std::cerr << "GPU " << dev0 + g << ": " << sfa_last_error(ctx) << std::endl; failed = true; }
                else for (size_t e = 0; e < idx.size(); e++) {
                    const Window &wd = todo[idx[e]];
                    const int f = wd.jet * steps;
                    std::cout << (wd.backward ? "Backward" : "Forward") << " flow from frame " << start + f * skip << " to " << start + f * skip + steps * skip
                              << " finished! (GPU " << dev0 + g << ", " << secs / idx.size() << " s per window in a batch of " << idx.size() << ")" << std::endl;
                }
            }
        }
        sfa_ctx_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int wk = 0; wk < nworkers; wk++) th.emplace_back(worker, wk);
    for (auto &t : th) t.join();

    // the gathered per-window timings (the only cross-GPU exchange of the path)
    {
        std::ofstream tj((params.output + "timings.json").c_str());
        tj << "[";
        for (size_t i = 0; i < todo.size(); i++)
            tj << (i ? "," : "") << "\n  {\"jet\": " << todo[i].jet << ", \"direction\": \"" << (todo[i].backward ? "backward" : "forward") << "\", \"gpu\": " << todo[i].gpu
               << ", \"seconds\": " << todo[i].seconds << ", \"flo\": \"" << todo[i].out << "\""
               << (todo[i].epe >= 0 ? ", \"epe\": " + std::to_string(todo[i].epe) + ", \"aae\": " + std::to_string(todo[i].aae) : string()) << "}";
        tj << "\n]\n";
    }
    for (unsigned f = start_f; f < end_f; f++) color_image_delete(seq[f]);
    for (auto g : gt) if (g) { image_delete(g[0]); image_delete(g[1]); free(g); }
    color_image_delete(channel_weights);
    std::cout << (failed ? "Failed!" : "Done!") << std::endl;
    return failed ? 5 : 0;
}
