// slow_flow.cpp -- the slow_flow driver of the drop-in: same cfg keys, command line and outputs as the reference's
// slow_flow.cpp (usage :58-62, defaults :64-128, adaptive frame rates :277-357, frame indexing :411-465, jet loop :706-1047,
// .flo outputs :908-911, :1027-1030, resume :794,958, config.cfg :684-688) for the path this repository implements:
//   frames -> normalize -> per jet: forward / backward Variational_MT::variational -> flow * steps -> .flo
// The jets of a sequence are independent; they are sharded over the node's GPUs (shard.h: `gpu_streams` host threads per
// GPU, each with its own context = HIP stream) and each worker refines `gpu_batch` frame windows in lockstep.  The driver is
// a pipeline: frames are decoded by a pool of host threads, sent to every GPU once and normalised there (sfa_sequence), every worker
// keeps ONE resident job for all its batches (device-to-device window copies / run / download), and results go to the output pool (.flo, colour PNG, occlusion images) while the worker
// already refines its next batch; with two workers per GPU the uploads of one overlap the kernels of the other.
// deep_matching 1 initialises the flow with EpicFlow's interpolation (epic.h) of match and edge files found at the reference's locations; producing those files
// (DeepMatching, the MATLAB SED detector) and the third-party Hamilton-Adams demosaicer (raw_demosaicing 1) are outside this build and reported as such.
//
// New, additive keys: gpus (default: all visible; gpu_oversubscribe 1 lets it exceed them: a rehearsal of the multi-GPU path), gpu_batch (windows refined in lockstep per job, default 32), gpu_streams
// (default 2), gpu_device (first device, default 0), io_threads (decode / output pool, default min(16, cores)),
// adaptive_fr_file (default: adaptiveFR.dat next to the executable, the reference's SOURCE_PATH).
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <string>
#include <future>
#include <thread>
#include <vector>

#include "epic.h"
#include "flow_vis.h"
#include "image.h"
#include "ingest.h"
#include "io.h"
#include "parameter_list.h"
#include "shard.h"
#include "variational_mt.h"

using std::string;

static void usage() {
    printf("usage:\n    ./slow_flow [cfg] -overwrite -resume -deep_settings [settings] -threads -fr [select one specific adaptive frame rate] "
           "-jet [select one specific high speed flow]\n\n"
           "    -deep_settings is accepted and ignored: the reference passes it to the DeepMatching binary it starts (slow_flow.cpp:180-181, :768); this build does\n"
           "    not start DeepMatching or the MATLAB edge detector -- with deep_matching 1 it reads their outputs from <output>tmp/ (matches_<a>_<b>.dat, edges_<n>.dat).\n"
           "    raw_demosaicing 1 (Hamilton-Adams, third-party code absent from the reference tree) is rejected; 0 (bilinear) and 2 (OpenCV's 8-bit conversion) work.\n\n");
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
static string exe_dir() {
    char buf[4096];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    string s(buf);
    return s.substr(0, s.find_last_of('/'));
}

struct Window {
    unsigned jet; bool backward; string out; double seconds; int gpu; double epe, aae;   // epe/aae < 0: no ground truth
    Window(unsigned j, bool b, const string &o) : jet(j), backward(b), out(o), seconds(0), gpu(-1), epe(-1), aae(-1) {}
};

// the labels of optimizeOcc as an 8-bit image, (occ + 1) / 2 * 255 (variational_mt.cpp:277-279, slow_flow.cpp:895-896)
static png_image occlusion_image(const image_t *occ) {
    png_image im;
    im.width = occ->width; im.height = occ->height; im.channels = 1; im.depth = 8;
    im.samples.resize((size_t)occ->width * occ->height);
    for (int y = 0; y < occ->height; y++)
        for (int x = 0; x < occ->width; x++) {
            const float v = (occ->data[(size_t)y * occ->stride + x] + 1) * 0.5f * 255.0f;
            im.samples[(size_t)y * occ->width + x] = (uint16_t)std::max(0.0f, std::min(255.0f, std::nearbyint(v)));
        }
    return im;
}

struct RunOptions {
    bool resume_frame = false;
    int selected_jet = -1;
};

// everything a finished window hands to the output pool
struct WindowResult {
    size_t index = 0;                            // into todo
    image_t *wx = nullptr, *wy = nullptr, *occ = nullptr;
    std::vector<image_t *> alt_occ;              // labels after alternation 1 .. niter_alter-1 (WRITE_FILES verbosity)
    ~WindowResult() {
        if (wx) image_delete(wx);
        if (wy) image_delete(wy);
        if (occ) image_delete(occ);
        for (auto *o : alt_occ) image_delete(o);
    }
};

// One pass over the sequence at one frame rate: the body of the reference's adFR loop (slow_flow.cpp:367-1060).
static int run_sequence(ParameterList &params, const string &sequence_path, const string &format, int skip, const RunOptions &opt) {
    const auto t_begin = std::chrono::steady_clock::now();
    const bool raw = params.exists("raw") && params.parameter<bool>("raw");
    const std::vector<int> red_loc = params.splitParameter<int>("raw_red_loc", "0,0");   // :439
    const int demosaicing = params.parameter<int>("raw_demosaicing", "0");
    const float scale = params.parameter<float>("scale", "1.0");
    const int steps = params.parameter<int>("slow_flow_S") - 1, ref = steps;         // :208-209
    const bool sintel = params.parameter<bool>("sintel", "0");
    const unsigned start = params.sequence_start;
    const string format_flow = format.substr(0, format.find_last_of('.'));
    const int io_threads = std::max(1, std::min(64, params.parameter<int>("io_threads", std::to_string(std::max(1u, std::min(16u, std::thread::hardware_concurrency()))))));

    {                                                                                // a job's windows-still-iterating mask is two 64-bit words: refused by name, not clamped
        const int gpu_batch = params.parameter<int>("gpu_batch", "32");
        if (gpu_batch < 1 || gpu_batch > 128) {
            std::cerr << "gpu_batch " << gpu_batch << " is out of range: a lockstep job takes 1 .. 128 windows (more windows run as several jobs: gpu_streams, batches per worker)" << std::endl;
            return 2;
        }
    }
    mkdirs(params.output);

    const int frames = 1 + (params.Jets + 2) * steps;                                // :411
    unsigned start_f = 0, end_f = frames, start_j = 0, end_j = params.Jets;
    if (opt.resume_frame && opt.selected_jet >= 0) {                                 // :418-424
        start_f = opt.selected_jet * steps;
        end_f = std::min(frames, 1 + (opt.selected_jet + 3) * steps);
        start_j = opt.selected_jet;
        end_j = std::min((int)params.Jets, opt.selected_jet + 1);
    }
    if (start_f > end_f) return 0;

    // the two output files of jet j, and whether the jet still has work (-resume skips existing flows, :712-716 / :934-938)
    auto flow_files = [&](unsigned j, string &fwd, string &bwd) {
        const int f = j * steps;
        fwd = sintel ? fmt2(params.output + format_flow + ".flo", start + f * skip, 0) : fmt1(params.output + format_flow + ".flo", start + f * skip);
        bwd = sintel ? fmt2(params.output + format_flow + "_back.flo", start + f * skip + steps * skip, 0)
                     : fmt1(params.output + format_flow + "_back.flo", start + f * skip + steps * skip);
    };
    auto jet_pending = [&](unsigned j) {
        if (!opt.resume_frame) return true;
        string fwd, bwd;
        flow_files(j, fwd, bwd);
        return !file_exists(fwd) || !file_exists(bwd);
    };
    // ---- deep_matching 1 (:744-863): the flow is initialised by EpicFlow's interpolation (epic.h) of DeepMatching matches along SED edges.  The reference starts
    //      both tools with system() (MATLAB + a binary, third-party); this build reads their outputs from the reference's own locations and says so if they are missing.
    const bool verbose_changes = params.verbosity(VER_CMD);       // the library prints the reference's "inner it / outer it ... avg change" lines (variational_mt.cpp:404-405, 431-432): per worker context
    const bool enable_dm = params.parameter<bool>("deep_matching");
    // dm_scale (:396-402, :571-586, :801-843): DeepMatching and the edge detector ran on frames blurred (sigma = 1 / sqrt(2 dm_scale)) and resized by dm_scale; the
    // interpolation then runs at that resolution (the saliency image, the edge map, the matches' coordinates) and its flow is resized to the frames and multiplied by the
    // INTEGER ratio of the widths (:827-828: `float fx = im[ref]->width / wx->width` divides two ints).  dm_scale_run = dm_scale, halved by main() when max_flow > 150
    const double dm_scale = params.exists("dm_scale_run") ? params.parameter<double>("dm_scale_run") : (double)params.parameter<float>("dm_scale", "1.0");
    auto edges_file = [&](int frame_number) { return params.output + "tmp/edges_" + std::to_string(frame_number) + ".dat"; };                    // :740-741
    auto matches_file = [&](int a, int b) { return params.output + "tmp/matches_" + std::to_string(a) + "_" + std::to_string(b) + ".dat"; };    // :742-743
    if (enable_dm) {
        for (unsigned j = start_j; j < end_j; j++) {
            if (!jet_pending(j)) continue;                                           // -resume: a jet whose two flows exist is skipped below and needs no inputs
            const int a = (int)params.sequence_start + (int)j * steps * skip, b = a + ref * skip;
            const string need[4] = {edges_file(a), edges_file(b), matches_file(a, b), matches_file(b, a)};
            for (const string &f : need)
                if (!file_exists(f)) {
                    std::cerr << "deep_matching 1: " << f << " is missing. The matches come from DeepMatching and the edge maps from the SED detector (matlab/detect_edges.m), "
                                 "third-party programs this build does not start; place their outputs there (raw float edge map of the frame size; one match 'x1 y1 x2 y2 ...' per line) "
                                 "and run with -resume or -overwrite, or set deep_matching 0" << std::endl;
                    return 2;
                }
        }
    }

    std::vector<color_image_t *> seq(frames, nullptr), seq_back(frames, nullptr);
    std::vector<int> seq_maxval(frames, 255);            // 255 / 65535 per frame: the 8-bit copy EpicFlow's saliency works on divides 16-bit samples by 255 (:472-474, :578)
    // ---- the GPUs (chosen before anything is read, so that the HIP runtime starts up -- 0.35 s for the first context of a process -- beside the decoding) ----
    const int ndev = sfa_device_count();
    int ngpu = ndev;
    if (ngpu <= 0) { std::cerr << "no HIP device: slowflow_amd has no CPU fallback" << std::endl; return 4; }
    // gpu_oversubscribe 1: `gpus` may exceed the devices of the box, GPU g then runs on device g mod devices -- the multi-GPU code path (one resident
    // sequence, one set of workers and one statistics slot per GPU) rehearsed on fewer cards than it is written for; never a way to get speed
    const bool oversubscribe = params.parameter<bool>("gpu_oversubscribe", "0");
    if (params.exists("gpus")) ngpu = std::max(1, oversubscribe ? std::min(16, params.parameter<int>("gpus")) : std::min(ngpu, params.parameter<int>("gpus")));
    const int dev0 = params.parameter<int>("gpu_device", "0");
    if (!oversubscribe) ngpu = std::max(1, std::min(ngpu, ndev - dev0));             // devices dev0 .. ndev-1 exist; asking for more would only fail after all frames are decoded
    if (!oversubscribe && dev0 >= ndev) { std::cerr << "gpu_device " << dev0 << " does not exist (" << ndev << " device(s))" << std::endl; return 4; }
    auto device_of = [&](int g) { return oversubscribe ? (dev0 + g) % ndev : dev0 + g; };
    const int n_loaded = (int)(end_f - start_f);

    // ---- the windows to refine: forward and backward of every jet (:706-1047), and who refines them -- known before a frame is read, so that every GPU is sent
    //      only the frames ITS windows read (until round 4 every GPU received and normalised the whole sequence) ----------------------------------
    std::vector<Window> todo;
    for (unsigned j = start_j; j < end_j; j++) {
        const int f = j * steps;
        string fwd, bwd;
        flow_files(j, fwd, bwd);
        if (!opt.resume_frame || !file_exists(fwd)) todo.push_back(Window(j, false, fwd));
        else std::cout << "Forward flow from frame " << start + f << " to " << start + f * skip + steps * skip << " already exist!" << std::endl;
        if (!opt.resume_frame || !file_exists(bwd)) todo.push_back(Window(j, true, bwd));
        else std::cout << "Backward flow from frame " << start + f * skip << " to " << start + f * skip + steps * skip << " already exist!" << std::endl;
    }
    // `gpu_streams` workers per GPU (default 2), each with its own context = HIP stream: two lockstep groups fill each other's
    // ramp-up / drain phases (the reference runs its windows from `threads` OpenMP threads the same way, slow_flow.cpp:706)
    const int streams = std::max(1, std::min(4, params.parameter<int>("gpu_streams", "2")));
    const std::vector<WorkerPlan> plan = plan_workers(todo.size(), ngpu, streams);
    std::vector<std::pair<int, int>> window_frames;                                  // relative to the loaded range [start_f, end_f)
    for (const Window &wd : todo) {
        const int f = (int)wd.jet * steps - (int)start_f;
        window_frames.push_back(wd.backward ? std::make_pair(f + steps, f + 3 * steps) : std::make_pair(f, f + 2 * ref));   // :721-724, :590-591
    }
    const std::vector<FrameRange> gpu_frames = plan_frames(window_frames, plan, ngpu, n_loaded);

    std::vector<sfa_ctx *> seq_ctx(ngpu, nullptr);
    std::vector<sfa_sequence *> seq_dev(ngpu, nullptr);
    std::vector<string> seq_err(ngpu);
    // The frames go to the GPUs once and are normalised there (:673; the three parts of sfa_sequence_normalize: same kernels, same statistics as the host-plane
    // normalize() of variational_mt.h).  GPU g holds the frames gpu_frames[g] of its own windows (+ the halo its boundary jets share with the neighbour): about 1 / N
    // of the sequence.  The statistics are those of ALL loaded frames: every GPU forms the six fp64 sums of the frames it holds (a deterministic kernel: a frame two
    // GPUs hold gets the same bits from both), the sums are collected per frame on the host, and the moment the last frame's sums are in, the statistics are formed
    // from them in frame order -- one set of doubles for every GPU.  One thread per GPU: it creates the context while the io pool decodes, uploads each of its frames the
    // moment its decode task has finished (`ready`), normalises its frames once the statistics are known and releases its workers (`gpu_ready`); no GPU waits for
    // another GPU's uploads beyond the sums of the frames it does not hold itself.
    std::mutex ready_mu;
    std::condition_variable ready_cv;
    std::vector<char> frame_ready(frames, 0);
    bool ingest_abort = false, size_known = false;
    int width = 0, height = 0;
    std::vector<double> frame_sums((size_t)6 * n_loaded, 0.0);
    std::vector<char> have_sums(n_loaded, 0), gpu_ready(ngpu, 0);
    int n_have = 0;
    bool stats_known = false;
    double stat_avg[3] = {0, 0, 0}, stat_std[3] = {1, 1, 1};
    std::vector<double> upload_bytes(ngpu, 0.0), upload_done_s(ngpu, 0.0), ready_s(ngpu, 0.0);
    auto seconds_since_begin = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    const bool upload_while_decoding = !(((!params.exists("raw") || params.parameter<float>("raw_weight", "1.0") == 1.0f)) && scale != 1);   // a rescaled sequence exists only after all frames are in
    std::vector<std::thread> up;
    for (int g = 0; g < ngpu; g++)
        up.emplace_back([&, g] {
            const FrameRange fr = gpu_frames[g];
            const int n_mine = fr.hi - fr.lo;
            int rc = sfa_ctx_create(device_of(g), &seq_ctx[g]);
            auto fail = [&] {                                                         // nobody may wait for this GPU's sums or readiness
                seq_err[g] = seq_ctx[g] ? sfa_last_error(seq_ctx[g]) : sfa_last_error(nullptr);
                { std::lock_guard<std::mutex> l(ready_mu); ingest_abort = true; }
                ready_cv.notify_all();
            };
            if (rc != SFA_OK) { fail(); return; }
            if (n_mine <= 0) return;                                                 // more GPUs than windows
            {
                std::unique_lock<std::mutex> l(ready_mu);
                ready_cv.wait(l, [&] { return size_known || ingest_abort; });
                if (ingest_abort) return;
            }
            rc = sfa_sequence_create(seq_ctx[g], width, height, n_mine, &seq_dev[g]);
            for (int f = fr.lo; f < fr.hi && rc == SFA_OK; f++) {
                {
                    std::unique_lock<std::mutex> l(ready_mu);
                    ready_cv.wait(l, [&] { return frame_ready[start_f + f] || ingest_abort; });
                    if (ingest_abort) return;
                }
                rc = sfa_sequence_upload(seq_dev[g], f - fr.lo, seq[start_f + f]->c1, seq[start_f + f]->stride);
            }
            std::vector<double> mine((size_t)6 * n_mine);
            if (rc == SFA_OK) rc = sfa_sequence_frame_sums(seq_dev[g], 0, n_mine, mine.data());      // (waits for the uploads: one stream)
            if (rc != SFA_OK) { fail(); return; }
            {
                std::unique_lock<std::mutex> l(ready_mu);
                upload_bytes[g] = (double)n_mine * 3 * width * height * sizeof(float);
                upload_done_s[g] = seconds_since_begin();
                for (int f = fr.lo; f < fr.hi; f++)
                    if (!have_sums[f]) {                                             // a halo frame: whoever is first; the other GPU's sums are the same bits
                        std::copy(mine.begin() + 6 * (f - fr.lo), mine.begin() + 6 * (f - fr.lo) + 6, frame_sums.begin() + 6 * f);
                        have_sums[f] = 1;
                        n_have++;
                    }
                if (n_have == n_loaded && !stats_known) {                            // variational_mt.cpp:41-52 over all loaded frames, in frame order
                    (void)sfa_normalize_statistics(frame_sums.data(), n_loaded, width, height, stat_avg, stat_std);
                    stats_known = true;
                }
                ready_cv.notify_all();
                ready_cv.wait(l, [&] { return stats_known || ingest_abort; });
                if (ingest_abort) return;
            }
            rc = sfa_sequence_apply_normalization(seq_dev[g], 0, n_mine, stat_avg, stat_std);
            if (rc != SFA_OK) { fail(); return; }
            { std::lock_guard<std::mutex> l(ready_mu); gpu_ready[g] = 1; ready_s[g] = seconds_since_begin(); }
            ready_cv.notify_all();
        });
    auto release_sequences = [&] {
        for (int g = 0; g < ngpu; g++) {
            if (seq_dev[g]) sfa_sequence_destroy(seq_dev[g]);
            if (seq_ctx[g]) sfa_ctx_destroy(seq_ctx[g]);
            seq_dev[g] = nullptr; seq_ctx[g] = nullptr;
        }
    };
    auto abort_ingest = [&] {
        { std::lock_guard<std::mutex> l(ready_mu); ingest_abort = true; }
        ready_cv.notify_all();
        for (auto &t : up) t.join();
        release_sequences();
    };
    // frames [f0, f1) are final: publish the size once, then the frames.  Every frame is held against the published size BEFORE it is marked ready: the uploaders copy
    // width x height of the published size out of each frame's buffer, so a smaller frame in the middle of the sequence would be read beyond its end; it is
    // left unpublished and reported through `size_mismatch` instead (the uploaders are then released by abort_ingest)
    string size_mismatch;
    auto frames_are = [&](unsigned f0, unsigned f1) {
        {
            std::lock_guard<std::mutex> l(ready_mu);
            if (!size_known) { width = seq[f0]->width; height = seq[f0]->height; size_known = true; }
            for (unsigned f = f0; f < f1; f++) {
                if (seq[f]->width != width || seq[f]->height != height) {
                    if (size_mismatch.empty())
                        size_mismatch = "frames of different sizes: frame " + std::to_string(f) + " of the sequence is " + std::to_string(seq[f]->width) + "x" + std::to_string(seq[f]->height) + ", the sequence " +
                                        std::to_string(width) + "x" + std::to_string(height);
                    continue;
                }
                frame_ready[f] = 1;
            }
        }
        ready_cv.notify_all();
    };

    // ---- read the image sequence (:447-592, without OpenCV: PNG, binary PPM / PGM / PFM): decoded, demosaiced and cropped by the
    //      io pool, one frame per task ----------------------------------------------------------------------------------------------
    std::vector<string> names(frames);
    for (unsigned f = start_f; f < end_f; f++) {
        if (!sintel) names[f] = fmt1(sequence_path + format, (int)start - ref * skip + (int)f * skip);
        else {
            int sintel_frame = start / 1000, hfr = (int)f * skip - ref * skip + (int)(start % 1000);
            while (hfr < 0) { sintel_frame--; hfr += 42; }
            while (hfr > 41) { sintel_frame++; hfr -= 42; }
            names[f] = fmt2(sequence_path + format, sintel_frame, hfr);
        }
        std::cout << "Reading " << names[f] << "..." << std::endl;
    }
    const bool preprocess = !params.exists("raw") || params.parameter<float>("raw_weight", "1.0") == 1.0f;   // :531
    const Point center = params.center, extent = params.extent;
    std::mutex err_mu;
    string load_error;
    {
        TaskPool pool(io_threads);
        for (unsigned f = start_f; f < end_f; f++)
            pool.submit([&, f] {
                int maxval = 255;
                color_image_t *img = color_image_load(names[f].c_str(), &maxval);
                if (!img) { std::lock_guard<std::mutex> l(err_mu); if (load_error.empty()) load_error = "cannot read frame " + names[f] + " (PNG, TIFF or binary PPM/PGM/PFM expected)"; return; }
                if (raw) {                                                           // demosaicing (:482-527): the mosaic is the grey image
                    image_t mosaic = {img->width, img->height, img->stride, img->c1};
                    color_image_t *rgb = color_image_new(img->width, img->height);
                    color_image_erase(rgb);
                    const int rx = red_loc.size() > 0 ? red_loc[0] : 0, ry = red_loc.size() > 1 ? red_loc[1] : 0;
                    if (demosaicing == 2) bayer2rgb_cv8u(&mosaic, rgb, rx, ry);          // :502-520
                    else bayer2rgbGR(&mosaic, rgb, rx, ry);                             // :488-491
                    color_image_delete(img);
                    img = rgb;
                }
                if (preprocess && (extent.x > 0 || extent.y > 0)) {                  // use only a part of the images (:533-536)
                    color_image_t *part = color_image_crop(img, center.x, center.y, extent.x, extent.y);
                    if (!part) {
                        std::lock_guard<std::mutex> l(err_mu);
                        if (load_error.empty()) load_error = "center / extent do not fit the " + std::to_string(img->width) + "x" + std::to_string(img->height) + " frames";
                        color_image_delete(img);
                        return;
                    }
                    color_image_delete(img);
                    img = part;
                }
                seq[f] = img;
                seq_maxval[f] = maxval;
                if (upload_while_decoding) frames_are(f, f + 1);
            });
        pool.wait_all();
    }
    if (!load_error.empty()) { std::cerr << load_error << std::endl; abort_ingest(); return 3; }
    { std::lock_guard<std::mutex> l(ready_mu); if (!size_mismatch.empty()) load_error = size_mismatch; }
    if (!load_error.empty()) { std::cerr << load_error << std::endl; abort_ingest(); return 3; }
    const double decode_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (preprocess && scale != 1) {                                                  // blur + resize against aliasing (:550-553), on the GPU
        sfa_ctx *ingest_ctx = nullptr;
        if (sfa_ctx_create(params.parameter<int>("gpu_device", "0"), &ingest_ctx) != SFA_OK) { std::cerr << sfa_last_error(nullptr) << std::endl; abort_ingest(); return 4; }
        for (unsigned f = start_f; f < end_f; f++) {
            color_image_t *small = color_image_rescale(ingest_ctx, seq[f], scale);
            if (!small) { std::cerr << "rescaling failed: " << sfa_last_error(ingest_ctx) << std::endl; sfa_ctx_destroy(ingest_ctx); abort_ingest(); return 4; }
            color_image_delete(seq[f]);
            seq[f] = small;
        }
        sfa_ctx_destroy(ingest_ctx);
    }
    if (!upload_while_decoding) {
        frames_are(start_f, end_f);
        std::lock_guard<std::mutex> l(ready_mu);
        load_error = size_mismatch;
    }
    if (!load_error.empty()) { std::cerr << load_error << std::endl; abort_ingest(); return 3; }
    for (unsigned f = start_f; f < end_f; f++) seq_back[frames - 1 - f] = seq[f];    // :590-591
    if (seq[start_f]->width != width || seq[start_f]->height != height) { std::cerr << "frames of different sizes" << std::endl; abort_ingest(); return 3; }
    color_image_t *channel_weights = color_image_new(width, height);                 // :597-598 (all ones without raw weighting)
    for (size_t i = 0; i < (size_t)3 * channel_weights->stride * height; i++) channel_weights->c1[i] = 1.0f;
    if (raw) rawWeighting(channel_weights, red_loc.size() > 0 ? red_loc[0] : 0, red_loc.size() > 1 ? red_loc[1] : 0, params.parameter<float>("raw_weight", "1.0"));   // :599-600

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
                    if (x0 < 0 || y0 < 0 || x0 + cw > g[c]->width || y0 + chh > g[c]->height) { std::cerr << "center / extent do not fit the ground truth" << std::endl; abort_ingest(); return 3; }
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

    // ---- the GPU threads started above have been uploading beside the decoding; what is left of the ingest is the normalisation ------------------
    const auto t_norm = std::chrono::steady_clock::now();
    auto ingest_failed = [&]() -> bool {
        bool any = false;
        for (int g = 0; g < ngpu; g++)
            if (!seq_err[g].empty()) { std::cerr << "GPU " << device_of(g) << ": " << seq_err[g] << std::endl; any = true; }
        return any;
    };
    if (todo.empty()) {                                                              // nothing to refine (-resume over finished jets): the statistics are not needed either
        std::lock_guard<std::mutex> l(ready_mu);
        ingest_abort = true;
    } else {
        std::unique_lock<std::mutex> l(ready_mu);                                    // the GPU threads uploaded their frames as they were decoded; the last frame's sums make the statistics
        ready_cv.wait(l, [&] { return stats_known || ingest_abort; });
    }
    ready_cv.notify_all();
    if (ingest_abort) {
        for (auto &t : up) t.join();
        const bool err = ingest_failed();
        release_sequences();
        if (err) return 4;
    }
    if (!todo.empty()) publish_normalization(params, stat_avg, stat_std);            // the slow_flow_img_norm_* parameters (variational_mt.cpp:71-84)
    const double normalize_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_norm).count();
    {
        std::ofstream infos((params.output + "config.cfg").c_str());                 // :684-688
        infos << "# SlowFlow variational estimation\n" << params;
    }
    const double ingest_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();

    const int batch = params.parameter<int>("gpu_batch", "32");                      // 1 .. 128: checked before anything was read
    const int F = 2 * ref + 1;
    const bool backward_forward_only = params.exists("method") && params.parameter("method") == "forward";   // :1019-1020
    const bool occ_on = params.parameter<bool>("slow_flow_occlusion_reasoning", "0");
    const bool out_occ = params.parameter<bool>("slow_flow_output_occlusions", "0");                          // :892
    // the per-alternation labels (:878-884: <output>tmp/frame_<n>_<alter>.png), with WRITE_FILES verbosity only
    const bool out_alt_occ = out_occ && occ_on && params.verbosity(WRITE_FILES) && params.parameter<int>("slow_flow_niter_alter", "1") > 1;
    if (out_occ) mkdirs(params.output + "occlusion/");                               // :676-677
    if (out_alt_occ || enable_dm) mkdirs(params.output + "tmp/");

    std::mutex io_mu;
    std::atomic<bool> failed(false);
    TaskPool out_pool(io_threads);
    std::vector<double> first_refine_s(ngpu, -1.0);                                  // when the first worker of a GPU started refining (seconds since the run began)

    // what happens to a finished window off the GPU worker's thread: flow * steps -> .flo, colour PNG, EPE / AAE, occlusion images
    auto emit = [&](std::shared_ptr<WindowResult> r) {
        Window &wd = todo[r->index];
        image_mul_scalar(r->wx, (float)steps);                                       // :908-909
        image_mul_scalar(r->wy, (float)steps);
        bool ok = writeFlowFile(wd.out.c_str(), r->wx, r->wy) == 0;
        const int fnum = (int)start + (int)wd.jet * steps * skip;
        if (ok && !wd.backward) {                                                    // the flow as a colour image next to the .flo (:913-925)
            ok = png_write((params.output + "frame_" + std::to_string(fnum) + ".png").c_str(), flowColorImg(r->wx, r->wy, 0));
            image_t **g = gt[wd.jet];
            if (g && g[0]->width == width && g[0]->height == height) {               // additive: error against the ground truth, reported in timings.json
                wd.epe = computeEPE(r->wx, r->wy, g[0], g[1]);
                wd.aae = computeAAE(r->wx, r->wy, g[0], g[1]);
            }
        }
        // the occlusion estimate of the forward window (:892-905; the reference writes a binary PXM named .pbm through OpenCV, here the
        // same 8-bit grey map as PGM) and, with WRITE_FILES verbosity, the labels after every alternation (:878-884 -> variational_mt.cpp:275-285)
        if (ok && r->occ) ok = writePGM((params.output + "occlusion/frame_" + std::to_string(fnum) + ".pgm").c_str(), r->occ, 1.0f, 127.5f) == 0;
        for (size_t a = 0; ok && a < r->alt_occ.size(); a++)
            ok = png_write((params.output + "tmp/frame_" + std::to_string(fnum) + "_" + std::to_string(a + 1) + ".png").c_str(), occlusion_image(r->alt_occ[a]));
        if (!ok) {
            std::lock_guard<std::mutex> l(io_mu);
            std::cerr << "cannot write the results of " << wd.out << std::endl;
            failed = true;
        }
    };

    auto worker = [&](const WorkerPlan &wp) {
        if (wp.lo >= wp.hi) return;
        const int device = device_of(wp.gpu);
        sfa_ctx *ctx = nullptr;
        if (sfa_ctx_create(device, &ctx) != SFA_OK) { std::lock_guard<std::mutex> l(io_mu); std::cerr << sfa_last_error(nullptr) << std::endl; failed = true; return; }
        sfa_ctx *ectx = nullptr;                                                     // the EpicFlow helper's context (deep_matching 1 only)
        if (enable_dm && sfa_ctx_create(device, &ectx) != SFA_OK) { std::lock_guard<std::mutex> l(io_mu); std::cerr << sfa_last_error(nullptr) << std::endl; failed = true; sfa_ctx_destroy(ctx); return; }
        (void)sfa_ctx_set_verbose(ctx, verbose_changes ? 1 : 0);
        {                                                                            // this GPU's frames are resident and normalised
            std::unique_lock<std::mutex> l(ready_mu);
            ready_cv.wait(l, [&] { return gpu_ready[wp.gpu] || ingest_abort; });
            if (ingest_abort) { failed = true; if (ectx) sfa_ctx_destroy(ectx); sfa_ctx_destroy(ctx); return; }
            if (first_refine_s[wp.gpu] < 0) first_refine_s[wp.gpu] = seconds_since_begin();
        }
        const int frame0 = gpu_frames[wp.gpu].lo;                                    // the GPU's resident sequence starts at this frame of the loaded range
        ParameterList tp(params);                                                    // one copy per thread (:708)
        // EpicFlow's interpolation of the matches of one window (:801-863 / :960-1004), per single frame step.  It runs on a helper thread with a context of its
        // own (the library's contexts are thread-compatible, not thread-safe), one batch ahead of the refinement: while the GPU refines batch n the host prepares
        // the initial flows of batch n + 1 (round 3 ran it inside the worker thread, in front of every batch -- ADVICE r2)
        struct EpicInit { image_t *wx = nullptr, *wy = nullptr; int rc = SFA_OK; };
        auto epic_init = [&](const Window &wd, sfa_ctx *ctx, EpicInit &out) {
            int rc = SFA_OK;
            image_t *iwx = nullptr, *iwy = nullptr;
            const int f = wd.jet * steps;
            {
                const int a = (int)params.sequence_start + f * skip, b = a + ref * skip;
                const color_image_t *un_ref = seq[wd.backward ? f + 2 * ref : f + ref];      // un_im[ref] / un_im_back[ref]: the window's reference frame, not normalised
                epic_matches mt;
                epic_edges ed;
                epic_params_t ep;
                epic_params_default(&ep);
                ep.pref_nn = 25; ep.nn = 160; ep.coef_kernel = 1.1f;         // :271-275
                // the resolution the interpolation runs at: cv::resize(img, img, Size(0, 0), dm_scale, dm_scale) makes cvRound(cols * dm_scale) x cvRound(rows * dm_scale)
                const int ew = dm_scale != 1 ? (int)std::lrint(width * dm_scale) : (int)width, eh = dm_scale != 1 ? (int)std::lrint(height * dm_scale) : (int)height;
                // ... while the reference allocates un_seq / wx and sizes the edge file with the TRUNCATED product (`color_image_new(width*dm_scale, height*dm_scale)`,
                // slow_flow.cpp:584, :801-806).  Where the two differ (1023 * 0.5: 512 against 511) the reference copies a 512-wide image into a 511-wide one -- no defined
                // result, and edge / match files laid out for either width would be misread by the other: refused by name
                if (dm_scale != 1 && (ew != (int)(width * dm_scale) || eh != (int)(height * dm_scale))) {
                    std::lock_guard<std::mutex> l(io_mu);
                    std::cerr << "dm_scale " << dm_scale << ": " << width << " x " << height << " frames give a rescaled image of " << ew << " x " << eh << " (cv::resize rounds) but buffers and edge files of "
                              << (int)(width * dm_scale) << " x " << (int)(height * dm_scale) << " (slow_flow.cpp:584 truncates): the reference has no defined result here; crop the frames to a multiple of 1 / dm_scale"
                              << std::endl;
                    rc = SFA_ERR_ARG;
                } else
                if (ew < 1 || eh < 1 || !read_matches((wd.backward ? matches_file(b, a) : matches_file(a, b)).c_str(), mt) ||
                    !read_edges((wd.backward ? edges_file(b) : edges_file(a)).c_str(), ew, eh, ed)) rc = SFA_ERR_ARG;
                else {
                    // the reference hands epic() un_seq: the frame after img.convertTo(CV_8U, norm), i.e. rounded to nearest and saturated to 0..255,
                    // 16-bit samples scaled by 1/255 first (slow_flow.cpp:472-474, :578-586) -- not the float frame the refinement reads
                    const int fi = wd.backward ? f + 2 * ref : f + ref;
                    // :571-573: GaussianBlur(sigma = 1 / sqrt(2 dm_scale), BORDER_REPLICATE) then resize by dm_scale, on the float frame (the pyramid's two
                    // operators: sfa_gaussian_blur, sfa_resize_linear_fx with cv::resize's fx / fy form of the coordinate scale)
                    color_image_t *small = nullptr;
                    if (dm_scale != 1) {
                        small = color_image_new(ew, eh);
                        std::vector<float> tmp((size_t)un_ref->stride * un_ref->height);
                        const float sigma = (float)(1 / sqrt(2 * dm_scale));
                        float *src3[3] = {un_ref->c1, un_ref->c2, un_ref->c3}, *dst3[3] = {small->c1, small->c2, small->c3};
                        for (int ch = 0; ch < 3 && rc == SFA_OK; ch++) {
                            rc = sfa_gaussian_blur(ctx, tmp.data(), src3[ch], un_ref->width, un_ref->height, un_ref->stride, sigma);
                            if (rc == SFA_OK) rc = sfa_resize_linear_fx(ctx, dst3[ch], ew, eh, small->stride, tmp.data(), un_ref->width, un_ref->height, un_ref->stride, dm_scale, dm_scale);
                        }
                    }
                    const color_image_t *un_src = small ? small : un_ref;
                    color_image_t *un8 = color_image_new(un_src->width, un_src->height);
                    {
                        const float norm = seq_maxval[fi] > 255 ? 1.0f / 255 : 1.0f;
                        const size_t n3 = (size_t)3 * un_src->stride * un_src->height;
                        for (size_t i = 0; i < n3; i++) {
                            const float v = nearbyintf(un_src->c1[i] * norm);              // cvRound: to nearest, ties to even
                            un8->c1[i] = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);       // saturate_cast<uchar>
                        }
                    }
                    if (small) color_image_delete(small);
                    color_image_t *lab = rgb_to_lab(un8);
                    color_image_delete(un8);
                    iwx = image_new(ew, eh); iwy = image_new(ew, eh);
                    image_erase(iwx); image_erase(iwy);
                    const int er = rc == SFA_OK ? epic(ctx, iwx, iwy, lab, mt, ed, &ep) : -1;
                    color_image_delete(lab);
                    if (er < 0) rc = rc == SFA_OK ? SFA_ERR_HIP : rc;
                    else if (er > 0) { image_erase(iwx); image_erase(iwy); }  // no usable match: start from zero like deep_matching 0
                    // :826-843: rescale flow.  fx, fy are quotients of ints; only when fx != 1 is the field resized (cv::resize to the frame's size: the ratio form)
                    const float fx = (float)((int)width / ew), fy = (float)((int)height / eh);
                    if (rc == SFA_OK && fx != 1) {
                        image_t *fwx = image_new(width, height), *fwy = image_new(width, height);
                        rc = sfa_resize_linear(ctx, fwx->data, width, height, fwx->stride, iwx->data, ew, eh, iwx->stride);
                        if (rc == SFA_OK) rc = sfa_resize_linear(ctx, fwy->data, width, height, fwy->stride, iwy->data, ew, eh, iwy->stride);
                        image_delete(iwx); image_delete(iwy);
                        iwx = fwx; iwy = fwy;
                    } else if (rc == SFA_OK && (ew != (int)width || eh != (int)height)) {
                        // the reference would hand a field of the reduced size to a solver of the frames' size here (1 < width ratio < 2): it has no defined result
                        std::lock_guard<std::mutex> l(io_mu);
                        std::cerr << "dm_scale " << dm_scale << ": the frames are " << width << " wide, the interpolation " << ew << " -- the reference rescales by the integer ratio "
                                  << (int)width / ew << " (slow_flow.cpp:827) and would keep the reduced field; use a dm_scale of 1 / n" << std::endl;
                        rc = SFA_ERR_ARG;
                    }
                    image_mul_scalar(iwx, fx / steps);                      // :842-843
                    image_mul_scalar(iwy, fy / steps);
                    if (rc == SFA_OK && params.verbosity(WRITE_FILES) && !wd.backward)     // :845-858
                        png_write((params.output + "tmp/frame_" + std::to_string(a) + "_INIT.png").c_str(), flowColorImg(iwx, iwy, 0));
                }
                if (rc != SFA_OK) { std::lock_guard<std::mutex> l(io_mu); std::cerr << "EpicFlow initialisation of jet " << wd.jet << " failed" << std::endl; }
            }
            out.wx = iwx; out.wy = iwy; out.rc = rc;
        };
        // forward and backward windows share one lockstep job (the channel weights are per window) unless the backward solver runs with
        // different parameters ("method forward", :1019-1020): then one pass per direction
        for (int dirpass = 0; dirpass < (backward_forward_only ? 2 : 1) && !failed; dirpass++) {
            std::vector<size_t> mine;
            for (size_t i = wp.lo; i < wp.hi; i++)
                if (!backward_forward_only || (int)todo[i].backward == dirpass) mine.push_back(i);
            if (mine.empty()) continue;
            sfa_params sp = sfa_params_from_cfg(tp, dirpass == 1 ? backward_forward_only : false);
            // one resident job for every full batch of this worker; a shorter last batch gets its own
            sfa_job *job = nullptr;
            int job_nb = 0;
            int rc = SFA_OK;
            auto batch_inits = [&](size_t b0) {
                const int nb = (int)std::min((size_t)batch, mine.size() - b0);
                std::vector<EpicInit> v(nb);
                for (int e = 0; e < nb; e++) epic_init(todo[mine[b0 + e]], ectx, v[e]);
                return v;
            };
            std::future<std::vector<EpicInit>> next_inits;
            if (enable_dm) next_inits = std::async(std::launch::async, batch_inits, (size_t)0);
            for (size_t b0 = 0; b0 < mine.size() && rc == SFA_OK && !failed; b0 += batch) {
                const int nb = (int)std::min((size_t)batch, mine.size() - b0);
                std::vector<EpicInit> inits;
                if (enable_dm) {
                    inits = next_inits.get();
                    if (b0 + batch < mine.size()) next_inits = std::async(std::launch::async, batch_inits, b0 + batch);
                }
                const auto t0 = std::chrono::steady_clock::now();
                if (!job || job_nb != nb) {
                    if (job) sfa_job_destroy(job);
                    job = nullptr;
                    rc = sfa_job_create(ctx, &sp, width, height, nb, &job);
                    job_nb = nb;
                    if (rc == SFA_OK && out_alt_occ) rc = sfa_job_keep_alternation_occlusions(job, 1);
                }
                for (int e = 0; e < nb && rc == SFA_OK; e++) {
                    const Window &wd = todo[mine[b0 + e]];
                    const int f = wd.jet * steps;
                    // the window's frames in the loaded sequence: forward seq[f + k]; backward seq_back[frames - 1 - f - 3*steps + k] = seq[f + 3*steps - k] (:590-591, :721-724)
                    std::vector<int> idx(F);
                    for (int k = 0; k < F; k++) idx[k] = (wd.backward ? f + 3 * steps - k : f + k) - (int)start_f - frame0;
                    const float *chw[3] = {channel_weights->c1, channel_weights->c2, channel_weights->c3};
                    image_t *iwx = enable_dm ? inits[e].wx : nullptr, *iwy = enable_dm ? inits[e].wy : nullptr;
                    if (enable_dm && inits[e].rc != SFA_OK) rc = inits[e].rc;
                    // only the forward solver gets the channel weights (:876 vs :1018); without raw weighting they are all ones (:597-598), which is
                    // what a NULL pointer means to the library (x * 1.0f is exact: same bits, three planes less to read per pixel)
                    if (rc == SFA_OK)
                        rc = sfa_job_upload_resident(job, e, seq_dev[wp.gpu], idx.data(), F, iwx ? iwx->data : nullptr, iwy ? iwy->data : nullptr, seq[start_f]->stride,
                                                     (wd.backward || !raw) ? nullptr : chw);
                }
                for (EpicInit &ei : inits) { if (ei.wx) image_delete(ei.wx); if (ei.wy) image_delete(ei.wy); }
                if (rc == SFA_OK) rc = sfa_job_run(job);
                std::vector<std::shared_ptr<WindowResult>> results;
                for (int e = 0; e < nb && rc == SFA_OK; e++) {
                    const Window &wd = todo[mine[b0 + e]];
                    std::shared_ptr<WindowResult> r(new WindowResult());
                    r->index = mine[b0 + e];
                    r->wx = image_new(width, height); r->wy = image_new(width, height);
                    image_erase(r->wx); image_erase(r->wy);
                    float change[2];
                    rc = sfa_job_download(job, e, r->wx->data, r->wy->data, r->wx->stride, change);
                    if (rc == SFA_OK && out_occ && !wd.backward) {                   // getOcclusions() of the forward solver (:893)
                        r->occ = image_new(width, height);
                        image_erase(r->occ);
                        rc = sfa_job_download_occlusions(job, e, r->occ->data, r->occ->stride);
                    }
                    for (int a = 1; a < sp.niter_alter && rc == SFA_OK && out_alt_occ && !wd.backward; a++) {
                        image_t *o = image_new(width, height);
                        image_erase(o);
                        r->alt_occ.push_back(o);
                        rc = sfa_job_download_alternation_occlusions(job, e, a, o->data, o->stride);
                    }
                    results.push_back(r);
                }
                const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rc == SFA_OK)
                    for (auto &r : results) {
                        todo[r->index].seconds = secs / nb;
                        todo[r->index].gpu = device;
                        out_pool.submit([&emit, r] { emit(r); });
                    }
                std::lock_guard<std::mutex> l(io_mu);
                if (rc != SFA_OK) { std::cerr << "GPU " << device << ": " << sfa_last_error(ctx) << std::endl; failed = true; }
                else for (int e = 0; e < nb; e++) {
                    const Window &wd = todo[mine[b0 + e]];
                    const int f = wd.jet * steps;
                    std::cout << (wd.backward ? "Backward" : "Forward") << " flow from frame " << start + f * skip << " to " << start + f * skip + steps * skip
                              << " finished! (GPU " << wp.gpu << " = device " << device << ", " << secs / nb << " s per window in a batch of " << nb << ")" << std::endl;
                }
            }
            if (next_inits.valid())                                                  // left early: the batch prepared ahead is not used
                for (EpicInit &ei : next_inits.get()) { if (ei.wx) image_delete(ei.wx); if (ei.wy) image_delete(ei.wy); }
            if (job) sfa_job_destroy(job);
        }
        if (ectx) sfa_ctx_destroy(ectx);
        sfa_ctx_destroy(ctx);
    };
    const auto t_compute = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (const WorkerPlan &wp : plan) th.emplace_back(worker, wp);
    for (auto &t : th) t.join();
    const double compute_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_compute).count();
    for (auto &t : up) if (t.joinable()) t.join();
    if (ingest_failed()) failed = true;
    out_pool.wait_all();
    const double total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();

    // the gathered per-window timings (the only cross-GPU exchange of the path) and where the wall time of the run went
    {
        std::ofstream tj((params.output + "timings.json").c_str());
        tj << "[";
        for (size_t i = 0; i < todo.size(); i++)
            tj << (i ? "," : "") << "\n  {\"jet\": " << todo[i].jet << ", \"direction\": \"" << (todo[i].backward ? "backward" : "forward") << "\", \"gpu\": " << todo[i].gpu
               << ", \"seconds\": " << todo[i].seconds << ", \"flo\": \"" << todo[i].out << "\""
               << (todo[i].epe >= 0 ? ", \"epe\": " + std::to_string(todo[i].epe) + ", \"aae\": " + std::to_string(todo[i].aae) : string()) << "}";
        tj << "\n]\n";
        std::ofstream rj((params.output + "run.json").c_str());
        rj << "{\"windows\": " << todo.size() << ", \"gpus\": " << ngpu << ", \"streams\": " << streams << ", \"batch\": " << batch << ", \"io_threads\": " << io_threads
           << ", \"decode_seconds\": " << decode_seconds << ", \"normalize_seconds\": " << normalize_seconds << ", \"ingest_seconds\": " << ingest_seconds << ", \"refine_seconds\": " << compute_seconds << ", \"total_seconds\": " << total_seconds;
        // per GPU: the frames it was sent (its windows' frames + halo), when the last of them had arrived, when its frames were normalised and when its first worker
        // began to refine -- all in seconds since the run began.  The whole sequence is "sequence_bytes"
        double last_upload = 0, first_refine = -1;
        rj << ", \"sequence_bytes\": " << (double)n_loaded * 3 * width * height * sizeof(float) << ", \"per_gpu\": [";
        for (int g = 0; g < ngpu; g++) {
            rj << (g ? ", " : "") << "{\"gpu\": " << g << ", \"device\": " << device_of(g) << ", \"frames\": [" << gpu_frames[g].lo << ", " << gpu_frames[g].hi << "], \"upload_bytes\": " << upload_bytes[g]
               << ", \"upload_done_s\": " << upload_done_s[g] << ", \"ready_s\": " << ready_s[g] << ", \"first_refine_s\": " << first_refine_s[g] << "}";
            last_upload = std::max(last_upload, upload_done_s[g]);
            if (first_refine_s[g] >= 0 && (first_refine < 0 || first_refine_s[g] < first_refine)) first_refine = first_refine_s[g];
        }
        rj << "], \"first_refine_s\": " << first_refine << ", \"last_upload_done_s\": " << last_upload << "}\n";
    }
    release_sequences();
    for (unsigned f = start_f; f < end_f; f++) color_image_delete(seq[f]);
    for (auto g : gt) if (g) { image_delete(g[0]); image_delete(g[1]); free(g); }
    color_image_delete(channel_weights);
    return failed ? 5 : 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { usage(); return 1; }
    ParameterList params;
    setDefault(params);
    if (argv[1][0] != '-' && file_exists(argv[1])) params.read(argv[1]);
    else { std::cerr << "Couldn't find " << argv[1] << "!" << std::endl; return -1; }

    bool overwrite_output = false;
    RunOptions opt;
    int selected_fr = -1;
    for (int i = 1; i < argc; i++) {                                                 // slow_flow.cpp:168-193
        const char *a = argv[i];
        if (a[0] != '-') continue;
        if (!strcmp(a, "-h") || !strcmp(a, "-help")) usage();
        else if (!strcmp(a, "-overwrite")) overwrite_output = true;
        else if (!strcmp(a, "-resume")) opt.resume_frame = true;
        else if (!strcmp(a, "-deep_settings") && i + 1 < argc) i++;
        else if (!strcmp(a, "-threads") && i + 1 < argc) params.insert("threads", argv[++i], true);
        else if (!strcmp(a, "-fr") && i + 1 < argc) selected_fr = atoi(argv[++i]);
        else if (!strcmp(a, "-jet") && i + 1 < argc) { opt.selected_jet = atoi(argv[++i]); opt.resume_frame = true; }
        else { fprintf(stderr, "unknown argument %s", a); usage(); }
    }
    const bool raw = params.exists("raw") && params.parameter<bool>("raw");
    const int demosaicing = params.parameter<int>("raw_demosaicing", "0");
    if (raw && demosaicing != 0 && demosaicing != 2) {
        std::cerr << "raw_demosaicing 1 (Hamilton-Adams, P. Getreuer's dmha) is third-party code absent from the reference tree: use raw_demosaicing 0 "
                     "(the reference's own bilinear / green-ratio routine), 2 (OpenCV's 8-bit bilinear conversion, restated) or provide demosaiced frames with raw 0" << std::endl;
        return 2;
    }

    const int steps = params.parameter<int>("slow_flow_S") - 1;                      // :208
    const int max_fps = params.parameter<int>("max_fps", "1");
    const int jet_fps = params.exists("jet_fps") ? params.parameter<int>("jet_fps") : max_fps;
    int skip = (int)((1.0f * max_fps) / jet_fps);                                    // :220
    const bool sintel = params.parameter<bool>("sintel", "0"), subframes = params.parameter<bool>("subframes", "0");
    unsigned start = params.sequence_start;
    const size_t cut = params.file.find_last_of('/') + 1;
    string sequence_path = params.file.substr(0, cut), format = params.file.substr(cut);
    if (sequence_path.empty() || params.output.empty()) { std::cerr << "cfg needs 'file' and 'output'" << std::endl; return -1; }
    if (sintel && !subframes) start *= 1000;
    params.sequence_start = start;

    if (!opt.resume_frame && !overwrite_output) {                                    // never overwrite a results folder (:254-265)
        string np = params.output;
        if (np.back() == '/') np.pop_back();
        const string base = np;
        int num = 1;
        while (file_exists(np)) { std::cerr << np << " already exists!" << std::endl; np = base + "_" + std::to_string(num++); }
        params.output = np;
    }
    if (params.output.back() != '/') params.output += "/";

    // ---- adaptive frame rates (:277-357): adaptiveFR.dat (next to the program: the reference's SOURCE_PATH) names the target quantile and the
    //      low-rate factor, <sequence>/quantil.dat (written by the adaptiveFR program) holds the sequence's flow-magnitude quantile ------------
    bool adaptive = false;
    double quantil = 1.0, hfr_quantil = 2.0;
    int lfr_factor = 4;
    AdaptiveRates rates = {1, 4};
    const string adfr = params.exists("adaptive_fr_file") ? params.parameter("adaptive_fr_file") : exe_dir() + "/adaptiveFR.dat";
    if (file_exists(adfr)) {
        std::ifstream f(adfr.c_str());
        string line;
        while (std::getline(f, line)) {
            const size_t tab = line.find('\t');
            if (tab == string::npos) continue;
            const string key = line.substr(0, tab), val = line.substr(tab + 1);
            if (key == "opt_hfr_quantil") hfr_quantil = atof(val.c_str());
            if (key == "opt_lfr_rate") lfr_factor = (int)atof(val.c_str());
        }
        adaptive = params.parameter<bool>("adaptive", "0");
    }
    // :197-200, :306: the bound on the flow DeepMatching may find -- it restricts the matcher's search (outside this build) and halves dm_scale beyond 150 px (:397-402)
    double max_flow = params.exists("max_flow") ? std::max(5.0f, params.parameter<float>("max_flow")) : 50.0, orig_max_flow = 0;
    const double scale_main = params.parameter<float>("scale", "1.0");
    const string qfstr = sequence_path + "/quantil.dat";
    if (!params.exists("max_flow") && file_exists(qfstr)) {
        std::ifstream f(qfstr.c_str());
        string line;
        std::getline(f, line);
        quantil = atof(line.c_str());
        // :315-319: the flow bound is three times the file's second line (the maximum) if there is one, else three times the quantile
        string line2;
        orig_max_flow = 3.0 * (std::getline(f, line2) ? atof(line2.c_str()) : quantil);
        if (!adaptive) max_flow = std::max(5.0, orig_max_flow * scale_main * steps * skip);   // :354 (ref = steps)
        if (adaptive) {
            const int keyframes = (int)(params.parameter<float>("max_fps") / params.parameter<float>("ref_fps"));   // :324
            rates = adaptive_rates(quantil, hfr_quantil, lfr_factor, keyframes, steps);
            if (keyframes == 0) std::cout << rates.hfr_rate << " " << rates.lfr_rate << std::endl;
            else std::cout << "hfr_rate " << rates.hfr_rate << std::endl << "lfr_rate " << rates.lfr_rate << std::endl;
        }
        // without `adaptive` the quantile only bounds DeepMatching's search radius (:353-355): outside this build
    } else
        adaptive = false;

    int start_fr = 0, end_fr = adaptive ? 2 : 1;                                     // :359-364
    if (selected_fr >= 0) { start_fr = selected_fr; end_fr = selected_fr + 1; }
    int rc_all = 0;
    for (int adFR = start_fr; adFR < end_fr; adFR++) {                               // :367-399
        ParameterList adaptCfg(params);
        if (adaptive) {
            const int rate = adFR == 0 ? rates.hfr_rate : rates.lfr_rate;
            adaptCfg.output += adFR == 0 ? "high_fr/" : "low_fr/";
            adaptCfg.insert("jet_fps", std::to_string(max_fps / rate), true);        // jet estimation fps
            skip = rate;
            max_flow = std::max(5.0, orig_max_flow * scale_main * steps * rate);     // :382, :392
        }
        {   // :396-402: smaller resolution for deep matching
            double dm_scale = params.parameter<float>("dm_scale", "1.0");
            if (params.parameter<bool>("deep_matching") && max_flow > 150) { dm_scale = 0.5 * dm_scale; max_flow = std::max(5.0, 0.5 * max_flow); }   // (:401: the bound at the halved size)
            std::ostringstream o;
            o.precision(17);
            o << dm_scale;
            adaptCfg.insert("dm_scale_run", o.str(), true);
        }
        const int rc = run_sequence(adaptCfg, sequence_path, format, skip, opt);
        if (rc != 0) rc_all = rc;
    }
    std::cout << (rc_all ? "Failed!" : "Done!") << std::endl;
    return rc_all;
}
