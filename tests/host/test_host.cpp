// CPU-side checks of the host mirror (no GPU needed): ParameterList syntax and accessors, the cfg -> sfa_params
// mapping of Variational_MT, the .flo wire format, the PPM/PGM/PFM/PNG loaders, ingest, flow colour coding and EPE/AAE.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "flow_vis.h"
#include "image.h"
#include "ingest.h"
#include "io.h"
#include "parameter_list.h"
#include "png.h"
#include "tiff.h"
#include "shard.h"
#include "variational_mt.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main(int argc, char **argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    // ---- ParameterList --------------------------------------------------------------------------------------
    {
        const std::string cfg = tmp + "/t.cfg";
        std::ofstream f(cfg.c_str());
        f << "# comment line\n"
          << "verbose\t\t1000\t\t# console\n"
          << "file\t\t/seq/dir/frame_%04i.ppm\t# sequence\n"
          << "output\t/out/dir\n"
          << "Jets\t\t12\n"
          << "start\t7\n"
          << "center\t100,50\n"
          << "slow_flow_S\t3\t\t\t\t# frames\n"
          << "slow_flow_alpha\t4.5\n"
          << "slow_flow_method\tforward\n"
          << "slow_flow_layers\t(1,3,5)\t# grid\n"
          << "missing_value\n"
          << "slow_flow_robust_color\t2\n";
        f.close();
        ParameterList p(cfg);
        CHECK(p.file == "/seq/dir/frame_%04i.ppm" && p.output == "/out/dir" && p.Jets == 12 && p.sequence_start == 7);
        CHECK(p.center.x == 100 && p.center.y == 50);
        CHECK(p.verbosity(VER_CMD) && !p.verbosity(VER_IN_GT) && !p.verbosity(99));
        CHECK(p.parameter<int>("slow_flow_S") == 3);
        CHECK(std::fabs(p.parameter<float>("slow_flow_alpha") - 4.5f) < 1e-6f);
        CHECK(p.parameter("slow_flow_method") == "forward");
        CHECK(!p.exists("missing_value") && !p.exists("file"));
        CHECK(p.parameter<int>("nope", "42") == 42 && p.parameter<bool>("nope", "0") == false && p.parameter<bool>("nope", "1") == true);
        CHECK(p.parameter<int>("nope") == 0);                       // error message + 0
        CHECK(p.experiments() == 3 && p.parameter<int>("slow_flow_layers") == 1);
        CHECK(p.nextExp() && p.parameter<int>("slow_flow_layers") == 3);
        CHECK(p.nextExp() && p.parameter<int>("slow_flow_layers") == 5 && !p.hasNextExp());
        p.reset();
        CHECK(p.parameter<int>("slow_flow_layers") == 1);
        p.insert("final", "0", true);
        p.setParameter<int>("final", 1);
        CHECK(p.parameter<int>("final") == 1);
        p.insert("x", "1");
        p.insert("x", "2");                                          // append -> a 2-valued grid
        CHECK(p.experiments() == 6);
        CHECK(p.splitParameter<int>("nope", "1,0").size() == 2 && p.splitParameter<int>("nope", "1,0")[0] == 1);
        ParameterList q(p);                                          // copy per thread (slow_flow.cpp:708)
        q.setParameter<int>("final", 0);
        CHECK(p.parameter<int>("final") == 1 && q.parameter<int>("final") == 0);
        // the 6-digit publish of normalize (variational_mt.cpp:71-84)
        std::stringstream s;
        s << 127.3464558342057;
        CHECK(s.str() == "127.346");
    }
    // ---- cfg -> sfa_params --------------------------------------------------------------------------------------
    {
        ParameterList p;
        const char *kv[][2] = {{"slow_flow_S", "3"}, {"slow_flow_niter_outer", "10"}, {"slow_flow_niter_inner", "1"}, {"slow_flow_niter_solver", "30"},
                               {"slow_flow_thres_outer", "1e-5"}, {"slow_flow_thres_inner", "1e-5"}, {"slow_flow_sor_omega", "1.9"}, {"slow_flow_alpha", "4.0"},
                               {"slow_flow_gamma", "6.0"}, {"slow_flow_delta", "1.0"}, {"slow_flow_robust_color", "1"}, {"slow_flow_robust_color_eps", "0.001"},
                               {"slow_flow_robust_color_truncation", "0.5"}, {"slow_flow_robust_reg", "2"}, {"slow_flow_robust_reg_eps", "0.05"},
                               {"slow_flow_robust_reg_truncation", "0.5"}, {"slow_flow_omega_1", "2"}, {"slow_flow_omega_0", "0"}, {"slow_flow_layers", "5"},
                               {"slow_flow_p_scale", "0.9"}, {"slow_flow_img_norm_avg_2", "127.368"}, {"slow_flow_img_norm_std_3", "0.177831"}, {"16bit", "1"}};
        for (auto &e : kv) p.insert(e[0], e[1], true);
        sfa_params sp = sfa_params_from_cfg(p, false);
        CHECK(sp.S == 3 && sp.niter_alter == 1 && sp.smoothing == 0 && sp.dataterm_norm == 1 && sp.one_direction == 0);   // library defaults of variational_mt.cpp
        CHECK(sp.robust_grad.id == 1 && sp.robust_grad.eps == 0.001f && sp.robust_reg.id == 2);                             // grad falls back to color
        CHECK(sp.rho[0] == 1.0f && sp.rho[1] == 1.0f && sp.omega[0] == 0.0f && sp.omega[1] == 2.0f);
        CHECK(sp.norm_avg[0] == 0.0f && std::fabs(sp.norm_avg[1] - 127.368f) < 1e-4f && sp.norm_std[0] == 1.0f && std::fabs(sp.norm_std[2] - 0.177831f) < 1e-7f);
        CHECK(sp.hbit == 1 && sp.occlusion_reasoning == 0 && sp.layers == 5 && sp.presmooth_sigma == 0.0f);
        CHECK(sp.occlusion_penalty == 1.0f && sp.occlusion_alpha == 0.5f && sp.niter_graphc == 10);                        // variational_mt.cpp:182,189-190
        p.insert("slow_flow_method", "forward", true);
        CHECK(sfa_params_from_cfg(p, false).one_direction == 1);
    }
    // ---- .flo and image files ---------------------------------------------------------------------------------------
    {
        image_t *u = image_new(5, 3), *v = image_new(5, 3);
        for (int y = 0; y < 3; y++) for (int x = 0; x < 5; x++) { u->data[y * u->stride + x] = x + 10 * y + 0.25f; v->data[y * v->stride + x] = -(x * 0.5f) - y; }
        CHECK(u->stride == 8);
        const std::string flo = tmp + "/t.flo";
        CHECK(writeFlowFile(flo.c_str(), u, v) == 0);
        std::ifstream f(flo.c_str(), std::ios::binary);
        float tag; int w, h; float first[2];
        f.read((char *)&tag, 4); f.read((char *)&w, 4); f.read((char *)&h, 4); f.read((char *)first, 8);
        CHECK(tag == 202021.25f && w == 5 && h == 3 && first[0] == 0.25f && first[1] == -0.0f);
        f.seekg(0, std::ios::end);
        CHECK((long)f.tellg() == 12 + 5 * 3 * 8);
        image_t **r = readFlowFile(flo.c_str());
        CHECK(r && r[0]->width == 5 && r[1]->data[2 * r[1]->stride + 4] == -4.0f && r[0]->data[1 * r[0]->stride + 3] == 13.25f);
        // 16-bit PPM, 8-bit PGM
        const std::string ppm = tmp + "/t.ppm";
        { std::ofstream o(ppm.c_str(), std::ios::binary); o << "P6\n# c\n2 1\n65535\n"; const unsigned char d[12] = {0x01, 0x02, 0, 5, 0xff, 0xff, 0, 0, 0, 1, 0x10, 0}; o.write((const char *)d, 12); }
        int maxv = 0;
        color_image_t *c = color_image_load(ppm.c_str(), &maxv);
        CHECK(c && maxv == 65535 && c->c1[0] == 258.0f && c->c2[0] == 5.0f && c->c3[0] == 65535.0f && c->c3[1] == 4096.0f);
        const std::string pgm = tmp + "/t.pgm";
        { std::ofstream o(pgm.c_str(), std::ios::binary); o << "P5 3 1 255\n"; const unsigned char d[3] = {7, 8, 9}; o.write((const char *)d, 3); }
        color_image_t *gcol = color_image_load(pgm.c_str(), &maxv);
        CHECK(gcol && maxv == 255 && gcol->c1[2] == 9.0f && gcol->c2[1] == 8.0f && gcol->c3[0] == 7.0f && gcol->c2 == gcol->c1 + gcol->stride);
    }
    // ---- without a GPU the drop-in must fail loudly ---------------------------------------------------------------------
    if (sfa_device_count() == 0) {
        bool threw = false;
        try {
            ParameterList p;
            color_image_t *im = color_image_new(8, 8);
            color_image_erase(im);
            color_image_t *seq[1] = {im};
            normalize(seq, 1, p);
        } catch (const std::runtime_error &e) { threw = std::string(e.what()).find("no HIP device") != std::string::npos; }
        CHECK(threw);
    }
    // ---- ingest: raw weighting, Bayer demosaicing, crop (utils.cpp:1241-1374, slow_flow.cpp:533-536) ----------------------
    {
        const int w = 10, h = 7;
        for (int red_y = 0; red_y < 2; red_y++)
            for (int red_x = 0; red_x < 2; red_x++) {
                color_image_t *cw = color_image_new(w, h);
                rawWeighting(cw, red_x, red_y, 2.0f);
                // a mosaic sampled from a constant colour (R,G,B) = (40, 100, 20): the measured channel of every pixel carries the
                // weight, the two others (3 - weight) / 2; the demosaiced image is that colour again
                image_t *mosaic = image_new(w, h);
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) {
                        const size_t o = (size_t)y * cw->stride + x;
                        const float r = cw->c1[o], g = cw->c2[o], b = cw->c3[o];
                        CHECK(std::fabs(r + g + b - 3.0f) < 1e-6f);
                        CHECK((r == 2.0f) + (g == 2.0f) + (b == 2.0f) == 1);
                        const bool is_red = (x % 2 == red_x) && (y % 2 == red_y), is_blue = (x % 2 != red_x) && (y % 2 != red_y);
                        // (for red_y == 1 the reference's weighting swaps green and blue/red columns relative to its own demosaicer,
                        //  utils.cpp:1345-1346,1360-1361: restated as written, checked against the Bayer layout for red_y == 0 only)
                        if (red_y == 0) CHECK(is_red ? r == 2.0f : is_blue ? b == 2.0f : g == 2.0f);
                        mosaic->data[(size_t)y * mosaic->stride + x] = is_red ? 40.0f : is_blue ? 20.0f : 100.0f;
                    }
                color_image_t *rgb = color_image_new(w, h);
                bayer2rgbGR(mosaic, rgb, red_x, red_y);
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) {
                        const size_t o = (size_t)y * rgb->stride + x;
                        CHECK(std::fabs(rgb->c1[o] - 40.0f) < 1e-4f && std::fabs(rgb->c2[o] - 100.0f) < 1e-4f && std::fabs(rgb->c3[o] - 20.0f) < 1e-4f);
                    }
                image_delete(mosaic); color_image_delete(rgb); color_image_delete(cw);
            }
        color_image_t *cw = color_image_new(4, 4);
        rawWeighting(cw, 0, 0, 7.0f);                                  // clamped to [0, 3]
        CHECK(cw->c1[0] == 3.0f && cw->c2[0] == 0.0f && cw->c3[0] == 0.0f);
        color_image_delete(cw);
        color_image_t *img = color_image_new(20, 12);
        for (int y = 0; y < 12; y++)
            for (int x = 0; x < 20; x++) { img->c1[y * img->stride + x] = (float)(100 * y + x); img->c2[y * img->stride + x] = 1; img->c3[y * img->stride + x] = 2; }
        color_image_t *part = color_image_crop(img, 10, 6, 8, 4);     // rows 4..7, columns 6..13
        CHECK(part && part->width == 8 && part->height == 4 && part->c1[0] == 406.0f && part->c1[3 * part->stride + 7] == 713.0f && part->c3[5] == 2.0f);
        CHECK(color_image_crop(img, 2, 2, 8, 4) == nullptr);          // does not fit
        if (part) color_image_delete(part);
        color_image_delete(img);
    }
    // ---- PNG in and out (png.h) -----------------------------------------------------------------------------------------
    {
        for (int depth = 8; depth <= 16; depth += 8)
            for (int ch = 1; ch <= 3; ch += 2) {
                png_image a;
                a.width = 13; a.height = 7; a.channels = ch; a.depth = depth;
                a.samples.resize((size_t)13 * 7 * ch);
                for (size_t i = 0; i < a.samples.size(); i++) a.samples[i] = (uint16_t)((i * 2654435761u >> 7) & (depth == 16 ? 0xffff : 0xff));
                const std::string f = tmp + "/rt.png";
                png_image b;
                CHECK(png_write(f.c_str(), a) && png_read(f.c_str(), b));
                CHECK(b.width == 13 && b.height == 7 && b.channels == ch && b.depth == depth && b.samples == a.samples);
                int maxv = 0;
                color_image_t *c = color_image_load(f.c_str(), &maxv);             // same loader entry as the PPM frames
                CHECK(c && maxv == (depth == 16 ? 65535 : 255) && c->width == 13 && c->height == 7);
                if (c) {
                    bool same = true;
                    for (int y = 0; y < 7; y++) for (int x = 0; x < 13; x++) for (int k = 0; k < 3; k++)
                        same &= (k == 0 ? c->c1 : k == 1 ? c->c2 : c->c3)[y * c->stride + x] == (float)a.samples[(y * 13 + x) * ch + (ch == 3 ? k : 0)];
                    CHECK(same);
                    color_image_delete(c);
                }
            }
        // files written by an independent encoder (the Python side of this test: all filters, palette, alpha, sub-byte depths)
        std::ifstream list((tmp + "/png_cases.txt").c_str());
        std::string name;
        int w, h, ch, depth, cases = 0;
        while (list >> name >> w >> h >> ch >> depth) {
            png_image b;
            CHECK(png_read((tmp + "/" + name + ".png").c_str(), b));
            std::ifstream rawf((tmp + "/" + name + ".raw").c_str(), std::ios::binary);
            std::vector<uint16_t> want((size_t)w * h * ch);
            rawf.read((char *)want.data(), want.size() * 2);
            CHECK(b.width == w && b.height == h && b.channels == ch && b.depth == depth && b.samples == want);
            cases++;
        }
        if (list.is_open()) CHECK(cases >= 10);
        { std::ofstream o((tmp + "/bad.png").c_str(), std::ios::binary); o << "\x89PNG\r\n\x1a\nnot really"; }
        png_image bad;
        CHECK(!png_read((tmp + "/bad.png").c_str(), bad) && !png_read((tmp + "/missing.png").c_str(), bad));
    }
    // ---- flow colour coding and error measures (flow_vis.h) -----------------------------------------------------------------
    {
        unsigned char pix[3];
        computeColor(0.0f, 0.0f, pix);   CHECK(pix[0] == 255 && pix[1] == 255 && pix[2] == 255);       // no motion: white
        computeColor(1.0f, 0.0f, pix);   CHECK(pix[0] == 255 && pix[1] == 0 && pix[2] == 0);           // right: red (hue 0 of the wheel)
        computeColor(-1.0f, 0.0f, pix);  CHECK(pix[0] == 0 && pix[1] > 200 && pix[2] == 255);          // left: cyan-blue
        computeColor(0.0f, 1.0f, pix);   CHECK(pix[0] > 200 && pix[1] > 200 && pix[2] == 0);           // down: yellow
        computeColor(0.0f, -1.0f, pix);  CHECK(pix[0] < 120 && pix[1] == 0 && pix[2] == 255);          // up: blue-violet
        computeColor(0.5f, 0.0f, pix);   CHECK(pix[0] == 255 && pix[1] == 127 && pix[2] == 127);       // half radius: half saturation
        computeColor(2.0f, 0.0f, pix);   CHECK(pix[0] == 191 && pix[1] == 0 && pix[2] == 0);           // beyond the radius: darkened
        image_t *u = image_new(6, 4), *v = image_new(6, 4), *gu = image_new(6, 4), *gv = image_new(6, 4);
        for (int y = 0; y < 4; y++) for (int x = 0; x < 6; x++) {
            u->data[y * u->stride + x] = 3.0f; v->data[y * v->stride + x] = 0.0f;
            gu->data[y * gu->stride + x] = 0.0f; gv->data[y * gv->stride + x] = 4.0f;
        }
        CHECK(std::fabs(computeEPE(u, v, gu, gv) - 5.0) < 1e-12);
        CHECK(std::fabs(computeAAE(u, v, gu, gv) - std::acos(1.0 / (std::sqrt(10.0) * std::sqrt(17.0)))) < 1e-6);
        gu->data[0] = 2e9f;                                             // unknown ground truth there: skipped, mean unchanged
        u->data[1] = 0.0f; v->data[1] = 4.0f;                           // one exact pixel among the 23 counted
        CHECK(std::fabs(computeEPE(u, v, gu, gv) - 5.0 * 22 / 23) < 1e-12);
        image_t *m = image_new(6, 4);
        for (int i = 0; i < m->stride * 4; i++) m->data[i] = 0.0f;
        m->data[1] = 1.0f;
        CHECK(computeEPE(u, v, gu, gv, m) == 0.0 && computeAAE(u, v, gu, gv, m) < 1e-3);
        image_t *other = image_new(5, 4);
        CHECK(computeEPE(other, other, gu, gv) == -1 && computeAAE(other, other, gu, gv) == -1);
        u->data[1] = 3.0f; v->data[1] = 0.0f;
        u->data[2 * u->stride + 3] = 1e4f;                              // outside the frame: black, and not part of the radius
        v->data[3] = NAN;
        png_image col = flowColorImg(u, v);
        CHECK(col.width == 6 && col.height == 4 && col.channels == 3 && col.depth == 8);
        CHECK(col.samples[0] == 255 && col.samples[1] == 0 && col.samples[2] == 0);                     // |(3,0)| is the radius: pure red
        CHECK(col.samples[(2 * 6 + 3) * 3] == 0 && col.samples[(2 * 6 + 3) * 3 + 2] == 0 && col.samples[3 * 3] == 0 && col.samples[3 * 3 + 1] == 0);
        png_image half = flowColorImg(u, v, 0, 6.0f);
        CHECK(half.samples[0] == 255 && half.samples[1] == 127 && half.samples[2] == 127);
        for (int i = 0; i < u->stride * 4; i++) { u->data[i] = 0.0f; v->data[i] = 0.0f; }
        CHECK(flowColorImg(u, v).samples[5] == 255);                    // zero flow: radius 1, all white
        image_t *src = image_new(4, 2);
        for (int y = 0; y < 2; y++) for (int x = 0; x < 4; x++) src->data[y * src->stride + x] = (float)(10 * y + x);
        image_t *halfsz = flow_resize_nearest(src, 0.5f), *dbl = flow_resize_nearest(src, 2.0f), *same = flow_resize_nearest(src, 1.0f);
        CHECK(halfsz && halfsz->width == 2 && halfsz->height == 1 && halfsz->data[0] == 0.0f && halfsz->data[1] == 1.0f);      // picks 0 and 2, times 0.5
        CHECK(dbl && dbl->width == 8 && dbl->height == 4 && dbl->data[3 * dbl->stride + 7] == 26.0f && dbl->data[1 * dbl->stride + 2] == 2.0f);
        CHECK(same && same->width == 4 && same->data[1 * same->stride + 3] == 13.0f);
        image_delete(u); image_delete(v); image_delete(gu); image_delete(gv); image_delete(m); image_delete(other);
        image_delete(src); image_delete(halfsz); image_delete(dbl); image_delete(same);
    }
    // ---- shard.h: who refines which window (the reference's OpenMP loop over jets, slow_flow.cpp:706) -----------------------------------
    {
        // config 4 on a full node: 64 jets x 2 directions over 8 GPUs x 2 streams
        const std::vector<WorkerPlan> plan = plan_workers(128, 8, 2);
        CHECK(plan.size() == 16);
        size_t next = 0;
        for (size_t i = 0; i < plan.size(); i++) {
            CHECK(plan[i].worker == (int)i && plan[i].gpu == (int)i / 2 && plan[i].stream == (int)i % 2);
            CHECK(plan[i].lo == next && plan[i].hi - plan[i].lo == 8);              // contiguous, 8 windows = 4 jets (fwd + bwd) per worker
            next = plan[i].hi;
        }
        CHECK(next == 128);
        for (int ngpu = 1; ngpu <= 8; ngpu++)
            for (int st = 1; st <= 4; st++)
                for (size_t n : {(size_t)0, (size_t)1, (size_t)7, (size_t)128, (size_t)129}) {
                    const std::vector<WorkerPlan> p = plan_workers(n, ngpu, st);
                    size_t cover = 0, mn = n, mx = 0;
                    for (const WorkerPlan &w : p) {
                        CHECK(w.lo == cover && w.hi >= w.lo && w.gpu >= 0 && w.gpu < ngpu && w.gpu == w.worker / st);
                        cover = w.hi;
                        mn = std::min(mn, w.hi - w.lo); mx = std::max(mx, w.hi - w.lo);
                    }
                    CHECK(cover == n && (int)p.size() == ngpu * st && mx - mn <= 1);
                    for (size_t i = 1; i < p.size(); i++) CHECK(p[i].gpu >= p[i - 1].gpu);   // a GPU's windows are contiguous
                }
        CHECK(plan_workers(5, 0, 0).size() == 1);
    }
    // ---- shard.h: which frames a GPU is sent.  Config 4 on a full node: 64 jets x 2 directions, S = 3 (steps = ref = 2): 133 frames; jet j reads
    //      2 j .. 2 j + 4 forwards and 2 j + 2 .. 2 j + 6 backwards (slow_flow.cpp:721-724, :590-591) ---------------------------------------------------------
    {
        const int jets = 64, steps = 2, ref = 2, n_frames = 1 + (jets + 2) * steps;
        std::vector<std::pair<int, int>> wf;
        for (int j = 0; j < jets; j++) { wf.push_back({j * steps, j * steps + 2 * ref}); wf.push_back({j * steps + steps, j * steps + 3 * steps}); }
        const std::vector<WorkerPlan> plan = plan_workers(wf.size(), 8, 2);
        const std::vector<FrameRange> fr = plan_frames(wf, plan, 8, n_frames);
        CHECK(fr.size() == 8 && n_frames == 133);
        long total = 0;
        for (int g = 0; g < 8; g++) {
            // GPU g: jets 8 g .. 8 g + 7 -> frames 16 g .. 16 g + 14 + 6 (inclusive): 21 frames, of which the neighbour holds 5 too (the halo)
            CHECK(fr[g].lo == 16 * g && fr[g].hi == (g == 7 ? n_frames : 16 * g + 21));
            total += fr[g].hi - fr[g].lo;
            for (const WorkerPlan &w : plan)
                if (w.gpu == g)
                    for (size_t i = w.lo; i < w.hi; i++) CHECK(wf[i].first >= fr[g].lo && wf[i].second < fr[g].hi);     // every window finds its frames on its GPU
        }
        CHECK(total == 8 * 21 && total * 8 < 2 * 8 * n_frames);                   // 168 frame uploads instead of 8 x 133 (16 %)
        // coverage without gaps whatever is left to do (-resume): every loaded frame is summed by somebody
        for (int ngpu = 1; ngpu <= 8; ngpu++)
            for (int st = 1; st <= 2; st++)
                for (int keep = 1; keep <= 7; keep++) {
                    std::vector<std::pair<int, int>> some;
                    for (size_t i = 0; i < wf.size(); i++) if ((int)(i * 2654435761u % 7) < keep) some.push_back(wf[i]);
                    const std::vector<WorkerPlan> p = plan_workers(some.size(), ngpu, st);
                    const std::vector<FrameRange> r = plan_frames(some, p, ngpu, n_frames);
                    std::vector<int> cover(n_frames, 0);
                    for (int g = 0; g < ngpu; g++) for (int f = r[g].lo; f < r[g].hi; f++) cover[f]++;
                    for (int f = 0; f < n_frames; f++) CHECK(cover[f] >= 1);
                    for (const WorkerPlan &w : p)
                        for (size_t i = w.lo; i < w.hi; i++) CHECK(some[i].first >= r[w.gpu].lo && some[i].second < r[w.gpu].hi);
                }
        const std::vector<FrameRange> none = plan_frames({}, plan_workers(0, 4, 2), 4, n_frames);
        for (const FrameRange &f : none) CHECK(f.lo == f.hi);
    }
    // ---- shard.h: adaptive frame rates (slow_flow.cpp:322-352), evaluated by hand from the reference's text -----------------------------
    {
        AdaptiveRates r = adaptive_rates(1.0, 2.0, 4, 10, 1);      // hfr = round(2/1) = 2 (10 % 2 == 0); lfr = min(10, 8) = 8 -> 9 -> 10; min(10/1, 10)
        CHECK(r.hfr_rate == 2 && r.lfr_rate == 10);
        r = adaptive_rates(0.5, 2.0, 4, 10, 2);                    // hfr = 4: 10 % 8 != 0 -> 5: 10 % 10 == 0; lfr = min(10, 20) = 10: 20 >= 10, 20 % 10 == 0; min(10/2, 10) = 5
        CHECK(r.hfr_rate == 5 && r.lfr_rate == 5);
        r = adaptive_rates(4.0, 2.0, 4, 10, 1);                    // slow sequence: hfr = max(1, round(.5)) = 1; lfr = min(10, 4) = 4 -> 5 (10 % 5 == 0, 5 % 1 == 0)
        CHECK(r.hfr_rate == 1 && r.lfr_rate == 5);
        r = adaptive_rates(0.7, 2.0, 4, 0, 1);                     // no keyframes: hfr = int(2.857) = 2; lfr = 2*4 = 8, then 2*8 = 16; m = round(16/2) = 8 -> 16
        CHECK(r.hfr_rate == 2 && r.lfr_rate == 16);
        r = adaptive_rates(5.0, 2.0, 4, 0, 1);                     // hfr = int(0.4) = 0 -> max(1, 0) = 1; lfr = 4, 4; m = 4
        CHECK(r.hfr_rate == 1 && r.lfr_rate == 4);
    }
    // ---- shard.h: the task pool runs everything it was given, wait_all() waits for it ---------------------------------------------------
    {
        std::atomic<int> sum(0);
        {
            TaskPool pool(4);
            for (int i = 1; i <= 200; i++) pool.submit([&sum, i] { sum += i; });
            pool.wait_all();
            CHECK(sum == 200 * 201 / 2);
            for (int i = 0; i < 50; i++) pool.submit([&sum] { sum += 1; });
        }                                                          // the destructor drains the queue
        CHECK(sum == 200 * 201 / 2 + 50);
    }
    // ---- ingest on inputs written by tests/test_host.py (checked there against a numpy evaluation of utils.cpp:1241-1374) ---------------
    {
        std::ifstream m((tmp + "/bayer.txt").c_str());
        int w = 0, h = 0, rx = 0, ry = 0;
        float weight = 0;
        if (m >> w >> h >> rx >> ry >> weight) {
            image_t *mosaic = image_new(w, h);
            std::ifstream f((tmp + "/bayer_in.bin").c_str(), std::ios::binary);
            f.read(reinterpret_cast<char *>(mosaic->data), (std::streamsize)((size_t)mosaic->stride * h * sizeof(float)));
            color_image_t *rgb = color_image_new(w, h), *cw = color_image_new(w, h);
            color_image_erase(rgb);
            bayer2rgbGR(mosaic, rgb, rx, ry);
            {                                                    // raw_demosaicing 2: OpenCV's 8-bit bilinear conversion, restated
                color_image_t *cv = color_image_new(w, h);
                color_image_erase(cv);
                bayer2rgb_cv8u(mosaic, cv, rx, ry);
                std::ofstream o3((tmp + "/bayer_rgb_cv.bin").c_str(), std::ios::binary);
                o3.write(reinterpret_cast<const char *>(cv->c1), (std::streamsize)((size_t)3 * cv->stride * h * sizeof(float)));
                color_image_delete(cv);
            }
            for (size_t i = 0; i < (size_t)3 * cw->stride * h; i++) cw->c1[i] = 1.0f;
            rawWeighting(cw, rx, ry, weight);
            std::ofstream o1((tmp + "/bayer_rgb.bin").c_str(), std::ios::binary), o2((tmp + "/bayer_w.bin").c_str(), std::ios::binary);
            o1.write(reinterpret_cast<const char *>(rgb->c1), (std::streamsize)((size_t)3 * rgb->stride * h * sizeof(float)));
            o2.write(reinterpret_cast<const char *>(cw->c1), (std::streamsize)((size_t)3 * cw->stride * h * sizeof(float)));
            image_delete(mosaic); color_image_delete(rgb); color_image_delete(cw);
        }
    }
    // ---- TIFF (tiff.h): own round trip, and files written by tests/test_host.py with Pillow (decoded there as well) --------------------------
    {
        for (int depth = 8; depth <= 16; depth += 8)
            for (int ch = 1; ch <= 3; ch += 2) {
                png_image a;
                a.width = 37; a.height = 21; a.channels = ch; a.depth = depth;
                a.samples.resize((size_t)a.width * a.height * ch);
                for (size_t i = 0; i < a.samples.size(); i++) a.samples[i] = (uint16_t)((i * 2654435761u >> 7) & (depth == 16 ? 65535u : 255u));
                const std::string fn = tmp + "/rt.tif";
                CHECK(tiff_write(fn.c_str(), a));
                png_image b;
                CHECK(tiff_read(fn.c_str(), b));
                CHECK(b.width == a.width && b.height == a.height && b.channels == ch && b.depth == depth && b.samples == a.samples);
                int maxval = 0;
                color_image_t *im = color_image_load(fn.c_str(), &maxval);     // the driver's loader recognises the magic
                CHECK(im && im->width == a.width && maxval == (depth == 16 ? 65535 : 255) && im->c1[5] == (float)a.samples[5 * ch]);
                if (im) color_image_delete(im);
            }
        std::ifstream list((tmp + "/tiff_list.txt").c_str());
        std::string name;
        while (list >> name) {
            png_image t;
            const bool ok = tiff_read((tmp + "/" + name).c_str(), t);
            std::ofstream o((tmp + "/" + name + ".txt").c_str());
            o << (ok ? 1 : 0) << " " << t.width << " " << t.height << " " << t.channels << " " << t.depth << "\n";
            if (ok) {
                std::ofstream ob((tmp + "/" + name + ".bin").c_str(), std::ios::binary);
                ob.write(reinterpret_cast<const char *>(t.samples.data()), (std::streamsize)(t.samples.size() * sizeof(uint16_t)));
            }
        }
    }
    printf(fails ? "host tests FAILED (%d)\n" : "host tests OK\n", fails);
    return fails ? 1 : 0;
}
