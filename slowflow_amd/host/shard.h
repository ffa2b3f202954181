// shard.h -- the pure host logic of the driver that decides who does what: the split of a sequence's frame windows over
// GPUs and streams (the reference's OpenMP loop over jets, slow_flow.cpp:706), the adaptive frame-rate selection
// (slow_flow.cpp:277-357) and a small task pool for frame decoding and result output.  Header-only so that
// tests/host/test_host.cpp exercises exactly the code the driver runs.
#ifndef SLOWFLOW_AMD_HOST_SHARD_H
#define SLOWFLOW_AMD_HOST_SHARD_H

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <utility>
#include <vector>

// One worker = one host thread with its own context (= HIP stream) on one GPU; it refines the windows [lo, hi) of the
// sequence's window list in lockstep batches.
struct WorkerPlan { int worker, gpu, stream; size_t lo, hi; };

// Contiguous blocks (neighbouring jets share frames; forward and backward of a jet stay together whenever the block
// boundaries allow), sizes differing by at most one, every window exactly once; GPU g owns workers g*streams ..
// g*streams+streams-1, so the windows of a GPU are contiguous too.
inline std::vector<WorkerPlan> plan_workers(size_t n_windows, int ngpu, int streams) {
    std::vector<WorkerPlan> plan;
    ngpu = std::max(1, ngpu);
    streams = std::max(1, streams);
    const int nworkers = ngpu * streams;
    for (int wk = 0; wk < nworkers; wk++) {
        WorkerPlan p;
        p.worker = wk; p.gpu = wk / streams; p.stream = wk % streams;
        p.lo = n_windows * wk / nworkers;
        p.hi = n_windows * (wk + 1) / nworkers;
        plan.push_back(p);
    }
    return plan;
}

// Which frames of the loaded sequence a GPU must hold: the union of the frames its windows read -- window i reads the frames [first, last] of `window_frames[i]`:
// jet j forwards j*steps .. j*steps + 2*ref, backwards j*steps + ref .. j*steps + 3*ref (slow_flow.cpp:721-724, :590-591) --, widened so that the ranges of the
// GPUs that have windows cover [0, n_frames) without a gap: every loaded frame enters normalize()'s statistics (slow_flow.cpp:673), so a frame no window reads
// (jets skipped by -resume) is still summed by one GPU.  Neighbouring GPUs overlap by the 2*ref..3*ref frames their boundary jets share (the halo).
// A GPU without windows gets lo == hi.
struct FrameRange { int lo, hi; };
inline std::vector<FrameRange> plan_frames(const std::vector<std::pair<int, int>> &window_frames, const std::vector<WorkerPlan> &plan, int ngpu, int n_frames) {
    ngpu = std::max(1, ngpu);
    std::vector<FrameRange> r((size_t)ngpu, FrameRange{0, 0});
    std::vector<char> has((size_t)ngpu, 0);
    for (const WorkerPlan &wp : plan)
        for (size_t i = wp.lo; i < wp.hi && i < window_frames.size(); i++) {
            FrameRange &g = r[(size_t)wp.gpu];
            const int lo = std::max(0, window_frames[i].first), hi = std::min(n_frames, window_frames[i].second + 1);
            if (!has[(size_t)wp.gpu]) { g.lo = lo; g.hi = hi; has[(size_t)wp.gpu] = 1; }
            else { g.lo = std::min(g.lo, lo); g.hi = std::max(g.hi, hi); }
        }
    int prev = -1;
    for (int g = 0; g < ngpu; g++) {
        if (!has[(size_t)g]) continue;
        if (prev < 0) r[(size_t)g].lo = 0;                                            // frames in front of the first window
        else if (r[(size_t)prev].hi < r[(size_t)g].lo) r[(size_t)prev].hi = r[(size_t)g].lo;   // a gap between two GPUs' windows goes to the earlier one
        prev = g;
    }
    if (prev >= 0) r[(size_t)prev].hi = n_frames;                                    // ... and behind the last one
    return r;
}

// Adaptive frame rates (slow_flow.cpp:322-352).  quantil = 0.99-quantile of the flow magnitude per frame at max_fps
// (quantil.dat, written by adaptiveFR), hfr_quantil / lfr_factor from adaptiveFR.dat (opt_hfr_quantil, opt_lfr_rate),
// keyframes = max_fps / ref_fps (integer division of the float quotient, :324), steps = S - 1.
// Returns the frame skips of the high and the low frame-rate pass.
struct AdaptiveRates { int hfr_rate, lfr_rate; };
inline AdaptiveRates adaptive_rates(double quantil, double hfr_quantil, int lfr_factor, int keyframes, int steps) {
    int hfr_rate = 1, lfr_rate = lfr_factor;
    if (keyframes == 0) {                                                            // exact rates (:326-338)
        hfr_rate = (int)(hfr_quantil / quantil);
        hfr_rate = (int)std::max(1.0, (double)std::round((double)hfr_rate));
        lfr_rate = hfr_rate * lfr_rate;
        lfr_rate = hfr_rate * lfr_rate;                                              // twice, as the reference does (:331,333)
        const double m = std::round((double)(lfr_rate / hfr_rate));                  // same keyframes (:336-337)
        lfr_rate = (int)(hfr_rate * m);
    } else {                                                                         // with keyframes (:340-351)
        hfr_rate = (int)std::max(1.0, (double)std::round(hfr_quantil / quantil));
        while (hfr_rate < keyframes && keyframes % (hfr_rate * steps) != 0) hfr_rate++;
        lfr_rate = std::min(keyframes, hfr_rate * lfr_rate);
        while ((lfr_rate * steps < keyframes &&
                (keyframes % (lfr_rate * steps) != 0 || (keyframes % (lfr_rate * steps) == 0 && (lfr_rate * steps) % (hfr_rate * steps) != 0))) ||
               (lfr_rate * steps >= keyframes && (lfr_rate * steps) % (hfr_rate * steps) != 0))
            lfr_rate++;
        lfr_rate = std::min(keyframes / steps, lfr_rate);
    }
    AdaptiveRates r = {hfr_rate, lfr_rate};
    return r;
}

// Fixed pool of host threads fed from one queue: frame decoding before the run, .flo / PNG output during it (a GPU worker
// hands its downloaded batch over and goes on with the next one).
class TaskPool {
public:
    explicit TaskPool(int n) {
        n = std::max(1, n);
        for (int i = 0; i < n; i++) threads_.emplace_back([this] { loop(); });
    }
    ~TaskPool() {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void submit(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> l(mu_);
            q_.push(std::move(f));
            pending_++;
        }
        cv_.notify_one();
    }
    void wait_all() {
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
    size_t size() const { return threads_.size(); }

private:
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop();
            }
            f();
            {
                std::lock_guard<std::mutex> l(mu_);
                pending_--;
            }
            done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::queue<std::function<void()>> q_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    size_t pending_ = 0;
    bool stop_ = false;
};

#endif
