// ba_host_driver.cpp -- the HOST half of the local-BA batch path under the CPU sanitizers (no GPU, no HIP call): ba_plan / ba_emit (structure
// analysis, pose re-ordering, point groups, staging layout) and the parked worker pool (BAPool) that two estimator threads share by design,
// driven through the host-only entry slam_debug_ba_host_time with threads = 0 (= the pool) from TWO caller threads at once, on ragged random
// windows incl. degenerate ones; plus slam_ba_plan_order on loop-closure windows.  Built twice by tests/host_sanitize/Makefile:
// -fsanitize=address,undefined and -fsanitize=thread.  Exit status 0 = the sanitizers had nothing to say.  (reference: src/estimator.jl:143-266
// fills these arrays; the planner itself has no counterpart there.)
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

extern "C" {
int slam_debug_ba_host_time(int S, const double *cams, const int32_t *Pn, const int32_t *Mn, const int32_t *On, const double *theta, const uint8_t *theta_const,
                            const double *pixels_yx, const int64_t *pose_ids, const int64_t *point_ids, int threads, double *out_us);
int slam_ba_plan_order(int P, int M, int O, const uint8_t *theta_const, const int64_t *pose_ids, const int64_t *point_ids, int32_t *order_out, int *hb_out);
}

struct Windows {
    std::vector<double> cams, theta, px; std::vector<int32_t> Pn, Mn, On; std::vector<uint8_t> tc; std::vector<int64_t> pi, li;
};

// S ragged windows: 2 .. 30 poses, a run of constant poses at the start (sometimes all of them, sometimes none), 0 .. 400 points each seen by
// 2 .. 8 consecutive poses (sometimes a loop closure: the first and the last poses share points), a few windows without observations
static Windows make(int S, unsigned seed, bool duplicate_obs)
{
    std::mt19937 g(seed);
    auto ri = [&](int a, int b) { return a + (int)(g() % (unsigned)(b - a + 1)); };
    Windows w;
    for (int z = 0; z < S; z++) {
        const int P = ri(2, 30), M = z % 7 == 6 ? 0 : ri(1, 400), nconst = z % 5 == 4 ? P : ri(0, P - 1);
        w.cams.insert(w.cams.end(), {700.0, 700.0, 600.0, 180.0});
        for (int p = 0; p < P; p++) { for (int k = 0; k < 6; k++) w.theta.push_back(0.01 * ri(-9, 9)); w.tc.push_back(p < nconst); }
        int O = 0;
        for (int m = 0; m < M; m++) {
            for (int k = 0; k < 3; k++) w.theta.push_back(1.0 + 0.1 * ri(0, 50));
            const int span = std::min(P, ri(2, 8)), first = ri(0, P - span);
            const bool loop = z % 3 == 0 && m % 9 == 0 && P > 10;
            for (int q = 0; q < span; q++) {
                const int p = loop ? (q < span / 2 ? q : P - span + q) : first + q;      // loop closure: the point is seen by the first and by the last key-frames of the window
                w.pi.push_back(p + 1); w.li.push_back(m + 1); w.px.push_back(10.0 + ri(0, 300)); w.px.push_back(10.0 + ri(0, 1000)); O++;
            }
            if (duplicate_obs && m == 3 && z == 1) { w.pi.push_back(w.pi.back()); w.li.push_back(m + 1); w.px.push_back(1.0); w.px.push_back(1.0); O++; }   // a point observed twice by one pose: a window error, not a crash
        }
        w.Pn.push_back(P); w.Mn.push_back(M); w.On.push_back(O);
    }
    return w;
}

// ONE large window (160 k observations in random order, a few constant poses, optionally one map point seen twice by a pose / one observation
// with a pose id out of range): slam_local_ba splits its passes over the observations into tasks of the pool (BAPlan::chunks)
static Windows big(unsigned seed, int flaw)
{
    std::mt19937 g(seed);
    auto ri = [&](int a, int b) { return a + (int)(g() % (unsigned)(b - a + 1)); };
    const int P = 30, M = 20000, span = 8;
    Windows w;
    w.cams.insert(w.cams.end(), {700.0, 700.0, 600.0, 180.0});
    for (int p = 0; p < P; p++) { for (int k = 0; k < 6; k++) w.theta.push_back(0.01 * ri(-9, 9)); w.tc.push_back(p < 3); }
    std::vector<int64_t> pi, li; std::vector<double> px;
    for (int m = 0; m < M; m++) {
        for (int k = 0; k < 3; k++) w.theta.push_back(1.0 + 0.1 * ri(0, 50));
        const int first = ri(0, P - span);
        for (int q = 0; q < span; q++) { pi.push_back(first + q + 1); li.push_back(m + 1); px.push_back(10.0 + ri(0, 300)); px.push_back(10.0 + ri(0, 1000)); }
    }
    if (flaw == 1) { pi.push_back(pi[8 * 15000 + 5]); li.push_back(15001); px.push_back(1.0); px.push_back(1.0); }
    if (flaw == 2) pi[8 * 12345] = P + 7;
    std::vector<int> ord(pi.size());
    for (size_t i = 0; i < ord.size(); i++) ord[i] = (int)i;
    std::shuffle(ord.begin(), ord.end(), g);
    for (int i : ord) { w.pi.push_back(pi[i]); w.li.push_back(li[i]); w.px.push_back(px[2 * i]); w.px.push_back(px[2 * i + 1]); }
    w.Pn.push_back(P); w.Mn.push_back(M); w.On.push_back((int)pi.size());
    return w;
}

int main()
{
    std::atomic<int> bad{0};
    // the split passes give the serial passes' staging and observation order, byte for byte, and the same verdict on flawed windows
    for (int flaw = 0; flaw < 3; flaw++) {
        Windows w = big(77u + (unsigned)flaw, flaw);
        double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        const int fa = slam_debug_ba_host_time(1, w.cams.data(), w.Pn.data(), w.Mn.data(), w.On.data(), w.theta.data(), w.tc.data(), w.px.data(), w.pi.data(), w.li.data(), -1, a);
        const int fb = slam_debug_ba_host_time(1, w.cams.data(), w.Pn.data(), w.Mn.data(), w.On.data(), w.theta.data(), w.tc.data(), w.px.data(), w.pi.data(), w.li.data(), -4, b);
        if (fa != (flaw ? 1 : 0) || fb != fa || a[2] != b[2]) { fprintf(stderr, "large window, flaw %d: serial %d (hash %.0f) vs split %d (hash %.0f)\n", flaw, fa, a[2], fb, b[2]); bad++; }
    }
    auto caller = [&](unsigned seed) {
        for (int round = 0; round < 12; round++) {
            const bool dup = round == 5;
            Windows w = make(24, seed * 100 + (unsigned)round, dup);
            double us[3] = {0, 0, 0};
            const int failed = slam_debug_ba_host_time(24, w.cams.data(), w.Pn.data(), w.Mn.data(), w.On.data(), w.theta.data(), w.tc.data(), w.px.data(), w.pi.data(), w.li.data(),
                                                       round % 3 == 2 ? 3 : 0, us);
            if (failed != (dup ? 1 : 0)) { fprintf(stderr, "caller %u round %d: %d windows failed their set-up (expected %d)\n", seed, round, failed, dup ? 1 : 0); bad++; }
            // the pose order of every window (ring folds, Cuthill-McKee): host-only, ctx-free
            size_t to = 0, oo = 0, po = 0;
            for (size_t z = 0; z < w.Pn.size(); z++) {
                std::vector<int32_t> order((size_t)w.Pn[z]); int hb = -1;
                const int rc = slam_ba_plan_order(w.Pn[z], w.Mn[z], w.On[z], w.tc.data() + po, w.On[z] ? w.pi.data() + oo : nullptr, w.On[z] ? w.li.data() + oo : nullptr, order.data(), &hb);
                if (rc < 0 && w.On[z] > 0 && !dup) { fprintf(stderr, "slam_ba_plan_order: window %zu -> %d\n", z, rc); bad++; }
                to += 6 * (size_t)w.Pn[z] + 3 * (size_t)w.Mn[z]; oo += (size_t)w.On[z]; po += (size_t)w.Pn[z];
            }
            (void)to;
        }
    };
    std::thread a(caller, 1u), b(caller, 2u);
    a.join(); b.join();
    printf("%s\n", bad.load() ? "FAILED" : "host half of the batch path: clean");
    return bad.load() ? 1 : 0;
}
