// ba_host.hpp -- the host-only half of the local BA set-up (no HIP call, no device needed): structure analysis of a window (pose order, point groups,
// pair lists), the layout of its three arena regions and the staging of its uploads.  Included at the end of ba_device.hpp (it uses slam_ba / BADev).
// Definitions: ba_host.hip.  reference: the arrays are those of src/estimator.jl:143-266 (_get_ba_parameters); the planner has no counterpart there.
#pragma once
inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
size_t band_lds_bytes(int n, int Ps, int hb);
bool ba_pose_order(int P, int M, int O, const uint8_t *theta_const, const int64_t *pose_ids, const int64_t *point_ids, std::vector<int> &order);
// ---- set-up of one window, in two host-only halves so that a batch of windows can be prepared by several threads:
//   ba_plan   the structure of the problem (map points sorted by first free observer, observation order, point groups or pair lists,
//             pose order) and the layout of its device memory in three regions -- uploaded arrays, zero-initialised state, work arrays;
//   ba_emit   binds the device pointers to the three region bases and writes the uploaded region into a host staging block.
// slam_ba_create / slam_local_ba give one window its own arena (the three regions back to back); slam_local_ba_batch lays the
// regions of all windows out region-major (one H2D copy, one memset for the whole batch).  Neither half makes a HIP call.
// fn(0 .. count-1) on the library's parked worker pool, the caller taking part (ba_batch.hip).  A batch prepares one window per task; ONE large window
// (slam_local_ba, BAPlan::nthreads > 1) splits its own passes over the observations instead.  Never called from inside a task of the pool.
void ba_parallel_for(int count, const std::function<void(int)> &fn);
int ba_pool_threads();
#define BA_PAR_MIN_OBS 32768        // observations per task; windows below 6 such tasks stay on the calling thread (measured: 100 k observations gain nothing from three tasks)
struct BAPlan {
    // inputs
    double fx = 0, fy = 0, cx = 0, cy = 0; int P = 0, M = 0, O = 0;
    const double *theta = nullptr; const uint8_t *theta_const_in = nullptr; const double *pixels_yx = nullptr;
    const int64_t *pose_ids = nullptr, *point_ids = nullptr;
    bool may_reorder = false, small_groups = false;
    int nthreads = 1;            // > 1: this one window's passes over the observations run on the worker pool (results identical to the serial passes)
    bool window = false;         // result: the window fits k_ba_window (<= 5 consecutive free poses, ...): no point groups are built for it
    int nfree_obs = 0;           // result: observations of free poses
    // results
    slam_ba *ba = nullptr;
    int err = 0; char msg[160] = {0};
    std::vector<int> cnt, pfirst, new_of, pt_id, rank, start, fgrp;
    std::unique_ptr<int[]> ccnt; int nchunk = 1;   // nchunk > 1: observations of point j in chunk t of the caller's list, [t * M + j] (spans -> fill_obs: the stable order without a serial walk)
    int chunks() const { return nthreads > 1 && O >= 6 * BA_PAR_MIN_OBS ? std::min(nthreads, O / BA_PAR_MIN_OBS) : 1; }
    std::vector<uint8_t> const_perm;
    std::vector<int4> grp;
    std::vector<int2> pairs, blk_pq; std::vector<int> blk_start;
    std::vector<int> v_opose, v_opoint, v_opk; std::vector<double> v_pix; bool filled = false;
    const uint8_t *theta_const = nullptr;
    size_t npairs = 0; int nblk = 0, ngrp = 0, wstride = 0, hb = 0, sg_ob = SG_OB, sg_sb = SG_SB;
    int twice_pt = -1, twice_pose = -1;
    // layout: offsets inside the three regions
    size_t o_pose, o_pts, o_const, o_pix, o_opose, o_opoint, o_start, o_ptid, o_opk, o_ohp, o_pfs, o_fobs, o_grp, o_fgrp, o_pairs, o_bs, o_bpq, up_bytes = 0;
    int sg_hp = SG_OB;
    size_t o_st, o_cf, o_outl, o_bwx = 0, zero_bytes = 0;
    int ksplit = 0;
    size_t o_sc0, o_sc1, o_pose_t, o_pts_t, o_hasp, o_f, o_ft, o_Jp, o_Jl, o_Vinv, o_bl, o_T, o_W, o_red, o_Sw, o_dp, o_dl, o_li, o_lf, o_part, o_band, o_wpart, o_xchg, work_bytes = 0;
    ~BAPlan() { delete ba; }
    int lab(int64_t id) const { return new_of.empty() ? (int)id - 1 : new_of[id - 1]; }
    int fail(int code, const char *fmt, long long a = 0, long long b = 0, long long c = 0) { err = code; snprintf(msg, sizeof msg, fmt, a, b, c); return code; }
    // the per-observation arrays (sorted by point): ONE walk over the caller's observations -- the sorted position of observation i is the
    // next free one of its point.  The walk also finds a map point observed twice by one free pose: it has no place in a pose block.
    void fill_obs(int *opose, int *opoint, int *opk, double *pix, int *ohp = nullptr, int *pfs = nullptr, int *fobs = nullptr)
    {
        const int T = nchunk;
        if (T > 1) {
            // chunk t of the caller's list places its observations behind those of the chunks before it: ccnt becomes the first sorted position of
            // (chunk, point) -- the same positions the serial walk hands out
            int *cc = ccnt.get();
            const size_t Ms = (size_t)M;
            ba_parallel_for(T, [&](int c) {
                for (int j = (int)((long long)M * c / T), j1 = (int)((long long)M * (c + 1) / T); j < j1; j++) {
                    int run = start[rank[j]];
                    for (int t = 0; t < T; t++) { const int v = cc[t * Ms + j]; cc[t * Ms + j] = run; run += v; }
                }
            });
            ba_parallel_for(T, [&](int t) {
                int *fill = cc + t * Ms;
                for (int i = (int)((long long)O * t / T), i1 = (int)((long long)O * (t + 1) / T); i < i1; i++) {
                    const int j = (int)point_ids[i] - 1, s = fill[j]++;
                    ba->perm[s] = i;
                    opose[s] = lab(pose_ids[i]); opoint[s] = j; opk[s] = rank[j];
                    pix[s] = pixels_yx[2 * i]; pix[(size_t)O + s] = pixels_yx[2 * i + 1];
                }
            });
            ccnt.reset();
        } else {
            std::vector<int> fill(start.begin(), start.end() - 1);
            for (int i = 0; i < O; i++) {
                const int j = (int)point_ids[i] - 1, k = rank[j], s = fill[k]++;
                ba->perm[s] = i;
                opose[s] = lab(pose_ids[i]); opoint[s] = j; opk[s] = k;
                pix[s] = pixels_yx[2 * i]; pix[(size_t)O + s] = pixels_yx[2 * i + 1];
            }
        }
        // a map point observed twice by one free pose (seen[p]: the last point, by sorted position, free pose p observed): the first such point in sorted order
        std::vector<int> tw_pt(T, -1), tw_pose(T, -1);
        auto twice = [&](int c) {
            std::vector<int> seen((size_t)P, -1);
            for (int k = (int)((long long)M * c / T), k1 = (int)((long long)M * (c + 1) / T); k < k1 && tw_pt[c] < 0; k++)
                for (int a = start[k]; a < start[k + 1]; a++) {
                    const int p = opose[a];
                    if (theta_const[p]) continue;
                    if (seen[p] == k) { tw_pt[c] = pt_id[k]; tw_pose[c] = new_of.empty() ? p : ba->pose_order[p]; break; }
                    seen[p] = k;
                }
        };
        if (T > 1) ba_parallel_for(T, twice); else twice(0);
        for (int c = 0; c < T && twice_pt < 0; c++) if (tw_pt[c] >= 0) { twice_pt = tw_pt[c]; twice_pose = tw_pose[c]; }
        if (pfs) {                                              // running count of free-pose observations by sorted point
            int c = 0;
            for (int k = 0; k < M; k++) { pfs[k] = c; for (int a = start[k]; a < start[k + 1]; a++) if (!theta_const[opose[a]]) { if (fobs) fobs[c] = a; c++; } }
            pfs[M] = c;
        }
        if (ohp) {                                              // index of an observation among its group's observations of free poses
            const int NG = (int)grp.size();
            std::vector<int> mxs(T, 0);
            auto groups = [&](int c) {
                int mx = 0;
                for (int gi = (int)((long long)NG * c / T), g1 = (int)((long long)NG * (c + 1) / T); gi < g1; gi++) {
                    const int4 &G = grp[gi];
                    int n = 0;
                    for (int a = G.y; a < G.y + G.w; a++) ohp[a] = theta_const[opose[a]] ? -1 : n++;
                    mx = std::max(mx, n);
                }
                mxs[c] = mx;
            };
            if (T > 1) ba_parallel_for(T, groups); else groups(0);
            int mx = 0; for (int c = 0; c < T; c++) mx = std::max(mx, mxs[c]);
            if (small_groups) sg_hp = std::max(8, (mx + 1) & ~1);
        }
    }
};
int ba_plan(BAPlan &pl);
int ba_emit(BAPlan &pl, char *Aup, char *Azero, char *Awork, char *stage);
