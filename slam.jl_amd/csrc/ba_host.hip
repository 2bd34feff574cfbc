// ba_host.hip -- definitions of ba_host.hpp: band_lds_bytes, ba_pose_order, ba_plan, ba_emit (host code only; compiled by hipcc for the HIP vector types
// and the shared structs, launches nothing).  Exercised without a GPU by tests/test_ba_plan_order.py and tests/host_sanitize/.
#include "ba_device.hpp"
// ---------------------------------------------------------------------------------


// dynamic LDS of k_band_solve for Ps block columns of half-bandwidth hb (n = 6 P: the damping / solution vectors span every pose)
size_t band_lds_bytes(int n, int Ps, int hb)
{
    const size_t band_fixed = (3 * (size_t)n + 36 * (size_t)Ps) * 8;      // x, damp, chat, L_kk^-1 of every column
    size_t band_lds = band_fixed + ((size_t)(hb + 1) * (hb + 1) * BS_WS + (size_t)(hb + 1) * (6 + BS_WS + 6) + 8 + 36 + 56) * 8 + (size_t)BS_PF * BS_PT * 8;
    if (hb * 6 <= 58) {      // narrow bands: room to stage the G blocks of (up to) all back-substitution steps, at least one
        const size_t want = band_fixed + (size_t)Ps * hb * 36 * 8, least = band_fixed + (size_t)hb * 36 * 8;
        band_lds = std::max(band_lds, std::max(std::min(want, (size_t)150 * 1024), least));
    }
    return band_lds;
}

// A pose order in which the reduced camera system is block-banded, for windows that are not in the caller's order: a loop closure
// (the first and the last key-frames of the window share map points, map_manager.jl:300-449) turns the covisibility chain into a ring, and
// a ring of half-bandwidth h is a band of half-bandwidth ~2 h once it is folded (c, c - 1, c + 1, c - 2, ...).  Candidates, judged by the
// half-bandwidth of the FREE poses' covisibility graph (the constant poses go first: they have no block in the system and only widen the
// span between free ones): the caller's order, every fold of it, Cuthill-McKee from a pose of lowest degree and its reverse.  Returns
// false if none fits the banded solver.  Host work: O(observations) + O(poses x edges); only windows headed for the general path get here.
bool ba_pose_order(int P, int M, int O, const uint8_t *theta_const, const int64_t *pose_ids, const int64_t *point_ids, std::vector<int> &order)
{
    std::vector<int> fidx(P, -1), fr;
    for (int p = 0; p < P; p++) if (!theta_const[p]) { fidx[p] = (int)fr.size(); fr.push_back(p); }
    const int F = (int)fr.size();
    if (F < 3) return false;
    // the free observers of every point (counting sort by point), then the graph: an edge per pair of free poses that share a point
    std::vector<int> st(M + 1, 0);
    for (int i = 0; i < O; i++) if (fidx[pose_ids[i] - 1] >= 0) st[point_ids[i]]++;
    for (int j = 0; j < M; j++) st[j + 1] += st[j];
    std::vector<int> lst(st[M]), fill(st.begin(), st.end() - 1);
    for (int i = 0; i < O; i++) { const int f = fidx[pose_ids[i] - 1]; if (f >= 0) lst[fill[point_ids[i] - 1]++] = f; }
    std::vector<uint8_t> adj((size_t)F * F, 0);
    for (int j = 0; j < M; j++) {
        const int a0 = st[j], a1 = st[j + 1], k = a1 - a0;
        if (j > 0 && st[j] - st[j - 1] == k && std::equal(lst.begin() + a0, lst.begin() + a1, lst.begin() + st[j - 1])) continue;   // same observers as the point before
        for (int a = a0; a < a1; a++)
            for (int b = a0; b < a1; b++) adj[(size_t)lst[a] * F + lst[b]] = 1;
    }
    std::vector<int2> edges; std::vector<int> deg(F, 0);
    for (int a = 0; a < F; a++)
        for (int b = a + 1; b < F; b++) if (adj[(size_t)a * F + b]) { edges.push_back(make_int2(a, b)); deg[a]++; deg[b]++; }
    std::vector<int> pos(F), cand(F), best;
    int best_bw = 1 << 30;
    auto judge = [&]() {                                       // cand[k] = free pose at position k
        for (int k = 0; k < F; k++) pos[cand[k]] = k;
        int bw = 0;
        for (const int2 &e : edges) bw = std::max(bw, std::abs(pos[e.x] - pos[e.y]));
        if (bw < best_bw) { best_bw = bw; best = cand; }
    };
    for (int k = 0; k < F; k++) cand[k] = k;
    judge();
    for (int c = 0; c < F; c++) {                              // folds: c, c - 1, c + 1, c - 2, ... around the ring
        for (int k = 0; k < F; k++) cand[k] = ((k & 1 ? c - (k + 1) / 2 : c + k / 2) % F + F) % F;
        judge();
    }
    {   // Cuthill-McKee (neighbours by ascending degree), every component from its pose of lowest degree; then reversed
        std::vector<uint8_t> seen(F, 0); std::vector<int> q; q.reserve(F);
        while ((int)q.size() < F) {
            int s0 = -1;
            for (int a = 0; a < F; a++) if (!seen[a] && (s0 < 0 || deg[a] < deg[s0])) s0 = a;
            seen[s0] = 1; size_t head = q.size(); q.push_back(s0);
            while (head < q.size()) {
                const int u = q[head++]; const size_t n0 = q.size();
                for (int v = 0; v < F; v++) if (!seen[v] && adj[(size_t)u * F + v]) { seen[v] = 1; q.push_back(v); }
                std::stable_sort(q.begin() + n0, q.end(), [&](int x, int y) { return deg[x] < deg[y]; });
            }
        }
        cand = q; judge();
        std::reverse(cand.begin(), cand.end()); judge();
    }
    const int hbq = std::min(std::max(best_bw, 1), F - 1);
    if (best_bw > BS_MAXHB || !sg_fold_fits(best_bw) || band_lds_bytes(6 * P, F, hbq) > 150 * 1024) return false;
    order.clear();
    for (int p = 0; p < P; p++) if (theta_const[p]) order.push_back(p);
    for (int k = 0; k < F; k++) order.push_back(fr[best[k]]);
    return true;
}


int ba_plan(BAPlan &pl)
{
    const int P = pl.P, M = pl.M, O = pl.O;
    const int64_t *pose_ids = pl.pose_ids, *point_ids = pl.point_ids;
    if (!(P > 0 && 6 * P <= SOLVE_MAX_N && M >= 0 && O >= 0 && pl.theta != nullptr && pl.theta_const_in != nullptr)) return pl.fail(SLAM_ERR_ARG, "slam_ba: bad arguments (P %lld, M %lld, O %lld)", P, M, O);
    if (!(O == 0 || (pl.pixels_yx != nullptr && pose_ids != nullptr && point_ids != nullptr))) return pl.fail(SLAM_ERR_ARG, "slam_ba: observations without arrays");
    pl.theta_const = pl.theta_const_in;
    const uint8_t *&theta_const = pl.theta_const;
    slam_ba *ba = pl.ba = new slam_ba();
    const int n = 6 * P;
    // --- host-side structure.  Map points sorted by (first free observing pose f, id); observations sorted by point in
    //     that order (stable).  hb = widest span of free observers of one point = block half-bandwidth of S.
    std::vector<int> &cnt = pl.cnt, &pfirst = pl.pfirst, plast(M), pany(M);
    cnt.assign(M, 0); pfirst.assign(M, 0);
    int hb = 0, bad_obs = -1, nfo = 0;
    const int T = pl.nchunk = pl.chunks();
    auto spans = [&]() {                                     // (the first pass also checks the ids: one walk over the observations, not two)
        if (T > 1) {
            // one large window: chunk t of the observations counts into its own arrays (the per-chunk counts are kept: fill_obs places the
            // observations of a chunk behind those of the chunks before it), merged by ranges of points
            const size_t Ms = (size_t)M;
            pl.ccnt.reset(new int[T * Ms]);
            std::unique_ptr<int[]> cf(new int[T * Ms]), cl(new int[T * Ms]), ca(new int[T * Ms]);
            std::vector<int> cbad(T, -1), cnfo(T, 0);
            ba_parallel_for(T, [&](int t) {
                int *c = pl.ccnt.get() + t * Ms, *f = cf.get() + t * Ms, *l = cl.get() + t * Ms, *a = ca.get() + t * Ms, nf = 0;
                std::fill(c, c + M, 0); std::fill(f, f + M, P); std::fill(l, l + M, -1); std::fill(a, a + M, P);
                for (int i = (int)((long long)O * t / T), i1 = (int)((long long)O * (t + 1) / T); i < i1; i++) {
                    if (pose_ids[i] < 1 || pose_ids[i] > P || point_ids[i] < 1 || point_ids[i] > M) { cbad[t] = i; break; }
                    const int j = (int)point_ids[i] - 1, p = pl.lab(pose_ids[i]);
                    c[j]++; a[j] = std::min(a[j], p);
                    if (!theta_const[p]) { f[j] = std::min(f[j], p); l[j] = std::max(l[j], p); nf++; }
                }
                cnfo[t] = nf;
            });
            for (int t = 0; t < T; t++) if (cbad[t] >= 0) { bad_obs = cbad[t]; return; }
            nfo = 0; for (int t = 0; t < T; t++) nfo += cnfo[t];
            ba_parallel_for(T, [&](int c) {
                for (int j = (int)((long long)M * c / T), j1 = (int)((long long)M * (c + 1) / T); j < j1; j++) {
                    int n = 0, f = P, l = -1, a = P;
                    for (int t = 0; t < T; t++) { n += pl.ccnt[t * Ms + j]; f = std::min(f, cf[t * Ms + j]); l = std::max(l, cl[t * Ms + j]); a = std::min(a, ca[t * Ms + j]); }
                    cnt[j] = n; pfirst[j] = f; plast[j] = l; pany[j] = a;
                }
            });
        } else {
        std::fill(cnt.begin(), cnt.end(), 0); std::fill(pfirst.begin(), pfirst.end(), P); std::fill(plast.begin(), plast.end(), -1); std::fill(pany.begin(), pany.end(), P);
        nfo = 0;
        for (int i = 0; i < O; i++) {
            if (pose_ids[i] < 1 || pose_ids[i] > P || point_ids[i] < 1 || point_ids[i] > M) { bad_obs = i; return; }
            const int j = (int)point_ids[i] - 1, p = pl.lab(pose_ids[i]);
            cnt[j]++; pany[j] = std::min(pany[j], p);
            if (!theta_const[p]) { pfirst[j] = std::min(pfirst[j], p); plast[j] = std::max(plast[j], p); nfo++; }
        }
        }
        hb = 0;
        for (int j = 0; j < M; j++) {
            if (plast[j] >= 0) hb = std::max(hb, plast[j] - pfirst[j]);
            else pfirst[j] = pany[j] < P ? pany[j] : 0;          // no free observer: any window will do (it gets no slot)
        }
    };
    spans();
    if (bad_obs >= 0) return pl.fail(SLAM_ERR_ARG, "slam_ba: observation %lld has pose id %lld / point id %lld out of range", bad_obs, pose_ids[bad_obs], point_ids[bad_obs]);
    static const bool no_reorder = getenv("SLAMHIP_BA_NO_REORDER") != nullptr;      // (measurement knob)
    if (pl.may_reorder && !no_reorder && M > 0 && O > 0 && (hb > BS_MAXHB || !sg_fold_fits(hb)) && ba_pose_order(P, M, O, pl.theta_const_in, pose_ids, point_ids, ba->pose_order)) {
        // not banded in the caller's pose order, banded in another one: the solver works on relabelled poses, ba_download restores the order
        pl.new_of.resize(P); pl.const_perm.resize(P);
        for (int k = 0; k < P; k++) { pl.new_of[ba->pose_order[k]] = k; pl.const_perm[k] = pl.theta_const_in[ba->pose_order[k]]; }
        theta_const = pl.const_perm.data();
        spans();
    }
    ba->hb = hb; pl.hb = hb;
    {   int f0 = P, f1 = -1;
        for (int p = 0; p < P; p++) if (!theta_const[p]) { f0 = std::min(f0, p); f1 = std::max(f1, p); }
        if (f1 - f0 + 1 >= 2) { ba->p0 = f0; ba->pspan = f1 - f0 + 1; } else { ba->p0 = 0; ba->pspan = P; } }
    std::vector<int> &pt_id = pl.pt_id, &rank = pl.rank, &start = pl.start;
    pt_id.assign(M, 0); rank.assign(M, 0); start.assign(M + 1, 0);
    { std::vector<int> fb(P + 1, 0);
      for (int j = 0; j < M; j++) fb[pfirst[j] + 1]++;
      for (int p = 0; p < P; p++) fb[p + 1] += fb[p];
      for (int j = 0; j < M; j++) { const int k = fb[pfirst[j]]++; pt_id[k] = j; rank[j] = k; } }
    for (int k = 0; k < M; k++) start[k + 1] = start[k] + cnt[pt_id[k]];
    ba->perm.assign(O, 0);
    // --- point groups of k_schur_groups: same f, <= SG_SB points, <= SG_OB observations; evenly sized within one f
    static const bool no_groups = getenv("SLAMHIP_NO_GROUPS") != nullptr;
    // (windows no order makes banded, hb > BS_MAXHB: the point groups still build the system -- their window is the whole block triangle -- when
    //  the free span is small enough for the dense one-workgroup solver, k_dense_solve, and the group's LDS layout fits)
    static const bool no_dense = getenv("SLAMHIP_NO_DENSE") != nullptr;
    const bool dense_ok = !no_dense && hb > BS_MAXHB && ba->pspan >= 2 && ba->pspan <= DS_MAXF && P <= DS_MAXF + 8 && sg_lds_bytes(hb, P) <= 150 * 1024;
    bool grouped = !no_groups && (hb <= BS_MAXHB || dense_ok) && M > 0 && O > 0 && sg_fold_fits(hb);
    pl.nfree_obs = nfo;
    {   // a window one workgroup can keep to itself (k_ba_window, batches only): the point groups of the launch-per-phase kernels are not built
        static const bool no_bw = getenv("SLAMHIP_NO_BA_WINDOW") != nullptr;
        int nfree = 0; for (int p = 0; p < P; p++) nfree += theta_const[p] ? 0 : 1;
        int cmax = 0; for (int j = 0; j < M; j++) cmax = std::max(cmax, cnt[j]);
        pl.window = pl.small_groups && !no_bw && grouped && nfree >= 1 && nfree <= 5 && nfree == ba->pspan && P <= 128 && O <= 40000 && cmax <= 168;
        if (pl.window) pl.sg_ob = std::max(cmax, 1);               // k_ba_window's tiles are sized from it (no point groups are built for such a window)
        static const bool no_ends = getenv("SLAMHIP_BA_WINDOW_SORTED") != nullptr;      // (knob: keep the order by first free observer)
        if (pl.window && M > 1 && !no_ends) {
            // k_ba_window splits a window over two workgroups by map points: the points that see a free pose at all (the Schur phase's records --
            // a fifth of the reference's window, and next to each other in the order by first free observer) go to BOTH ends of the order,
            // alternately, the others between them: each half then holds half of the records and half of the observations (phase clocks per half:
            // A 65 k, Schur 43 k, C 47 k cycles; 1.64 -> 1.62 ms per 128 windows).  Nothing in that kernel depends on the order by first free
            // observer: it is the point groups' window structure, which such a window has not.
            std::vector<int> np_(M); int lo = 0, hi = M - 1, alt = 0, mid = 0;
            for (int k = 0; k < M; k++) { const int j = pt_id[k]; if (plast[j] >= 0) { if (alt++ & 1) np_[hi--] = j; else np_[lo++] = j; } }
            mid = lo;
            for (int k = 0; k < M; k++) { const int j = pt_id[k]; if (plast[j] < 0) np_[mid++] = j; }
            std::reverse(np_.begin() + hi + 1, np_.end());          // (the far end in ascending order of the old ranks, like the near one)
            for (int k = 0; k < M; k++) { pt_id[k] = np_[k]; rank[np_[k]] = k; }
            for (int k = 0; k < M; k++) start[k + 1] = start[k] + cnt[pt_id[k]];
        }
    }
    std::vector<int4> &grp = pl.grp; std::vector<int> &fgrp = pl.fgrp;
    fgrp.assign(P + 1, 0);
    int max_no = 0, max_np = 0;
    if (grouped && !pl.window) {
        static const int sg_points = [] { const char *v = getenv("SLAMHIP_SG_POINTS"); return v ? atoi(v) : 0; }();      // (measurement knob)
        // points per group: a small window in full groups occupies a few compute units and each workgroup walks 7 points per subset; with
        // 16-point groups the reference-shaped window (800 points: 18 -> 50 groups) builds in 0.75 instead of 0.83 ms per 15 iterations,
        // while anything that already fills the chip gets slower with more, smaller groups (more partials for k_schur_reduce, more than one
        // round of workgroups: P = 50 +12 % at 40 points per group) -- so: M / 96, between 16 and SG_SB
        // a batch of windows (small_groups) fills the chip whatever the group size: groups of <= 256 observations run as 256-thread
        // workgroups, three to a compute unit (128 reference-shaped windows: 7.5 ms with the single-window sizes, 3.7 ms so)
        int ob_cap = pl.small_groups ? 256 : SG_OB;
        int sb_eff = sg_points > 0 ? std::min(sg_points, SG_SB)
                   : pl.small_groups ? std::min(std::max(ob_cap / std::max(1, (O + M - 1) / M), 8), SG_SB)
                   : std::min(std::max((M + 95) / 96, 16), SG_SB);
        if (pl.small_groups) for (int j = 0; j < M; j++) if (cnt[j] > ob_cap) { ob_cap = SG_OB; sb_eff = std::min(sb_eff, SG_SB); break; }
        int k = 0;
        for (int f = 0; f < P && grouped; f++) {
            fgrp[f] = (int)grp.size();
            int ke = k;
            while (ke < M && pfirst[pt_id[ke]] == f) ke++;
            const int nf = ke - k, ng = (nf + sb_eff - 1) / sb_eff, tgt = ng ? (nf + ng - 1) / ng : 0;
            while (k < ke) {
                int k1 = k, no = 0;
                while (k1 < ke && k1 - k < tgt && no + cnt[pt_id[k1]] <= ob_cap) { no += cnt[pt_id[k1]]; k1++; }
                if (k1 == k) { grouped = false; break; }          // one point with more than SG_OB observations: pair lists
                grp.push_back(make_int4(k, start[k], f | ((k1 - k) << 16), no));
                max_no = std::max(max_no, no); max_np = std::max(max_np, k1 - k);
                k = k1;
            }
        }
        fgrp[P] = (int)grp.size();
    }
    ba->grouped = grouped;
    if (grouped && pl.small_groups && !pl.window) { pl.sg_ob = std::max(64, (max_no + 7) & ~7); pl.sg_sb = std::max(8, (max_np + 1) & ~1); }
    const int *opose = nullptr;                              // sorted observation -> pose, host copy (needed by the pair lists)
    if (!grouped) {
        pl.v_opose.resize(O); pl.v_opoint.resize(O); pl.v_opk.resize(O); pl.v_pix.resize(2 * (size_t)O);
        pl.fill_obs(pl.v_opose.data(), pl.v_opoint.data(), pl.v_opk.data(), pl.v_pix.data());
        pl.filled = true;
        if (pl.twice_pt >= 0) return pl.fail(SLAM_ERR_ARG, "slam_ba: map point %lld is observed twice by pose %lld", pl.twice_pt + 1, pl.twice_pose + 1);
        opose = pl.v_opose.data();
    }
    // --- pair lists sorted by upper pose block (p <= q), both poses free: only where the groups do not apply
    std::vector<int2> &pairs = pl.pairs, &blk_pq = pl.blk_pq; std::vector<int> &blk_start = pl.blk_start;
    size_t npairs = 0;
    if (!grouped) {
        std::vector<int> bcount((size_t)P * P + 1, 0);
        for (int j = 0; j < M; j++)
            for (int a = start[j]; a < start[j + 1]; a++) {
                if (theta_const[opose[a]]) continue;
                for (int b = start[j]; b < start[j + 1]; b++) {
                    if (theta_const[opose[b]]) continue;
                    const int p = opose[a], q = opose[b];
                    if (p > q || (p == q && a > b)) continue;   // upper blocks; within a diagonal block keep a <= b once
                    bcount[(size_t)p * P + q + 1]++; npairs++;
                }
            }
        std::vector<int> boff((size_t)P * P + 1, 0);
        for (size_t k = 0; k < (size_t)P * P; k++) boff[k + 1] = boff[k] + bcount[k + 1];
        pairs.resize(npairs);
        { std::vector<int> fill(boff.begin(), boff.end() - 1);
          for (int j = 0; j < M; j++)
              for (int a = start[j]; a < start[j + 1]; a++) {
                  if (theta_const[opose[a]]) continue;
                  for (int b = start[j]; b < start[j + 1]; b++) {
                      if (theta_const[opose[b]]) continue;
                      const int p = opose[a], q = opose[b];
                      if (p > q || (p == q && a > b)) continue;
                      pairs[fill[(size_t)p * P + q]++] = make_int2(a, b);
                  }
              } }
        for (int p = 0; p < P; p++)
            for (int q = p; q < P; q++) {
                const size_t k = (size_t)p * P + q;
                if (boff[k + 1] > boff[k]) { blk_start.push_back(boff[k]); blk_pq.push_back(make_int2(p, q)); }
            }
    }
    blk_start.push_back((int)npairs);
    pl.npairs = npairs;
    const int nblk = pl.nblk = (int)blk_pq.size();
    const int ngrp = pl.ngrp = (int)grp.size(), hbw = hb + 1;
    pl.wstride = grouped ? hbw * (hbw + 1) / 2 * 36 + hbw * 12 : 0;
    const int nbo = (O + 255) / 256, nbp = (std::max(M, n) + 255) / 256;
    ba->nblocks_obs = std::max(nbo, 1); ba->nblocks_pts = std::max(nbp, 1);
    // --- layout: uploaded arrays (one contiguous block: a single copy from the staging buffer), the zero-initialised ones (one memset), the rest
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    pl.o_pose = take(n * 8); pl.o_pts = take((size_t)3 * M * 8 + 8); pl.o_const = take(P); pl.o_pix = take((size_t)2 * O * 8 + 8);
    pl.o_opose = take((size_t)O * 4 + 4); pl.o_opoint = take((size_t)O * 4 + 4); pl.o_start = take((size_t)(M + 1) * 4);
    pl.o_ptid = take((size_t)M * 4 + 4); pl.o_opk = take((size_t)O * 4 + 4); pl.o_ohp = take(pl.window ? 8 : (size_t)O * 4 + 4); pl.o_pfs = take(pl.small_groups ? (size_t)(M + 1) * 4 : 8); pl.o_fobs = take(pl.small_groups ? (size_t)pl.nfree_obs * 4 + 4 : 8); pl.o_grp = take((size_t)ngrp * 16 + 16); pl.o_fgrp = take((size_t)(P + 1) * 4);      // (pfs / fobs: k_ba_window's lists, batches only)
    pl.o_pairs = take(npairs * 8 + 8); pl.o_bs = take((size_t)(nblk + 1) * 4); pl.o_bpq = take((size_t)nblk * 8 + 8);
    pl.up_bytes = off; off = 0;
    pl.o_st = take(sizeof(LMState)); pl.o_cf = take(64); pl.o_outl = take((size_t)O + 1);
    pl.o_dp = take(n * 8);                                       // dp of the constant poses outside the solve's span stays zero
    if (pl.window) {                                             // k_ba_window on two workgroups: flags + two double-buffered partials of 32 x 25 doubles each (zeroed: the flags count exchanges)
        pl.o_bwx = take(BWX_DOUBLES * 8);
        pl.ksplit = (int)(std::lower_bound(start.begin(), start.end(), (O + 1) / 2) - start.begin());
        pl.ksplit = std::min(std::max(pl.ksplit, 0), M);
    }
    pl.o_red = take(((size_t)n * n + 2 * n + 8) * 8);            // the private reduce buffer: only its band is ever rewritten
    pl.zero_bytes = off; off = 0;
    pl.o_sc0 = take(n * 8); pl.o_sc1 = take(n * 8);
    pl.o_pose_t = take(n * 8); pl.o_pts_t = take((size_t)3 * M * 8 + 8); pl.o_hasp = take((size_t)O + 1);
    pl.o_f = take((size_t)2 * O * 8 + 8); pl.o_ft = take(8);    // (trial residuals are not kept: every build re-evaluates d.f)
    pl.o_Jp = take((size_t)12 * O * 8 + 8); pl.o_Jl = take((size_t)6 * O * 8 + 8);
    pl.o_Vinv = take((size_t)6 * M * 8 + 8); pl.o_bl = take((size_t)3 * M * 8 + 8);
    pl.o_T = take(grouped ? 8 : (size_t)18 * O * 8 + 8); pl.o_W = take(grouped ? 8 : (size_t)18 * O * 8 + 8);   // T / W records: pair-list path only
    // (the tiled Cholesky's working matrices exist for every window: a window whose band does not fit k_band_solve's LDS takes that path)
    pl.o_Sw = take((size_t)(n + 1) * n * 8); pl.o_dl = take((size_t)3 * M * 8 + 8);
    pl.o_li = take((size_t)((n + CT - 1) / CT) * CT * CT * 8); pl.o_lf = take((size_t)(n + 1) * n * 8);
    pl.o_part = take(((size_t)std::max(ba->nblocks_pts, ngrp) + 2 * (size_t)std::max(ba->nblocks_obs, ngrp) + 8) * 8);
    pl.o_band = take((size_t)P * ((size_t)(BS_MAXHB + 1) * 36 + 8) * 8);
    pl.o_wpart = take((size_t)ngrp * pl.wstride * 8 + 8);
    pl.o_xchg = take(2048 * 8);
    pl.work_bytes = off;
    return SLAM_OK;
}

// bind the device pointers (region bases Aup / Azero / Awork) and write the uploaded region into `stage` (host memory, up_bytes)
int ba_emit(BAPlan &pl, char *Aup, char *Azero, char *Awork, char *stage)
{
    slam_ba *ba = pl.ba;
    const int P = pl.P, M = pl.M, O = pl.O, n = 6 * P;
    BADev &d = ba->d;
    d.cam = {pl.fx, pl.fy, pl.cx, pl.cy}; d.P = P; d.M = M; d.O = O; d.n = n;
    d.sc0 = (double *)(Awork + pl.o_sc0); d.sc1 = (double *)(Awork + pl.o_sc1);
    d.pose = (double *)(Aup + pl.o_pose); d.pose_t = (double *)(Awork + pl.o_pose_t); d.pts = (double *)(Aup + pl.o_pts); d.pts_t = (double *)(Awork + pl.o_pts_t);
    d.pconst = (const uint8_t *)(Aup + pl.o_const); d.pix = (const double *)(Aup + pl.o_pix);
    d.opose = (const int *)(Aup + pl.o_opose); d.opoint = (const int *)(Aup + pl.o_opoint); d.pt_start = (const int *)(Aup + pl.o_start);
    d.outl = (uint8_t *)(Azero + pl.o_outl); d.hasp = (uint8_t *)(Awork + pl.o_hasp);
    d.f = (double *)(Awork + pl.o_f); d.ft = (double *)(Awork + pl.o_ft); d.Jp = (double *)(Awork + pl.o_Jp); d.Jl = (double *)(Awork + pl.o_Jl);
    d.Vinv = (double *)(Awork + pl.o_Vinv); d.bl = (double *)(Awork + pl.o_bl); d.T = (double *)(Awork + pl.o_T); d.Wm = (double *)(Awork + pl.o_W);
    d.pairs = (const int2 *)(Aup + pl.o_pairs); d.blk_start = (const int *)(Aup + pl.o_bs); d.blk_pq = (const int2 *)(Aup + pl.o_bpq); d.nblk = pl.nblk;
    ba->reduce = (double *)(Azero + pl.o_red);
    ba->zeroed = ba->reduce;                                   // (the set-up's memset of the zero region covers it)
    d.S = ba->reduce; d.g = ba->reduce + (size_t)n * n; d.udiag = d.g + n;
    d.Swork = (double *)(Awork + pl.o_Sw); d.dp = (double *)(Azero + pl.o_dp); d.dl = (double *)(Awork + pl.o_dl);
    d.part = (double *)(Awork + pl.o_part); d.st = (LMState *)(Azero + pl.o_st);
    ba->chol_flag = (int *)(Azero + pl.o_cf); ba->linv = (double *)(Awork + pl.o_li); ba->lfac = (double *)(Awork + pl.o_lf); ba->band = (double *)(Awork + pl.o_band);
    d.pt_id = (const int *)(Aup + pl.o_ptid); d.opk = (const int *)(Aup + pl.o_opk); d.grp = (const int4 *)(Aup + pl.o_grp); d.fgrp = (const int *)(Aup + pl.o_fgrp);
    d.ngrp = pl.ngrp; d.whb = pl.hb; d.wstride = pl.wstride; d.wpart = (double *)(Awork + pl.o_wpart);
    d.sg_ob = pl.sg_ob; d.sg_sb = pl.sg_sb; d.ohp = (const int *)(Aup + pl.o_ohp); d.pfs = (const int *)(Aup + pl.o_pfs); d.fobs = (const int *)(Aup + pl.o_fobs);
    ba->nparts = ba->grouped ? pl.ngrp : ba->nblocks_obs;
    ba->xchg = (double *)(Awork + pl.o_xchg);
#define UP(o, src, bytes) do { if ((bytes) > 0) memcpy(stage + (o), (src), (bytes)); } while (0)
    if (!pl.new_of.empty()) { double *dst = (double *)(stage + pl.o_pose); for (int k = 0; k < P; k++) memcpy(dst + 6 * k, pl.theta + 6 * ba->pose_order[k], 48); }
    else UP(pl.o_pose, pl.theta, (size_t)n * 8);
    UP(pl.o_pts, pl.theta + n, (size_t)3 * M * 8);
    UP(pl.o_const, pl.theta_const, (size_t)P); UP(pl.o_start, pl.start.data(), (size_t)(M + 1) * 4);
    if (pl.filled) { UP(pl.o_pix, pl.v_pix.data(), (size_t)2 * O * 8); UP(pl.o_opose, pl.v_opose.data(), (size_t)O * 4); UP(pl.o_opoint, pl.v_opoint.data(), (size_t)O * 4); UP(pl.o_opk, pl.v_opk.data(), (size_t)O * 4); }
    else {                                                   // (grouped: nothing on the host needs these arrays) written in place
        pl.fill_obs((int *)(stage + pl.o_opose), (int *)(stage + pl.o_opoint), (int *)(stage + pl.o_opk), (double *)(stage + pl.o_pix), pl.window ? nullptr : (int *)(stage + pl.o_ohp), pl.small_groups ? (int *)(stage + pl.o_pfs) : nullptr, pl.small_groups ? (int *)(stage + pl.o_fobs) : nullptr);
        if (pl.twice_pt >= 0) return pl.fail(SLAM_ERR_ARG, "slam_ba: map point %lld is observed twice by pose %lld", pl.twice_pt + 1, pl.twice_pose + 1);
    }
    UP(pl.o_pairs, pl.pairs.data(), pl.npairs * 8); UP(pl.o_bs, pl.blk_start.data(), (size_t)(pl.nblk + 1) * 4); UP(pl.o_bpq, pl.blk_pq.data(), (size_t)pl.nblk * 8);
    UP(pl.o_ptid, pl.pt_id.data(), (size_t)M * 4); UP(pl.o_grp, pl.grp.data(), (size_t)pl.ngrp * 16); UP(pl.o_fgrp, pl.fgrp.data(), (size_t)(P + 1) * 4);
#undef UP
    d.sg_hp = pl.sg_hp;
    return SLAM_OK;
}

