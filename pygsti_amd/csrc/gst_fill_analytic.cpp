// gst_fill_analytic.cpp -- exact derivatives behind the C ABI (matrixforwardsim.py:1047-1287): the reversed plan and its
// pair / block tables, the two state caches and the MFMA contraction, and the chain rule for general parameterisations
// (gst_set_derivs / gst_set_second_derivs).
#include "gst_state.hpp"

#include <map>

using namespace gst_impl;

namespace gst_impl {

// ---- tiles of the D = 16 contraction (gst_kernels_tiles.hip) ----------------------------------------------------------------
// Circuits whose strings are P_i . W . Q_m for rows i and columns m: over the middle segment W the forward states depend on
// the row only and the backward states on the column only.  Found from the two tries alone:
//   * in suffix order (circuits sorted by their state in the reversed plan) neighbours that share most of their suffix form
//     a RUN (one Q_m behind all the P_i); T = the suffix length common to the whole run; a member's segment starts at
//     s = n - T and its row key is the forward state it is in there;
//   * runs with the same set of row keys are the columns of one product; rows x columns are cut into tiles of 8 x 4;
//   * the segment length K of a tile is verified, not assumed: the longest stretch from s on which every row's forward ids
//     agree across the tile's columns (inside the common suffix the backward ids agree across rows by construction).
// Everything a circuit has outside [s, s + K) goes to its remnant lists.  Tiles with K < 16 or fewer than 2 rows / 2 columns
// are not formed; their circuits stay with the item kernel.
// (measured with 4: 98.6 % of the 2Q design's circuits tiled, and slower -- a tile's fixed costs, ~45 us of dependent loads,
//  barriers and stores per tile, are not worth it for middles of a few gates: 2.89 + 0.11 ms against 2.08 + 0.70 ms)
constexpr int TILE_MIN_SEGMENT = 16;
struct TileSet {
    std::vector<int32_t> order, cid, blk, tsf, tsr, rem_ptr, rem_f, rem_r;
    std::vector<uint8_t> tiled;
    int32_t n_tiles = 0;
    int64_t n_tiled = 0, seg_slots = 0, rem_slots = 0;
};

static void build_tiles(const gst::HostPlan& h, const gst::HostPlan& R, const std::vector<int32_t>& suffix_order, TileSet& T)
{
    constexpr int TRW = gst::TILE_ROWS, TCL = gst::TILE_COLS, TRC = TRW * TCL;
    const int nG = h.n_gates;
    const int64_t nC = h.n_circuits;
    T.tiled.assign((size_t)nC, 0);
    auto len = [&](int32_t c) { return (int32_t)(h.circ_ptr[c + 1] - h.circ_ptr[c]); };
    auto gates = [&](int32_t c) { return h.circ_gates.data() + h.circ_ptr[c]; };
    auto plain4 = [&](int32_t c) {
        if (h.eff_ptr[c + 1] - h.eff_ptr[c] != 4) return false;
        for (int x = 0; x < 4; x++) if (h.eff_label[(size_t)h.eff_ptr[c] + x] != x) return false;
        return true;
    };
    auto fpath = [&](int32_t c, std::vector<int32_t>& f) {      // f[k] = forward state after k gates
        const int32_t n = len(c);
        f.resize((size_t)n + 1);
        int32_t id = h.circ_leaf[c];
        for (int32_t k = n; k >= 0; k--) { f[(size_t)k] = id; if (k > 0) id = h.node_parent[id]; }
    };
    auto rpath = [&](int32_t c, std::vector<int32_t>& r) {      // r[d] = reversed-plan state after the last d gates
        const int32_t n = len(c);
        r.resize((size_t)n + 1);
        int32_t id = R.circ_leaf[c];
        for (int32_t d = n; d >= 0; d--) { r[(size_t)d] = id; if (d > 0) id = R.node_parent[id]; }
    };
    auto mix = [](uint64_t x, uint64_t v) { x ^= v + 0x9E3779B97F4A7C15ull + (x << 6) + (x >> 2); x *= 0xff51afd7ed558ccdull; return x ^ (x >> 33); };
    auto prefix_hash = [&](int32_t c, int32_t n_gates_in) {      // (preparation, first n gates): names a forward state whatever task holds it
        uint64_t x = mix(0x1234567ull, (uint64_t)h.circ_rho[(size_t)c]);
        const int32_t* g = gates(c);
        for (int32_t k = 0; k < n_gates_in; k++) x = mix(x, (uint64_t)g[k] + 1);
        return x;
    };
    // ---- runs of the suffix order: one ending (and middle) behind different SHORT beginnings ---------------------------------
    // (strings are compared, not state ids: both plans are cut into tasks, and a state shared across a cut has two ids -- with
    //  equal values; the tile then reads row 0's backward states for the column and column 0's forward states for the row)
    struct Run { std::vector<int32_t> members; int32_t T = 0x7fffffff; std::vector<uint64_t> keys; };
    std::vector<Run> runs;
    {
        int32_t prev = -1;
        for (int64_t k = 0; k < (int64_t)suffix_order.size(); k++) {
            const int32_t c = suffix_order[(size_t)k];
            if (!plain4(c)) { prev = -1; continue; }
            bool joined = false;
            if (prev >= 0) {
                const int32_t n1 = len(prev), n2 = len(c), lim = std::min(n1, n2);
                const int32_t *g1 = gates(prev) + n1, *g2 = gates(c) + n2;
                int32_t lcs = 0;
                while (lcs < lim && g1[-1 - lcs] == g2[-1 - lcs]) lcs++;
                if (lcs >= TILE_MIN_SEGMENT && std::max(n1, n2) - lcs <= 12) {
                    runs.back().members.push_back(c);
                    runs.back().T = std::min(runs.back().T, lcs);
                    joined = true;
                }
            }
            if (!joined) { runs.emplace_back(); runs.back().members.push_back(c); }
            prev = c;
        }
    }
    // ---- row keys (the beginning each member has in front of the common suffix); runs with the same rows AND the same
    //      middle string are the columns of one product -------------------------------------------------------------------------
    std::map<std::vector<uint64_t>, std::vector<int32_t>> groups;
    for (size_t q = 0; q < runs.size(); q++) {
        Run& r = runs[q];
        if (r.members.size() < 2) continue;
        std::vector<std::pair<uint64_t, int32_t>> km;
        for (int32_t c : r.members) km.emplace_back(prefix_hash(c, len(c) - r.T), c);
        std::sort(km.begin(), km.end());
        bool dup = false;
        for (size_t k = 1; k < km.size(); k++) dup = dup || km[k].first == km[k - 1].first;
        if (dup) continue;
        r.keys.clear(); r.members.clear();
        for (auto& x : km) { r.keys.push_back(x.first); r.members.push_back(x.second); }
        const std::vector<uint64_t>& key = r.keys;
        groups[key].push_back((int32_t)q);
    }
    // ---- tiles ---------------------------------------------------------------------------------------------------------------
    std::vector<int64_t> tile_work;
    std::vector<std::vector<int32_t>> fp((size_t)TRC), rcol((size_t)TCL);
    std::vector<std::vector<int32_t>> g_slots((size_t)nG);
    for (auto& kv : groups) {
        // the columns of one product: runs whose common suffixes agree but for their last few gates.  Sorted by suffix string,
        // such runs are neighbours; a chunk of up to TILE_COLS of them is extended while the next one shares all but <= 8
        // gates of the LONGER suffix with the chunk's first
        std::vector<int32_t> sorted_runs = kv.second;
        auto suffix = [&](int32_t q, const int32_t*& g, int32_t& n) { const Run& r = runs[(size_t)q]; const int32_t c = r.members[0]; n = r.T; g = gates(c) + (len(c) - r.T); };
        std::sort(sorted_runs.begin(), sorted_runs.end(), [&](int32_t x, int32_t y) {
            const int32_t *gx, *gy; int32_t nx, ny;
            suffix(x, gx, nx); suffix(y, gy, ny);
            return std::lexicographical_compare(gx, gx + nx, gy, gy + ny);
        });
        const int n_rows_all = (int)kv.first.size();
        std::vector<std::vector<int32_t>> chunks;
        for (size_t k = 0; k < sorted_runs.size();) {
            std::vector<int32_t> ch{sorted_runs[k]};
            const int32_t *g0; int32_t n0;
            suffix(sorted_runs[k], g0, n0);
            size_t k2 = k + 1;
            while (k2 < sorted_runs.size() && (int)ch.size() < TCL) {
                const int32_t *g1; int32_t n1;
                suffix(sorted_runs[k2], g1, n1);
                int32_t l = 0;
                while (l < n0 && l < n1 && g0[l] == g1[l]) l++;
                if (l < TILE_MIN_SEGMENT || l < std::max(n0, n1) - 8) break;
                ch.push_back(sorted_runs[k2]); k2++;
            }
            chunks.push_back(ch);
            k = k2;
        }
        for (const std::vector<int32_t>& cols : chunks) {
            const size_t c0 = 0;
            const int ncol = (int)cols.size();
            if (ncol < 2 && n_rows_all < 4) continue;          // (a single column is still worth a tile when it has rows to share it)
            for (int r0 = 0; r0 < n_rows_all; r0 += TRW) {
                const int nrow = std::min(TRW, n_rows_all - r0);
                if (nrow < 2) continue;
                auto circ = [&](int i, int m) { return runs[(size_t)cols[c0 + (size_t)m]].members[(size_t)(r0 + i)]; };
                auto seg0 = [&](int i, int m) { return len(circ(i, m)) - runs[(size_t)cols[c0 + (size_t)m]].T; };
                int32_t K = 0x7fffffff;
                for (int m = 0; m < ncol; m++) K = std::min(K, runs[(size_t)cols[c0 + (size_t)m]].T);
                // every row: the same beginning in every column (exactly, not by hash), then the same gates from s on
                bool ok = true;
                for (int i = 0; i < nrow && ok && K >= TILE_MIN_SEGMENT; i++)
                    for (int m = 1; m < ncol && ok; m++) {
                        const int32_t ca = circ(i, 0), cb = circ(i, m), s0 = seg0(i, 0), sm = seg0(i, m);
                        ok = s0 == sm && h.circ_rho[(size_t)ca] == h.circ_rho[(size_t)cb] && std::equal(gates(ca), gates(ca) + s0, gates(cb));
                        if (!ok) break;
                        const int32_t *ga = gates(ca) + s0, *gb = gates(cb) + sm;
                        int32_t u = 0;
                        while (u < K && ga[u] == gb[u]) u++;
                        K = std::min(K, u);
                    }
                if (!ok || K < TILE_MIN_SEGMENT) continue;
                const int32_t tile = T.n_tiles++;
                T.cid.resize((size_t)T.n_tiles * TRC, -1);
                for (int i = 0; i < nrow; i++)
                    for (int m = 0; m < ncol; m++) { T.cid[(size_t)tile * TRC + (size_t)i * TCL + m] = circ(i, m); T.tiled[(size_t)circ(i, m)] = 1; T.n_tiled++; }
                for (int i = 0; i < nrow; i++) fpath(circ(i, 0), fp[(size_t)i * TCL]);      // the row's forward states: column 0's circuit
                for (int m = 0; m < ncol; m++) rpath(circ(0, m), rcol[(size_t)m]);          // the column's backward states: row 0's circuit
                for (auto& v : g_slots) v.clear();
                {
                    const int32_t cc = circ(0, 0), s = seg0(0, 0);
                    for (int32_t u = 0; u < K; u++) g_slots[(size_t)gates(cc)[s + u]].push_back(u);
                }
                T.blk.resize((size_t)T.n_tiles * (nG + 1));
                for (int g = 0; g < nG; g++) {
                    T.blk[(size_t)tile * (nG + 1) + g] = (int32_t)(T.tsf.size() / (4 * TRW));
                    const auto& us = g_slots[(size_t)g];
                    for (size_t q = 0; q < us.size(); q += 4) {
                        const size_t fb = T.tsf.size(), rb = T.tsr.size();
                        T.tsf.resize(fb + 4 * TRW, -1); T.tsr.resize(rb + 4 * TCL, -1);
                        for (size_t sl = 0; sl < 4 && q + sl < us.size(); sl++) {
                            const int32_t u = us[q + sl];
                            for (int i = 0; i < nrow; i++) T.tsf[fb + sl * TRW + (size_t)i] = fp[(size_t)i * TCL][(size_t)(seg0(i, 0) + u)];
                            for (int m = 0; m < ncol; m++) {
                                const int32_t n = len(circ(0, m)), s = seg0(0, m);
                                T.tsr[rb + sl * TCL + (size_t)m] = rcol[(size_t)m][(size_t)(n - (s + u + 1))];
                            }
                        }
                    }
                    T.seg_slots += (int64_t)us.size();
                }
                T.blk[(size_t)tile * (nG + 1) + nG] = (int32_t)(T.tsf.size() / (4 * TRW));
                // remnants: per gate, per circuit slot, the applications outside [s, s + K) -- the circuit's OWN state ids
                T.rem_ptr.resize((size_t)T.n_tiles * nG * (TRC + 1));
                std::vector<std::vector<std::pair<int32_t, int32_t>>> per_gate((size_t)nG * TRC);
                std::vector<int32_t> ff, rr;
                for (int i = 0; i < nrow; i++)
                    for (int m = 0; m < ncol; m++) {
                        const int32_t cc = circ(i, m), n = len(cc), s = seg0(i, m);
                        fpath(cc, ff); rpath(cc, rr);
                        for (int32_t k = 1; k <= n; k++) {
                            if (k > s && k <= s + K) continue;
                            const int g = gates(cc)[k - 1];
                            per_gate[(size_t)g * TRC + (size_t)i * TCL + m].emplace_back(ff[(size_t)(k - 1)], rr[(size_t)(n - k)]);
                        }
                    }
                for (int g = 0; g < nG; g++) {
                    int32_t* rp = &T.rem_ptr[((size_t)tile * nG + g) * (TRC + 1)];
                    for (int sl = 0; sl < TRC; sl++) {
                        rp[sl] = (int32_t)(T.rem_f.size() / 4);
                        const auto& v = per_gate[(size_t)g * TRC + sl];
                        for (size_t q = 0; q < v.size(); q++) { T.rem_f.push_back(v[q].first); T.rem_r.push_back(v[q].second); }
                        while (T.rem_f.size() % 4) { T.rem_f.push_back(-1); T.rem_r.push_back(v.back().second); }
                        T.rem_slots += (int64_t)v.size();
                    }
                    rp[TRC] = (int32_t)(T.rem_f.size() / 4);
                }
                tile_work.push_back((int64_t)K * nrow * ncol);
            }
        }
    }
    if (std::getenv("GST_TILE_DEBUG")) {
        size_t big = 0, g2 = 0;
        for (auto& r : runs) big += r.members.size() >= 2;
        for (auto& kv : groups) g2 += kv.second.size() >= 2;
        std::fprintf(stderr, "[tiles] %zu runs (%zu with >= 2 members), %zu groups (%zu with >= 2 columns), %d tiles, %lld circuits\n",
                     runs.size(), big, groups.size(), g2, T.n_tiles, (long long)T.n_tiled);
    }
    T.order.resize((size_t)T.n_tiles);
    for (int32_t k = 0; k < T.n_tiles; k++) T.order[(size_t)k] = k;
    std::stable_sort(T.order.begin(), T.order.end(), [&](int32_t x, int32_t y) { return tile_work[(size_t)x] > tile_work[(size_t)y]; });
    if (T.tsf.empty()) { T.tsf.assign(4 * TRW, -1); T.tsr.assign(4 * TCL, -1); }
    if (T.rem_f.empty()) { T.rem_f.assign(4, -1); T.rem_r.assign(4, 0); }
}


// Analytic mode, D = 16: reversed plan + pair tables, built and uploaded once per plan.
int ensure_reverse(gst_plan* p)
{
    if (p->rev_ready) return GST_OK;
    const gst::HostPlan& h = p->hp;
    const int32_t rev_tasks = 0;
    std::string err = gst::build_reverse_plan(h, p->rev, rev_tasks, h.D == 16 ? 1 : (h.D == 64 ? 8 : 4));
    if (!err.empty()) return fail(GST_EINVAL, "reversed plan: " + err);
    if (p->rev.max_slots > (h.D == 64 ? 32 : 4)) return fail(GST_EUNSUPPORTED, "reversed plan needs too many save slots");
    std::vector<int32_t> pf, pr;
    std::vector<int64_t> pos_ptr;
    gst::build_pair_tables(h, p->rev, pf, pr, pos_ptr);
    HIP_TRY(p->d_rprog.ensure(p->rev.prog.size() + 64));
    HIP_TRY(hipMemsetAsync(p->d_rprog.p, 0, (p->rev.prog.size() + 64) * 4, p->stream));
    H2D_TRY(p, p->d_rprog.p, p->rev.prog.data(), p->rev.prog.size() * 4);
    HIP_TRY(p->d_rtask_off.ensure(p->rev.task_off.size()));
    H2D_TRY(p, p->d_rtask_off.p, p->rev.task_off.data(), p->rev.task_off.size() * 8);
    HIP_TRY(p->d_pos_ptr.ensure(pos_ptr.size()));
    H2D_TRY(p, p->d_pos_ptr.p, pos_ptr.data(), pos_ptr.size() * 8);
    int rc;
    std::vector<int32_t> zeros((size_t)h.n_circuits + 1, 0);
    if ((rc = upload_i32(p, p->d_reff_ptr, zeros))) return rc;
    if ((rc = upload_i32(p, p->d_rev_leaf, p->rev.circ_leaf))) return rc;
    if ((rc = upload_i32(p, p->d_pair_f, pf))) return rc;
    if ((rc = upload_i32(p, p->d_pair_r, pr))) return rc;
    if ((rc = upload_i32(p, p->d_circ_rho, h.circ_rho))) return rc;
    std::vector<int32_t> order((size_t)h.n_circuits);
    for (int64_t c = 0; c < h.n_circuits; c++) order[(size_t)c] = (int32_t)c;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return p->rev.circ_leaf[x] < p->rev.circ_leaf[y]; });
    const int nG = h.n_gates;
    // applications of every gate that two circuits have in common at their ends (equal backward-state ids)
    auto common_tail = [&](int32_t c, int32_t c2, int32_t* per_gate) {
        int64_t common = 0;
        for (int g = 0; g < nG; g++) {
            const int64_t a0 = pos_ptr[(size_t)c * nG + g], a1 = pos_ptr[(size_t)c * nG + g + 1];
            const int64_t b0 = pos_ptr[(size_t)c2 * nG + g], b1 = pos_ptr[(size_t)c2 * nG + g + 1];
            int64_t n = 0;
            while (n < a1 - a0 && n < b1 - b0 && pr[(size_t)(a1 - 1 - n)] == pr[(size_t)(b1 - 1 - n)]) n++;
            if (per_gate) per_gate[g] = (int32_t)n;
            common += n;
        }
        return common;
    };
    auto similar = [&](int32_t c, int32_t c2, int64_t common) {
        const int64_t longer = std::max(h.circ_ptr[c + 1] - h.circ_ptr[c], h.circ_ptr[c2 + 1] - h.circ_ptr[c2]);
        return common >= 8 && 2 * common >= longer;
    };
    // The work items of the item kernel over a set of circuits (suffix order in, locality order / pairing / ranges / block
    // stream out): once for ALL circuits (what every mode of the contraction can run on), and -- when tiles were formed --
    // once more for the circuits no tile holds.
    struct ItemBufs { DevBuf<int32_t>*order, *partner, *common, *bf1, *bf2, *br, *bptr; DevBuf<uint32_t>*range, *counter; };
    auto build_items = [&](std::vector<int32_t> order, const ItemBufs& B, int64_t* n_items_out) -> int {
        const int64_t nC = (int64_t)order.size();
        int rc = GST_OK;
        if (h.D == 16 && p->ana_germ_order && nC > 1) {
            // Locality of the FORWARD states.  Pure suffix order keeps the backward chains of neighbours together but walks
            // through every prefix family (preparation fiducial x germ) for each measurement fiducial and germ power, so the
            // forward chains -- 128 bytes per application of every item -- never stay in an XCD's 4 MB L2.  Runs of
            // neighbours that end alike (one germ power and measurement fiducial behind all the preparation fiducials)
            // are kept whole, and the runs are ordered by the forward-trie family their first member belongs to (the root
            // of its state's parent chain = the task of the forward plan): all the runs of one germ become consecutive, their
            // 16 forward chains (2 MB) stay in L2 while the germ's backward chains stream through once.
            std::vector<int32_t> root((size_t)h.n_state_ids, -2);
            auto root_of = [&](int32_t id) {
                int32_t r = id;
                while (root[(size_t)r] == -2 && h.node_parent[(size_t)r] >= 0) r = h.node_parent[(size_t)r];
                const int32_t top = root[(size_t)r] == -2 ? r : root[(size_t)r];
                for (int32_t q = id; q != r; q = h.node_parent[(size_t)q]) root[(size_t)q] = top;
                root[(size_t)r] = top;
                return top;
            };
            std::vector<int32_t> run_of((size_t)nC, 0), run_key;
            int32_t run = 0;
            run_key.push_back(root_of(h.circ_leaf[(size_t)order[0]]));
            for (int64_t k = 1; k < nC; k++) {
                const int32_t c = order[(size_t)k - 1], c2 = order[(size_t)k];
                if (!similar(c, c2, common_tail(c, c2, nullptr))) { run++; run_key.push_back(0x7fffffff); }
                run_of[(size_t)k] = run;
                run_key[(size_t)run] = std::min(run_key[(size_t)run], root_of(h.circ_leaf[(size_t)c2]));
            }
            std::vector<int32_t> posn((size_t)nC);
            for (int64_t k = 0; k < nC; k++) posn[(size_t)k] = (int32_t)k;
            std::stable_sort(posn.begin(), posn.end(), [&](int32_t x, int32_t y) { return run_key[(size_t)run_of[(size_t)x]] < run_key[(size_t)run_of[(size_t)y]]; });
            std::vector<int32_t> reordered((size_t)nC);
            for (int64_t k = 0; k < nC; k++) reordered[(size_t)k] = order[(size_t)posn[(size_t)k]];
            order.swap(reordered);
        }
        // Work items of the D = 16 contraction: a circuit, or TWO neighbours of the suffix order whose last applications
        // coincide (same germ power and measurement fiducial behind different preparation fiducials): over the common
        // tail their backward states are the same vectors and the kernel gathers them once for both.
        std::vector<int32_t> item_first, item_partner, item_common;
        const bool pairing = h.D == 16 && h.n_effects == 4 && p->ana_pairs;
        auto plain4 = [&](int32_t c) {
            if (h.eff_ptr[c + 1] - h.eff_ptr[c] != 4) return false;
            for (int x = 0; x < 4; x++) if (h.eff_label[(size_t)h.eff_ptr[c] + x] != x) return false;
            return true;
        };
        for (int64_t k = 0; k < nC; k++) {
            const int32_t c = order[(size_t)k];
            bool paired = false;
            if (pairing && k + 1 < nC) {
                const int32_t c2 = order[(size_t)k + 1];
                if (plain4(c) && plain4(c2)) {
                    std::vector<int32_t> cg((size_t)nG, 0);
                    const int64_t common = common_tail(c, c2, cg.data());
                    if (similar(c, c2, common)) {
                        item_first.push_back(c); item_partner.push_back(c2);
                        item_common.insert(item_common.end(), cg.begin(), cg.end());
                        paired = true;
                        k++;
                    }
                }
            }
            if (!paired) {
                item_first.push_back(c); item_partner.push_back(-1);
                item_common.insert(item_common.end(), (size_t)nG, 0);
            }
        }
        const int64_t n_items = (int64_t)item_first.size();
        if (h.D == 16 && p->ana_lpt && n_items > 1) {
            // Longest first: an item of depth 1,030 occupies its wavefront for a quarter of the whole launch, and the launch
            // ends with whichever long item was started last.  Items are bucketed by the power of two of their work and the
            // buckets dispatched in descending order; inside a bucket the locality order above (germ-major runs) is kept.
            std::vector<int32_t> bucket((size_t)n_items), perm((size_t)n_items);
            for (int64_t k = 0; k < n_items; k++) {
                int64_t w = h.circ_ptr[item_first[(size_t)k] + 1] - h.circ_ptr[item_first[(size_t)k]];
                if (item_partner[(size_t)k] >= 0) w += h.circ_ptr[item_partner[(size_t)k] + 1] - h.circ_ptr[item_partner[(size_t)k]];
                int b = 0;
                while ((w >> b) > 1) b++;
                bucket[(size_t)k] = b; perm[(size_t)k] = (int32_t)k;
            }
            std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) { return bucket[(size_t)x] > bucket[(size_t)y]; });
            std::vector<int32_t> f2((size_t)n_items), p2((size_t)n_items), c2(item_common.size());
            for (int64_t k = 0; k < n_items; k++) {
                f2[(size_t)k] = item_first[(size_t)perm[(size_t)k]]; p2[(size_t)k] = item_partner[(size_t)perm[(size_t)k]];
                std::copy(item_common.begin() + (size_t)perm[(size_t)k] * nG, item_common.begin() + ((size_t)perm[(size_t)k] + 1) * nG, c2.begin() + (size_t)k * nG);
            }
            item_first.swap(f2); item_partner.swap(p2); item_common.swap(c2);
        }
        if (h.D == 16) {
            if ((rc = upload_i32(p, (*B.order), item_first))) return rc;
            if ((rc = upload_i32(p, (*B.partner), item_partner))) return rc;
            if ((rc = upload_i32(p, (*B.common), item_common))) return rc;
        } else {
            if ((rc = upload_i32(p, (*B.order), order))) return rc;
        }
        // 8 contiguous ranges of the item list with equal numbers of gate applications (+ a constant per circuit)
        std::vector<uint32_t> range_begin(9, 0);
        {
            auto work = [&](int64_t k) {
                double w = (double)(h.circ_ptr[item_first[(size_t)k] + 1] - h.circ_ptr[item_first[(size_t)k]]) + 24.0;
                if (item_partner[(size_t)k] >= 0) w += (double)(h.circ_ptr[item_partner[(size_t)k] + 1] - h.circ_ptr[item_partner[(size_t)k]]) + 24.0;
                return w;
            };
            double total = 0;
            for (int64_t k = 0; k < n_items; k++) total += work(k);
            double acc = 0;
            int r = 1;
            for (int64_t k = 0; k < n_items && r < 8; k++) {
                acc += work(k);
                while (r < 8 && acc >= total * r / 8.0) range_begin[r++] = (uint32_t)(k + 1);
            }
            for (; r < 8; r++) range_begin[r] = (uint32_t)n_items;
            range_begin[8] = (uint32_t)n_items;
        }
        if (h.D != 16) {            // (the other contraction kernels index the plain permutation; their ranges are unused)
            for (int r = 0; r <= 8; r++) range_begin[r] = (uint32_t)(nC * r / 8);
        }
        HIP_TRY((*B.range).ensure(9));
        H2D_TRY(p, (*B.range).p, range_begin.data(), 9 * 4);
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY((*B.counter).ensure(8));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (h.D == 16 && pairing && p->ana_stream && nG <= 63) {
            // Two-circuit items as ONE stream of blocks of 4 "slots": per gate the common tail (slot = one application of
            // both circuits: their two forward ids and the shared backward id), then what each circuit has before the tail
            // (the other circuit's forward id = -1: its operand is zeroed), padded to a multiple of 4 with dead slots.  The
            // contraction's gather pipeline then runs through a whole item without draining at every gate and segment.
            const size_t blk_slots = 4 * (size_t)gst::analytic_stream_chunks();
            std::vector<int32_t> bf1, bf2, br, bptr((size_t)n_items * (size_t)nG + 1, 0);
            bf1.reserve(pf.size()); bf2.reserve(pf.size()); br.reserve(pf.size());
            for (int64_t k = 0; k < n_items; k++) {
                const int32_t c = item_first[(size_t)k], c2 = item_partner[(size_t)k];
                for (int g = 0; g < nG; g++) {
                    bptr[(size_t)k * nG + g] = (int32_t)(bf1.size() / blk_slots);
                    if (c2 < 0) continue;
                    const int64_t p0 = pos_ptr[(size_t)c * nG + g], p1 = pos_ptr[(size_t)c * nG + g + 1];
                    const int64_t q0 = pos_ptr[(size_t)c2 * nG + g], q1 = pos_ptr[(size_t)c2 * nG + g + 1];
                    const int64_t cg = item_common[(size_t)k * nG + g];
                    for (int64_t t = 0; t < cg; t++) { bf1.push_back(pf[(size_t)(p1 - cg + t)]); bf2.push_back(pf[(size_t)(q1 - cg + t)]); br.push_back(pr[(size_t)(p1 - cg + t)]); }
                    for (int64_t j = p0; j < p1 - cg; j++) { bf1.push_back(pf[(size_t)j]); bf2.push_back(-1); br.push_back(pr[(size_t)j]); }
                    for (int64_t j = q0; j < q1 - cg; j++) { bf1.push_back(-1); bf2.push_back(pf[(size_t)j]); br.push_back(pr[(size_t)j]); }
                    while (bf1.size() % blk_slots) { bf1.push_back(-1); bf2.push_back(-1); br.push_back(br.empty() ? 0 : br.back()); }
                }
                if (bf1.size() / 4 > 0x1ffffff0u) return fail(GST_EUNSUPPORTED, "analytic block stream too long");
            }
            bptr[(size_t)n_items * nG] = (int32_t)(bf1.size() / blk_slots);
            if (bf1.empty()) { bf1.assign(blk_slots, -1); bf2.assign(blk_slots, -1); br.assign(blk_slots, 0); }
            if ((rc = upload_i32(p, (*B.bf1), bf1))) return rc;
            if ((rc = upload_i32(p, (*B.bf2), bf2))) return rc;
            if ((rc = upload_i32(p, (*B.br), br))) return rc;
            if ((rc = upload_i32(p, (*B.bptr), bptr))) return rc;
            HIP_TRY(hipStreamSynchronize(p->stream));
        }
        *n_items_out = n_items;
        return GST_OK;
    };
    // tiles first (they need the reversed plan's state graph, like the level programs below)
    std::vector<int32_t> leftover;
    p->n_tiles = 0; p->n_lo_circuits = 0;
    if (h.D == 16 && h.n_effects == 4 && p->ana_pairs && p->ana_stream && p->ana_tiles && nG <= 63 && h.n_circuits >= 64) {
        TileSet T;
        build_tiles(h, p->rev, order, T);
        if (T.n_tiles > 0) {
            if ((rc = upload_i32(p, p->d_tile_order, T.order)) || (rc = upload_i32(p, p->d_tile_cid, T.cid)) || (rc = upload_i32(p, p->d_tile_blk, T.blk)) ||
                (rc = upload_i32(p, p->d_tsf, T.tsf)) || (rc = upload_i32(p, p->d_tsr, T.tsr)) || (rc = upload_i32(p, p->d_rem_ptr, T.rem_ptr)) ||
                (rc = upload_i32(p, p->d_rem_f, T.rem_f)) || (rc = upload_i32(p, p->d_rem_r, T.rem_r))) return rc;
            HIP_TRY(p->d_tile_counter.ensure(1));
            HIP_TRY(hipStreamSynchronize(p->stream));
            p->n_tiles = T.n_tiles;
            p->tile_stats[0] = T.n_tiled; p->tile_stats[1] = T.seg_slots; p->tile_stats[2] = T.rem_slots;
            for (int32_t c : order) if (!T.tiled[(size_t)c]) leftover.push_back(c);
            p->n_lo_circuits = (int64_t)leftover.size();
        }
    }
    int64_t n_items = 0;
    {
        ItemBufs B{&p->d_circ_order, &p->d_circ_partner, &p->d_pair_common, &p->d_blk_f1, &p->d_blk_f2, &p->d_blk_r, &p->d_blk_ptr, &p->d_range_begin, &p->d_work_counter};
        if ((rc = build_items(order, B, &n_items))) return rc;
    }
    if (p->n_tiles > 0 && !leftover.empty()) {
        ItemBufs B{&p->d_lo_order, &p->d_lo_partner, &p->d_lo_common, &p->d_lo_blk_f1, &p->d_lo_blk_f2, &p->d_lo_blk_r, &p->d_lo_blk_ptr, &p->d_lo_range_begin, &p->d_lo_counter};
        int64_t n_lo_items = 0;
        if ((rc = build_items(leftover, B, &n_lo_items))) return rc;
    }
    if (h.D == 16) build_levels_host(p, true);      // (needs the reversed plan's state graph, dropped below; whatever GST_OPT_FAST_CHAINS says NOW)
    // (the host copies of the reversed programs are not needed any more)
    p->rev.prog.clear(); p->rev.prog.shrink_to_fit();
    p->rev.node_parent.clear(); p->rev.node_parent.shrink_to_fit();
    p->rev.node_sym.clear(); p->rev.node_sym.shrink_to_fit();
    p->rev_ready = true;
    return GST_OK;
}

// Exact Jacobian columns (GST_DERIV_ANALYTIC) into device memory.
int run_dprobs_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                        int64_t n_param, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    if (p->comp_index >= 0 && !p->derivs_set)
        return fail(GST_EUNSUPPORTED, "a complement effect is declared: exact derivatives of TP POVMs need gst_set_derivs");
    if (h.D != 4 && h.D != 16 && h.D != 64) return fail(GST_EUNSUPPORTED, "the analytic mode supports D = 4, 16 and 64");
    if (h.D == 64 && !p->ana_mfma) return fail(GST_EUNSUPPORTED, "D = 64 analytic derivatives exist on the MFMA path only");
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    // (the backward chain pass needs only the model arrays already on their way: it forks here, onto the second stream --
    //  only where the two-cache contraction will run: a plain 1Q Jacobian is launch-bound and takes the single kernel)
    const bool will_fork = p->ana_mfma && (h.D != 4 || p->want_cache_path);
    if (will_fork) HIP_TRY(hipEventRecord(p->ev_fork, p->stream));
    // Forward states: the sequential walk (bit-identical probabilities), or -- this mode has no ordering contract -- the
    // log-depth level pass where the plan's germ-power paths make it pay (GST_OPT_FAST_CHAINS; probabilities <= 1e-10)
    int rc;
    bool lv_f = false;
    p->last_levels = false;
    if (h.D == 16 && will_fork && n_param > 0 && p->fast_chains) {
        if ((rc = ensure_levels(p, false))) return rc;
        lv_f = levels_wanted(p, p->lv_fwd);
    }
    rc = lv_f ? run_levels_forward(p, d_base)
              : run_probs(p, d_base, n_param > 0, will_fork && n_param > 0 ? 2 : 1, nullptr, will_fork && n_param > 0);   // probabilities + every forward state
    if (rc) return rc;
    if (n_param == 0) return GST_OK;
    const bool request_was_cached = p->request_cached(2, param_idx, dest_idx, n_param);
    if (!request_was_cached) {
        p->request_serial++;
        const int D = h.D, DD = D * D;
        std::vector<int32_t> cm_gate((size_t)std::max(h.n_gates, 1) * DD, -1), cm_rho((size_t)h.n_rhos * D, -1),
            cm_eff((size_t)h.n_effects * D, -1), col0(std::max(h.n_gates, 1), -2);
        std::vector<int64_t> none_cols;
        for (int64_t c = 0; c < n_param; c++) {
            const int64_t pi = param_idx[c];
            const int32_t col = (int32_t)(dest_idx ? dest_idx[c] : c);
            switch (p->pkind[pi]) {
            case GST_KIND_GATE: cm_gate[(size_t)p->pobj[pi] * DD + p->pelem[pi]] = col; break;
            case GST_KIND_RHO: cm_rho[(size_t)p->pobj[pi] * D + p->pelem[pi]] = col; break;
            case GST_KIND_EFFECT: cm_eff[(size_t)p->pobj[pi] * D + p->pelem[pi]] = col; break;
            default: none_cols.push_back(col);
            }
        }
        for (int g = 0; g < h.n_gates; g++) {
            const int32_t* m = cm_gate.data() + (size_t)g * DD;
            bool any = false, contiguous = m[0] >= 0;
            for (int k = 0; k < DD; k++) { any = any || m[k] >= 0; contiguous = contiguous && m[k] == m[0] + k; }
            col0[g] = contiguous ? m[0] : (any ? -1 : -2);
        }
        if (!p->graph_uploaded) {
            if ((rc = upload_i32(p, p->d_node_parent, h.node_parent))) return rc;
            if ((rc = upload_i32(p, p->d_node_sym, h.node_sym))) return rc;
            {   // run[id] = 1 + run[id-1] while parent(id) == id-1 is a gate state reached by a consecutive id
                std::vector<int32_t> run(h.n_state_ids, 0);
                for (int64_t i = 1; i < h.n_state_ids; i++)
                    if (h.node_parent[i] == i - 1) run[i] = 1 + ((h.node_parent[i - 1] >= 0 && h.node_parent[i - 1] == i - 2) ? run[i - 1] : 0);
                if ((rc = upload_i32(p, p->d_node_run, run))) return rc;
                HIP_TRY(hipStreamSynchronize(p->stream));
            }
            if ((rc = upload_i32(p, p->d_circ_leaf, h.circ_leaf))) return rc;
            p->graph_uploaded = true;
        }
        p->cached_kind = 0;
        if ((rc = upload_i32(p, p->d_gate_col0, col0))) return rc;
        if ((rc = upload_i32(p, p->d_cm_gate, cm_gate))) return rc;
        if ((rc = upload_i32(p, p->d_cm_rho, cm_rho))) return rc;
        if ((rc = upload_i32(p, p->d_cm_eff, cm_eff))) return rc;
        HIP_TRY(hipStreamSynchronize(p->stream));          // host vectors above go out of scope
        p->remember_request(2, param_idx, dest_idx, n_param);
        p->cached_none_cols = none_cols;
    }
    const int D = h.D;
    const std::vector<int64_t>& none_cols = p->cached_none_cols;
    for (int64_t col : none_cols)                      // parameters of objects this atom never applies: exact zeros
        HIP_TRY(hipMemset2DAsync(d_out + col, (size_t)ld * 8, 0, 8, (size_t)h.n_elements, p->stream));
    gst::AnaArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_circuits = h.n_circuits;
    a.circ_leaf = p->d_circ_leaf.p; a.node_parent = p->d_node_parent.p; a.node_sym = p->d_node_sym.p; a.node_run = p->d_node_run.p;
    a.eff_ptr = p->d_eff_ptr.p; a.eff_label = p->d_eff_label.p; a.eff_dest = p->d_eff_dest.p;
    a.gates_t = p->d_gates_t.p; a.effects = p->d_effects.p; a.base_cache = p->d_base_cache.p;
    a.n_gates = h.n_gates; a.n_rhos = h.n_rhos; a.n_effects = h.n_effects;
    a.gate_col0 = p->d_gate_col0.p; a.colmap_gate = p->d_cm_gate.p; a.colmap_rho = p->d_cm_rho.p; a.colmap_eff = p->d_cm_eff.p;
    a.out = d_out; a.ld = ld;
    // The MFMA kernels address both state caches with a uniform 64-bit base + 32-bit per-lane byte offsets; a cache of
    // 4 GB or more selects their WIDE instantiation (64-bit lane offsets: two more address registers per gather in
    // flight), nothing is refused.  (GST_TEST_FORCE cache_limit=: tests lower the 4 GB so that a small plan takes that form.)
    const double cache_limit = p->test_cache_limit > 0 ? p->test_cache_limit : 4.0e9;
    const bool caches_small = (double)h.n_state_ids * D * 8 < cache_limit;
    // the two-cache contraction: MFMA at D = 16 / 64; at D = 4 (VALU) only when a Hessian needs its tables -- a plain 1Q
    // Jacobian is launch-bound and the single backward-walking kernel below is one launch instead of three
    // D <= 16: the backward pass needs the chain kernel, whose tables (all gates, effects, emit ring) live in LDS; a gate
    // set too large for it takes the single-kernel path below (Jacobians) or is refused (Hessians need the caches)
    const bool chain_ok = D == 64 || gst::chain_kernel_fits(D, h.n_gates, h.n_effects, 4);
    if (!chain_ok && p->want_cache_path)
        return fail(GST_EUNSUPPORTED, "exact Hessians at D <= 16 need the gate set in LDS (at most " +
                                          std::to_string(128 * 1024 / (D * D * 8)) + " gates at this D)");
    bool rev_small = true;
    if (will_fork && chain_ok) {
        if ((rc = ensure_reverse(p))) return rc;
        rev_small = (double)p->rev.n_state_ids * h.n_effects * D * 8 < cache_limit;
        if (h.D == 64 && will_fork && p->chain_resident && p->fast_chains && !p->d_c64_order.p &&
            gst::chain64_resident_fits(std::min(16, h.n_effects), p->rev.max_slots, h.n_gates)) {
            // backward walk with resident gates (chain64_resident_kernel): its tasks longest first, the pop counter
            const int64_t nT = p->rev.n_tasks();
            std::vector<uint32_t> order((size_t)nT);
            for (int64_t t = 0; t < nT; t++) order[(size_t)t] = (uint32_t)t;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return p->rev.task_applies[x] > p->rev.task_applies[y]; });
            HIP_TRY(p->d_c64_order.ensure((size_t)nT));
            HIP_TRY(p->d_c64_counter.ensure(1));
            H2D_TRY(p, p->d_c64_order.p, order.data(), (size_t)nT * sizeof(uint32_t));
            HIP_TRY(hipStreamSynchronize(p->stream));          // once per plan: the second stream's fork event is already recorded
        }
    }
    if (will_fork && chain_ok) {
        a.wide = (caches_small && rev_small) ? 0 : 1;
        // backward states: the chain kernel over the reversed plan, transposed gates (= the row-major array), one lane
        // group per effect (64/D effects per pass)
        gst::WalkArgs w;
        std::memset(&w, 0, sizeof(w));
        w.prog = p->d_rprog.p; w.task_off = p->d_rtask_off.p;
        w.eff_ptr = p->d_reff_ptr.p; w.eff_label = p->d_reff_ptr.p; w.eff_dest = p->d_reff_ptr.p;
        w.gates = p->d_gates_t.p; w.gates_t = p->d_gates.p;          // (G^T)^T = G: the roles of the two layouts swap
        w.rhos = p->d_effects.p; w.effects = p->d_effects.p;
        w.n_gates = h.n_gates; w.n_effects = 0;
        w.n_pwaves = 1; w.rows_S = 0; w.mode = gst::EMIT_PROBS; w.out = p->d_pbase.p;
        HIP_TRY(p->d_rev_cache.ensure((size_t)p->rev.n_state_ids * h.n_effects * D));
        w.base_cache_w = p->d_rev_cache.p;
        w.multi_start = h.n_effects;
        w.chain_share = 2;                      // (the forward pass runs beside this one)
        TIME_REC(p, evk0);
        // Both chain passes are latency-bound (one wavefront per task, a fraction of the SIMDs): the backward one runs
        // on the second stream beside the forward pass launched above, and the contraction waits for both.
        HIP_TRY(hipStreamWaitEvent(p->stream2, p->ev_fork, 0));
        bool lv_r = false;
        if (D == 16 && p->fast_chains) {
            if ((rc = ensure_levels(p, true))) return rc;
            lv_r = levels_wanted(p, p->lv_rev);
        }
        if (lv_r) {                                            // backward states by the level pass: one launch, all effects
            gst::LevelArgs ra;
            level_args(p->lv_rev, ra);
            ra.bmats = p->d_gates.p; ra.starts = p->d_effects.p; ra.cache = p->d_rev_cache.p;
            HIP_TRY(gst::launch_level_pass(ra, p->rev.n_tasks(), p->stream2));
            p->last_launches++;
        } else if (D == 64 && p->fast_chains && gst::chain64_fits(std::min(16, h.n_effects), p->rev.max_slots)) {      // all effects of a task as one row block on the matrix cores
            // opt-in (GST_TEST_FORCE chain_resident=1): <= 10 gates resident in registers, one persistent workgroup per CU
            const bool resident = p->chain_resident && p->d_c64_order.p != nullptr;          // (task order uploaded above)
            for (int e0 = 0; e0 < h.n_effects; e0 += 16) {
                w.start0 = e0;
                if (resident && std::min(16, h.n_effects - e0) > 1) {
                    w.block_order = p->d_c64_order.p; w.bin_head = p->d_c64_counter.p;
                    // after the forward pass, not beside it: a 452-register wavefront shares its SIMD with nothing
                    hipStream_t sb = p->stream;
                    HIP_TRY(hipMemsetAsync(p->d_c64_counter.p, 0, sizeof(uint32_t), sb));
                    HIP_TRY(gst::launch_chain64_resident(w, p->rev.n_tasks(), p->rev.max_slots, p->n_cus, sb));
                    w.block_order = nullptr; w.bin_head = nullptr;
                } else {
                    HIP_TRY(gst::launch_chain64(w, p->rev.n_tasks(), p->rev.max_slots, p->stream2));
                }
                p->last_launches++;
            }
        } else if (D == 64) {                                  // one wavefront per (task, effect), a single launch
            w.start0 = 0; w.n_pwaves = h.n_effects;
            HIP_TRY(gst::launch_walk_rows(D, w, p->rev.n_tasks(), p->rev.max_slots, p->stream2));
            p->last_launches++;
        } else {
            for (int e0 = 0; e0 < h.n_effects; e0 += 64 / D) {  // four effects (lane groups) per pass of the chain kernel
                w.start0 = e0;
                HIP_TRY(gst::launch_walk_rows(D, w, p->rev.n_tasks(), p->rev.max_slots, p->stream2));
                p->last_launches++;
            }
        }
        HIP_TRY(hipEventRecord(p->ev_join, p->stream2));
        HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_join, 0));
        a.rev_cache = p->d_rev_cache.p; a.rev_leaf = p->d_rev_leaf.p; a.pair_f = p->d_pair_f.p; a.pair_r = p->d_pair_r.p;
        a.circ_partner = D == 16 ? p->d_circ_partner.p : nullptr; a.pair_common = D == 16 ? p->d_pair_common.p : nullptr;
        a.pos_ptr = p->d_pos_ptr.p; a.circ_rho = p->d_circ_rho.p; a.circ_order = p->d_circ_order.p; a.work_counter = p->d_work_counter.p; a.range_begin = p->d_range_begin.p;
        a.group_fetch = p->ana_group_fetch ? 1 : 0;
        if (D == 16 && p->d_blk_ptr.p) { a.blk_f1 = p->d_blk_f1.p; a.blk_f2 = p->d_blk_f2.p; a.blk_r = p->d_blk_r.p; a.blk_ptr = p->d_blk_ptr.p; }
        // GST_OPT_ANALYTIC_KEEP_ZEROS: the blocks of gates an item never applies are exact zeros; when THIS destination got
        // THIS request last time (and the caller promised, by setting the option, to write nothing but row scalings into it
        // in between) they are zero already and are not stored again -- a third of the D = 16 contraction's stores
        // Without the option (value 2, the default) the same holds for destinations the library can vouch for: memory from
        // gst_device_malloc_tracked and the plan's own staging buffer, whose every other writer reports to gst_track.cpp.
        const bool same_dest = p->ana_zero_out == (const void*)d_out && p->ana_zero_ld == ld && p->ana_zero_valid && request_was_cached;
        const bool zero_form = D == 16 && (!p->derivs_set || p->jelem_call) && !p->want_cache_path;
        const size_t extent = jac_extent(nE_total(p), ld, dest_idx, n_param);
        const uint64_t sig = (p->uid * 0x9E3779B97F4A7C15ull) ^ ((p->jelem_call ? ~0ull : p->request_serial) * 0xC2B2AE3D27D4EB4Full) ^ (uint64_t)ld;
        bool claim = false;
        a.zeros_resident = 0; a.zeros_ok = nullptr;
        if (zero_form && p->ana_keep_zeros == 1) a.zeros_resident = same_dest ? 1 : 0;
        else if (zero_form && p->ana_keep_zeros == 2 && (d_out == p->d_out.p || d_out == p->d_jelem.p || gst::track_owned(d_out, extent))) {
            claim = true;
            if (const uint32_t* w = gst::track_claim_find(d_out, extent, sig)) { a.zeros_resident = 1; a.zeros_ok = w; }
        }
        if (!claim) gst::track_touch(d_out, extent);
        p->last_zeros_resident = a.zeros_resident != 0;
        p->ana_zero_out = d_out; p->ana_zero_ld = ld; p->ana_zero_valid = (D == 16);
        // D = 16, a plain Jacobian fill: the circuits of the tiles on the tile kernel, the rest on the item kernel (disjoint rows)
        const bool use_tiles = D == 16 && p->n_tiles > 0 && !a.wide && !p->want_cache_path && a.blk_ptr != nullptr;
        p->last_tiles = use_tiles;
        if (use_tiles) {
            gst::TileArgs t;
            std::memset(&t, 0, sizeof(t));
            t.a = a;
            t.n_tiles = p->n_tiles; t.debug = p->tile_dbg;
            t.tile_order = p->d_tile_order.p; t.tile_cid = p->d_tile_cid.p; t.tile_blk = p->d_tile_blk.p; t.tsf = p->d_tsf.p; t.tsr = p->d_tsr.p;
            t.rem_ptr = p->d_rem_ptr.p; t.rem_f = p->d_rem_f.p; t.rem_r = p->d_rem_r.p; t.counter = p->d_tile_counter.p;
            HIP_TRY(hipMemsetAsync(p->d_tile_counter.p, 0, sizeof(uint32_t), p->stream));
            HIP_TRY(gst::launch_analytic_tiles(t, p->n_cus, p->stream));
            p->last_launches++;
            if (p->n_lo_circuits > 0) {
                gst::AnaArgs lo = a;
                lo.n_circuits = p->n_lo_circuits;
                lo.circ_order = p->d_lo_order.p; lo.circ_partner = p->d_lo_partner.p; lo.pair_common = p->d_lo_common.p;
                lo.blk_f1 = p->d_lo_blk_f1.p; lo.blk_f2 = p->d_lo_blk_f2.p; lo.blk_r = p->d_lo_blk_r.p; lo.blk_ptr = p->d_lo_blk_ptr.p;
                lo.range_begin = p->d_lo_range_begin.p; lo.work_counter = p->d_lo_counter.p;
                HIP_TRY(hipMemsetAsync(p->d_lo_counter.p, 0, 8 * sizeof(uint32_t), p->stream));
                HIP_TRY(gst::launch_analytic_mfma(lo, p->stream));
                p->last_launches++;
            }
        } else {
        HIP_TRY(hipMemsetAsync(p->d_work_counter.p, 0, 8 * sizeof(uint32_t), p->stream));
        if (D == 64) HIP_TRY(gst::launch_analytic_mfma64(a, p->stream));
        else if (D == 16) HIP_TRY(gst::launch_analytic_mfma(a, p->stream));
        else HIP_TRY(gst::launch_analytic_small(a, p->stream));
        p->last_launches++;
        }
        TIME_REC(p, evk1);
        if (claim) {         // what this fill leaves behind; the word reads 1 again whatever a row scaling did to it before
            if (uint32_t* w = gst::track_claim_set(d_out, extent, sig, p->device, p->uid)) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)w, 1, 1, p->stream));
        }
        p->last_ana = a; p->last_ana_valid = true;     // (the Hessian rows re-launch the contraction with other caches)
        p->last_ana.zeros_resident = 0; p->last_ana.zeros_ok = nullptr;
        return GST_OK;
    }
    if (D == 64) return fail(GST_EUNSUPPORTED, "D = 64 analytic derivatives exist on the MFMA path only");
    gst::track_touch(d_out, jac_extent(h.n_elements, ld, dest_idx, n_param));
    p->last_zeros_resident = false;
    TIME_REC(p, evk0);
    HIP_TRY(gst::launch_analytic(D, a, p->stream));
    TIME_REC(p, evk1);
    p->last_launches++;
    return GST_OK;
}


// The element Jacobian [nE][n_el] ([rhos | effects | gates] of the `full` layout) of the current model into d_jelem,
// through the ordinary analytic path with the identity element map.
int run_element_jacobian(gst_plan* p, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements;
    const int64_t n_el = (int64_t)h.n_rhos * D + (int64_t)h.n_effects * D + (int64_t)h.n_gates * D * D;
    std::vector<int32_t> ek((size_t)n_el), eo((size_t)n_el), ee((size_t)n_el);
    {
        int64_t q = 0;
        for (int r = 0; r < h.n_rhos; r++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_RHO; eo[q] = r; ee[q] = j; }
        for (int e = 0; e < h.n_effects; e++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_EFFECT; eo[q] = e; ee[q] = j; }
        for (int g = 0; g < h.n_gates; g++) for (int j = 0; j < D * D; j++, q++) { ek[q] = GST_KIND_GATE; eo[q] = g; ee[q] = j; }
    }
    std::vector<int64_t> all((size_t)n_el);
    for (int64_t q = 0; q < n_el; q++) all[(size_t)q] = q;
    if (p->d_jelem.p && (size_t)(nE * n_el) > p->d_jelem.n) gst::track_touch(p->d_jelem.p, p->d_jelem.n * 8);      // (about to be re-allocated)
    HIP_TRY(p->d_jelem.ensure((size_t)std::max<int64_t>(nE * n_el, 1)));
    p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
    p->cached_kind = 0;
    // (the scratch is written by this call only and always with the same column map: its structural zeros stay resident)
    p->jelem_call = true;
    int rc = run_dprobs_analytic(p, p->d_jelem.p, n_el, all.data(), nullptr, n_el, d_probs_out);
    p->jelem_call = false;
    p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
    p->cached_kind = 0;
    return rc;
}

// Which rows of the element Jacobian can be non-zero in which gate's columns: bit g of word r <=> some circuit with an
// outcome among rows 32 r .. 32 r + 31 applies gate g (the chain-rule product skips the rest: launch_chain_rule_gemm).
// Built once per plan from the expanded circuits; padded to whole 128-row workgroup tiles.
int ensure_rowmask(gst_plan* p)
{
    if (p->rowmask_built) return GST_OK;
    p->rowmask_built = true;
    const gst::HostPlan& h = p->hp;
    if (h.n_gates > 64 || h.circ_ptr.size() != (size_t)h.n_circuits + 1 || h.eff_ptr.size() != (size_t)h.n_circuits + 1) return GST_OK;
    std::vector<uint64_t> mask((size_t)((h.n_elements + 127) / 128) * 4, 0);
    for (int64_t c = 0; c < h.n_circuits; c++) {
        uint64_t gs = 0;
        for (int64_t k = h.circ_ptr[(size_t)c]; k < h.circ_ptr[(size_t)c + 1]; k++) gs |= 1ull << h.circ_gates[(size_t)k];
        for (int32_t x = h.eff_ptr[(size_t)c]; x < h.eff_ptr[(size_t)c + 1]; x++) mask[(size_t)(h.eff_dest[(size_t)x] >> 5)] |= gs;
    }
    HIP_TRY(p->d_rowmask.ensure(mask.size()));
    H2D_TRY(p, p->d_rowmask.p, mask.data(), mask.size() * 8);
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->rowmask_ok = true;
    return GST_OK;
}

// GST_DERIV_ANALYTIC with gst_set_derivs: element Jacobian (the `full` layout [rhos | effects | gates]) into scratch,
// then one MFMA chain-rule product per object, accumulated into the requested parameter columns.
int run_dprobs_general(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                       int64_t n_param, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements;
    const int64_t n_el = (int64_t)h.n_rhos * D + (int64_t)h.n_effects * D + (int64_t)h.n_gates * D * D;
    for (int64_t c = 0; c < n_param; c++)
        if (param_idx[c] < 0 || param_idx[c] >= p->dv_n_params) return fail(GST_EINVAL, "parameter index out of range");
    std::vector<int32_t> dest_of((size_t)p->dv_n_params, -1);
    for (int64_t c = 0; c < n_param; c++) {
        if (dest_of[(size_t)param_idx[c]] >= 0) return fail(GST_EINVAL, "a parameter is requested twice (not supported with gst_set_derivs)");
        dest_of[(size_t)param_idx[c]] = (int32_t)(dest_idx ? dest_idx[c] : c);
    }
    int rc = run_element_jacobian(p, d_probs_out);
    if (rc) return rc;
    if (n_param == 0) return GST_OK;
    // Object by object: the first object that maps onto a destination column STORES its product there, later ones (shared
    // parameters: the effects of a POVM) add to it.  Only when some requested column is reached by no object at all are
    // the columns zeroed first (8.4 GB of memset and as much read-modify-write traffic for the 2Q CPTPLND Jacobian otherwise).
    std::vector<uint8_t> first_writer(p->dv_kind.size(), 0);
    bool all_covered = true;
    {
        std::vector<uint8_t> seen((size_t)p->dv_n_params, 0);
        for (size_t o = 0; o < p->dv_kind.size(); o++) {
            bool any_seen = false, any_new = false;
            for (int64_t c = p->dv_off_cols[o]; c < p->dv_off_cols[o + 1]; c++) {
                const int64_t gp = p->dv_param_idx[(size_t)c];
                if (dest_of[(size_t)gp] < 0) continue;
                if (seen[(size_t)gp]) any_seen = true; else any_new = true;
            }
            first_writer[o] = (any_new && !any_seen) ? 1 : 0;
            // an object that is neither purely first nor purely later (it shares SOME columns) must add: those columns need zeros
            if (any_new && any_seen) all_covered = false;
            for (int64_t c = p->dv_off_cols[o]; c < p->dv_off_cols[o + 1]; c++) seen[(size_t)p->dv_param_idx[(size_t)c]] = 1;
        }
        for (int64_t c = 0; c < n_param; c++) all_covered = all_covered && seen[(size_t)param_idx[c]];
        if (!all_covered) std::fill(first_writer.begin(), first_writer.end(), (uint8_t)0);
    }
    bool window = true;
    for (int64_t c = 1; dest_idx && c < n_param; c++) window = window && dest_idx[c] == dest_idx[0] + c;
    if (all_covered) {
        // (every requested column gets its first value by a store)
    } else if (window) {
        const int64_t d0 = dest_idx ? dest_idx[0] : 0;
        HIP_TRY(hipMemset2DAsync(d_out + d0, (size_t)ld * 8, 0, (size_t)n_param * 8, (size_t)nE, p->stream));
    } else {
        for (int64_t c = 0; c < n_param; c++) HIP_TRY(hipMemset2DAsync(d_out + dest_idx[c], (size_t)ld * 8, 0, 8, (size_t)nE, p->stream));
    }
    const int64_t base_rho = 0, base_eff = (int64_t)h.n_rhos * D, base_gate = base_eff + (int64_t)h.n_effects * D;
    HIP_TRY(p->d_dv_colmap.ensure((size_t)std::max<int64_t>(p->dv_off_cols.back(), 1)));
    std::vector<int32_t> colmap((size_t)p->dv_off_cols.back());
    for (size_t c = 0; c < colmap.size(); c++) colmap[c] = dest_of[(size_t)p->dv_param_idx[c]];
    if (!colmap.empty()) H2D_TRY(p, p->d_dv_colmap.p, colmap.data(), colmap.size() * 4);
    HIP_TRY(hipStreamSynchronize(p->stream));          // `colmap` goes out of scope
    if ((rc = ensure_rowmask(p))) return rc;
    for (size_t o = 0; o < p->dv_kind.size(); o++) {
        const int k = p->dv_kind[o];
        int K = k == GST_KIND_GATE ? D * D : D;
        const int64_t a0 = (k == GST_KIND_GATE ? base_gate : k == GST_KIND_RHO ? base_rho : base_eff) + (int64_t)p->dv_obj[o] * K;
        // The effects of one POVM share their parameters: as separate objects each of them would read-modify-write the same
        // nE x n columns (1 GB apiece at 2Q).  Consecutive objects of one kind whose element columns, derivative matrices
        // and destination columns line up are ONE product with the K's stacked.
        size_t last = o;
        while (last + 1 < p->dv_kind.size() && p->dv_kind[last + 1] == k && p->dv_obj[last + 1] == p->dv_obj[last] + 1 &&
               p->dv_ncols[last + 1] == p->dv_ncols[o] && !first_writer[last + 1] &&
               p->dv_off_deriv[last + 1] == p->dv_off_deriv[last] + (int64_t)(k == GST_KIND_GATE ? D * D : D) * p->dv_ncols[o] &&
               std::equal(colmap.begin() + p->dv_off_cols[o], colmap.begin() + p->dv_off_cols[o + 1], colmap.begin() + p->dv_off_cols[last + 1]))
            last++;
        K *= (int)(last - o + 1);
        HIP_TRY(gst::launch_chain_rule_gemm(p->d_jelem.p, n_el, a0, K, p->d_dv_deriv.p + p->dv_off_deriv[o], p->dv_ncols[o],
                                            p->d_dv_colmap.p + p->dv_off_cols[o], d_out, ld, nE, p->stream, first_writer[o] != 0,
                                            (k == GST_KIND_GATE && last == o && p->rowmask_ok) ? p->d_rowmask.p : nullptr, p->dv_obj[o]));
        p->last_launches++;
        o = last;
    }
    return GST_OK;
}

}  // namespace gst_impl

extern "C" {

int gst_set_derivs(gst_plan* p, int32_t n_params, int32_t n_objs, const int32_t* kind, const int32_t* obj,
                   const int32_t* n_cols, const int64_t* param_idx, const double* deriv)
{
    return guarded([&]() -> int {
    if (!p || n_params < 0 || n_objs < 0) return fail(GST_EINVAL, "bad argument");
    p->cached_kind = 0;
    if (n_objs == 0) { p->derivs_set = false; p->dv2_set = false; p->dv2_off.clear(); return GST_OK; }
    if (!kind || !obj || !n_cols || !param_idx || !deriv) return fail(GST_EINVAL, "bad argument");
    const int D = p->hp.D, Du = p->user_D();
    std::vector<double> padded;            // (a padded dimension: the caller's [Du*Du or Du][n] matrices with their rows moved)
    if (Du != D) {
        int64_t nd = 0, ns = 0;
        for (int32_t o = 0; o < n_objs; o++) nd += (int64_t)(kind[o] == GST_KIND_GATE ? D * D : D) * std::max(n_cols[o], 0);
        padded.assign((size_t)nd, 0.0);
        nd = 0;
        for (int32_t o = 0; o < n_objs; o++) {
            const int64_t n = std::max(n_cols[o], 0);
            const bool gate = kind[o] == GST_KIND_GATE;
            for (int e = 0; e < (gate ? Du * Du : Du); e++) {
                const int ep = gate ? (e / Du) * D + e % Du : e;
                std::memcpy(&padded[(size_t)(nd + (int64_t)ep * n)], deriv + ns + (int64_t)e * n, (size_t)n * 8);
            }
            nd += (int64_t)(gate ? D * D : D) * n; ns += (int64_t)(gate ? Du * Du : Du) * n;
        }
        deriv = padded.data();
    }
    std::vector<int64_t> off_c((size_t)n_objs + 1, 0), off_d((size_t)n_objs + 1, 0);
    for (int32_t o = 0; o < n_objs; o++) {
        const int k = kind[o];
        const int nobj = k == GST_KIND_GATE ? p->hp.n_gates : k == GST_KIND_RHO ? p->hp.n_rhos : k == GST_KIND_EFFECT ? p->hp.n_effects : -1;
        if (nobj < 0 || obj[o] < 0 || obj[o] >= nobj || n_cols[o] < 0) return fail(GST_EINVAL, "derivative object " + std::to_string(o) + " out of range");
        off_c[o + 1] = off_c[o] + n_cols[o];
        off_d[o + 1] = off_d[o] + (int64_t)(k == GST_KIND_GATE ? D * D : D) * n_cols[o];
    }
    for (int64_t c = 0; c < off_c[n_objs]; c++)
        if (param_idx[c] < 0 || param_idx[c] >= n_params) return fail(GST_EINVAL, "derivative parameter index out of range");
    int rc = ensure_device(p);
    if (rc) return rc;
    p->dv_kind.assign(kind, kind + n_objs); p->dv_obj.assign(obj, obj + n_objs); p->dv_ncols.assign(n_cols, n_cols + n_objs);
    p->dv_param_idx.assign(param_idx, param_idx + off_c[n_objs]);
    p->dv_off_cols = off_c; p->dv_off_deriv = off_d;
    p->dv_n_params = n_params;
    p->dv_deriv_h.assign(deriv, deriv + off_d[n_objs]);
    p->dv2_set = false; p->dv2_off.clear();
    HIP_TRY(p->d_dv_deriv.ensure((size_t)std::max<int64_t>(off_d[n_objs], 1)));
    if (off_d[n_objs] > 0) {
        H2D_TRY(p, p->d_dv_deriv.p, deriv, (size_t)off_d[n_objs] * 8);
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    p->derivs_set = true;
    return GST_OK;
    });
}

int gst_set_second_derivs(gst_plan* p, int32_t n_objs, const int32_t* nonzero, const double* hess)
{
    return guarded([&]() -> int {
    if (!p || n_objs < 0) return fail(GST_EINVAL, "bad argument");
    if (n_objs == 0) { p->dv2_set = false; p->dv2_off.clear(); return GST_OK; }
    if (!p->derivs_set || (size_t)n_objs != p->dv_kind.size()) return fail(GST_ESTATE, "gst_set_second_derivs follows gst_set_derivs, object for object");
    if (!nonzero) return fail(GST_EINVAL, "bad argument");
    const int D = p->hp.D;
    if (p->D_user) return fail(GST_EUNSUPPORTED, "second derivative matrices with a padded state dimension");
    std::vector<int64_t> off((size_t)n_objs, -1);
    int64_t total = 0;
    for (int32_t o = 0; o < n_objs; o++) {
        if (!nonzero[o]) continue;
        const int64_t K = p->dv_kind[o] == GST_KIND_GATE ? D * D : D;
        off[(size_t)o] = total;
        total += K * p->dv_ncols[o] * (int64_t)p->dv_ncols[o];
    }
    if (total > 0 && !hess) return fail(GST_EINVAL, "hess is NULL");
    int rc = ensure_device(p);
    if (rc) return rc;
    HIP_TRY(p->d_dv2.ensure((size_t)std::max<int64_t>(total, 1)));
    if (total > 0) {
        H2D_TRY(p, p->d_dv2.p, hess, (size_t)total * 8);
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    p->dv2_off = off;
    p->dv2_set = total > 0;
    return GST_OK;
    });
}


// Host-only: form the tiles of the D = 16 exact contraction for this plan and CHECK them against the pair tables the item
// kernel reads -- for every tiled circuit and every gate, the (forward id, backward id) pairs of its segment slots and of its
// remnant lists together must be exactly the circuit's applications of that gate (build_pair_tables), as a multiset.
int gst_get_tile_stats(gst_plan* p, int64_t* out)
{
    return guarded([&]() -> int {
    if (!p || !out) return fail(GST_EINVAL, "NULL argument");
    for (int k = 0; k < 8; k++) out[k] = 0;
    const gst::HostPlan& h = p->hp;
    if (h.D != 16 || h.n_effects != 4 || h.n_gates > 63) return GST_OK;          // (no tiles for such plans)
    gst::HostPlan R;
    std::string err = gst::build_reverse_plan(h, R, 0, 1);
    if (!err.empty()) return fail(GST_EINVAL, "reversed plan: " + err);
    std::vector<int32_t> pf, pr;
    std::vector<int64_t> pos_ptr;
    gst::build_pair_tables(h, R, pf, pr, pos_ptr);
    std::vector<int32_t> order((size_t)h.n_circuits);
    for (int64_t c = 0; c < h.n_circuits; c++) order[(size_t)c] = (int32_t)c;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return R.circ_leaf[x] < R.circ_leaf[y]; });
    TileSet T;
    build_tiles(h, R, order, T);
    constexpr int TRW = gst::TILE_ROWS, TCL = gst::TILE_COLS, TRC = TRW * TCL;
    const int nG = h.n_gates;
    int64_t bad = 0, max_K = 0;
    // A state is named by its string, not by its id: both plans are cut into tasks, and a tile reads a row's forward states /
    // a column's backward states through ONE member's ids (equal strings, equal values).  Path hashes of both state graphs:
    auto path_hashes = [](const gst::HostPlan& P, std::vector<uint64_t>& out) {
        auto mix = [](uint64_t x, uint64_t v) { x ^= v + 0x9E3779B97F4A7C15ull + (x << 6) + (x >> 2); x *= 0xff51afd7ed558ccdull; return x ^ (x >> 33); };
        out.assign((size_t)P.n_state_ids, 0);
        std::vector<uint8_t> done((size_t)P.n_state_ids, 0);
        std::vector<int32_t> stack;
        for (int64_t id0 = 0; id0 < P.n_state_ids; id0++) {
            int32_t id = (int32_t)id0;
            while (!done[(size_t)id]) {
                const int32_t par = P.node_parent[(size_t)id];
                if (par >= 0 && !done[(size_t)par]) { stack.push_back(id); id = par; continue; }
                out[(size_t)id] = par < 0 ? mix(0xABCDEFull, (uint64_t)P.node_sym[(size_t)id]) : mix(out[(size_t)par], (uint64_t)P.node_sym[(size_t)id] + 1);
                done[(size_t)id] = 1;
                if (stack.empty()) break;
                id = stack.back(); stack.pop_back();
            }
        }
    };
    std::vector<uint64_t> hF, hR;
    path_hashes(h, hF); path_hashes(R, hR);
    std::vector<std::pair<uint64_t, uint64_t>> got, want;
    for (int32_t t = 0; t < T.n_tiles; t++) {
        int64_t K = 0;
        for (int g = 0; g < nG; g++) {
            const int32_t b0 = T.blk[(size_t)t * (nG + 1) + g], b1 = T.blk[(size_t)t * (nG + 1) + g + 1];
            for (int sl = 0; sl < TRC; sl++) {
                const int32_t c = T.cid[(size_t)t * TRC + sl];
                const int i = sl / TCL, m = sl % TCL;
                const int32_t r0 = T.rem_ptr[((size_t)t * nG + g) * (TRC + 1) + sl], r1 = T.rem_ptr[((size_t)t * nG + g) * (TRC + 1) + sl + 1];
                if (c < 0) { if (r1 != r0) bad++; continue; }
                got.clear(); want.clear();
                for (int32_t b = b0; b < b1; b++)
                    for (int q = 0; q < 4; q++) {
                        const int32_t f = T.tsf[((size_t)b * 4 + q) * TRW + i], r = T.tsr[((size_t)b * 4 + q) * TCL + m];
                        if ((f < 0) != (r < 0)) { bad++; continue; }
                        if (f >= 0) got.emplace_back(hF[(size_t)f], hR[(size_t)r]);
                    }
                if (sl == 0) K += (int64_t)got.size();
                for (int32_t b = r0; b < r1; b++)
                    for (int q = 0; q < 4; q++) {
                        const int32_t f = T.rem_f[(size_t)b * 4 + q], r = T.rem_r[(size_t)b * 4 + q];
                        if (r < 0) bad++;
                        if (f >= 0) got.emplace_back(hF[(size_t)f], hR[(size_t)r]);
                    }
                for (int64_t a = pos_ptr[(size_t)c * nG + g]; a < pos_ptr[(size_t)c * nG + g + 1]; a++) want.emplace_back(hF[(size_t)pf[(size_t)a]], hR[(size_t)pr[(size_t)a]]);
                std::sort(got.begin(), got.end()); std::sort(want.begin(), want.end());
                if (got != want) bad++;
            }
        }
        max_K = std::max(max_K, K);
    }
    int64_t n_once = 0;
    {
        std::vector<int32_t> seen((size_t)h.n_circuits, 0);
        for (int32_t c : T.cid) if (c >= 0) seen[(size_t)c]++;
        for (int64_t c = 0; c < h.n_circuits; c++) { if (seen[(size_t)c] > 1 || (seen[(size_t)c] == 1) != (T.tiled[(size_t)c] != 0)) bad++; n_once += seen[(size_t)c] == 1; }
    }
    out[0] = T.n_tiles; out[1] = T.n_tiled; out[2] = T.seg_slots; out[3] = T.rem_slots; out[4] = bad; out[5] = max_K;
    out[6] = h.n_circuits; out[7] = n_once;
    return GST_OK;
    });
}

}  // extern "C"
