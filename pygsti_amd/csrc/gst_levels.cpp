// gst_levels.cpp -- see gst_levels.hpp.  Pure host C++; the programs are interpreted in numpy by tests/test_levels.py.
#include "gst_levels.hpp"

#include <algorithm>
#include <map>

namespace gst {

namespace {

constexpr int MAX_PERIOD = 16;     // longest germ recognised (gates)
constexpr int MIN_CHAIN = 12;      // shortest periodic path given a doubling schedule (gates; at least three periods)

struct Tile { int32_t kind, n_nodes, mref, a, b; };

struct Chain {
    int32_t start;                 // local node the path leaves from (its state is the chain's X_0)
    int32_t m;                     // period
    std::vector<int32_t> nodes;    // local nodes at positions 1 .. T
    int32_t done = -1;             // stage after which every node of the chain is available
};

}  // namespace

std::string build_level_program(const HostPlan& P, int32_t nv, LevelProgram& out, const std::vector<int32_t>* keep_ids)
{
    out = LevelProgram();
    out.D = P.D; out.nv = nv;
    if (P.D != 16) return "level programs exist for D = 16";
    if (nv < 1 || nv > 16 || 16 % nv) return "the vectors per state must divide 16";
    if ((int64_t)P.task_node0.size() != P.n_tasks() || (int64_t)P.task_nodes.size() != P.n_tasks()) return "plan carries no task node ranges";
    if (P.n_gates >= (1 << 20)) return "too many gates";
    const int per_tile = 16 / nv;
    const int64_t nT = P.n_tasks();
    // keep_ids: only these states (and what they are computed from) are produced -- the probability-only program
    std::vector<uint8_t> keep_mark;
    if (keep_ids) {
        keep_mark.assign((size_t)P.n_state_ids, 0);
        for (int32_t id : *keep_ids) if (id >= 0 && id < P.n_state_ids) keep_mark[(size_t)id] = 1;
    }
    std::vector<uint8_t> need;
    out.task_off.assign(1, 0);
    out.task_ids_off.assign(1, 0);
    std::vector<int32_t> par, sym, depth, ready, chain_of, pos_in, run, anc, best_len, best_m;
    std::vector<uint32_t> cont;
    std::vector<uint8_t> covered;
    std::vector<std::vector<int32_t>> runs;      // runs[m][i]
    for (int64_t t = 0; t < nT; t++) {
        const int64_t base = P.task_node0[(size_t)t];
        const int32_t n = P.task_nodes[(size_t)t];
        par.assign((size_t)n, 0); sym.assign((size_t)n, 0); depth.assign((size_t)n, 0);
        int32_t deepest = 0;
        for (int32_t i = 1; i < n; i++) {
            const int32_t gp = P.node_parent[(size_t)(base + i)];
            par[(size_t)i] = gp < 0 ? 0 : (int32_t)(gp - base);
            if (par[(size_t)i] < 0 || par[(size_t)i] >= i) return "state graph is not in creation order";
            sym[(size_t)i] = P.node_sym[(size_t)(base + i)];
            depth[(size_t)i] = depth[(size_t)par[(size_t)i]] + 1;
            deepest = std::max(deepest, depth[(size_t)i]);
        }
        out.sum_task_depth += deepest;
        out.n_nodes += n - 1;
        // ---- periodic runs: run_m[i] = number of consecutive nodes ending at i whose gate equals the one m levels up ----
        runs.assign((size_t)MAX_PERIOD + 1, std::vector<int32_t>());
        anc = par;                                                  // anc = ancestor m levels up (m = 1 first)
        cont.assign((size_t)n, 0);
        for (int m = 1; m <= MAX_PERIOD; m++) {
            if (m > 1) {
                // ancestor m levels up = (ancestor m - 1 levels up of the parent)
                std::vector<int32_t> next((size_t)n, -1);
                for (int32_t i = 1; i < n; i++) next[(size_t)i] = anc[(size_t)par[(size_t)i]] ;
                // (anc currently holds the (m-1)-ancestors; the root's is -1)
                anc.swap(next);
            } else {
                anc[0] = -1;
            }
            std::vector<int32_t>& r = runs[(size_t)m];
            r.assign((size_t)n, 0);
            for (int32_t i = 1; i < n; i++) {
                const int32_t a = anc[(size_t)i];
                if (a > 0 && depth[(size_t)a] >= 2 && depth[(size_t)i] >= 2 && sym[(size_t)a] == sym[(size_t)i]) {
                    r[(size_t)i] = r[(size_t)par[(size_t)i]] + 1;
                    cont[(size_t)par[(size_t)i]] |= 1u << m;       // the parent's run of period m goes on below it
                }
            }
        }
        // ---- candidate chains: ends of maximal runs, longest first; a node belongs to one chain at most ----
        struct Cand { int32_t len, m, end; };
        std::vector<Cand> cands;
        for (int32_t i = 1; i < n; i++) {
            int32_t seen_len = 0;
            for (int m = 1; m <= MAX_PERIOD; m++) {
                const int32_t rm = runs[(size_t)m][(size_t)i];
                if (rm == 0) continue;
                const int32_t len = rm + m;
                if (len <= seen_len) continue;                     // (a multiple of a shorter period: the same path)
                seen_len = len;
                // does the run of THIS period stop here?  (a child continues it iff the child's gate equals the gate m-1 levels above i)
                bool goes_on = false;
                if (cont[(size_t)i] & (1u << m)) {
                    // some child matches period m: the run continues only if that child's run is rm + 1 (always true when it matches)
                    goes_on = true;
                }
                if (!goes_on && len >= MIN_CHAIN && len >= 3 * m) cands.push_back(Cand{len, m, i});
            }
        }
        std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) {
            if (x.len != y.len) return x.len > y.len;
            if (x.m != y.m) return x.m < y.m;
            return x.end < y.end;
        });
        covered.assign((size_t)n, 0);
        chain_of.assign((size_t)n, -1); pos_in.assign((size_t)n, 0);
        std::vector<Chain> chains;
        for (const Cand& c : cands) {
            // the uncovered tail of the run (walking up from its end)
            int32_t T = 0, v = c.end;
            while (T < c.len && !covered[(size_t)v]) { T++; v = par[(size_t)v]; }
            if (T < MIN_CHAIN || T < 3 * c.m) continue;
            Chain ch;
            ch.m = c.m;
            ch.nodes.resize((size_t)T);
            v = c.end;
            for (int32_t k = T - 1; k >= 0; k--) { ch.nodes[(size_t)k] = v; v = par[(size_t)v]; }
            ch.start = v;
            if (ch.start <= 0) continue;                           // (cannot happen: a run consists of gate nodes, depth >= 2)
            const int32_t ci = (int32_t)chains.size();
            for (int32_t k = 0; k < T; k++) { covered[(size_t)ch.nodes[(size_t)k]] = 1; chain_of[(size_t)ch.nodes[(size_t)k]] = ci; pos_in[(size_t)ch.nodes[(size_t)k]] = k + 1; }
            out.chain_nodes += T;
            chains.push_back(std::move(ch));
        }
        out.n_chains += (int64_t)chains.size();
        // ---- which states are produced: all of them, or the kept ones and their sources (children before parents) ----
        need.assign((size_t)n, keep_ids ? 0 : 1);
        if (keep_ids) {
            for (int32_t i = 1; i < n; i++) need[(size_t)i] = keep_mark[(size_t)(base + i)];
            for (int32_t i = n - 1; i >= 1; i--) {
                if (!need[(size_t)i]) continue;
                const int32_t ci = chain_of[(size_t)i];
                if (ci < 0) { need[(size_t)par[(size_t)i]] = 1; continue; }
                const Chain& ch = chains[(size_t)ci];
                const int32_t pos = pos_in[(size_t)i], r = pos / ch.m;
                int32_t src_pos;                                    // the position this node is computed from (see schedule_chain)
                if (pos % ch.m) src_pos = r * ch.m;                 // inside a period: from its period's boundary X_r
                else { int32_t k = 0; while ((2 << k) <= r) k++; src_pos = (r - (1 << k)) * ch.m; }     // X_r from X_{r - 2^k}
                need[(size_t)(src_pos == 0 ? ch.start : ch.nodes[(size_t)src_pos - 1])] = 1;
            }
        }
        // ---- stages ----
        std::vector<std::vector<Tile>> stages;
        auto stage_ref = [&](int32_t s) -> std::vector<Tile>& {
            if ((size_t)s >= stages.size()) stages.resize((size_t)s + 1);
            return stages[(size_t)s];
        };
        auto gid = [&](int32_t local) { return (int32_t)(base + local); };
        auto rows_tile = [&](int32_t stage, int32_t mref, const std::vector<int32_t>& src, const std::vector<int32_t>& dst) {
            for (size_t k0 = 0; k0 < src.size(); k0 += (size_t)per_tile) {
                const int32_t cnt = (int32_t)std::min<size_t>((size_t)per_tile, src.size() - k0);
                const int32_t a = (int32_t)out.ids.size();
                out.ids.insert(out.ids.end(), src.begin() + (long)k0, src.begin() + (long)k0 + cnt);
                const int32_t b = (int32_t)out.ids.size();
                out.ids.insert(out.ids.end(), dst.begin() + (long)k0, dst.begin() + (long)k0 + cnt);
                stage_ref(stage).push_back(Tile{LV_KIND_ROWS, cnt, mref, a, b});
                out.n_produced += cnt;
            }
        };
        int32_t n_slots = 0;
        ready.assign((size_t)n, -1);
        std::map<std::pair<int32_t, int32_t>, std::pair<std::vector<int32_t>, std::vector<int32_t>>> buckets;   // (stage, matrix) -> (src ids, dst ids)
        auto schedule_chain = [&](Chain& ch) {
            const int32_t m = ch.m, T = (int32_t)ch.nodes.size();
            const int32_t q = T / m;
            const int32_t r0 = ready[(size_t)ch.start];
            std::vector<int32_t> g((size_t)m);
            for (int32_t s = 0; s < m; s++) g[(size_t)s] = sym[(size_t)ch.nodes[(size_t)s]];
            auto pos_id = [&](int32_t pos) { return pos == 0 ? gid(ch.start) : gid(ch.nodes[(size_t)pos - 1]); };
            int32_t K = 0;
            while ((1 << K) < q + 1) K++;
            // matrices: pow[k] = (M(g0) ... M(g_{m-1}))^(2^k)
            std::vector<int32_t> pow_ref((size_t)K), t_pow((size_t)K);
            std::vector<int32_t> prefix_ref((size_t)m + 1, LV_BMAT_IDENT);       // prefix_ref[s] = M(g0) ... M(g_{s-1})
            prefix_ref[1] = g[0];
            if (m == 1) { pow_ref[0] = g[0]; t_pow[0] = -1; }
            else {
                int32_t prev = g[0];
                for (int32_t s = 1; s < m; s++) {
                    const int32_t slot = n_slots++;
                    stage_ref(s - 1).push_back(Tile{LV_KIND_MAT, 16, g[(size_t)s], prev, slot});
                    prev = lv_slot_ref(slot);
                    prefix_ref[(size_t)s + 1] = prev;
                }
                pow_ref[0] = prev; t_pow[0] = m - 2;
            }
            for (int32_t k = 1; k < K; k++) {
                const int32_t slot = n_slots++;
                stage_ref(t_pow[(size_t)k - 1] + 1).push_back(Tile{LV_KIND_MAT, 16, pow_ref[(size_t)k - 1], pow_ref[(size_t)k - 1], slot});
                pow_ref[(size_t)k] = lv_slot_ref(slot); t_pow[(size_t)k] = t_pow[(size_t)k - 1] + 1;
            }
            // doubling over the period boundaries X_r = state at position r m
            int32_t st = r0;
            std::vector<int32_t> src, dst;
            for (int32_t k = 0; k < K; k++) {
                const int32_t have = 1 << k, cnt = std::min(have, q + 1 - have);
                if (cnt <= 0) break;
                st = std::max(st, t_pow[(size_t)k]) + 1;
                src.clear(); dst.clear();
                for (int32_t j = 0; j < cnt; j++) {
                    const int32_t dn = ch.nodes[(size_t)((have + j) * m) - 1];
                    ready[(size_t)dn] = st;       // (what hangs off an early node need not wait for the whole chain)
                    if (!need[(size_t)dn]) continue;
                    src.push_back(pos_id(j * m)); dst.push_back(pos_id((have + j) * m));
                }
                rows_tile(st, pow_ref[(size_t)k], src, dst);
            }
            // the positions inside a period: F[r m + s] = X_r x (M(g0) ... M(g_{s-1})), every r and every s in ONE stage (the
            // prefix products are the intermediates of the germ's product chain, ready since stage m - 2 <= st)
            if (m > 1) {
                st = std::max(st, m - 2) + 1;
                for (int32_t s = 1; s < m; s++) {
                    src.clear(); dst.clear();
                    for (int32_t r = 0; r * m + s <= T; r++) {
                        const int32_t dn = ch.nodes[(size_t)(r * m + s) - 1];
                        ready[(size_t)dn] = st;
                        if (!need[(size_t)dn]) continue;
                        src.push_back(pos_id(r * m)); dst.push_back(pos_id(r * m + s));
                    }
                    rows_tile(st, prefix_ref[(size_t)s], src, dst);
                }
            }
            ch.done = st;
        };
        for (int32_t i = 1; i < n; i++) {
            if (depth[(size_t)i] == 1) {                                // a start vector: copied into its node at stage 0
                ready[(size_t)i] = 0;
                if (!need[(size_t)i]) continue;
                auto& b = buckets[{0, LV_BMAT_IDENT}];
                b.first.push_back(-(sym[(size_t)i] + 1)); b.second.push_back(gid(i));
                continue;
            }
            const int32_t ci = chain_of[(size_t)i];
            if (ci >= 0) {
                Chain& ch = chains[(size_t)ci];
                if (ch.done < 0) schedule_chain(ch);      // (sets `ready` of every node of the chain)
                continue;
            }
            const int32_t st = ready[(size_t)par[(size_t)i]] + 1;
            ready[(size_t)i] = st;
            if (!need[(size_t)i]) continue;
            auto& b = buckets[{st, sym[(size_t)i]}];
            b.first.push_back(gid(par[(size_t)i])); b.second.push_back(gid(i));
        }
        for (auto& kv : buckets) rows_tile(kv.first.first, kv.first.second, kv.second.first, kv.second.second);
        // ---- serialise (empty stages dropped: only the order matters) ----
        const size_t w0 = out.words.size();
        out.words.push_back(0);
        int32_t ns = 0;
        for (auto& sv : stages) {
            if (sv.empty()) continue;
            ns++;
            out.words.push_back((int32_t)sv.size());
            for (const Tile& tl : sv) {
                out.words.push_back(tl.kind | (tl.n_nodes << 8));
                out.words.push_back(tl.mref);
                out.words.push_back(tl.a);
                out.words.push_back(tl.b);
            }
            out.n_tiles += (int64_t)sv.size();
        }
        out.words[w0] = ns;
        out.n_stages += ns;
        out.max_stages = std::max(out.max_stages, ns);
        out.max_mats = std::max(out.max_mats, n_slots);
        out.task_off.push_back((int64_t)out.words.size());
        out.task_ids_off.push_back((int64_t)out.ids.size());
        out.max_task_ints = std::max<int64_t>(out.max_task_ints, (int64_t)(out.words.size() - w0) + (out.task_ids_off.back() - out.task_ids_off[out.task_ids_off.size() - 2]));
        if (out.words.size() > 0x7fff0000u || out.ids.size() > 0x7fff0000u) return "level program too large";
    }
    out.worthwhile = out.n_stages * 4 <= out.sum_task_depth;
    return "";
}

}  // namespace gst
