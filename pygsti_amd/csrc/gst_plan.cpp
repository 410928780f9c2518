// gst_plan.cpp -- see gst_plan.hpp.  Pure host C++ (no HIP), unit-testable without a GPU through
// gst_plan_create_* + gst_get_program (tests/test_plan_compiler.py interprets the programs in numpy).
#include "gst_plan.hpp"

#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <set>
#include <queue>

#include "../../include/gstfwd.h"

namespace gst {

std::string expand_table(HostPlan& P, int32_t n_rows, int32_t cache_size, const int32_t* t_dest,
                         const int32_t* t_start, const int32_t* t_cache, const int32_t* t_rho,
                         const int64_t* row_ptr, const int32_t* gate_idx)
{
    // circuit[t_dest[k]] = cached[t_start[k]] + gates(row k); cached[t_cache[k]] = circuit[t_dest[k]]
    // (mapforwardsim_calc_densitymx.pyx:228-277).  Rows are in evaluation order.
    if (n_rows < 0 || cache_size < 0) return "negative n_rows/cache_size";
    std::vector<int64_t> len(n_rows, -1);
    std::vector<int32_t> cache_owner(cache_size, -1);   // expanded-circuit index held in each slot
    P.n_circuits = n_rows;
    P.circ_rho.assign(n_rows, -1);
    // first pass: lengths
    for (int32_t k = 0; k < n_rows; k++) {
        const int32_t i = t_dest[k];
        if (i < 0 || i >= n_rows || len[i] != -1) return "t_dest is not a permutation of the rows";
        if (row_ptr[k + 1] < row_ptr[k]) return "row_ptr not monotone";
        int64_t base = 0;
        if (t_start[k] == -1) {
            if (t_rho[k] < 0 || t_rho[k] >= P.n_rhos) return "t_rho out of range";
            P.circ_rho[i] = t_rho[k];
        } else {
            if (t_start[k] < 0 || t_start[k] >= cache_size || cache_owner[t_start[k]] < 0)
                return "t_start refers to a cache slot that has not been written yet";
            const int32_t src = cache_owner[t_start[k]];
            base = len[src];
            P.circ_rho[i] = P.circ_rho[src];
        }
        len[i] = base + (row_ptr[k + 1] - row_ptr[k]);
        if (t_cache[k] != -1) {
            if (t_cache[k] < 0 || t_cache[k] >= cache_size) return "t_cache out of range";
            cache_owner[t_cache[k]] = i;
        }
    }
    P.circ_ptr.assign(n_rows + 1, 0);
    for (int32_t i = 0; i < n_rows; i++) P.circ_ptr[i + 1] = P.circ_ptr[i] + len[i];
    P.circ_gates.resize(P.circ_ptr[n_rows]);
    std::fill(cache_owner.begin(), cache_owner.end(), -1);
    for (int32_t k = 0; k < n_rows; k++) {
        const int32_t i = t_dest[k];
        int32_t* dst = P.circ_gates.data() + P.circ_ptr[i];
        int64_t base = 0;
        if (t_start[k] != -1) {
            const int32_t src = cache_owner[t_start[k]];
            base = len[src];
            std::memcpy(dst, P.circ_gates.data() + P.circ_ptr[src], sizeof(int32_t) * base);
        }
        const int64_t n = row_ptr[k + 1] - row_ptr[k];
        if (n) std::memcpy(dst + base, gate_idx + row_ptr[k], sizeof(int32_t) * n);
        if (t_cache[k] != -1) cache_owner[t_cache[k]] = i;
    }
    return "";
}

namespace {

struct Node {
    int32_t sym;          // rho index at depth 1, gate index deeper
    int32_t parent = -1;  // local index of the parent node
    int32_t first_child = -1, last_child = -1, next_sibling = -1;
    int32_t n_children = 0;
    int32_t emit_head = -1;   // index into emit list (linked through emit_next)
    int32_t need = 0;         // save slots needed below this node
    int32_t weight = 0;       // nodes in subtree (tie-break)
    int64_t bsum = 0;         // T: sum over branching nodes M of this subtree of (children(M) - 1)
    int64_t bdist = 0;        // U: the same sum weighted by the distance from this node to M
};

struct TaskCompiler {
    const HostPlan& P;
    std::vector<Node> nodes;
    std::vector<int32_t> emit_circ, emit_next;
    std::vector<std::pair<int32_t, int32_t>> leaf_of_circuit;   // (circuit, local node of its final state)
    std::vector<uint32_t>& prog;
    int32_t max_slot_used = 0;

    TaskCompiler(const HostPlan& p, std::vector<uint32_t>& out) : P(p), prog(out) {}

    int32_t sym_at(int32_t c, int64_t pos) const   // pos 0 = rho
    {
        return pos == 0 ? P.circ_rho[c] : P.circ_gates[P.circ_ptr[c] + pos - 1];
    }

    // Build the local trie of sorted circuits order[a..b) (lcp[k] = common symbols of k-1 and k).
    void build(const std::vector<int32_t>& order, const std::vector<int64_t>& lcp, int64_t a, int64_t b)
    {
        nodes.clear(); emit_circ.clear(); emit_next.clear(); leaf_of_circuit.clear();
        nodes.push_back(Node{-1});                 // virtual root (depth 0)
        std::vector<int32_t> path{0};              // path[d] = node at depth d
        for (int64_t k = a; k < b; k++) {
            const int32_t c = order[k];
            const int64_t L = 1 + (P.circ_ptr[c + 1] - P.circ_ptr[c]);
            const int64_t common = (k == a) ? 0 : std::min<int64_t>(lcp[k], L);
            path.resize(common + 1);
            for (int64_t d = common; d < L; d++) {
                Node n{sym_at(c, d)};
                n.parent = path.back();
                const int32_t id = (int32_t)nodes.size();
                Node& par = nodes[path.back()];
                if (par.last_child >= 0) nodes[par.last_child].next_sibling = id; else par.first_child = id;
                par.last_child = id;
                par.n_children++;
                nodes.push_back(n);
                path.push_back(id);
            }
            leaf_of_circuit.push_back({c, path.back()});
            const int32_t e = (int32_t)emit_circ.size();
            emit_circ.push_back(c);
            emit_next.push_back(-1);
            // append at tail of the node's emit list (keeps circuit order deterministic)
            Node& end = nodes[path.back()];
            if (end.emit_head < 0) end.emit_head = e;
            else { int32_t t = end.emit_head; while (emit_next[t] >= 0) t = emit_next[t]; emit_next[t] = e; }
        }
        // children are created after their parents: reverse index order is a post-order
        for (int32_t i = (int32_t)nodes.size() - 1; i >= 0; i--) {
            Node& n = nodes[i];
            n.weight = 1;
            n.bsum = n.n_children > 1 ? n.n_children - 1 : 0;
            n.bdist = 0;
            int32_t best = -1, second = -1;   // largest and second-largest child need
            for (int32_t ch = n.first_child; ch >= 0; ch = nodes[ch].next_sibling) {
                n.weight += nodes[ch].weight;
                n.bsum += nodes[ch].bsum;
                n.bdist += nodes[ch].bdist + nodes[ch].bsum;
                const int32_t nd = nodes[ch].need;
                if (nd > best) { second = best; best = nd; } else if (nd > second) second = nd;
            }
            if (n.n_children <= 1) n.need = std::max(best, 0);
            else n.need = std::max(best, second + 1);   // max-need child goes last and reuses the slot
        }
    }

    int64_t node_base = 0;              // global id of this task's local node 0 (base-state cache index)

    // the op that creates `node`'s state, followed by its NODE marker (global state id)
    void op(int32_t node, int depth)
    {
        prog.push_back(((depth == 1 ? GST_OP_RHO : GST_OP_APPLY) << 28) | (uint32_t)nodes[node].sym);
        prog.push_back((GST_OP_NODE << 28) | (uint32_t)(node_base + node));
    }
    void replay_add_last_op() { replay.push_back(prog[prog.size() - 2]); replay.push_back(prog.back()); }
    void replay_set_last_op() { replay.clear(); replay_add_last_op(); }

    int32_t slot_limit = 1 << 20;       // save slots the device offers (gst_options.max_slots)
    std::vector<uint32_t> replay;       // program words that re-create the CURRENT node's state from its anchor
                                        // (a LOAD of a live slot, or a RHO) -- used where no slot is free

    // Emit the program for the subtree below `node` (whose state is in v); slots >= `base` are free.
    // A branching node keeps its state in slot `base` while its children are walked; when the slot
    // budget is exhausted the state is instead re-derived for each child by replaying the (short)
    // path from the nearest saved ancestor -- same gates in the same order, so bit-identical.
    void walk(int32_t node, int depth, int32_t base)
    {
        for (;;) {
            for (int32_t e = nodes[node].emit_head; e >= 0; e = emit_next[e])
                prog.push_back((GST_OP_EMIT << 28) | (uint32_t)emit_circ[e]);
            const Node& n = nodes[node];
            if (n.n_children == 0) return;
            if (n.n_children == 1) {             // chain: iterate, do not recurse
                node = n.first_child; depth++;
                op(node, depth);
                if (depth == 1) replay.clear();
                replay_add_last_op();
                continue;
            }
            // pick the child that goes last (it inherits slot `base`): the heaviest subtree, because with a
            // bounded slot budget everything walked BEFORE it has one slot less and may have to replay
            int32_t last = -1;
            for (int32_t ch = n.first_child; ch >= 0; ch = nodes[ch].next_sibling)
                if (last < 0 || nodes[ch].weight > nodes[last].weight ||
                    (nodes[ch].weight == nodes[last].weight && nodes[ch].need > nodes[last].need))
                    last = ch;
            const std::vector<uint32_t> here = replay;       // re-creates THIS node's state
            if (depth == 0) {                    // children of the root start from a rho: nothing to keep
                for (int32_t ch = n.first_child; ch >= 0; ch = nodes[ch].next_sibling) {
                    if (ch == last) continue;
                    op(ch, 1);
                    replay_set_last_op();
                    walk(ch, 1, base);
                }
                node = last; depth = 1;
                op(node, depth);
                replay_set_last_op();
                continue;
            }
            bool have_slot = base < slot_limit;
            if (have_slot && base == slot_limit - 1) {
                // About to take the LAST free slot.  Holding it here denies it to everything walked before
                // the last child: each branching node M in those subtrees then replays dist(this, M) gates per
                // extra child.  If re-deriving THIS node's state per child is cheaper than that, replay here
                // and leave the slot to the descendants (typical: a shallow fork above several long chains).
                const int64_t replay_cost = (int64_t)(n.n_children - 1) * (int64_t)(here.size() / 2 + 1);
                int64_t deny_cost = 0;
                for (int32_t ch = n.first_child; ch >= 0; ch = nodes[ch].next_sibling)
                    if (ch != last) deny_cost += nodes[ch].bdist + nodes[ch].bsum;
                if (replay_cost < deny_cost) have_slot = false;
            }
            if (have_slot) {
                prog.push_back((GST_OP_SAVE << 28) | (uint32_t)base);
                max_slot_used = std::max(max_slot_used, base + 1);
            }
            bool first = true;
            for (int32_t ch = n.first_child; ch >= 0; ch = nodes[ch].next_sibling) {
                if (ch == last) continue;
                if (!first) {
                    if (have_slot) prog.push_back((GST_OP_LOAD << 28) | (uint32_t)base);
                    else prog.insert(prog.end(), here.begin(), here.end());
                }
                first = false;
                op(ch, depth + 1);
                if (have_slot) replay.assign(1, (GST_OP_LOAD << 28) | (uint32_t)base);
                else replay = here;
                replay_add_last_op();
                walk(ch, depth + 1, have_slot ? base + 1 : base);
            }
            if (have_slot) prog.push_back((GST_OP_LOAD << 28) | (uint32_t)base);
            else prog.insert(prog.end(), here.begin(), here.end());
            node = last; depth++;
            op(node, depth);                     // continue into the last child; slot `base` is free again,
            replay = here;                       // so its anchor is this node's own anchor
            replay_add_last_op();
        }
    }
};

}  // namespace

std::string compile_plan(HostPlan& P, int32_t target_tasks, int32_t max_slots)
{
    const int64_t nC = P.n_circuits;
    if (P.D <= 0 || P.n_gates < 0 || P.n_rhos <= 0 || P.n_effects <= 0) return "bad dimensions";
    if ((int64_t)P.circ_rho.size() != nC || (int64_t)P.circ_ptr.size() != nC + 1) return "circuit arrays have wrong size";
    if ((int64_t)P.eff_ptr.size() != nC + 1) return "effect CSR has wrong size";
    if (P.n_gates >= (1 << 28) || nC >= (1 << 28)) return "plan too large for the 28-bit program argument";
    for (int64_t i = 0; i < nC; i++) {
        if (P.circ_rho[i] < 0 || P.circ_rho[i] >= P.n_rhos) return "circ_rho out of range";
        if (P.circ_ptr[i + 1] < P.circ_ptr[i]) return "circ_ptr not monotone";
        if (P.eff_ptr[i + 1] < P.eff_ptr[i]) return "eff_ptr not monotone";
    }
    if ((int64_t)P.circ_gates.size() != P.circ_ptr[nC]) return "circ_gates has wrong size";
    for (int32_t g : P.circ_gates) if (g < 0 || g >= P.n_gates) return "gate index out of range";
    if ((int64_t)P.eff_label.size() != P.eff_ptr[nC] || P.eff_dest.size() != P.eff_label.size())
        return "effect arrays have wrong size";
    std::vector<uint8_t> seen(P.n_elements, 0);
    for (size_t x = 0; x < P.eff_label.size(); x++) {
        if (P.eff_label[x] < 0 || P.eff_label[x] >= P.n_effects) return "effect label out of range";
        const int32_t d = P.eff_dest[x];
        if (d < 0 || d >= P.n_elements) return "element index out of range";
        if (seen[d]) return "element index assigned twice";
        seen[d] = 1;
    }

    const bool tm_ = std::getenv("GST_PLAN_TIMING") != nullptr;      // development aid: seconds per stage on stderr
    auto t0_ = std::chrono::steady_clock::now();
    auto lap_ = [&](const char* w) { if (tm_) { auto t1 = std::chrono::steady_clock::now(); std::fprintf(stderr, "[plan] %s %.3f s\n", w, std::chrono::duration<double>(t1 - t0_).count()); t0_ = t1; } };
    lap_("validate");
    // --- sort circuits lexicographically on (rho, gates...) --------------------------------------
    std::vector<int32_t> order(nC);
    std::iota(order.begin(), order.end(), 0);
    const int32_t* G = P.circ_gates.data();
    const int64_t* ptr = P.circ_ptr.data();
    auto less = [&](int32_t a, int32_t b) {
        if (P.circ_rho[a] != P.circ_rho[b]) return P.circ_rho[a] < P.circ_rho[b];
        const int64_t la = ptr[a + 1] - ptr[a], lb = ptr[b + 1] - ptr[b];
        const int32_t *pa = G + ptr[a], *pb = G + ptr[b];
        const int64_t n = std::min(la, lb);
        for (int64_t i = 0; i < n; i++) if (pa[i] != pb[i]) return pa[i] < pb[i];
        if (la != lb) return la < lb;
        return a < b;
    };
    std::sort(order.begin(), order.end(), less);
    lap_("sort");
    std::vector<int64_t> lcp(nC, 0);   // common symbols (rho counts as one) with the previous circuit
    P.sum_depth = 0; P.trie_nodes = 0; P.max_depth = 0;
    for (int64_t k = 0; k < nC; k++) {
        const int32_t c = order[k];
        const int64_t L = ptr[c + 1] - ptr[c];
        P.sum_depth += L;
        P.max_depth = std::max<int32_t>(P.max_depth, (int32_t)L);
        if (k > 0) {
            const int32_t p = order[k - 1];
            if (P.circ_rho[p] == P.circ_rho[c]) {
                const int64_t Lp = ptr[p + 1] - ptr[p], n = std::min(L, Lp);
                int64_t i = 0;
                while (i < n && G[ptr[p] + i] == G[ptr[c] + i]) i++;
                lcp[k] = 1 + i;
            }
        }
        P.trie_nodes += (1 + L) - lcp[k];
    }

    // --- cut the sorted list into tasks ------------------------------------------------------------
    // A task restarts from rho, so cutting before circuit k re-computes lcp[k]-1 gate applications.
    // Cut only where that is small against the task being closed (<= 5% of its size; the task keeps growing
    // past the grain until such a place comes, so a small atom of deep germ-power families is still cut between
    // families instead of collapsing into a few huge tasks).
    if (target_tasks <= 0) target_tasks = 2048;
    int64_t grain = std::max<int64_t>(32, P.trie_nodes / target_tasks);
    // Small plans (a 1Q design is ~4,000 states for 1,024 SIMDs) are pure latency: the time of a pass is the longest
    // task's dependent chain, not the work.  There a cut is allowed wherever a quarter of the deepest circuit's worth
    // of new states has accumulated, whatever it re-computes -- many short walks instead of a few long ones.
    int64_t deepest = 0;
    for (int64_t k = 0; k < nC; k++) deepest = std::max<int64_t>(deepest, 1 + (ptr[k + 1] - ptr[k]));
    const bool small = P.trie_nodes <= 65536;
    if (small) grain = std::max<int64_t>(16, deepest / 4);
    std::vector<int64_t> cuts{0};
    int64_t cur = 0;
    for (int64_t k = 0; k < nC; k++) {
        const int32_t c = order[k];
        const int64_t L = 1 + (ptr[c + 1] - ptr[c]);
        if (k > 0 && cur >= grain && (small || lcp[k] - 1 <= std::max<int64_t>(2, cur / 20))) { cuts.push_back(k); cur = 0; }
        cur += (cur == 0) ? L : L - lcp[k];
    }
    cuts.push_back(nC);

    lap_("lcp+cuts");
    // --- compile each task ----------------------------------------------------------------------------
    struct Built { std::vector<uint32_t> words; int64_t applies; int64_t node0; int32_t nodes; };
    std::vector<Built> built;
    built.reserve(cuts.size());
    P.max_slots = 0;
    P.applies_per_pass = 0;
    P.n_state_ids = 0;
    P.node_parent.clear(); P.node_sym.clear();
    P.circ_leaf.assign(nC, -1);
    for (size_t t = 0; t + 1 < cuts.size(); t++) {
        if (cuts[t] == cuts[t + 1]) continue;
        Built b;
        TaskCompiler tc(P, b.words);
        if (max_slots > 0) tc.slot_limit = max_slots;
        tc.node_base = P.n_state_ids;
        tc.build(order, lcp, cuts[t], cuts[t + 1]);
        b.node0 = tc.node_base; b.nodes = (int32_t)tc.nodes.size();
        P.n_state_ids += (int64_t)tc.nodes.size();
        // state-id graph for the analytic (backward) sweeps: parent id and symbol of every state
        P.node_parent.resize(P.n_state_ids, -1);
        P.node_sym.resize(P.n_state_ids, 0);
        for (size_t i = 1; i < tc.nodes.size(); i++) {
            const int32_t par = tc.nodes[i].parent;
            P.node_parent[tc.node_base + i] = par <= 0 ? -1 : (int32_t)(tc.node_base + par);   // -1: a rho state
            P.node_sym[tc.node_base + i] = tc.nodes[i].sym;
        }
        for (auto& lc : tc.leaf_of_circuit) P.circ_leaf[lc.first] = (int32_t)(tc.node_base + lc.second);
        tc.walk(0, 0, 0);
        b.words.push_back(GST_OP_END << 28);
        b.applies = 0;
        for (uint32_t w : b.words) b.applies += (GST_OP(w) == GST_OP_APPLY);
        P.max_slots = std::max(P.max_slots, tc.max_slot_used);
        P.applies_per_pass += b.applies;
        built.push_back(std::move(b));
    }
    lap_("tasks");
    // heaviest first: the device takes tasks in launch order, long ones should not start last
    std::stable_sort(built.begin(), built.end(), [](const Built& a, const Built& b) { return a.applies > b.applies; });
    P.prog.clear(); P.task_off.assign(1, 0); P.task_applies.clear();
    P.task_node0.clear(); P.task_nodes.clear();
    for (auto& b : built) {
        P.prog.insert(P.prog.end(), b.words.begin(), b.words.end());
        P.task_off.push_back((int64_t)P.prog.size());
        P.task_applies.push_back(b.applies);
        P.task_node0.push_back(b.node0); P.task_nodes.push_back(b.nodes);
    }
    return "";
}

void task_gate_costs(const HostPlan& P, std::vector<int32_t>& cost)
{
    cost.clear();
    const int nG = P.n_gates;
    if (nG > 64) return;
    const int64_t nT = P.n_tasks();
    const int stride = nG + 2;
    cost.assign((size_t)nT * stride, 0);
    std::vector<uint64_t> slot(64, 0);
    for (int64_t t = 0; t < nT; t++) {
        int32_t* c = cost.data() + (size_t)t * stride;
        uint64_t dirty = 0;                       // bit g: the walk perturbing gate g holds a perturbed state here
        for (int64_t k = P.task_off[t]; k < P.task_off[t + 1]; k++) {
            const uint32_t w = P.prog[k], op = GST_OP(w), arg = GST_ARG(w);
            if (op == GST_OP_APPLY) {
                dirty |= 1ull << arg;
                for (uint64_t m = dirty; m; m &= m - 1) c[__builtin_ctzll(m)]++;
                c[nG]++;
            } else if (op == GST_OP_SAVE) { if (arg < 64) slot[arg] = dirty; }
            else if (op == GST_OP_LOAD) { if (arg < 64) dirty = slot[arg]; }
            else if (op == GST_OP_RHO) dirty = 0;
            else if (op == GST_OP_EMIT) c[nG + 1]++;
        }
    }
}

void fd_executed_work(const HostPlan& P, const std::vector<uint64_t>& wave_gates, const std::vector<uint8_t>& wave_rho,
                      const std::vector<uint8_t>& wave_eff, const std::vector<int32_t>& wave_cols, int64_t out[6])
{
    for (int i = 0; i < 6; i++) out[i] = 0;
    const int64_t nT = P.n_tasks();
    const size_t nW = wave_gates.size();
    std::vector<uint64_t> slot(64, 0);
    for (size_t w0 = 0; w0 < nW; w0 += 64) {                  // 64 wavefronts at a time: one bit each
        const size_t nb = std::min<size_t>(64, nW - w0);
        std::vector<uint64_t> gate_waves((size_t)std::max(P.n_gates, 1), 0);
        uint64_t rho_waves = 0, eff_waves = 0;
        int64_t cols_all = 0;
        for (size_t b = 0; b < nb; b++) {
            for (uint64_t m = wave_gates[w0 + b]; m; m &= m - 1) gate_waves[(size_t)__builtin_ctzll(m)] |= 1ull << b;
            if (wave_rho[w0 + b]) rho_waves |= 1ull << b;
            if (wave_eff[w0 + b]) eff_waves |= 1ull << b;
            cols_all += wave_cols[w0 + b];
        }
        auto cols_of = [&](uint64_t mask) { int64_t c = 0; for (uint64_t m = mask; m; m &= m - 1) c += wave_cols[w0 + (size_t)__builtin_ctzll(m)]; return c; };
        for (int64_t t = 0; t < nT; t++) {
            uint64_t dirty = 0;
            uint64_t last_mask = ~0ull; int64_t last_cols = 0;          // (the mask changes rarely along a chain)
            for (int64_t k = P.task_off[t]; k < P.task_off[t + 1]; k++) {
                const uint32_t w = P.prog[k], op = GST_OP(w), arg = GST_ARG(w);
                if (op == GST_OP_APPLY) {
                    dirty |= gate_waves[arg];
                    if (dirty != last_mask) { last_mask = dirty; last_cols = cols_of(dirty); }
                    out[0] += __builtin_popcountll(dirty); out[1] += last_cols;
                    out[4] += (int64_t)nb; out[5] += cols_all;
                } else if (op == GST_OP_SAVE) { if (arg < 64) slot[arg] = dirty; }
                else if (op == GST_OP_LOAD) { if (arg < 64) dirty = slot[arg]; }
                else if (op == GST_OP_RHO) dirty = rho_waves;
                else if (op == GST_OP_EMIT) {
                    const int64_t n_out = P.eff_ptr[arg + 1] - P.eff_ptr[arg];
                    const uint64_t real = dirty | eff_waves;
                    out[2] += n_out * __builtin_popcountll(real); out[3] += n_out * cols_of(real);
                }
            }
        }
    }
}

void build_dirty_programs(const HostPlan& P, DirtyPrograms& out)
{
    const int nG = P.n_gates, nR = P.n_rhos, nC = nG + nR;
    const int64_t nT = P.n_tasks();
    out.n_classes = nC;
    out.words.clear();
    out.off.assign((size_t)nT * nC + 1, 0);
    out.applies.assign((size_t)nT * nC, 0);
    out.emits.assign((size_t)nT * nC, 0);
    std::vector<int32_t> tag;
    for (int64_t t = 0; t < nT; t++) {
        for (int c = 0; c < nC; c++) {
            const bool is_rho = c >= nG;
            const uint32_t obj = (uint32_t)(is_rho ? c - nG : c);
            const size_t start = out.words.size();
            bool dirty = false;
            int32_t cur_id = 0;
            int32_t n_app = 0, n_emit = 0;
            tag.assign(64, -1);
            for (int64_t k = P.task_off[t]; k < P.task_off[t + 1]; k++) {
                const uint32_t w = P.prog[k], op = GST_OP(w), arg = GST_ARG(w);
                if (op == GST_OP_RHO) {
                    if (is_rho && arg == obj) { out.words.push_back(w); dirty = true; }
                    else dirty = false;
                } else if (op == GST_OP_NODE) {
                    cur_id = (int32_t)arg;
                } else if (op == GST_OP_APPLY) {
                    if (!dirty && !is_rho && arg == obj) {
                        out.words.push_back((GST_OP_CACHE << 28) | (uint32_t)cur_id);      // start from the base pass's state
                        dirty = true;
                    }
                    if (dirty) { out.words.push_back(w); n_app++; }
                } else if (op == GST_OP_SAVE) {
                    if (arg < 64) {
                        if (dirty) { out.words.push_back(w); tag[arg] = -1; }
                        else tag[arg] = cur_id;
                    }
                } else if (op == GST_OP_LOAD) {
                    if (arg < 64) {
                        if (tag[arg] >= 0) { dirty = false; cur_id = tag[arg]; }
                        else { out.words.push_back(w); dirty = true; }
                    }
                } else if (op == GST_OP_EMIT) {
                    if (dirty) { out.words.push_back(w); n_emit++; }
                }
            }
            if (n_emit == 0) out.words.resize(start);      // nothing observable changes: no program at all
            else out.words.push_back(GST_OP_END << 28);
            out.off[(size_t)t * nC + c + 1] = (int64_t)out.words.size();
            out.applies[(size_t)t * nC + c] = n_emit ? n_app : 0;
            out.emits[(size_t)t * nC + c] = n_emit;
        }
    }
}

void task_split_candidates(const HostPlan& P, std::vector<int32_t>& cand_ptr, std::vector<int32_t>& cand_pc, std::vector<float>& cand_frac,
                           int max_per_task, std::vector<uint32_t>* cand_live)
{
    const int64_t nT = P.n_tasks();
    cand_ptr.assign((size_t)nT + 1, 0);
    cand_pc.clear(); cand_frac.clear();
    if (cand_live) cand_live->clear();
    std::vector<uint64_t> live;
    std::vector<std::pair<int32_t, int64_t>> all;      // (pc, gate applications before it)
    for (int64_t t = 0; t < nT; t++) {
        cand_ptr[(size_t)t + 1] = (int32_t)cand_pc.size();
        const int64_t k0 = P.task_off[t], k1 = P.task_off[t + 1];
        const int64_t n = k1 - k0;
        if (n < 64) continue;
        // backward liveness of the save slots: live[k] = slots that hold data some later LOAD still needs, before word k
        live.assign((size_t)n + 1, 0);
        bool ok = true;
        for (int64_t k = n - 1; k >= 0; k--) {
            const uint32_t w = P.prog[k0 + k], op = GST_OP(w), arg = GST_ARG(w);
            uint64_t l = live[(size_t)k + 1];
            if (op == GST_OP_LOAD || op == GST_OP_SAVE) {
                if (arg >= 64) { ok = false; break; }
                if (op == GST_OP_LOAD) l |= 1ull << arg; else l &= ~(1ull << arg);
            }
            live[(size_t)k] = l;
        }
        if (!ok) continue;
        const int64_t total = P.task_applies[(size_t)t];
        if (total < 32) continue;
        all.clear();
        int64_t before = 0;
        for (int64_t k = 0; k < n; k++) {
            const uint32_t w = P.prog[k0 + k], op = GST_OP(w);
            const bool after_load = k > 0 && GST_OP(P.prog[k0 + k - 1]) == GST_OP_LOAD;
            // a position where the 64 lane states are the walk's entire state: no save slot live, and the interpreter is
            // at an instruction boundary it can resume from (an EMIT, or the word after a LOAD)
            // (cand_live given: the hand-over carries the live save slots along, so liveness does not restrict the cut)
            //  and a cut may fall INSIDE a chain, in front of any APPLY: a germ-power family's last trunk segment is half of
            //  its work with no EMIT or LOAD inside)
            if (k > 2 && (live[(size_t)k] == 0 || (cand_live && live[(size_t)k] < 16)) &&
                (op == GST_OP_EMIT || after_load || (cand_live && op == GST_OP_APPLY)) && op != GST_OP_END &&
                10 * before >= total && 10 * before <= 9 * total && (all.empty() || all.back().second != before))
                all.emplace_back((int32_t)k, before);
            if (op == GST_OP_APPLY) before++;
        }
        // thin out to at most max_per_task, evenly spaced in work
        const size_t m = all.size();
        const size_t keep = std::min<size_t>(m, (size_t)std::max(max_per_task, 1));
        for (size_t i = 0; i < keep; i++) {
            const auto& c = all[keep == m ? i : (size_t)(((double)i + 0.5) * (double)m / (double)keep)];
            if (!cand_pc.empty() && (int32_t)cand_pc.size() > cand_ptr[(size_t)t] && cand_pc.back() == c.first) continue;
            cand_pc.push_back(c.first);
            cand_frac.push_back((float)c.second / (float)total);
            if (cand_live) cand_live->push_back((uint32_t)live[(size_t)c.first]);
        }
        cand_ptr[(size_t)t + 1] = (int32_t)cand_pc.size();
    }
}

void task_split_points(const HostPlan& P, std::vector<int32_t>& split_pc, std::vector<float>& split_frac)
{
    const int64_t nT = P.n_tasks();
    split_pc.assign((size_t)nT, -1);
    split_frac.assign((size_t)nT, 0.0f);
    std::vector<int32_t> ptr, pc;
    std::vector<float> fr;
    task_split_candidates(P, ptr, pc, fr, 1 << 20, nullptr);
    for (int64_t t = 0; t < nT; t++) {
        int32_t best = -1;
        for (int32_t i = ptr[(size_t)t]; i < ptr[(size_t)t + 1]; i++)
            if (best < 0 || std::fabs(fr[(size_t)i] - 0.5f) < std::fabs(fr[(size_t)best] - 0.5f)) best = i;
        if (best < 0 || fr[(size_t)best] < 0.3f || fr[(size_t)best] > 0.7f) continue;     // (no useful middle)
        split_pc[(size_t)t] = pc[(size_t)best];
        split_frac[(size_t)t] = fr[(size_t)best];
    }
}

void pack_fd_queues(const std::vector<std::pair<int32_t, uint32_t>>& items, int32_t n_units, int64_t nT, int n_bins, int handover,
                    const std::vector<int32_t>& cand_ptr, const std::vector<int32_t>& cand_pc, const std::vector<float>& cand_frac,
                    const std::vector<uint32_t>& cand_live, FdQueues& out)
{
    struct BinItem { uint32_t id; int64_t cost; int32_t part; };
    std::vector<std::vector<BinItem>> bins((size_t)n_bins);
    std::vector<int64_t>& load = out.load;
    load.assign((size_t)n_bins, 0);
    typedef std::pair<int64_t, int32_t> LB;                  // (load, queue)
    {
        std::priority_queue<LB, std::vector<LB>, std::greater<LB>> heap;
        for (int b = 0; b < n_bins; b++) heap.emplace(0, b);
        for (const auto& it : items) {
            LB t = heap.top(); heap.pop();
            const int64_t c = (int64_t)(-it.first) + 8;
            bins[(size_t)t.second].push_back(BinItem{it.second, c, 0});
            load[(size_t)t.second] = t.first + c;
            heap.emplace(t.first + c, t.second);
        }
    }
    out.n_split = 0;
    out.ho_index.clear(); out.ho_pc.clear(); out.ho_live.clear();
    const bool have_cands = cand_ptr.size() == (size_t)nT + 1;
    if (handover != 0 && have_cands) {
        out.ho_index.assign((size_t)nT * (size_t)n_units, -1);
        int64_t total = 0;
        for (int64_t l : load) total += l;
        const int64_t mean = total / n_bins;
        const bool force = handover == 2;
        // the candidate of task t closest to fraction `want`
        auto pick = [&](int64_t t, double want) -> int32_t {
            int32_t best = -1;
            for (int32_t i = cand_ptr[(size_t)t]; i < cand_ptr[(size_t)t + 1]; i++)
                if (best < 0 || std::fabs(cand_frac[(size_t)i] - want) < std::fabs(cand_frac[(size_t)best] - want)) best = i;
            return best;
        };
        const int64_t tol = std::max<int64_t>(16, mean / 200);
        // where a walk is cut: 0 = in its middle (the second part is then long enough to be worth a wavefront's wait and
        // short enough to end with its queue); 1 = where the donor's load becomes the mean; 2 = where donor and receiver
        // become equal.  (development switch; the default is what measured best on a 1/8 atom of the 2Q design)
        const int cut_policy = 0;
        const double cut_frac = 0.5;
        const int64_t min_gain = 16;
        // donors: queues above the mean, fullest first; receivers: all queues, emptiest first.  A donor that has nothing
        // (more) to give leaves the donor set only -- it may still receive.
        std::set<LB> donors, recv;
        for (int b = 0; b < n_bins; b++) { donors.emplace(load[(size_t)b], b); recv.emplace(load[(size_t)b], b); }
        for (int iter = 0; iter < 16 * n_bins && !donors.empty(); iter++) {
            const LB top = *donors.rbegin();
            const int bmax = top.second;
            if (!force && load[(size_t)bmax] - mean <= tol) break;      // the fullest queue is at the mean: done
            int best = -1;
            for (size_t k = 0; k < bins[(size_t)bmax].size(); k++) {
                const BinItem& bi = bins[(size_t)bmax][k];
                const int64_t t = (int64_t)(bi.id / (uint32_t)n_units);
                if (bi.part != 0 || cand_ptr[(size_t)t + 1] == cand_ptr[(size_t)t]) continue;
                if (!force && bi.cost < std::max<int64_t>(64, mean / 4)) continue;
                if (best < 0 || bi.cost > bins[(size_t)bmax][(size_t)best].cost) best = (int)k;
            }
            if (best < 0) { donors.erase(top); continue; }  // nothing to cut in the fullest queue: look at the next
            LB low = *recv.begin();
            if (low.second == bmax) { if (recv.size() < 2) break; low = *std::next(recv.begin()); }
            const int bmin = low.second;
            BinItem whole = bins[(size_t)bmax][(size_t)best];
            const int64_t tsk = (int64_t)(whole.id / (uint32_t)n_units);
            // shed x: down to the mean, but never so much that the receiver ends above the donor
            double want = force ? 0.5 : cut_frac;
            if (!force && cut_policy != 0) {
                const double half = 0.5 * (double)(load[(size_t)bmax] - load[(size_t)bmin]);
                const double x = cut_policy == 2 ? half : std::min((double)(load[(size_t)bmax] - mean), half);
                want = 1.0 - x / (double)whole.cost;
            }
            const int32_t ci = pick(tsk, want);
            const double frac = cand_frac[(size_t)ci];
            const int64_t c1 = (int64_t)((double)whole.cost * frac) + 24, c2 = whole.cost - (int64_t)((double)whole.cost * frac) + 24;
            const int64_t new_hi = std::max(load[(size_t)bmax] - whole.cost + c1, load[(size_t)bmin] + c2);
            if (!force && new_hi + min_gain >= load[(size_t)bmax]) { donors.erase(top); continue; }   // no gain here: next queue
            donors.erase(top); donors.erase(LB(load[(size_t)bmin], bmin));
            recv.erase(LB(load[(size_t)bmax], bmax)); recv.erase(low);
            bins[(size_t)bmax].erase(bins[(size_t)bmax].begin() + best);
            bins[(size_t)bmax].insert(bins[(size_t)bmax].begin(), BinItem{whole.id, c1, 1});
            {   // second part: in cost order behind the receiver's longer items (not behind its short ones: picked up
                // late, its work would land at the very end of that queue)
                auto& rb = bins[(size_t)bmin];
                size_t at = 0;
                while (at < rb.size() && (rb[at].part == 1 || rb[at].cost >= c2)) at++;
                rb.insert(rb.begin() + (long)at, BinItem{whole.id, c2, 2});
            }
            load[(size_t)bmax] += c1 - whole.cost;
            load[(size_t)bmin] += c2;
            out.ho_index[(size_t)whole.id] = out.n_split++;
            out.ho_pc.push_back(cand_pc[(size_t)ci]);
            out.ho_live.push_back(cand_live.size() == cand_pc.size() ? cand_live[(size_t)ci] : 0u);
            donors.emplace(load[(size_t)bmin], bmin); donors.emplace(load[(size_t)bmax], bmax);
            recv.emplace(load[(size_t)bmin], bmin); recv.emplace(load[(size_t)bmax], bmax);
        }
    }
    out.bin_ptr.assign((size_t)n_bins + 1, 0);
    out.bin_items.clear();
    out.bin_items.reserve(items.size() + (size_t)out.n_split);
    for (int b = 0; b < n_bins; b++) {
        for (const BinItem& bi : bins[(size_t)b]) out.bin_items.push_back(bi.id | ((uint32_t)bi.part << 30));
        out.bin_ptr[(size_t)b + 1] = (int32_t)out.bin_items.size();
    }
    if (out.n_split == 0) out.ho_index.clear();
}

std::string build_reverse_plan(const HostPlan& P, HostPlan& R, int32_t target_tasks, int32_t max_slots)
{
    R = HostPlan();
    R.D = P.D; R.n_gates = P.n_gates; R.n_rhos = 1; R.n_effects = 1;     // (one nominal effect; no circuit emits)
    R.n_circuits = P.n_circuits; R.n_elements = 0;
    R.circ_rho.assign((size_t)P.n_circuits, 0);
    R.circ_ptr = P.circ_ptr;
    R.circ_gates.resize(P.circ_gates.size());
    for (int64_t c = 0; c < P.n_circuits; c++) {
        const int64_t a = P.circ_ptr[c], b = P.circ_ptr[c + 1];
        for (int64_t k = a; k < b; k++) R.circ_gates[k] = P.circ_gates[a + (b - 1 - k)];
    }
    R.eff_ptr.assign((size_t)P.n_circuits + 1, 0);
    return compile_plan(R, target_tasks, max_slots);
}

void build_pair_tables(const HostPlan& P, const HostPlan& R, std::vector<int32_t>& pf, std::vector<int32_t>& pr,
                       std::vector<int64_t>& pos_ptr)
{
    const int nG = P.n_gates;
    const int64_t nC = P.n_circuits;
    const int64_t total = P.circ_ptr[nC];
    pf.assign((size_t)total, 0);
    pr.assign((size_t)total, 0);
    pos_ptr.assign((size_t)nC * nG + 1, 0);
    // counts per (circuit, gate) -> offsets
    for (int64_t c = 0; c < nC; c++)
        for (int64_t k = P.circ_ptr[c]; k < P.circ_ptr[c + 1]; k++) pos_ptr[(size_t)c * nG + P.circ_gates[k] + 1]++;
    for (size_t i = 1; i < pos_ptr.size(); i++) pos_ptr[i] += pos_ptr[i - 1];
    std::vector<int32_t> f, r;
    std::vector<int64_t> cursor(nG);
    for (int64_t c = 0; c < nC; c++) {
        const int64_t a = P.circ_ptr[c];
        const int64_t n = P.circ_ptr[c + 1] - a;
        f.resize((size_t)n + 1); r.resize((size_t)n + 1);
        int32_t id = P.circ_leaf[c];
        for (int64_t k = n; k >= 0; k--) { f[(size_t)k] = id; if (k > 0) id = P.node_parent[id]; }      // f[k] = state of F_k
        id = R.circ_leaf[c];
        for (int64_t d = n; d >= 0; d--) { r[(size_t)d] = id; if (d > 0) id = R.node_parent[id]; }      // r[d] = state of B_{n-d}
        for (int g = 0; g < nG; g++) cursor[g] = pos_ptr[(size_t)c * nG + g];
        for (int64_t k = 1; k <= n; k++) {
            const int g = P.circ_gates[a + k - 1];
            const int64_t at = cursor[g]++;
            pf[(size_t)at] = f[(size_t)k - 1];
            pr[(size_t)at] = r[(size_t)(n - k)];
        }
    }
}

}  // namespace gst
