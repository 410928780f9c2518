// gst_fill_fd.cpp -- probabilities and finite-difference Jacobians behind the C ABI: the base pass (sequential walk, or
// the log-depth level passes for the modes without an ordering contract), lane packing and launch forms of the FD walk
// (dispatcher-placed workgroups, persistent per-SIMD queues with hand-overs, base pass inside the launch, fused base
// lane), and finite differences over whole dense model sets (mapforwardsim_calc_densitymx.pyx:149-383).
#include "gst_state.hpp"

using namespace gst_impl;

namespace gst_impl {

void base_args(gst_plan* p, gst::WalkArgs& a)
{
    std::memset(&a, 0, sizeof(a));
    a.prog = p->d_prog.p; a.task_off = p->d_task_off.p;
    a.eff_ptr = p->d_eff_ptr.p; a.eff_label = p->d_eff_label.p; a.eff_dest = p->d_eff_dest.p;
    a.gates = p->d_gates.p; a.gates_t = p->d_gates_t.p; a.rhos = p->d_rhos.p; a.effects = p->d_effects.p;
    a.n_gates = p->hp.n_gates; a.n_effects = p->hp.n_effects;
    a.n_pwaves = 1;
}

// Base probabilities into d_dst (device), S = 0 walk: one wavefront per task.  With `fill_cache` the
// pass also stores every state it produces (the derivative passes start from them).
// `reassoc`: the caller has no ordering contract (exact derivatives, GST_OPT_FAST_PROBS) -- a D = 64 plan then walks on
// the matrix cores (gst_kernels_chain64.hip), 5x faster per dependent step, results equal up to re-association.
int run_probs(gst_plan* p, double* d_dst, bool fill_cache, int chain_share, const uint32_t* guard, bool reassoc)
{
    gst::WalkArgs a;
    base_args(p, a);
    a.chain_share = chain_share;
    a.guard = guard;
    a.mode = gst::EMIT_PROBS;
    a.out = d_dst;
    if (fill_cache) {
        HIP_TRY(p->d_base_cache.ensure((size_t)p->hp.n_state_ids * p->hp.D));
        a.base_cache_w = p->d_base_cache.p;
    }
    a.rows_S = 0;
    if (reassoc && p->hp.D == 64 && p->fast_chains && !guard && gst::chain64_fits(1, p->hp.max_slots)) {
        HIP_TRY(gst::launch_chain64(a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
        p->last_levels = true;
    } else HIP_TRY(gst::launch_walk_rows(p->hp.D, a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
    p->last_launches++;
    return GST_OK;
}

// ---- log-depth chain passes (gst_levels.hpp / gst_kernels_levels.hip) -------------------------------------------------------

// The level program of the forward plan (`rev` false) or of the reversed plan, built once on the host (the reversed
// plan's while its state graph still exists: ensure_reverse).
void build_levels_host(gst_plan* p, bool rev, bool probs_only)
{
    gst_plan::Levels& L = rev ? p->lv_rev : (probs_only ? p->lv_probs : p->lv_fwd);
    if (L.built) return;
    L.built = true;
    const gst::HostPlan& h = rev ? p->rev : p->hp;
    L.why = gst::build_level_program(h, rev ? p->hp.n_effects : 1, L.prog, probs_only ? &p->hp.circ_leaf : nullptr);
    L.usable = L.why.empty();
}

int ensure_levels(gst_plan* p, bool rev, bool probs_only)
{
    gst_plan::Levels& L = rev ? p->lv_rev : (probs_only ? p->lv_probs : p->lv_fwd);
    const gst::HostPlan& h = rev ? p->rev : p->hp;
    build_levels_host(p, rev, probs_only);
    if (!L.usable || L.uploaded) return GST_OK;
    int rc;
    if ((rc = upload_i32(p, L.d_words, L.prog.words))) return rc;
    if ((rc = upload_i32(p, L.d_ids, L.prog.ids))) return rc;
    HIP_TRY(L.d_task_off.ensure(L.prog.task_off.size()));
    H2D_TRY(p, L.d_task_off.p, L.prog.task_off.data(), L.prog.task_off.size() * 8);
    HIP_TRY(L.d_ids_off.ensure(L.prog.task_ids_off.size()));
    H2D_TRY(p, L.d_ids_off.p, L.prog.task_ids_off.data(), L.prog.task_ids_off.size() * 8);
    HIP_TRY(L.d_mats.ensure(std::max<size_t>((size_t)h.n_tasks() * (size_t)std::max(L.prog.max_mats, 1) * 256, 1)));
    HIP_TRY(hipStreamSynchronize(p->stream));
    L.uploaded = true;
    return GST_OK;
}

void level_args(const gst_plan::Levels& L, gst::LevelArgs& a)
{
    std::memset(&a, 0, sizeof(a));
    a.words = L.d_words.p; a.task_off = L.d_task_off.p; a.ids = L.d_ids.p; a.task_ids_off = L.d_ids_off.p; a.mats = L.d_mats.p;
    a.nv = L.prog.nv; a.max_mats = std::max(L.prog.max_mats, 1);
    a.stage_lds = L.prog.max_task_ints * 4 <= 60 * 1024 ? 1 : 0;       // (a plan with huger tasks reads its program from memory)
    a.lds_ints = (int32_t)std::min<int64_t>(L.prog.max_task_ints, 15 * 1024);
}

bool levels_wanted(const gst_plan* p, const gst_plan::Levels& L)
{
    return L.usable && (p->fast_chains == 2 || (p->fast_chains == 1 && L.prog.worthwhile));
}

// Every state of the forward trie into the base-state cache by the level pass, then (d_dst != NULL) the probabilities from
// the circuits' final states.  The caller has checked levels_wanted(p, p->lv_fwd).
int run_levels_forward(gst_plan* p, double* d_dst, bool probs_only)
{
    const gst::HostPlan& h = p->hp;
    int rc;
    HIP_TRY(p->d_base_cache.ensure((size_t)h.n_state_ids * h.D));
    gst::LevelArgs a;
    level_args(probs_only ? p->lv_probs : p->lv_fwd, a);
    a.bmats = p->d_gates_t.p; a.starts = p->d_rhos.p; a.cache = p->d_base_cache.p;
    HIP_TRY(gst::launch_level_pass(a, h.n_tasks(), p->stream));
    p->last_launches++;
    if (d_dst) {
        if (!p->leaf_uploaded) {
            if ((rc = upload_i32(p, p->d_circ_leaf, h.circ_leaf))) return rc;
            HIP_TRY(hipStreamSynchronize(p->stream));
            p->leaf_uploaded = true;
        }
        HIP_TRY(gst::launch_probs_from_cache(p->d_base_cache.p, p->d_circ_leaf.p, p->d_eff_ptr.p, p->d_eff_label.p, p->d_eff_dest.p,
                                             p->d_effects.p, h.n_circuits, h.D, d_dst, p->stream));
        p->last_launches++;
    }
    p->last_levels = true;
    return GST_OK;
}

// gst_fill_probs*: the sequential walk (bit-identical to the reference), or with GST_OPT_FAST_PROBS the level pass
int run_probs_any(gst_plan* p, double* d_dst)
{
    p->last_levels = false;
    if (p->fast_probs && p->hp.D == 16 && p->fast_chains) {
        int rc = ensure_levels(p, false, true);
        if (rc) return rc;
        if (levels_wanted(p, p->lv_probs)) return run_levels_forward(p, d_dst, true);
    }
    return run_probs(p, d_dst, false, 1, nullptr, p->fast_probs);
}

// Pack the requested parameter columns into wavefronts of 64 lanes.  Lanes of one wavefront should
// perturb the same gate (the special-row work is paid per gate per wavefront): SPAM parameters are
// packed together, each gate's parameters start on a wavefront boundary when the gate has >= 32
// requested parameters.
void pack_lanes(const gst_plan* p, const int64_t* param_idx, const int64_t* dest_idx, int64_t n, LaneLayout& L,
                bool keep_lane63_idle)
{
    struct Item { int32_t kind, obj, elem, col; };
    std::vector<Item> spam, none;
    std::vector<std::vector<Item>> per_gate(p->hp.n_gates);
    for (int64_t c = 0; c < n; c++) {
        const int64_t pi = param_idx[c];
        Item it{p->pkind[pi], p->pobj[pi], p->pelem[pi], (int32_t)(dest_idx ? dest_idx[c] : c)};
        if (it.kind == GST_KIND_GATE) per_gate[it.obj].push_back(it);
        else if (it.kind == GST_KIND_NONE) none.push_back(it);
        else spam.push_back(it);
    }
    auto idle = [&]() { L.col.push_back(-1); L.kind[0].push_back(GST_KIND_NONE); L.obj[0].push_back(0); L.elem[0].push_back(0); };
    auto push = [&](const Item& it) {
        if (keep_lane63_idle && L.col.size() % 64 == 63) idle();       // (fused base lane: see WalkArgs::fused)
        L.col.push_back(it.col); L.kind[0].push_back(it.kind); L.obj[0].push_back(it.obj); L.elem[0].push_back(it.elem);
    };
    auto pad = [&]() {
        while (L.col.size() % 64) { L.col.push_back(-1); L.kind[0].push_back(GST_KIND_NONE); L.obj[0].push_back(0); L.elem[0].push_back(0); }
    };
    for (auto& it : spam) push(it);
    for (auto& it : none) push(it);
    for (auto& g : per_gate) {
        if (g.size() >= 32) pad();
        for (auto& it : g) push(it);
    }
    if (keep_lane63_idle && L.col.empty()) idle();
    pad();
    L.n_waves = (int32_t)(L.col.size() / 64);
}

// Row-per-lane kernel: one wavefront per requested column.
void pack_waves(const gst_plan* p, const int64_t* param_idx, const int64_t* dest_idx, int64_t n, LaneLayout& L)
{
    for (int64_t c = 0; c < n; c++) {
        const int64_t pi = param_idx[c];
        L.col.push_back((int32_t)(dest_idx ? dest_idx[c] : c));
        L.kind[0].push_back(p->pkind[pi]); L.obj[0].push_back(p->pobj[pi]); L.elem[0].push_back(p->pelem[pi]);
    }
    L.n_waves = (int32_t)n;
}

// Estimated cost of every (task, wavefront) pair of an FD request, longest first: a wavefront's work in a task is set
// by the objects its lanes perturb (a gate the task never applies costs almost nothing, a gate of the germ costs the
// whole chain; gst::task_gate_costs).  D = 64: the unit is a group of rows_group() consecutive wavefronts of one task.
void fd_items(gst_plan* p, const LaneLayout& L, bool rows, std::vector<std::pair<int32_t, uint32_t>>& items, int32_t& n_units)
{
    const int nG = p->hp.n_gates;
    const int64_t nT = p->hp.n_tasks();
    const int stride = nG + 2;
    std::vector<uint64_t> wave_gates(L.n_waves, 0);
    std::vector<uint8_t> wave_rho(L.n_waves, 0);
    for (size_t q = 0; q < L.col.size(); q++) {
        if (L.col[q] < 0) continue;
        const size_t w = rows ? q : q / 64;
        if (L.kind[0][q] == GST_KIND_GATE) wave_gates[w] |= 1ull << L.obj[0][q];
        else if (L.kind[0][q] == GST_KIND_RHO) wave_rho[w] = 1;
    }
    const int32_t grp = rows ? gst::rows_group(p->hp.D, p->hp.max_slots) : 1;
    n_units = (L.n_waves + grp - 1) / grp;
    items.clear();
    items.reserve((size_t)nT * n_units);
    for (int64_t t = 0; t < nT; t++) {
        const int32_t* c = p->task_cost.data() + (size_t)t * stride;
        for (int32_t u = 0; u < n_units; u++) {
            int32_t best = c[nG + 1] / 4;
            for (int32_t w = u * grp; w < std::min<int32_t>((u + 1) * grp, L.n_waves); w++) {
                if (wave_rho[w]) best = std::max(best, c[nG + 1] / 4 + c[nG]);
                else
                    for (uint64_t m = wave_gates[w]; m; m &= m - 1) best = std::max(best, c[nG + 1] / 4 + c[__builtin_ctzll(m)]);
            }
            items.emplace_back(-best, (uint32_t)(t * n_units + u));
        }
    }
    std::stable_sort(items.begin(), items.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
}

// FD Jacobian columns into device memory.  d_raw (optional) receives the perturbed probabilities.
int run_dprobs_fd(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                  int64_t n_param, double eps, double* d_probs_out, double* d_raw, int64_t ldraw)
{
    // base probabilities (pyx:349)
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    const bool rows = (p->hp.D == 64);
    // Launch-bound plans (1Q fits: a fill is two ~30 us chains behind each other): ONE launch -- lane 63 of every
    // wavefront walks the unperturbed model, so neither the base pass nor its state cache is needed.  Not for
    // complement effects (their columns are evaluated on the cached final states) and not for the Hessian driver's
    // passes (d_raw), which reuse the base pass's by-products.
    const bool fused = !rows && n_param > 0 && p->hp.n_state_ids <= 65536 && p->comp_index < 0 && !d_raw && p->fd_fused;
    int rc = GST_OK;
    p->last_overlap = false;
    p->last_fd_form = 0;
    if (n_param == 0) return run_probs(p, d_base, false);
    // (the base pass itself is enqueued below, once the launch form is known: the persistent launch of a small atom runs
    //  it inside its own kernel)
    if (p->cached_fused != fused) p->cached_kind = 0;
    if (!p->request_cached(1, param_idx, dest_idx, n_param)) {
        // (an optimizer asks for the same columns every iteration: pack and upload the lane tables once)
        p->cached_kind = 0;            // nothing below may leave a half-updated request looking cached when it fails
        LaneLayout L;
        // a declared complement effect: effect-parameter columns leave the walk (effect_fd_kernel below does them on
        // the cached final states, where the recomputed complement can be substituted)
        std::vector<int64_t> w_idx, w_dest;
        p->ecol_tab.clear();
        if (p->comp_index >= 0) {
            std::vector<int32_t> eo, ee, ed, et;
            for (int64_t c = 0; c < n_param; c++) {
                const int64_t pi = param_idx[c];
                const int64_t dst = dest_idx ? dest_idx[c] : c;
                if (p->pkind[pi] != GST_KIND_EFFECT) { w_idx.push_back(pi); w_dest.push_back(dst); continue; }
                const int32_t e = p->pobj[pi];
                if (e == p->comp_index) return fail(GST_EINVAL, "a parameter maps to the complement effect");
                eo.push_back(e); ee.push_back(p->pelem[pi]); ed.push_back((int32_t)dst);
                et.push_back(std::find(p->comp_others.begin(), p->comp_others.end(), e) != p->comp_others.end() ? 1 : 0);
            }
            p->ecol_tab.insert(p->ecol_tab.end(), eo.begin(), eo.end());
            p->ecol_tab.insert(p->ecol_tab.end(), ee.begin(), ee.end());
            p->ecol_tab.insert(p->ecol_tab.end(), ed.begin(), ed.end());
            p->ecol_tab.insert(p->ecol_tab.end(), et.begin(), et.end());
            if ((rc = upload_i32(p, p->d_ecol_tab, p->ecol_tab))) return rc;
        }
        const bool filtered = p->comp_index >= 0;
        const int64_t* l_idx = filtered ? w_idx.data() : param_idx;
        const int64_t* l_dest = filtered ? w_dest.data() : dest_idx;
        const int64_t l_n = filtered ? (int64_t)w_idx.size() : n_param;
        if (rows) pack_waves(p, l_idx, l_dest, l_n, L);     // one perturbed model per wavefront
        else pack_lanes(p, l_idx, l_dest, l_n, L, fused);
        p->cached_kind = 0;
        p->cached_fused = fused;
        if ((rc = upload_i32(p, p->d_lane[0], L.col))) return rc;
        if ((rc = upload_i32(p, p->d_lane[1], L.kind[0]))) return rc;
        if ((rc = upload_i32(p, p->d_lane[2], L.obj[0]))) return rc;
        if ((rc = upload_i32(p, p->d_lane[3], L.elem[0]))) return rc;
        // Launch order: longest (task, wavefront) pairs first.  A wavefront's work in a task is set by the objects
        // its lanes perturb (a gate the task never applies costs almost nothing, a gate of the germ costs the whole
        // chain), so with only a few pairs per SIMD -- a 1/8 atom of the 2Q design has 4.4 -- the order decides how
        // long the last SIMD runs.
        p->have_block_order = false;
        const int64_t nT = p->hp.n_tasks();
        if (p->task_cost.empty()) gst::task_gate_costs(p->hp, p->task_cost);
        if (!p->task_cost.empty() && nT * L.n_waves < 0x7fffffffLL && nT * L.n_waves > 1) {
            std::vector<std::pair<int32_t, uint32_t>> items;       // (-cost, task * n_units + unit), longest first
            int32_t n_units = 0;
            fd_items(p, L, rows, items, n_units);
            std::vector<uint32_t> order(items.size());
            for (size_t i = 0; i < items.size(); i++) order[i] = items[i].second;
            HIP_TRY(p->d_block_order.ensure(order.size()));
            H2D_TRY(p, p->d_block_order.p, order.data(), order.size() * 4);
            HIP_TRY(hipStreamSynchronize(p->stream));
            p->have_block_order = true;
            if (const char* tp = std::getenv("GST_FD_TRACE")) {          // development aid: the estimates next to the trace
                if (FILE* f = std::fopen((std::string(tp) + ".cost").c_str(), "wb")) {
                    for (auto& it : items) { int64_t r[2] = {(int64_t)it.second, (int64_t)-it.first}; std::fwrite(r, 8, 2, f); }
                    std::fclose(f);
                }
            }
            // Persistent launch (D <= 16): pack the pairs into one queue per SIMD with equal estimated work -- longest
            // first, each into the least loaded queue -- instead of leaving the placement to the dispatcher.
            // Measured on MI355X (2Q design, kernel ms, interleaved repeats, queues vs dispatcher): 1/8 atom 4.54 vs
            // 4.63, 1/4 atom 7.17 vs 7.37, 1/2 atom 14.3 vs 13.6 -- with many pairs per SIMD the dispatcher's dynamic
            // placement is as good or better, so the queues are used below 10 pairs per SIMD -- and not below one pair
            // per SIMD, where a fill is launch-bound and the extra memset and pops cost 10 us (1Q L<=128: 100 vs 110 us).
            p->have_bins = false;
            if (!rows && p->fd_persist && p->hp.max_slots <= 4 && items.size() < (1u << 30) && (p->fd_persist_always || (items.size() <= (size_t)40 * p->n_cus && items.size() >= (size_t)4 * p->n_cus)) &&
                (size_t)16 * std::max(p->hp.max_slots, 1) * p->hp.D * 64 * 8 <= 160 * 1024) {
                const int n_bins = 4 * p->n_cus;
                if (p->fd_handover != 0 && !p->split_ready) {
                    // cuts only where no save slot is live and the interpreter's outer loop stands (an EMIT, the word after a
                    // LOAD), the one nearest a walk's middle.  (Cuts at any position with live slots travelling along balance the
                    // ESTIMATED loads to 2 % and make the kernel slower -- 4.45 against 4.05 ms on a 1/8 atom: a second part
                    // popped before its first part is done occupies a wavefront.  Measured in round 3, removed in round 4.)
                    gst::task_split_candidates(p->hp, p->cand_ptr, p->cand_pc, p->cand_frac, 1 << 20, nullptr);
                    p->cand_live.clear();
                    p->split_ready = true;
                }
                gst::FdQueues Q;
                gst::pack_fd_queues(items, n_units, nT, n_bins, p->fd_handover, p->cand_ptr, p->cand_pc, p->cand_frac, p->cand_live, Q);
                const int32_t n_split = Q.n_split;
                const std::vector<int32_t>& bptr = Q.bin_ptr;
                const std::vector<uint32_t>& bitems = Q.bin_items;
                const std::vector<int32_t>& ho_index = Q.ho_index;
                const std::vector<int32_t>& ho_pc = Q.ho_pc;
                std::vector<int32_t> ho_live(Q.ho_live.begin(), Q.ho_live.end());
                if ((rc = upload_i32(p, p->d_bin_ptr, bptr))) return rc;
                HIP_TRY(p->d_bin_items.ensure(bitems.size()));
                H2D_TRY(p, p->d_bin_items.p, bitems.data(), bitems.size() * 4);
                HIP_TRY(p->d_bin_head.ensure((size_t)n_bins + 1));            // (+ the abort flag)
                p->n_split = n_split;
                if (n_split > 0) {
                    if ((rc = upload_i32(p, p->d_ho_index, ho_index))) return rc;
                    if ((rc = upload_i32(p, p->d_task_split_pc, ho_pc))) return rc;
                    if ((rc = upload_i32(p, p->d_ho_live, ho_live))) return rc;
                    HIP_TRY(p->d_ho_state.ensure((size_t)n_split * (size_t)(1 + std::max(p->hp.max_slots, 0)) * p->hp.D * 64));
                    HIP_TRY(p->d_ho_tag.ensure((size_t)n_split * 4));
                    HIP_TRY(p->d_ho_id.ensure((size_t)n_split));
                    HIP_TRY(p->d_ho_flag.ensure((size_t)n_split));
                }
                HIP_TRY(hipStreamSynchronize(p->stream));
                p->n_bins = n_bins;
                p->have_bins = true;
            }
        }
        HIP_TRY(hipStreamSynchronize(p->stream));       // the host vectors go out of scope
        p->remember_request(1, param_idx, dest_idx, n_param);
        p->cached_n_waves = L.n_waves;
        p->fd_request_serial++;
    }
    // ---- the base pass ----------------------------------------------------------------------------------------------
    // Persistent launches (small atoms, D = 16) walk the base chains INSIDE the FD kernel: ~0.45 ms of pure latency that
    // nothing overlapped (a 1/8 atom of the 2Q design: 4.4 ms per step).  Needs the chain kernel's tables in LDS next to
    // the walks' save slots and at most two chains per workgroup.
    const int split_req = (p->fd_split == 2 || p->fd_split == 4) && p->hp.D == 16 ? p->fd_split : 1;
    const bool persist = !rows && split_req == 1 && p->have_bins && p->have_block_order && p->cached_n_waves > 0;
    bool overlap = false;
    int32_t ovl_chain_doubles = 0;
    if (persist && !fused && p->fd_overlap && p->hp.D == 16 && p->comp_index < 0 && !d_raw &&
        gst::chain_kernel_fits(p->hp.D, p->hp.n_gates, p->hp.n_effects, p->hp.max_slots)) {
        const int waves = gst::persistent_waves(p->hp.D);
        const int64_t chains = (p->hp.n_tasks() + p->n_cus - 1) / p->n_cus;
        ovl_chain_doubles = (int32_t)((gst::chain_lds_doubles(p->hp.D, p->hp.n_gates, p->hp.n_effects, p->hp.max_slots) + 1) & ~(size_t)1);
        const size_t lds = ((size_t)waves * std::max(p->hp.max_slots, 1) * p->hp.D * 64 + (size_t)chains * ovl_chain_doubles) * 8;
        overlap = chains <= 2 && lds <= 160 * 1024;
    }
    if (!fused && (!overlap || p->fd_overlap_diag) && (rc = run_probs(p, d_base, true))) return rc;
    gst::WalkArgs a;
    base_args(p, a);
    a.mode = gst::EMIT_FD;
    a.out = d_out; a.ld = ld; a.eps = eps; a.pbase = d_base;
    a.raw = d_raw; a.ldraw = ldraw;
    a.base_cache = p->d_base_cache.p;
    a.fused = fused ? 1 : 0; a.probs_out = fused ? d_base : nullptr;
    a.lanes.col = p->d_lane[0].p; a.lanes.kind[0] = p->d_lane[1].p; a.lanes.obj[0] = p->d_lane[2].p; a.lanes.elem[0] = p->d_lane[3].p;
    a.n_pwaves = p->cached_n_waves;
    a.block_order = p->have_block_order ? p->d_block_order.p : nullptr;
    const char* trace_path = std::getenv("GST_FD_TRACE");          // development aid: per-pair timestamps (tools/trace_stats.py)
    const size_t n_trace = (size_t)p->hp.n_tasks() * ((size_t)std::max(p->cached_n_waves, 1) + 1) + (size_t)std::max(p->n_split, 0);
    if (trace_path && !rows) {
        HIP_TRY(p->d_trace.ensure(1 + 4 * n_trace));
        HIP_TRY(hipMemsetAsync(p->d_trace.p, 0, 8, p->stream));
        a.trace = (unsigned long long*)p->d_trace.p;
    }
    TIME_REC(p, evk0);
    if (p->comp_index >= 0 && !p->ecol_tab.empty()) {
        const int D = p->hp.D;
        const int32_t nc = (int32_t)(p->ecol_tab.size() / 4);
        // this call's perturbed values: theta + eps for the effect itself; identity - sum(others), the others summed
        // from 0 in the declared order (Python's sum()), for the complement (complementeffect.py:72-78)
        p->ecol_val.assign((size_t)2 * nc, 0.0);
        for (int32_t k = 0; k < nc; k++) {
            const int32_t e = p->ecol_tab[k], i = p->ecol_tab[nc + k];
            const double own = p->h_effects[(size_t)e * D + i] + eps;
            double sum = 0.0;
            for (int32_t o : p->comp_others) sum = sum + (o == e ? own : p->h_effects[(size_t)o * D + i]);
            p->ecol_val[k] = own;
            p->ecol_val[nc + k] = p->comp_identity[i] - sum;
        }
        HIP_TRY(p->d_ecol_val.ensure(p->ecol_val.size()));
        H2D_TRY(p, p->d_ecol_val.p, p->ecol_val.data(), p->ecol_val.size() * 8);
        if (!p->leaf_uploaded) {
            if ((rc = upload_i32(p, p->d_circ_leaf, p->hp.circ_leaf))) return rc;
            p->leaf_uploaded = true;
        }
        gst::EffectFDArgs ea;
        std::memset(&ea, 0, sizeof(ea));
        ea.n_circuits = p->hp.n_circuits; ea.n_cols = nc; ea.D = D; ea.comp_index = p->comp_index;
        ea.circ_leaf = p->d_circ_leaf.p; ea.eff_ptr = p->d_eff_ptr.p; ea.eff_label = p->d_eff_label.p; ea.eff_dest = p->d_eff_dest.p;
        ea.effects = p->d_effects.p; ea.base_cache = p->d_base_cache.p; ea.pbase = d_base;
        ea.col_obj = p->d_ecol_tab.p; ea.col_elem = p->d_ecol_tab.p + nc; ea.col_dest = p->d_ecol_tab.p + 2 * nc;
        ea.col_touches_comp = p->d_ecol_tab.p + 3 * nc;
        ea.col_own = p->d_ecol_val.p; ea.col_comp = p->d_ecol_val.p + nc;
        ea.out = d_out; ea.ld = ld; ea.raw = d_raw; ea.ldraw = ldraw; ea.eps = eps;
        HIP_TRY(gst::launch_effect_fd(ea, p->stream));
        p->last_launches++;
    }
    if (a.n_pwaves == 0) {
        // (every requested column was an effect parameter)
    } else if (rows) {
        a.rows_S = 1;
        HIP_TRY(gst::launch_walk_rows(p->hp.D, a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
    } else {
        // gst_options.fd_split > 1 splits every (task, 64 columns) pair's rows over 2 or 4 wavefronts (walk_kernel's
        // NW).  Bit-identical, but measured on MI355X it costs 1.35-1.45x the SIMD time per pair (one barrier per gate
        // application) and that cancels the balance it buys on a 1/8 atom (4.63 vs 4.70 ms), so "auto" is 1.
        const int split = split_req;
        if (persist) {
            // persistent launch: one workgroup per CU, pairs popped from the per-SIMD queues
            gst::WalkArgs sb = a;                  // (the stand-by launches' arguments: the plain dispatcher-placed form)
            a.bin_ptr = p->d_bin_ptr.p; a.bin_items = p->d_bin_items.p; a.bin_head = p->d_bin_head.p; a.n_bins = p->n_bins;
            if (p->n_split > 0) {
                a.ho_pc = p->d_task_split_pc.p; a.ho_index = p->d_ho_index.p; a.ho_state = p->d_ho_state.p;
                a.ho_id = p->d_ho_id.p; a.ho_flag = p->d_ho_flag.p;
                a.ho_live = (const uint32_t*)p->d_ho_live.p; a.ho_tag = p->d_ho_tag.p; a.ho_blocks = 1 + std::max(p->hp.max_slots, 0);
                HIP_TRY(hipMemsetAsync(p->d_ho_flag.p, 0, (size_t)p->n_split * 4, p->stream));
            }
            a.lds_wave_doubles = std::max(p->hp.max_slots, 1) * p->hp.D * 64;
            // queue heads, and behind them the abort flag of the bounded waits (hand-over, overlap)
            HIP_TRY(hipMemsetAsync(p->d_bin_head.p, 0, ((size_t)p->n_bins + 1) * 4, p->stream));
            uint32_t* const d_abort = p->d_bin_head.p + p->n_bins;
            const bool can_wait = p->n_split > 0 || overlap;
            a.abort_flag = can_wait ? d_abort : nullptr;
            if (overlap) {
                HIP_TRY(p->d_base_cache.ensure((size_t)p->hp.n_state_ids * p->hp.D));
                a.base_cache = p->d_base_cache.p;
                // consumers tell "not produced yet" from a value by this bit pattern (gst_chain.hpp)
                if (!p->fd_overlap_diag) {
                    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_base, (int)gst::OVL_SENTINEL32, (size_t)p->hp.n_elements * 2, p->stream));
                    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p->d_base_cache.p, (int)gst::OVL_SENTINEL32, (size_t)p->hp.n_state_ids * p->hp.D * 2, p->stream));
                }
                a.ovl_n_tasks = (int32_t)p->hp.n_tasks();
                a.ovl_chain_doubles = ovl_chain_doubles;
                a.pbase_w = d_base;
                a.ovl_test_skip = (p->test_skip_chains || p->fd_overlap_diag) ? 1 : 0;
                p->last_overlap = true;
            }
            HIP_TRY(gst::launch_walk_persistent(p->hp.D, a, p->n_cus, p->hp.max_slots, p->stream));
            p->last_fd_form = overlap ? 2 : 1;
            if (can_wait && p->fd_standby) {
                // Stand-by launches: the same work in the form that waits for nothing -- separate base pass, one workgroup
                // per pair, no hand-over -- guarded by the abort flag: every workgroup leaves at once unless a bounded
                // wait of the launch above ran out (its producer not resident: a shared device).  Costs two empty
                // launches per fill; buys "never hangs, never returns a half-written Jacobian" without a host round trip.
                if (overlap) {
                    if ((rc = run_probs(p, d_base, true, 1, d_abort))) return rc;
                    p->last_launches--;            // (counted below, once, like every FD fill)
                }
                sb.base_cache = p->d_base_cache.p;
                sb.guard = d_abort;
                HIP_TRY(gst::launch_walk(p->hp.D, 1, sb, p->hp.n_tasks(), p->hp.max_slots, p->stream, 1));
            }
        } else
            HIP_TRY(gst::launch_walk(p->hp.D, 1, a, p->hp.n_tasks(), p->hp.max_slots, p->stream, split));
    }
    TIME_REC(p, evk1);
    p->last_launches++;
    if (a.trace) {
        std::vector<uint64_t> h(1 + 4 * n_trace);
        int rc_t = d2h_bytes(p, h.data(), p->d_trace.p, h.size() * 8);
        if (rc_t) return rc_t;
        if (FILE* f = std::fopen(trace_path, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
    }
    return GST_OK;
}

// Finite differences over ANY parameterisation (gst_fill_dprobs_models): column m = (p(model set m) - p(base)) / eps, the
// base being gst_set_model's model.  Every model set is a complete dense model (what set_parameter_value + to_dense give
// on the host), so nothing is assumed about which elements a parameter moves; the price is that no state is shared
// with the base pass -- each (task, model set) pair is a full probability walk (chain kernel at D <= 16, row-per-lane
// kernel at D = 64).  Model sets are processed in chunks that bound the scratch (probability vectors) to 2 GB.
// `nm` model sets resident in d_mm_models ([gates_t | rhos | effects] each): one independent probability walk per (walk
// program, set) into d_mm_raw, then columns m0 .. m0 + nm (or d_dest) of d_out = (p_set - p_base) / eps.
int run_models_chunk(gst_plan* p, int64_t nm, int64_t m0, const double* d_base, double* d_out, int64_t ld, const int32_t* d_dest, double eps)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements, nT = h.n_tasks();
    const size_t ng = (size_t)h.n_gates * D * D, nr = (size_t)h.n_rhos * D, ne = (size_t)h.n_effects * D;
    gst::WalkArgs w;
    base_args(p, w);
    w.gates = nullptr;
    w.gates_t = p->d_mm_models.p; w.rhos = p->d_mm_models.p + ng; w.effects = p->d_mm_models.p + ng + nr;
    w.n_models = (int32_t)nm; w.model_stride = (int64_t)(ng + nr + ne); w.out_model_stride = nE; w.mm_tasks = (int32_t)nT;
    w.n_pwaves = (int32_t)nm;
    w.mode = gst::EMIT_PROBS; w.rows_S = 0; w.out = p->d_mm_raw.p;
    HIP_TRY(gst::launch_walk_rows(D, w, nT, h.max_slots, p->stream));
    HIP_TRY(gst::launch_fd_from_models(p->d_mm_raw.p, nE, d_base, nE, (int32_t)nm, d_dest, (int32_t)m0, eps, d_out, ld, p->stream));
    p->last_launches += 2;
    return GST_OK;
}

int run_dprobs_models(gst_plan* p, int64_t n_models, const double* gates, const double* rhos, const double* effects,
                      double* d_out, int64_t ld, const int64_t* dest_idx, double eps, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D, Du = p->user_D();
    const int64_t nE = h.n_elements;
    const size_t ng = (size_t)h.n_gates * D * D, nr = (size_t)h.n_rhos * D, ne = (size_t)h.n_effects * D;
    const size_t ngu = (size_t)h.n_gates * Du * Du, nru = (size_t)h.n_rhos * Du, neu = (size_t)h.n_effects * Du;   // the caller's sets
    const size_t stride = ng + nr + ne;
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    int rc = run_probs(p, d_base, false);
    if (rc) return rc;
    if (n_models == 0) return GST_OK;
    if (n_models > 0x7fffffffLL) return fail(GST_EINVAL, "too many model sets");
    const int64_t nT = h.n_tasks();
    int64_t chunk = std::max<int64_t>(1, (int64_t)(2.0e9 / (8.0 * (double)std::max<int64_t>(nE, 1))));
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, 0x7fffffffLL / std::max<int64_t>(nT, 1)));
    chunk = std::min<int64_t>(chunk, n_models);
    HIP_TRY(p->d_mm_models.ensure((size_t)chunk * stride));
    HIP_TRY(p->d_mm_raw.ensure((size_t)chunk * (size_t)std::max<int64_t>(nE, 1)));
    std::vector<int32_t> dest32;
    if (dest_idx) {
        dest32.resize((size_t)n_models);
        for (int64_t m = 0; m < n_models; m++) {
            if (dest_idx[m] < 0 || dest_idx[m] >= ld) return fail(GST_EINVAL, "destination column out of range");
            dest32[(size_t)m] = (int32_t)dest_idx[m];
        }
        if ((rc = upload_i32(p, p->d_mm_dest, dest32))) return rc;
    } else if (n_models > ld) return fail(GST_EINVAL, "more model sets than columns");
    std::vector<double> stage((size_t)chunk * stride);
    TIME_REC(p, evk0);
    for (int64_t m0 = 0; m0 < n_models; m0 += chunk) {
        const int64_t nm = std::min<int64_t>(chunk, n_models - m0);
        if (Du != D) std::fill(stage.begin(), stage.begin() + (size_t)nm * stride, 0.0);
        for (int64_t m = 0; m < nm; m++) {
            double* dst = stage.data() + (size_t)m * stride;
            const double* G = gates + (size_t)(m0 + m) * ngu;
            for (int g = 0; g < h.n_gates; g++)
                for (int i = 0; i < Du; i++)
                    for (int j = 0; j < Du; j++) dst[((size_t)g * D + j) * D + i] = G[((size_t)g * Du + i) * Du + j];
            for (int r = 0; r < h.n_rhos; r++) std::memcpy(dst + ng + (size_t)r * D, rhos + (size_t)(m0 + m) * nru + (size_t)r * Du, (size_t)Du * 8);
            for (int e = 0; e < h.n_effects; e++) std::memcpy(dst + ng + nr + (size_t)e * D, effects + (size_t)(m0 + m) * neu + (size_t)e * Du, (size_t)Du * 8);
        }
        H2D_TRY(p, p->d_mm_models.p, stage.data(), (size_t)nm * stride * 8);
        if ((rc = run_models_chunk(p, nm, m0, d_base, d_out, ld, dest_idx ? p->d_mm_dest.p + m0 : nullptr, eps))) return rc;
        HIP_TRY(hipStreamSynchronize(p->stream));          // the staging vector is refilled by the next chunk
    }
    TIME_REC(p, evk1);
    return GST_OK;
}

}  // namespace gst_impl

