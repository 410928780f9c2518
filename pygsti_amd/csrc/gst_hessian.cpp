// gst_hessian.cpp -- Hessian blocks behind the C ABI: finite differences of finite differences (mapforwardsim.py:394-438),
// the composed route for host-stepped models, exact blocks (matrixforwardsim.py:1190-1381) incl. general
// parameterisations, and the objective-Hessian rectangles (distforwardsim.py:304-340, objectivefns.py:4914-4968).
#include "gst_state.hpp"

using namespace gst_impl;

extern "C" {

// The host copy of the model with parameter `pi` stepped by eps, as `model.from_vector(vec)` leaves a model of
// one-parameter-per-element members (mapforwardsim.py:425-428); a TP POVM's complement is re-derived from the stepped
// effect in the reference's summation order (complementeffect.py:72-78).
static void step_host_model(gst_plan* p, int64_t pi, double eps)
{
    const int D = p->hp.D;
    const int32_t k = p->pkind[pi], o = p->pobj[pi], el = p->pelem[pi];
    if (k == GST_KIND_GATE) {
        const size_t at = (size_t)o * D * D + el;
        const double v = p->h_gates[at] + eps;
        p->h_gates[at] = v;
        p->h_gates_t[((size_t)o * D + el % D) * D + el / D] = v;
    } else if (k == GST_KIND_RHO) {
        p->h_rhos[(size_t)o * D + el] = p->h_rhos[(size_t)o * D + el] + eps;
    } else if (k == GST_KIND_EFFECT) {
        p->h_effects[(size_t)o * D + el] = p->h_effects[(size_t)o * D + el] + eps;
        if (p->comp_index >= 0 && std::find(p->comp_others.begin(), p->comp_others.end(), o) != p->comp_others.end()) {
            double sum = 0.0;
            for (int32_t q : p->comp_others) sum = sum + p->h_effects[(size_t)q * D + el];
            p->h_effects[(size_t)p->comp_index * D + el] = p->comp_identity[el] - sum;
        }
    }
    p->model_dirty = true;
}

// FD-of-FD Hessian block COMPOSED from FD Jacobians, literally as MapForwardSimulator._mapfill_hprobs_atom does it
// (mapforwardsim.py:420-436): dprobs over block 2 at theta; for every row parameter i the model is stepped to
// theta + eps e_i, dprobs2 = its FD Jacobian over block 2 (own base pass, `(orig + eps) + eps` where i is in block 2),
// row i = (dprobs2 - dprobs) / eps.  Every Jacobian is bit-identical to the reference's on its model, hence so is the
// block.  The route for plans the fused two-perturbation kernels do not cover (D = 64 with a complement effect): n1 + 2
// Jacobian passes instead of one fused launch.  Leaves behind what run_hprobs_dev leaves behind.
static int run_hprobs_composed(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                               int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2, double eps)
{
    int rc;
    const int64_t nE = p->hp.n_elements;
    HIP_TRY(p->d_dcol.ensure((size_t)nE * std::max<int64_t>(n2, 1)));
    HIP_TRY(p->d_hrow.ensure((size_t)nE * std::max<int64_t>(n2, 1)));
    HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * std::max<int64_t>(n1, 1)));
    if ((rc = run_dprobs_fd(p, p->d_dcol.p, n2, idx2, nullptr, n2, eps, nullptr, nullptr, 0))) return rc;
    const int32_t* d_dest2 = nullptr;
    if (dest2) {
        std::vector<int32_t> d2(dest2, dest2 + n2);
        if ((rc = upload_i32(p, p->d_hdest, d2))) return rc;
        d_dest2 = p->d_hdest.p;
    }
    const std::vector<double> g0 = p->h_gates, gt0 = p->h_gates_t, r0 = p->h_rhos, e0 = p->h_effects;
    auto restore = [&]() { p->h_gates = g0; p->h_gates_t = gt0; p->h_rhos = r0; p->h_effects = e0; p->model_dirty = true; };
    for (int64_t a = 0; a < n1; a++) {
        step_host_model(p, idx1[a], eps);
        if ((rc = upload_model(p)) || (rc = run_dprobs_fd(p, p->d_hrow.p, n2, idx2, nullptr, n2, eps, nullptr, nullptr, 0))) { restore(); return rc; }
        hipError_t he = gst::launch_hess_compose(p->d_hrow.p, p->d_dcol.p, nE, (int32_t)n2, eps, d_H, ld1, ld2, dest1 ? dest1[a] : a, d_dest2, p->stream);
        restore();
        if (he != hipSuccess) return fail(GST_EHIP, std::string("hess_compose: ") + hipGetErrorString(he));
    }
    if ((rc = upload_model(p))) return rc;
    // (the by-products the objective-Hessian rectangle reads: probabilities at theta, dprobs over block 1)
    return run_dprobs_fd(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, eps, nullptr, nullptr, 0);
}

// FD-of-FD Hessian block into the device buffer d_H [nE][ld1][ld2] (mapforwardsim.py:394-438).  Leaves behind, on the
// device: probabilities (d_pbase), FD dprobs over block 2 (d_dcol, [nE][n2]) and over block 1 (d_probs_tmp, [nE][n1]).
static int run_hprobs_dev(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                          int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2, double eps)
{
    int rc;
    if (p->comp_index >= 0) {
        for (int64_t c = 0; c < n1 + n2; c++) {
            const int64_t pi = c < n1 ? idx1[c] : idx2[c - n1];
            if (p->pkind[pi] == GST_KIND_EFFECT && p->pobj[pi] == p->comp_index) return fail(GST_EINVAL, "a parameter maps to the complement effect");
        }
    }
    // the fused two-perturbation kernels have no D = 64 form that re-derives a complement effect: that block is composed
    if (p->hess_composed || (p->comp_index >= 0 && p->hp.D == 64))
        return run_hprobs_composed(p, d_H, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2, eps);
    const int64_t nE = p->hp.n_elements;
    // (1) dprobs over block 2 at theta (mapforwardsim.py:420-421), FD step = eps
    HIP_TRY(p->d_dcol.ensure((size_t)nE * n2));
    if ((rc = run_dprobs_fd(p, p->d_dcol.p, n2, idx2, nullptr, n2, eps, nullptr, nullptr, 0))) return rc;
    // (2) probabilities at theta + eps e_i for every i of block 1 (the `probs` of the inner FD, pyx:349)
    HIP_TRY(p->d_raw.ensure((size_t)nE * n1));
    HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * n1));
    if ((rc = run_dprobs_fd(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, eps, nullptr, p->d_raw.p, n1))) return rc;
    // (3) all (i, j) pairs: wavefront = (row i, 64 columns j)
    const bool rows = (p->hp.D == 64);
    LaneLayout L2;
    if (rows) pack_waves(p, idx2, nullptr, n2, L2);
    else pack_lanes(p, idx2, nullptr, n2, L2);   // col = position in block 2
    const int32_t w2 = L2.n_waves;
    LaneLayout L;
    std::vector<int32_t> wave_row, wave_rowidx, lane_colidx;
    for (int64_t a = 0; rows && a < n1; a++) {       // one wavefront per (i, j) pair
        const int64_t pi = idx1[a];
        for (int32_t w = 0; w < w2; w++) {
            wave_row.push_back((int32_t)(dest1 ? dest1[a] : a));
            wave_rowidx.push_back((int32_t)a);
            const int32_t c = L2.col[w];
            L.col.push_back((int32_t)(dest2 ? dest2[c] : c));
            lane_colidx.push_back(c);
            L.kind[0].push_back(p->pkind[pi]); L.obj[0].push_back(p->pobj[pi]); L.elem[0].push_back(p->pelem[pi]);
            L.kind[1].push_back(L2.kind[0][w]); L.obj[1].push_back(L2.obj[0][w]); L.elem[1].push_back(L2.elem[0][w]);
        }
    }
    for (int64_t a = 0; !rows && a < n1; a++) {
        const int64_t pi = idx1[a];
        for (int32_t w = 0; w < w2; w++) {
            wave_row.push_back((int32_t)(dest1 ? dest1[a] : a));
            wave_rowidx.push_back((int32_t)a);
            for (int q = 0; q < 64; q++) {
                const size_t s = (size_t)w * 64 + q;
                const int32_t c = L2.col[s];
                L.col.push_back(c < 0 ? -1 : (int32_t)(dest2 ? dest2[c] : c));
                lane_colidx.push_back(c < 0 ? 0 : c);
                L.kind[0].push_back(p->pkind[pi]); L.obj[0].push_back(p->pobj[pi]); L.elem[0].push_back(p->pelem[pi]);
                L.kind[1].push_back(L2.kind[0][s]); L.obj[1].push_back(L2.obj[0][s]); L.elem[1].push_back(L2.elem[0][s]);
            }
        }
    }
    L.n_waves = rows ? (int32_t)L.col.size() : (int32_t)(L.col.size() / 64);
    p->cached_kind = 0;      // the shared lane tables are about to hold the (i, j) pairs
    if ((rc = upload_i32(p, p->d_lane[0], L.col))) return rc;
    for (int s = 0; s < 2; s++) {
        if ((rc = upload_i32(p, p->d_lane[1 + 3 * s], L.kind[s]))) return rc;
        if ((rc = upload_i32(p, p->d_lane[2 + 3 * s], L.obj[s]))) return rc;
        if ((rc = upload_i32(p, p->d_lane[3 + 3 * s], L.elem[s]))) return rc;
    }
    if ((rc = upload_i32(p, p->d_wave_row, wave_row))) return rc;
    if ((rc = upload_i32(p, p->d_wave_rowidx, wave_rowidx))) return rc;
    if ((rc = upload_i32(p, p->d_lane_colidx, lane_colidx))) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    gst::WalkArgs a;
    base_args(p, a);
    a.mode = gst::EMIT_HESS;
    a.out = d_H; a.ld = ld1; a.ld2 = ld2; a.eps = eps;
    a.prow = p->d_raw.p; a.ldrow = n1; a.dcol = p->d_dcol.p; a.lddcol = n2;
    a.pbase = p->d_pbase.p; a.base_cache = p->d_base_cache.p;
    a.lanes.col = p->d_lane[0].p;
    for (int s = 0; s < 2; s++) {
        a.lanes.kind[s] = p->d_lane[1 + 3 * s].p; a.lanes.obj[s] = p->d_lane[2 + 3 * s].p; a.lanes.elem[s] = p->d_lane[3 + 3 * s].p;
    }
    a.wave_row = p->d_wave_row.p; a.wave_rowidx = p->d_wave_rowidx.p; a.lane_colidx = p->d_lane_colidx.p;
    a.n_pwaves = L.n_waves;
    TIME_REC(p, evk0);
    if (rows) {
        a.rows_S = 2;
        HIP_TRY(gst::launch_walk_rows(p->hp.D, a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
    } else {
        const bool comp = p->comp_index >= 0;
        if (comp) {       // the complement description rides in the (otherwise unused here) effect-column tables
            if ((rc = upload_i32(p, p->d_ecol_tab, p->comp_others))) return rc;
            HIP_TRY(p->d_ecol_val.ensure(p->comp_identity.size()));
            H2D_TRY(p, p->d_ecol_val.p, p->comp_identity.data(), p->comp_identity.size() * 8);
            a.comp_index = p->comp_index; a.n_others = (int32_t)p->comp_others.size();
            a.comp_others = p->d_ecol_tab.p; a.comp_identity = p->d_ecol_val.p;
            p->cached_kind = 0;
        }
        HIP_TRY(gst::launch_walk(p->hp.D, 2, a, p->hp.n_tasks(), p->hp.max_slots, p->stream, 1, comp));
    }
    TIME_REC(p, evk1);
    p->last_launches++;
    return GST_OK;
}

// Exact Hessian block (what MatrixForwardSimulator returns) into the device buffer d_H [nE][ld1][ld2], D = 16 / 64:
//   H[e, t1, t2] = sum_{k: g_k = g2} B_k[a2] dF^{t1}_{k-1}[b2] + dB^{t1}_k[a2] F_{k-1}[b2]     (+ the SPAM columns)
// with the derivative states dF^{t1} (forward plan) and dB^{t1} (reversed plan, per effect) of four rows t1 at a time
// (dwalk_kernel) and the Jacobian's MFMA contraction run twice per row with one of the two caches swapped.
static int run_hprobs_analytic(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                               int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2)
{
    const gst::HostPlan& h = p->hp;
    if (!p->ana_mfma) return fail(GST_EUNSUPPORTED, "analytic Hessians need the two-cache contraction path");
    const int D = h.D, nEf = h.n_effects;
    const int64_t nE = h.n_elements;
    int rc;
    // set-up through the Jacobian path: column maps of block 2 (in the caller's column numbering), F and B caches
    HIP_TRY(p->d_hscratch.ensure((size_t)nE * ld2));
    p->last_ana_valid = false;
    p->want_cache_path = true;
    rc = run_dprobs_analytic(p, p->d_hscratch.p, ld2, idx2, dest2, n2, nullptr);
    p->want_cache_path = false;
    if (rc) return rc;
    if (!p->last_ana_valid) return fail(GST_EUNSUPPORTED, "analytic Hessians need the two-cache contraction path");
    gst::AnaArgs base = p->last_ana;
    const std::vector<int64_t> none_cols = p->cached_none_cols;
    {   // derivative-state caches (four row parameters per state) of 4 GB or more: the contraction's wide form
        const double cache_limit = p->test_cache_limit > 0 ? p->test_cache_limit : 4.0e9;
        if ((double)p->rev.n_state_ids * 4 * D * nEf * 8 >= cache_limit || (double)h.n_state_ids * 4 * D * 8 >= cache_limit) base.wide = 1;
    }
    HIP_TRY(p->d_dF.ensure((size_t)h.n_state_ids * 4 * D));
    HIP_TRY(p->d_dB.ensure((size_t)p->rev.n_state_ids * 4 * D * nEf));
    HIP_TRY(p->d_theta.ensure(5 * 4 * (size_t)(1 + nEf)));
    for (int64_t i0 = 0; i0 < n1; i0 += 4) {
        const int nt = (int)std::min<int64_t>(4, n1 - i0);
        // parameter tables: block 0 = forward walk, block 1 + x = backward walk from effect x
        std::vector<int32_t> th((size_t)5 * 4 * (1 + nEf), 0);
        for (int v = 0; v < 1 + nEf; v++) {
            int32_t* t = th.data() + (size_t)v * 20;          // inj_gate[4] inj_dst[4] inj_src[4] start_obj[4] start_idx[4]
            for (int q = 0; q < 4; q++) {
                t[q] = -1; t[12 + q] = -2;
                if (q >= nt) continue;
                const int64_t pi = idx1[i0 + q];
                const int k = p->pkind[pi], o = p->pobj[pi], el = p->pelem[pi];
                if (k == GST_KIND_GATE) {
                    t[q] = o;
                    t[4 + q] = (v == 0) ? el / D : el % D;       // forward: row a1 receives F[b1]; backward: row b1 receives B[a1]
                    t[8 + q] = (v == 0) ? el % D : el / D;
                } else if (k == GST_KIND_RHO && v == 0) { t[12 + q] = o; t[16 + q] = el; }
                else if (k == GST_KIND_EFFECT && v > 0 && o == v - 1) { t[12 + q] = -1; t[16 + q] = el; }
            }
        }
        H2D_TRY(p, p->d_theta.p, th.data(), th.size() * 4);
        HIP_TRY(hipStreamSynchronize(p->stream));
        gst::DWalkArgs w;
        std::memset(&w, 0, sizeof(w));
        w.n_gates = h.n_gates; w.n_theta = nt;
        auto tables = [&](int v) {
            const int32_t* t = p->d_theta.p + (size_t)v * 20;
            w.inj_gate = t; w.inj_dst = t + 4; w.inj_src = t + 8; w.start_obj = t + 12; w.start_idx = t + 16;
        };
        // dF over the forward plan
        tables(0);
        w.prog = p->d_prog.p; w.task_off = p->d_task_off.p; w.tile = p->d_gates_t.p;
        w.base = p->d_base_cache.p; w.bstride = D; w.bmul = 1; w.boff = 0;
        w.out = p->d_dF.p; w.ostride = D; w.omul = 1; w.ooff = 0;
        HIP_TRY(gst::launch_dwalk(D, w, h.n_tasks(), h.max_slots, p->stream));
        // dB over the reversed plan, one pass per effect
        w.prog = p->d_rprog.p; w.task_off = p->d_rtask_off.p; w.tile = p->d_gates.p;
        for (int x = 0; x < nEf; x++) {
            tables(1 + x);
            // backward-state layouts: D <= 16 [state][component][effect], D = 64 [state][effect][component]
            w.base = p->d_rev_cache.p; w.bstride = (int64_t)D * nEf;
            w.out = p->d_dB.p; w.ostride = (int64_t)D * nEf;
            if (D <= 16) { w.bmul = nEf; w.boff = x; w.omul = nEf; w.ooff = x; }
            else { w.bmul = 1; w.boff = (int64_t)x * D; w.omul = 1; w.ooff = (int64_t)x * D; }
            HIP_TRY(gst::launch_dwalk(D, w, p->rev.n_tasks(), p->rev.max_slots, p->stream));
        }
        p->last_launches += 1 + nEf;
        for (int q = 0; q < nt; q++) {
            const int64_t row = dest1 ? dest1[i0 + q] : i0 + q;
            gst::AnaArgs a = base;
            a.out = d_H + row * ld2; a.ld = ld1 * ld2;
            // theta_1 earlier than theta_2: derivative forward states against the backward states
            a.base_cache = p->d_dF.p + (size_t)q * D; a.fwd_stride = 4 * D * 8;
            a.rev_cache = p->d_rev_cache.p; a.rev_stride = 0;
            a.rho_zero = 1; a.eff_zero = 0; a.accumulate = 0;
            HIP_TRY(hipMemsetAsync(p->d_work_counter.p, 0, 8 * sizeof(uint32_t), p->stream));
            if (D == 64) HIP_TRY(gst::launch_analytic_mfma64(a, p->stream));
            else if (D == 16) HIP_TRY(gst::launch_analytic_mfma(a, p->stream));
            else HIP_TRY(gst::launch_analytic_small(a, p->stream));
            // theta_1 later: forward states against the derivative backward states, added
            a.base_cache = p->d_base_cache.p; a.fwd_stride = 0;
            a.rev_cache = p->d_dB.p + (size_t)q * D * nEf; a.rev_stride = (uint32_t)(4 * D * nEf * 8);
            a.rho_zero = 0; a.eff_zero = 1; a.accumulate = 1;
            HIP_TRY(hipMemsetAsync(p->d_work_counter.p, 0, 8 * sizeof(uint32_t), p->stream));
            if (D == 64) HIP_TRY(gst::launch_analytic_mfma64(a, p->stream));
            else if (D == 16) HIP_TRY(gst::launch_analytic_mfma(a, p->stream));
            else HIP_TRY(gst::launch_analytic_small(a, p->stream));
            p->last_launches += 2;
            for (int64_t col : none_cols)
                HIP_TRY(hipMemset2DAsync(d_H + row * ld2 + col, (size_t)ld1 * ld2 * 8, 0, 8, (size_t)nE, p->stream));
        }
    }
    return GST_OK;
}

// Exact Hessian block with gst_set_derivs, for parameterisations whose dense elements are LINEAR in the parameters
// (TP, ...): H_param[p1][p2] = sum_{a, b} (d elem_a / d p1) H_elem[a][b] (d elem_b / d p2), what
// MatrixForwardSimulator._hprobs_from_rho_e assembles when the members' hessian_wrt_params vanish
// (matrixforwardsim.py:1190-1287).  The element block is computed for the elements the requested parameters touch
// (identity element map, as run_dprobs_general does) and contracted with the sparse derivative columns on the device.
static int run_hprobs_general(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                              int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements;
    const int64_t n_el = (int64_t)h.n_rhos * D + (int64_t)h.n_effects * D + (int64_t)h.n_gates * D * D;
    const int64_t base_rho = 0, base_eff = (int64_t)h.n_rhos * D, base_gate = base_eff + (int64_t)h.n_effects * D;
    struct Csc { std::vector<int32_t> ptr, row, dest; std::vector<double> w; std::vector<int64_t> elems; };
    auto build = [&](const int64_t* idx, const int64_t* dest, int64_t n, Csc& c) -> int {
        std::vector<int32_t> pos((size_t)p->dv_n_params, -1);
        for (int64_t k = 0; k < n; k++) {
            if (idx[k] < 0 || idx[k] >= p->dv_n_params) return fail(GST_EINVAL, "parameter index out of range");
            if (pos[(size_t)idx[k]] >= 0) return fail(GST_EINVAL, "a parameter is requested twice (not supported with gst_set_derivs)");
            pos[(size_t)idx[k]] = (int32_t)k;
        }
        std::vector<std::vector<std::pair<int64_t, double>>> cols((size_t)n);      // (global element, weight) per requested parameter
        for (size_t o = 0; o < p->dv_kind.size(); o++) {
            const int k = p->dv_kind[o];
            const int K = k == GST_KIND_GATE ? D * D : D;
            const int64_t a0 = (k == GST_KIND_GATE ? base_gate : k == GST_KIND_RHO ? base_rho : base_eff) + (int64_t)p->dv_obj[o] * K;
            const int nc = p->dv_ncols[o];
            const double* dm = p->dv_deriv_h.data() + p->dv_off_deriv[o];
            for (int c2 = 0; c2 < nc; c2++) {
                const int32_t at = pos[(size_t)p->dv_param_idx[(size_t)p->dv_off_cols[o] + c2]];
                if (at < 0) continue;
                for (int r = 0; r < K; r++)
                    if (dm[(size_t)r * nc + c2] != 0.0) cols[(size_t)at].emplace_back(a0 + r, dm[(size_t)r * nc + c2]);
            }
        }
        std::vector<int64_t> el;
        for (auto& v : cols) for (auto& e : v) el.push_back(e.first);
        std::sort(el.begin(), el.end());
        el.erase(std::unique(el.begin(), el.end()), el.end());
        c.elems = el;
        c.ptr.assign((size_t)n + 1, 0);
        for (int64_t k = 0; k < n; k++) {
            for (auto& e : cols[(size_t)k]) {
                c.row.push_back((int32_t)(std::lower_bound(el.begin(), el.end(), e.first) - el.begin()));
                c.w.push_back(e.second);
            }
            c.ptr[(size_t)k + 1] = (int32_t)c.row.size();
            c.dest.push_back((int32_t)(dest ? dest[k] : k));
        }
        return GST_OK;
    };
    Csc c1, c2;
    int rc;
    if ((rc = build(idx1, dest1, n1, c1)) || (rc = build(idx2, dest2, n2, c2))) return rc;
    const int64_t m1 = (int64_t)c1.elems.size(), m2 = (int64_t)c2.elems.size();
    // (m1 == 0 or m2 == 0: nothing the atom applies depends on one of the blocks; the contraction then writes exact zeros)
    if ((double)nE * (double)std::max<int64_t>(m1, 1) * (double)std::max<int64_t>(m2, 1) * 8.0 > 64.0e9)
        return fail(GST_ENOMEM, "element-Hessian block too large: request smaller parameter blocks");
    HIP_TRY(p->d_helem.ensure((size_t)std::max<int64_t>(nE * m1 * m2, 1)));
    if (m1 > 0 && m2 > 0) {
        // element Hessian through the `full` path with the identity element map
        std::vector<int32_t> ek((size_t)n_el), eo((size_t)n_el), ee((size_t)n_el);
        int64_t q = 0;
        for (int r = 0; r < h.n_rhos; r++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_RHO; eo[q] = r; ee[q] = j; }
        for (int e = 0; e < h.n_effects; e++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_EFFECT; eo[q] = e; ee[q] = j; }
        for (int g = 0; g < h.n_gates; g++) for (int j = 0; j < D * D; j++, q++) { ek[q] = GST_KIND_GATE; eo[q] = g; ee[q] = j; }
        p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
        p->cached_kind = 0;
        const bool ds = p->derivs_set;
        const int32_t ci = p->comp_index;          // (a complement declared for the FD modes plays no role here: the
        p->derivs_set = false;                     //  derivative columns already carry its -1 entries)
        p->comp_index = -1;
        rc = run_hprobs_analytic(p, p->d_helem.p, m1, m2, c1.elems.data(), nullptr, m1, c2.elems.data(), nullptr, m2);
        p->derivs_set = ds;
        p->comp_index = ci;
        p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
        p->cached_kind = 0;
        if (rc) return rc;
    }
    // CSC tables: [ptr1 | row1 | dest1 | ptr2 | row2 | dest2], weights [w1 | w2]
    std::vector<int32_t> tab;
    const size_t o_p1 = 0, o_r1 = o_p1 + c1.ptr.size(), o_d1 = o_r1 + c1.row.size(), o_p2 = o_d1 + c1.dest.size(),
                 o_r2 = o_p2 + c2.ptr.size(), o_d2 = o_r2 + c2.row.size();
    tab.insert(tab.end(), c1.ptr.begin(), c1.ptr.end()); tab.insert(tab.end(), c1.row.begin(), c1.row.end());
    tab.insert(tab.end(), c1.dest.begin(), c1.dest.end());
    tab.insert(tab.end(), c2.ptr.begin(), c2.ptr.end()); tab.insert(tab.end(), c2.row.begin(), c2.row.end());
    tab.insert(tab.end(), c2.dest.begin(), c2.dest.end());
    std::vector<double> w(c1.w);
    w.insert(w.end(), c2.w.begin(), c2.w.end());
    if (w.empty()) w.push_back(0.0);
    if ((rc = upload_i32(p, p->d_hcsc, tab))) return rc;
    HIP_TRY(p->d_hw.ensure(w.size()));
    H2D_TRY(p, p->d_hw.p, w.data(), w.size() * 8);
    HIP_TRY(gst::launch_hessian_chain_rule(p->d_helem.p, nE, (int)m1, (int)m2, p->d_hcsc.p + o_p1, p->d_hcsc.p + o_r1, p->d_hw.p,
                                           p->d_hcsc.p + o_d1, (int)n1, p->d_hcsc.p + o_p2, p->d_hcsc.p + o_r2,
                                           p->d_hw.p + c1.w.size(), p->d_hcsc.p + o_d2, (int)n2, d_H, ld1, ld2, p->stream));
    p->last_launches++;
    HIP_TRY(hipStreamSynchronize(p->stream));          // the host tables go out of scope
    if (p->dv2_set) {
        // members that are not linear in their parameters: + sum_a (d p / d elem_a) d^2 elem_a / d p1 d p2, one MFMA
        // product per such object: [nE x K] (element Jacobian) . [K x (n_o x n_o)] scattered into the block's entries
        if (ld1 * ld2 > 0x7fffffffLL) return fail(GST_EINVAL, "Hessian block too wide");
        if ((rc = run_element_jacobian(p, nullptr))) return rc;
        std::vector<int32_t> pos1((size_t)p->dv_n_params, -1), pos2((size_t)p->dv_n_params, -1);
        for (int64_t k = 0; k < n1; k++) pos1[(size_t)idx1[k]] = (int32_t)(dest1 ? dest1[k] : k);
        for (int64_t k = 0; k < n2; k++) pos2[(size_t)idx2[k]] = (int32_t)(dest2 ? dest2[k] : k);
        for (size_t o = 0; o < p->dv_kind.size(); o++) {
            if (p->dv2_off[o] < 0) continue;
            const int k = p->dv_kind[o];
            const int K = k == GST_KIND_GATE ? D * D : D;
            const int nc = p->dv_ncols[o];
            const int64_t a0 = (k == GST_KIND_GATE ? base_gate : k == GST_KIND_RHO ? base_rho : base_eff) + (int64_t)p->dv_obj[o] * K;
            std::vector<int32_t> cmap((size_t)nc * nc, -1);
            bool any = false;
            for (int ca = 0; ca < nc; ca++) {
                const int32_t i = pos1[(size_t)p->dv_param_idx[(size_t)p->dv_off_cols[o] + ca]];
                if (i < 0) continue;
                for (int cb = 0; cb < nc; cb++) {
                    const int32_t j = pos2[(size_t)p->dv_param_idx[(size_t)p->dv_off_cols[o] + cb]];
                    if (j >= 0) { cmap[(size_t)ca * nc + cb] = (int32_t)((int64_t)i * ld2 + j); any = true; }
                }
            }
            if (!any) continue;
            if ((rc = upload_i32(p, p->d_dv_colmap, cmap))) return rc;
            HIP_TRY(gst::launch_chain_rule_gemm(p->d_jelem.p, n_el, a0, K, p->d_dv2.p + p->dv2_off[o], nc * nc, p->d_dv_colmap.p,
                                                d_H, ld1 * ld2, nE, p->stream));
            p->last_launches++;
            HIP_TRY(hipStreamSynchronize(p->stream));      // `cmap` goes out of scope; the next object reuses the buffer
        }
    }
    return GST_OK;
}

int gst_fill_hprobs_analytic(gst_plan* p, double* out, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                             int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (!p->derivs_set && !p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if (!out && n1 > 0 && n2 > 0) return fail(GST_EINVAL, "out is NULL");
    if (n1 < 0 || n2 < 0 || (n1 > 0 && !idx1) || (n2 > 0 && !idx2)) return fail(GST_EINVAL, "bad parameter list");
    if (!p->derivs_set && ((rc = check_params(p, idx1, n1)) || (rc = check_params(p, idx2, n2)))) return rc;
    const int64_t nE = p->hp.n_elements;
    if (n1 == 0 || n2 == 0) return end_call(p, true);
    if ((rc = stage_out(p, (size_t)nE * ld1 * ld2))) return rc;
    const bool dense = (!dest1 && !dest2 && ld1 == n1 && ld2 == n2);
    if (!dense && (rc = h2d_bytes(p, p->d_out.p, out, (size_t)nE * ld1 * ld2 * 8))) return rc;
    if (p->derivs_set) rc = run_hprobs_general(p, p->d_out.p, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2);
    else rc = run_hprobs_analytic(p, p->d_out.p, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2);
    if (rc) return rc;
    if ((rc = d2h_bytes(p, out, p->d_out.p, (size_t)nE * ld1 * ld2 * 8))) return rc;
    return end_call(p, true);
    });
}

int gst_fill_hprobs(gst_plan* p, double* out, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                    int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2, double eps)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (p->derivs_set) return fail(GST_EUNSUPPORTED, "Hessians need the one-parameter-per-element map (gst_set_derivs is set)");
    if (!p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if (!out && n1 > 0 && n2 > 0) return fail(GST_EINVAL, "out is NULL");
    if ((rc = check_params(p, idx1, n1)) || (rc = check_params(p, idx2, n2))) return rc;
    const int64_t nE = p->hp.n_elements;
    if (n1 == 0 || n2 == 0) return end_call(p, true);
    // dense device output [nE][n1'][n2'] in the caller's leading dimensions
    if ((rc = stage_out(p, (size_t)nE * ld1 * ld2))) return rc;
    // rows/columns of the caller's block that this call does not own must survive: start from the caller's data
    const bool dense = (!dest1 && !dest2 && ld1 == n1 && ld2 == n2);
    if (!dense && (rc = h2d_bytes(p, p->d_out.p, out, (size_t)nE * ld1 * ld2 * 8))) return rc;
    if ((rc = run_hprobs_dev(p, p->d_out.p, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2, eps))) return rc;
    if ((rc = d2h_bytes(p, out, p->d_out.p, (size_t)nE * ld1 * ld2 * 8))) return rc;
    return end_call(p, true);
    });
}

int gst_objective_hessian_block(gst_plan* p, const gst_objective_desc* d, const double* d_counts, const double* d_totals,
                                const int64_t* idx1, int64_t n1, const int64_t* idx2, int64_t n2, double eps, double* out)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (!d || !d_counts || !d_totals) return fail(GST_EINVAL, "bad argument");
    if (d->kind != GST_OBJ_CHI2 && d->kind != GST_OBJ_POISSON_DLOGL) return fail(GST_EINVAL, "unknown objective kind");
    if (!(d->min_prob_clip > 0.0) || (d->kind == GST_OBJ_POISSON_DLOGL && !(d->radius > 0.0)))
        return fail(GST_EINVAL, "min_prob_clip and radius must be positive");
    if (p->derivs_set && d->hessian_mode != GST_DERIV_ANALYTIC)
        return fail(GST_EUNSUPPORTED, "general parameterisations (gst_set_derivs) exist in GST_DERIV_ANALYTIC only");
    if (!p->derivs_set && !p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if (!out && n1 > 0 && n2 > 0) return fail(GST_EINVAL, "out is NULL");
    if (n1 < 0 || n2 < 0 || (n1 > 0 && !idx1) || (n2 > 0 && !idx2)) return fail(GST_EINVAL, "bad parameter list");
    if (!p->derivs_set && ((rc = check_params(p, idx1, n1)) || (rc = check_params(p, idx2, n2)))) return rc;
    if (n1 > 0x7fffffff || n2 > 0x7fffffff) return fail(GST_EINVAL, "block too large");
    const int64_t nE = p->hp.n_elements;
    if (n1 == 0 || n2 == 0) return end_call(p, true);
    if ((rc = stage_out(p, (size_t)nE * n1 * n2))) return rc;
    const double* d_d1 = nullptr;
    const double* d_d2 = nullptr;
    if (d->hessian_mode == GST_DERIV_ANALYTIC && p->derivs_set) {
        // linear general parameterisation (TP): chain-ruled Jacobians of both blocks and the chain-ruled Hessian block
        HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * n1));
        HIP_TRY(p->d_dcol.ensure((size_t)nE * n2));
        if ((rc = run_hprobs_general(p, p->d_out.p, n1, n2, idx1, nullptr, n1, idx2, nullptr, n2))) return rc;
        if ((rc = run_dprobs_general(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, nullptr))) return rc;
        if ((rc = run_dprobs_general(p, p->d_dcol.p, n2, idx2, nullptr, n2, nullptr))) return rc;
        d_d1 = p->d_probs_tmp.p; d_d2 = p->d_dcol.p;
    } else if (d->hessian_mode == GST_DERIV_ANALYTIC) {
        HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * n1));
        if ((rc = run_dprobs_analytic(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, nullptr))) return rc;
        if ((rc = run_hprobs_analytic(p, p->d_out.p, n1, n2, idx1, nullptr, n1, idx2, nullptr, n2))) return rc;
        d_d1 = p->d_probs_tmp.p; d_d2 = p->d_hscratch.p;     // (the Hessian driver leaves the block-2 Jacobian in its scratch)
    } else if (d->hessian_mode == GST_DERIV_FD) {
        if ((rc = run_hprobs_dev(p, p->d_out.p, n1, n2, idx1, nullptr, n1, idx2, nullptr, n2, eps))) return rc;
        d_d1 = p->d_probs_tmp.p; d_d2 = p->d_dcol.p;
    } else return fail(GST_EINVAL, "unknown hessian_mode");
    // objective coefficients on the (optionally clipped) probabilities, then the contraction over elements
    HIP_TRY(p->d_obj_dt.ensure((size_t)nE)); HIP_TRY(p->d_obj_ht.ensure((size_t)nE));
    double* d_probs = p->d_pbase.p;
    if (d->prob_clip_lo < d->prob_clip_hi) {
        // _clip_probs (objectivefns.py:4766-4774) through the element-wise objective kernel, on a copy
        HIP_TRY(p->d_obj_pc.ensure((size_t)nE)); HIP_TRY(p->d_obj_tmp.ensure((size_t)2 * nE));
        HIP_TRY(hipMemcpyAsync(p->d_obj_pc.p, p->d_pbase.p, (size_t)nE * 8, hipMemcpyDeviceToDevice, p->stream));
        const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (nE + 255) / 256));
        HIP_TRY(p->d_obj_part.ensure((size_t)nb));
        HIP_TRY(gst::launch_objective_rows(d->kind, p->d_obj_pc.p, d_counts, d_totals, nE, d->min_prob_clip, d->radius, d->prob_clip_lo,
                                           d->prob_clip_hi, p->d_obj_tmp.p, p->d_obj_tmp.p + nE, nullptr, p->d_obj_part.p, nb, p->stream));
        d_probs = p->d_obj_pc.p;
    }
    HIP_TRY(gst::launch_objective_coeffs(d->kind, d_probs, d_counts, d_totals, nE, d->min_prob_clip, d->radius, p->d_obj_dt.p,
                                         p->d_obj_ht.p, p->stream));
    const int n_slabs = gst::hessian_block_slabs(nE, (int)n1, (int)n2);
    HIP_TRY(p->d_hess_part.ensure((size_t)n_slabs * n1 * n2));
    HIP_TRY(p->d_hess_out.ensure((size_t)n1 * n2));
    HIP_TRY(gst::launch_hessian_block(p->d_out.p, d_d1, d_d2, p->d_obj_dt.p, p->d_obj_ht.p, nE, (int)n1, (int)n2,
                                      p->d_hess_part.p, n_slabs, p->d_hess_out.p, p->stream));
    p->last_launches += 2;
    if ((rc = d2h_bytes(p, out, p->d_hess_out.p, (size_t)n1 * n2 * 8))) return rc;
    return end_call(p, true);
    });
}

}  // extern "C"
