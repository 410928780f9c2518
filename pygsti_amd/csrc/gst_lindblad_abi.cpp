// gst_lindblad_abi.cpp -- Lindblad-parameterised members on the device (gst_set_lindblad): upload of the description,
// base members per parameter vector, FD Jacobians through the dirty programs (gst_kernels_pert.hip) and exact ones
// through the Frechet derivatives + chain rule (lindbladerrorgen.py:658-742, experrorgenop.py:114-123).
#include "gst_state.hpp"

using namespace gst_impl;

namespace gst_impl {

// ---- Lindblad-parameterised members (gst_set_lindblad) -------------------------------------------------------------------
size_t lb_set_stride(const gst_plan* p)
{
    const int D = p->hp.D;
    return (size_t)p->hp.n_gates * D * D + (size_t)p->hp.n_rhos * D + (size_t)p->hp.n_effects * D;
}

int lb_upload(gst_plan* p)
{
    gst_plan::Lindblad& L = p->lb;
    if (L.uploaded) return GST_OK;
    std::vector<int32_t> i32;
    for (const auto* v : {&L.kind, &L.obj, &L.n_eff, &L.n_par, &L.n_blocks, &L.blk_type, &L.blk_mode, &L.blk_n}) i32.insert(i32.end(), v->begin(), v->end());
    std::vector<int64_t> i64;
    for (const auto* v : {&L.param0, &L.term_off, &L.static_off}) i64.insert(i64.end(), v->begin(), v->end());
    int rc;
    if ((rc = upload_i32(p, p->d_lb_i32, i32))) return rc;
    HIP_TRY(p->d_lb_i64.ensure(i64.size()));
    H2D_TRY(p, p->d_lb_i64.p, i64.data(), i64.size() * 8);
    HIP_TRY(p->d_lb_statics.ensure(L.statics.size()));
    H2D_TRY(p, p->d_lb_statics.p, L.statics.data(), L.statics.size() * 8);
    HIP_TRY(p->d_lb_term_re.ensure(L.term_re.size()));
    H2D_TRY(p, p->d_lb_term_re.p, L.term_re.data(), L.term_re.size() * 8);
    HIP_TRY(p->d_lb_term_im.ensure(L.term_im.size()));
    H2D_TRY(p, p->d_lb_term_im.p, L.term_im.data(), L.term_im.size() * 8);
    HIP_TRY(p->d_lb_theta.ensure((size_t)std::max(L.n_params, 1)));
    HIP_TRY(p->d_lb_base.ensure(lb_set_stride(p)));
    HIP_TRY(p->d_lb_gates_rm.ensure(std::max<size_t>((size_t)p->hp.n_gates * p->hp.D * p->hp.D, 1)));
    if (p->hp.D == 64) HIP_TRY(p->d_lb_ws64.ensure(gst::lindblad64_workspace_doubles(L.n_members)));
    HIP_TRY(hipStreamSynchronize(p->stream));
    L.uploaded = true;
    return GST_OK;
}

void lb_args(gst_plan* p, gst::LbArgs& a)
{
    const gst_plan::Lindblad& L = p->lb;
    std::memset(&a, 0, sizeof(a));
    const size_t nm = (size_t)L.n_members;
    a.n_members = L.n_members; a.n_gates = p->hp.n_gates; a.n_rhos = p->hp.n_rhos; a.n_effects = p->hp.n_effects;
    const int32_t* i = p->d_lb_i32.p;
    a.kind = i; a.obj = i + nm; a.n_eff = i + 2 * nm; a.n_params = i + 3 * nm; a.n_blocks = i + 4 * nm;
    a.blk_type = i + 5 * nm; a.blk_mode = a.blk_type + nm * gst::LB_MAX_BLOCKS; a.blk_n = a.blk_mode + nm * gst::LB_MAX_BLOCKS;
    const int64_t* l = p->d_lb_i64.p;
    a.param0 = l; a.term_off = l + nm; a.static_off = l + 2 * nm;
    a.theta = p->d_lb_theta.p; a.term_re = p->d_lb_term_re.p; a.term_im = p->d_lb_term_im.p; a.statics = p->d_lb_statics.p;
    a.base_set = p->d_lb_base.p;
    a.set_stride = (int64_t)lb_set_stride(p);
}

// FD Jacobian columns of a Lindblad-parameterised model WITH state sharing (gst_kernels_pert.hip): the device builds the
// one changed member of every column; the base pass fills the state cache; every (task, member) pair whose dirty program
// is not empty is walked for 64/D columns per wavefront; POVM columns come from the circuits' final base states; every
// entry no walk reaches is an exact zero, written by one streaming pass.
int run_dprobs_lindblad_shared(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                               int64_t n_param, double eps, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D, G = 64 / D;
    const gst_plan::Lindblad& L = p->lb;
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    int rc = run_probs(p, d_base, n_param > 0);
    if (rc || n_param == 0) return rc;
    const int64_t nT = h.n_tasks();
    if (!p->dirty_ready) {
        gst::build_dirty_programs(h, p->dirty);
        HIP_TRY(p->d_dirty_words.ensure(p->dirty.words.size() + 64));
        HIP_TRY(hipMemsetAsync(p->d_dirty_words.p, 0, (p->dirty.words.size() + 64) * 4, p->stream));
        H2D_TRY(p, p->d_dirty_words.p, p->dirty.words.data(), p->dirty.words.size() * 4);
        HIP_TRY(p->d_dirty_off.ensure(p->dirty.off.size()));
        H2D_TRY(p, p->d_dirty_off.p, p->dirty.off.data(), p->dirty.off.size() * 8);
        HIP_TRY(hipStreamSynchronize(p->stream));
        p->dirty_ready = true;
    }
    if (!p->leaf_uploaded) {
        if ((rc = upload_i32(p, p->d_circ_leaf, h.circ_leaf))) return rc;
        p->leaf_uploaded = true;
    }
    if (!p->request_cached(5, param_idx, dest_idx, n_param)) {
        p->cached_kind = 0;
        // columns grouped by member, G per wavefront
        std::vector<std::vector<int64_t>> by_member((size_t)L.n_members);
        std::vector<int64_t> no_member;
        for (int64_t c = 0; c < n_param; c++) {
            const int64_t gp = param_idx[c];
            int m = -1;
            for (int mm = 0; mm < L.n_members; mm++)
                if (gp >= L.param0[(size_t)mm] && gp < L.param0[(size_t)mm] + L.n_par[(size_t)mm]) m = mm;
            const int64_t dst = dest_idx ? dest_idx[c] : c;
            if (dst < 0 || dst >= ld) return fail(GST_EINVAL, "destination column out of range");
            // a parameter of no member of THIS plan (an object the atom never applies, cf. GST_KIND_NONE): the reference's
            // step changes no probability of the atom -- the column is an exact zero, written by the zero-fill pass
            if (m < 0) no_member.push_back(c);
            else by_member[(size_t)m].push_back(c);
        }
        std::vector<int64_t> set_param;
        std::vector<int32_t> wk, wo, wn, w0, wc, cdest, zero_dest;
        std::vector<int32_t> rl_col, rl_kind, rl_obj, rl_elem;       // preparation columns: lanes of the lane-per-model FD kernel
        p->lb_povm_cols.clear();
        for (int m = 0; m < L.n_members; m++) {
            const auto& cols = by_member[(size_t)m];
            if (cols.empty()) continue;
            if (L.kind[(size_t)m] == GST_KIND_RHO && p->lb_rho_lanes) {
                // A perturbed preparation changes every state of every circuit but no gate: these columns are exactly what
                // walk_kernel's lanes are for -- 64 columns per wavefront, every gate coefficient a scalar operand -- once a
                // lane may start from a whole perturbed vector (WalkArgs::rho_models) instead of one stepped element.
                for (size_t k = 0; k < cols.size(); k++) {
                    const int64_t c = cols[k];
                    rl_col.push_back((int32_t)(dest_idx ? dest_idx[c] : c)); rl_kind.push_back(GST_KIND_RHO);
                    rl_obj.push_back(L.obj[(size_t)m]); rl_elem.push_back((int32_t)set_param.size());
                    set_param.push_back(param_idx[c]);
                    cdest.push_back((int32_t)(dest_idx ? dest_idx[c] : c));
                }
                while (rl_col.size() % 64) { rl_col.push_back(-1); rl_kind.push_back(GST_KIND_NONE); rl_obj.push_back(0); rl_elem.push_back(0); }
                continue;
            }
            const bool povm = L.kind[(size_t)m] == GST_KIND_EFFECT;
            if (povm) {
                p->lb_povm_cols.push_back(L.obj[(size_t)m]); p->lb_povm_cols.push_back(L.n_eff[(size_t)m]);
                p->lb_povm_cols.push_back((int32_t)set_param.size()); p->lb_povm_cols.push_back((int32_t)cols.size());
            }
            for (size_t k = 0; k < cols.size(); k += (size_t)G) {
                const size_t n = std::min<size_t>((size_t)G, cols.size() - k);
                if (!povm) {
                    wk.push_back(L.kind[(size_t)m]); wo.push_back(L.obj[(size_t)m]); wn.push_back(L.n_eff[(size_t)m]);
                    w0.push_back((int32_t)set_param.size()); wc.push_back((int32_t)n);
                }
                for (size_t q = 0; q < n; q++) {
                    const int64_t c = cols[k + q];
                    set_param.push_back(param_idx[c]);
                    const int32_t dst = (int32_t)(dest_idx ? dest_idx[c] : c);
                    cdest.push_back(dst);
                    if (!povm) zero_dest.push_back(dst);
                }
            }
        }
        p->lb_n_sets = (int64_t)set_param.size();          // (member-less columns come last: no perturbed set is built for them)
        for (int64_t c : no_member) {
            const int32_t dst = (int32_t)(dest_idx ? dest_idx[c] : c);
            cdest.push_back(dst); zero_dest.push_back(dst);
        }
        std::sort(zero_dest.begin(), zero_dest.end());
        const int32_t n_pw = (int32_t)wk.size();
        // wave tables: kind | obj | n_eff | col0 | ncols (n_pw each), then col_dest (n_param), then the zero-fill list
        std::vector<int32_t> tab;
        for (const auto* v : {&wk, &wo, &wn, &w0, &wc, &cdest, &zero_dest}) tab.insert(tab.end(), v->begin(), v->end());
        if ((rc = upload_i32(p, p->d_lb_waves, tab))) return rc;
        p->lb_n_zero = (int32_t)zero_dest.size();
        HIP_TRY(p->d_lb_setparam.ensure(set_param.size()));
        H2D_TRY(p, p->d_lb_setparam.p, set_param.data(), set_param.size() * 8);
        HIP_TRY(p->d_lb_pert.ensure((size_t)n_param * D * D));
        // work items: (dirty program of (task, the wavefront's class), wavefront), expensive first; empty programs -- the task
        // never shows the member to an outcome -- are no items at all
        const int nC = p->dirty.n_classes;
        std::vector<std::pair<int64_t, std::pair<uint32_t, int32_t>>> items;
        items.reserve((size_t)nT * std::max(n_pw, 1));
        for (int64_t t = 0; t < nT; t++)
            for (int32_t w = 0; w < n_pw; w++) {
                const int cls = wk[(size_t)w] == GST_KIND_GATE ? wo[(size_t)w] : h.n_gates + wo[(size_t)w];
                const size_t pi = (size_t)t * nC + cls;
                if (p->dirty.off[pi + 1] == p->dirty.off[pi]) continue;
                items.push_back({-((int64_t)p->dirty.applies[pi] + p->dirty.emits[pi] / 2 + 8), {(uint32_t)pi, w}});
            }
        std::stable_sort(items.begin(), items.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        std::vector<uint32_t> iprog(items.size());
        std::vector<int32_t> ipw(items.size());
        for (size_t i = 0; i < items.size(); i++) { iprog[i] = items[i].second.first; ipw[i] = items[i].second.second; }
        HIP_TRY(p->d_block_order.ensure(iprog.size() + 1));
        if (!iprog.empty()) H2D_TRY(p, p->d_block_order.p, iprog.data(), iprog.size() * 4);
        if ((rc = upload_i32(p, p->d_lb_item_pw, ipw))) return rc;
        p->lbr_n_waves = (int32_t)(rl_col.size() / 64);
        if (p->lbr_n_waves > 0) {
            if ((rc = upload_i32(p, p->d_lbr_lane[0], rl_col)) || (rc = upload_i32(p, p->d_lbr_lane[1], rl_kind)) ||
                (rc = upload_i32(p, p->d_lbr_lane[2], rl_obj)) || (rc = upload_i32(p, p->d_lbr_lane[3], rl_elem))) return rc;
            // (task, wavefront) pairs, longest programs first: every pair walks its whole task
            std::vector<int64_t> order((size_t)nT);
            for (int64_t t = 0; t < nT; t++) order[(size_t)t] = t;
            std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return h.task_off[(size_t)x + 1] - h.task_off[(size_t)x] > h.task_off[(size_t)y + 1] - h.task_off[(size_t)y]; });
            std::vector<uint32_t> bo;
            bo.reserve((size_t)nT * p->lbr_n_waves);
            for (int64_t t : order) for (int32_t w = 0; w < p->lbr_n_waves; w++) bo.push_back((uint32_t)(t * p->lbr_n_waves + w));
            HIP_TRY(p->d_lbr_order.ensure(bo.size() + 1));
            H2D_TRY(p, p->d_lbr_order.p, bo.data(), bo.size() * 4);
        }
        HIP_TRY(hipStreamSynchronize(p->stream));          // the host vectors go out of scope
        p->lb_n_pwaves = n_pw;
        p->lb_n_items = (int64_t)items.size();
        p->remember_request(5, param_idx, dest_idx, n_param);
    }
    // the changed member of every column
    gst::LbArgs b;
    lb_args(p, b);
    b.set_param = p->d_lb_setparam.p; b.sets = p->d_lb_pert.p; b.set_stride = (int64_t)D * D; b.member_only = 1; b.eps = eps;
    HIP_TRY(gst::launch_lindblad_build(D, b, p->lb_n_sets, p->stream));
    gst::PertArgs a;
    std::memset(&a, 0, sizeof(a));
    a.prog = p->d_dirty_words.p; a.prog_off = p->d_dirty_off.p;
    a.item_prog = p->d_block_order.p; a.item_pw = p->d_lb_item_pw.p;
    a.eff_ptr = p->d_eff_ptr.p; a.eff_label = p->d_eff_label.p; a.eff_dest = p->d_eff_dest.p;
    a.gates_t = p->d_gates_t.p; a.rhos = p->d_rhos.p; a.effects = p->d_effects.p;
    a.n_gates = h.n_gates; a.n_effects = h.n_effects;
    a.base_cache = p->d_base_cache.p; a.pbase = d_base;
    a.pert = p->d_lb_pert.p; a.pert_stride = (int64_t)D * D;
    const int32_t npw = p->lb_n_pwaves;
    const int32_t* t = p->d_lb_waves.p;
    a.wave_kind = t; a.wave_obj = t + npw; a.wave_neff = t + 2 * npw; a.wave_col0 = t + 3 * npw; a.wave_ncols = t + 4 * npw;
    a.col_dest = t + 5 * npw;
    a.n_pwaves = npw;
    a.out = d_out; a.ld = ld; a.eps = eps;
    TIME_REC(p, evk0);
    HIP_TRY(gst::launch_zero_columns(d_out, ld, h.n_elements, a.col_dest + n_param, p->lb_n_zero, -1, p->stream));
    HIP_TRY(gst::launch_walk_pert(D, a, p->lb_n_items, h.max_slots, p->stream));
    if (p->lbr_n_waves > 0) {          // preparation columns: the lane-per-model FD kernel, each lane starting from its own vector
        gst::WalkArgs w;
        base_args(p, w);
        w.mode = gst::EMIT_FD;
        w.out = d_out; w.ld = ld; w.eps = eps; w.pbase = d_base; w.base_cache = p->d_base_cache.p;
        w.lanes.col = p->d_lbr_lane[0].p; w.lanes.kind[0] = p->d_lbr_lane[1].p; w.lanes.obj[0] = p->d_lbr_lane[2].p; w.lanes.elem[0] = p->d_lbr_lane[3].p;
        w.n_pwaves = p->lbr_n_waves;
        w.block_order = p->d_lbr_order.p;
        w.rho_models = p->d_lb_pert.p; w.rho_model_stride = (int64_t)D * D;
        HIP_TRY(gst::launch_walk(D, 1, w, nT, h.max_slots, p->stream, 1));
        p->last_launches++;
    }
    for (size_t k = 0; k + 3 < p->lb_povm_cols.size(); k += 4)
        HIP_TRY(gst::launch_effect_columns(D, a, p->d_circ_leaf.p, h.n_circuits, p->lb_povm_cols[k], p->lb_povm_cols[k + 1],
                                           p->lb_povm_cols[k + 2], p->lb_povm_cols[k + 3], p->stream));
    TIME_REC(p, evk1);
    p->last_launches += 3 + (int64_t)p->lb_povm_cols.size() / 4;
    return GST_OK;
}

// FD Jacobian columns of a Lindblad-parameterised model: the device builds the dense model after every parameter step
// (one workgroup per column) and walks every (program, model set) pair; base probabilities from the base model, which
// gst_set_lindblad_params built.
int run_dprobs_lindblad(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param,
                        double eps, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int64_t nE = h.n_elements, nT = h.n_tasks();
    if (!p->lb.have_theta) return fail(GST_ESTATE, "gst_set_lindblad_params has not been called");
    if (h.D == 64) return fail(GST_EUNSUPPORTED, "three-qubit Lindblad members: exact derivatives (GST_DERIV_ANALYTIC) only");
    if (!(eps != 0.0)) return fail(GST_EINVAL, "eps must be non-zero");
    for (int64_t c = 0; c < n_param; c++)
        if (param_idx[c] < 0 || param_idx[c] >= p->lb.n_params) return fail(GST_EINVAL, "parameter index out of range");
    if (p->lb_share && gst::pert_kernel_fits(h.D, h.n_gates, h.n_effects, h.max_slots) && h.n_gates <= 64)
        return run_dprobs_lindblad_shared(p, d_out, ld, param_idx, dest_idx, n_param, eps, d_probs_out);
    p->cached_kind = 0;
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    int rc = run_probs(p, d_base, false);
    if (rc || n_param == 0) return rc;
    const size_t stride = lb_set_stride(p);
    int64_t chunk = std::max<int64_t>(1, (int64_t)(2.0e9 / (8.0 * (double)std::max<int64_t>(nE, 1))));
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, 0x7fffffffLL / std::max<int64_t>(nT, 1)));
    chunk = std::min<int64_t>(chunk, n_param);
    HIP_TRY(p->d_mm_models.ensure((size_t)chunk * stride));
    HIP_TRY(p->d_mm_raw.ensure((size_t)chunk * (size_t)std::max<int64_t>(nE, 1)));
    HIP_TRY(p->d_lb_setparam.ensure((size_t)n_param));
    H2D_TRY(p, p->d_lb_setparam.p, param_idx, (size_t)n_param * 8);
    std::vector<int32_t> dest32;
    if (dest_idx) {
        dest32.resize((size_t)n_param);
        for (int64_t m = 0; m < n_param; m++) {
            if (dest_idx[m] < 0 || dest_idx[m] >= ld) return fail(GST_EINVAL, "destination column out of range");
            dest32[(size_t)m] = (int32_t)dest_idx[m];
        }
        if ((rc = upload_i32(p, p->d_mm_dest, dest32))) return rc;
    } else if (n_param > ld) return fail(GST_EINVAL, "more columns than the leading dimension");
    HIP_TRY(hipStreamSynchronize(p->stream));              // (param_idx / dest32 are the caller's / a local)
    TIME_REC(p, evk0);
    for (int64_t m0 = 0; m0 < n_param; m0 += chunk) {
        const int64_t nm = std::min<int64_t>(chunk, n_param - m0);
        gst::LbArgs a;
        lb_args(p, a);
        a.set_param = p->d_lb_setparam.p + m0;
        a.sets = p->d_mm_models.p;
        a.eps = eps;
        HIP_TRY(gst::launch_lindblad_build(h.D, a, nm, p->stream));
        p->last_launches++;
        if ((rc = run_models_chunk(p, nm, m0, d_base, d_out, ld, dest_idx ? p->d_mm_dest.p + m0 : nullptr, eps))) return rc;
    }
    TIME_REC(p, evk1);
    return GST_OK;
}

// GST_DERIV_ANALYTIC for a Lindblad-parameterised model: the members' d(dense)/d(parameter) matrices are computed ON THE
// DEVICE (lindblad_deriv_kernel: Frechet derivative of the exponential, composed with the static factor) into the
// buffers gst_set_derivs would have filled from the host's deriv_wrt_params(), then the ordinary chain rule runs.
int run_dprobs_lindblad_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                                 int64_t n_param, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    gst_plan::Lindblad& L = p->lb;
    if (!L.have_theta) return fail(GST_ESTATE, "gst_set_lindblad_params has not been called");
    // objects in gst_set_derivs' terms: a gate / preparation member is one object, a POVM member one per effect
    std::vector<int32_t> kind, obj, ncols;
    std::vector<int64_t> pidx, member_off((size_t)L.n_members, 0), set_param;
    int64_t doff = 0;
    for (int m = 0; m < L.n_members; m++) {
        const int np = L.n_par[(size_t)m];
        member_off[(size_t)m] = doff;
        const int reps = L.kind[(size_t)m] == GST_KIND_EFFECT ? L.n_eff[(size_t)m] : 1;
        for (int e = 0; e < reps; e++) {
            kind.push_back(L.kind[(size_t)m]); obj.push_back(L.obj[(size_t)m] + e); ncols.push_back(np);
            for (int q = 0; q < np; q++) pidx.push_back(L.param0[(size_t)m] + q);
            doff += (int64_t)(L.kind[(size_t)m] == GST_KIND_GATE ? D * D : D) * np;
        }
        for (int q = 0; q < np; q++) set_param.push_back(L.param0[(size_t)m] + q);
    }
    const int32_t n_objs = (int32_t)kind.size();
    p->dv_kind = kind; p->dv_obj = obj; p->dv_ncols = ncols; p->dv_param_idx = pidx;
    p->dv_off_cols.assign((size_t)n_objs + 1, 0); p->dv_off_deriv.assign((size_t)n_objs + 1, 0);
    for (int32_t o = 0; o < n_objs; o++) {
        p->dv_off_cols[(size_t)o + 1] = p->dv_off_cols[(size_t)o] + ncols[(size_t)o];
        p->dv_off_deriv[(size_t)o + 1] = p->dv_off_deriv[(size_t)o] + (int64_t)(kind[(size_t)o] == GST_KIND_GATE ? D * D : D) * ncols[(size_t)o];
    }
    p->dv_n_params = L.n_params;
    p->dv_deriv_h.clear();                 // (no host copy: exact Hessian blocks of Lindblad models still come through gst_set_derivs)
    p->dv2_set = false; p->dv2_off.clear();
    HIP_TRY(p->d_dv_deriv.ensure((size_t)std::max<int64_t>(doff, 1)));
    HIP_TRY(p->d_lb_setparam.ensure(set_param.size() + member_off.size()));
    H2D_TRY(p, p->d_lb_setparam.p, set_param.data(), set_param.size() * 8);
    H2D_TRY(p, p->d_lb_setparam.p + set_param.size(), member_off.data(), member_off.size() * 8);
    gst::LbArgs a;
    lb_args(p, a);
    a.set_param = p->d_lb_setparam.p; a.deriv_out = p->d_dv_deriv.p; a.deriv_off = p->d_lb_setparam.p + set_param.size(); a.eps = 0.0;
    if (D == 64) HIP_TRY(gst::launch_lindblad64_derivs(a, p->d_lb_ws64.p, (int64_t)set_param.size(), p->stream));
    else HIP_TRY(gst::launch_lindblad_derivs(D, a, (int64_t)set_param.size(), p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));          // (the host vectors above go out of scope)
    p->last_launches++;
    p->cached_kind = 0;
    return run_dprobs_general(p, d_out, ld, param_idx, dest_idx, n_param, d_probs_out);
}

}  // namespace gst_impl

extern "C" {

int gst_set_lindblad(gst_plan* p, int32_t n_params, int32_t n_members, const gst_lindblad_member* members, int64_t n_terms,
                     const double* term_re, const double* term_im)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    gst_plan::Lindblad& L = p->lb;
    // (the request tables of the last fill -- wave tables, set_param order, POVM columns, dirty items -- are derived from
    //  the members' kinds, objects and parameter ranges: a new description, or none, invalidates them)
    if (n_members == 0) { L = gst_plan::Lindblad(); p->cached_kind = 0; return GST_OK; }
    const int D = p->hp.D;
    if ((D != 4 && D != 16 && D != 64) || p->D_user) return fail(GST_EUNSUPPORTED, "Lindblad members are built on the device for D = 4, 16 and 64");
    if (n_members < 0 || n_params < 0 || !members || n_terms <= 0 || !term_re || !term_im) return fail(GST_EINVAL, "bad argument");
    gst_plan::Lindblad N;
    N.n_params = n_params; N.n_members = n_members;
    const int nb_max = D - 1;                      // Pauli basis of n qubits without the identity: 4^n - 1 = D - 1
    std::vector<uint8_t> gate_seen((size_t)p->hp.n_gates, 0), rho_seen((size_t)p->hp.n_rhos, 0), eff_seen((size_t)p->hp.n_effects, 0);
    for (int32_t m = 0; m < n_members; m++) {
        const gst_lindblad_member& M = members[m];
        const std::string who = "member " + std::to_string(m) + ": ";
        if (!M.static_part) return fail(GST_EINVAL, who + "static_part is NULL");
        if (M.n_blocks < 1 || M.n_blocks > gst::LB_MAX_BLOCKS) return fail(GST_EINVAL, who + "1.." + std::to_string(gst::LB_MAX_BLOCKS) + " coefficient blocks");
        int64_t np = 0, nc = 0;
        for (int b = 0; b < M.n_blocks; b++) {
            const int bt = M.block_type[b], md = M.block_mode[b], n = M.block_n[b];
            if (bt < 0 || bt > 2 || md < 0 || md > 1 || n < 1 || n > nb_max) return fail(GST_EINVAL, who + "bad coefficient block");
            np += bt == 2 ? (int64_t)n * n : n; nc += bt == 2 ? (int64_t)n * n : n;
        }
        // (D = 64: the kernels keep one member's parameters and coefficients in LDS -- a Hamiltonian block plus ONE 'other' block)
        if (np > (D == 64 ? 63 + 63 * 63 : gst::lb_max_coeffs(D))) return fail(GST_EUNSUPPORTED, who + "too many parameters for one member");
        if (M.param0 < 0 || M.param0 + np > n_params) return fail(GST_EINVAL, who + "parameter range outside the model's");
        if (M.term_offset < 0 || M.term_offset + nc > n_terms) return fail(GST_EINVAL, who + "term range outside the term table");
        size_t n_static = 0;
        if (M.kind == GST_KIND_GATE) {
            if (M.obj < 0 || M.obj >= p->hp.n_gates || gate_seen[(size_t)M.obj]++) return fail(GST_EINVAL, who + "bad or repeated gate index");
            n_static = (size_t)D * D;
        } else if (M.kind == GST_KIND_RHO) {
            if (M.obj < 0 || M.obj >= p->hp.n_rhos || rho_seen[(size_t)M.obj]++) return fail(GST_EINVAL, who + "bad or repeated state index");
            n_static = (size_t)D;
        } else if (M.kind == GST_KIND_EFFECT) {
            if (M.n_eff < 1 || M.obj < 0 || M.obj + M.n_eff > p->hp.n_effects) return fail(GST_EINVAL, who + "bad effect range");
            // (a perturbed member occupies one D*D slot of the per-column member sets: n_eff * D doubles must fit)
            if (M.n_eff > D) return fail(GST_EUNSUPPORTED, who + "a POVM member with more than D effects (over-complete POVM) is not built on the device");
            for (int e = M.obj; e < M.obj + M.n_eff; e++) if (eff_seen[(size_t)e]++) return fail(GST_EINVAL, who + "effect listed twice");
            n_static = (size_t)M.n_eff * D;
        } else return fail(GST_EINVAL, who + "unknown kind");
        N.kind.push_back(M.kind); N.obj.push_back(M.obj); N.n_eff.push_back(M.kind == GST_KIND_EFFECT ? M.n_eff : 1);
        N.n_par.push_back((int32_t)np); N.n_blocks.push_back(M.n_blocks);
        N.param0.push_back(M.param0); N.term_off.push_back(M.term_offset); N.static_off.push_back((int64_t)N.statics.size());
        N.statics.insert(N.statics.end(), M.static_part, M.static_part + n_static);
    }
    // a parameter belongs to exactly ONE member: the build / derivative kernels and the column tables step the member whose
    // range holds the column's parameter, so members that share an error generator (same gpindices for two gates) would
    // have only one of them stepped, where the reference's set_parameter_value moves both -- refuse, the caller then
    // takes the host-stepped dense-model route (gst_fill_dprobs_models)
    {
        std::vector<std::pair<int64_t, int64_t>> rng;
        for (int32_t m = 0; m < n_members; m++) if (N.n_par[(size_t)m] > 0) rng.emplace_back(N.param0[(size_t)m], N.param0[(size_t)m] + N.n_par[(size_t)m]);
        std::sort(rng.begin(), rng.end());
        for (size_t k = 1; k < rng.size(); k++)
            if (rng[k].first < rng[k - 1].second)
                return fail(GST_EUNSUPPORTED, "two Lindblad members share parameters (overlapping parameter ranges): not built on the device");
    }
    // every object of the plan must belong to a member: the device builds the WHOLE model
    for (uint8_t v : gate_seen) if (!v) return fail(GST_EINVAL, "a gate of the plan belongs to no Lindblad member");
    for (uint8_t v : rho_seen) if (!v) return fail(GST_EINVAL, "a state preparation of the plan belongs to no Lindblad member");
    for (uint8_t v : eff_seen) if (!v) return fail(GST_EINVAL, "an effect of the plan belongs to no Lindblad member");
    N.blk_type.assign((size_t)n_members * gst::LB_MAX_BLOCKS, 0); N.blk_mode = N.blk_type; N.blk_n = N.blk_type;
    for (int32_t m = 0; m < n_members; m++)
        for (int b = 0; b < members[m].n_blocks; b++) {
            N.blk_type[(size_t)m * gst::LB_MAX_BLOCKS + b] = members[m].block_type[b];
            N.blk_mode[(size_t)m * gst::LB_MAX_BLOCKS + b] = members[m].block_mode[b];
            N.blk_n[(size_t)m * gst::LB_MAX_BLOCKS + b] = members[m].block_n[b];
        }
    N.term_re.assign(term_re, term_re + (size_t)n_terms * D * D);
    N.term_im.assign(term_im, term_im + (size_t)n_terms * D * D);
    N.set = true;
    L = std::move(N);
    p->cached_kind = 0;
    return GST_OK;
    });
}

int gst_set_lindblad_params(gst_plan* p, const double* theta)
{
    return guarded([&]() -> int {
    if (!p || !theta) return fail(GST_EINVAL, "NULL argument");
    if (!p->lb.set) return fail(GST_ESTATE, "gst_set_lindblad has not been called");
    int rc = ensure_device(p);
    if (rc) return rc;
    if ((rc = lb_upload(p))) return rc;
    gst_plan::Lindblad& L = p->lb;
    L.theta.assign(theta, theta + L.n_params);
    H2D_TRY(p, p->d_lb_theta.p, L.theta.data(), (size_t)L.n_params * 8);
    gst::LbArgs a;
    lb_args(p, a);
    a.set_param = nullptr; a.sets = p->d_lb_base.p; a.gates_rowmajor = p->d_lb_gates_rm.p; a.eps = 0.0;
    if (p->hp.D == 64) {
        int max_np = 1;
        for (int32_t v : L.n_par) max_np = std::max(max_np, (int)v);
        HIP_TRY(gst::launch_lindblad64_build(a, p->d_lb_ws64.p, max_np, p->stream));
    } else
        HIP_TRY(gst::launch_lindblad_build(p->hp.D, a, L.n_members, p->stream));
    // the base model also becomes the plan's model (what gst_set_model would have been given): 13 KB back over PCIe
    const int D = p->hp.D;
    const size_t ng = (size_t)p->hp.n_gates * D * D, nr = (size_t)p->hp.n_rhos * D, ne = (size_t)p->hp.n_effects * D;
    std::vector<double> set(ng + nr + ne);
    p->h_gates.resize(ng);
    int rc2 = d2h_bytes(p, set.data(), p->d_lb_base.p, set.size() * 8);
    if (!rc2 && ng) rc2 = d2h_bytes(p, p->h_gates.data(), p->d_lb_gates_rm.p, ng * 8);
    if (rc2) return rc2;
    p->h_gates_t.assign(set.begin(), set.begin() + (long)ng);
    p->h_rhos.assign(set.begin() + (long)ng, set.begin() + (long)(ng + nr));
    p->h_effects.assign(set.begin() + (long)(ng + nr), set.end());
    p->have_model = true;
    p->model_dirty = true;
    L.have_theta = true;
    return GST_OK;
    });
}

int gst_get_lindblad_model_sets(gst_plan* p, const int64_t* param_idx, int64_t n_param, double eps, double* gates, double* rhos, double* effects)
{
    return guarded([&]() -> int {
    if (!p || n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad argument");
    if (!p->lb.set || !p->lb.have_theta) return fail(GST_ESTATE, "gst_set_lindblad / gst_set_lindblad_params have not been called");
    int rc = ensure_device(p);
    if (rc) return rc;
    for (int64_t c = 0; c < n_param; c++)
        if (param_idx[c] < 0 || param_idx[c] >= p->lb.n_params) return fail(GST_EINVAL, "parameter index out of range");
    if (n_param == 0) return GST_OK;
    const int D = p->hp.D;
    if (D == 64) return fail(GST_EUNSUPPORTED, "perturbed model sets of three-qubit Lindblad members are not built (exact derivatives only)");
    const size_t ng = (size_t)p->hp.n_gates * D * D, nr = (size_t)p->hp.n_rhos * D, ne = (size_t)p->hp.n_effects * D, stride = ng + nr + ne;
    HIP_TRY(p->d_mm_models.ensure((size_t)n_param * stride));
    HIP_TRY(p->d_lb_setparam.ensure((size_t)n_param));
    H2D_TRY(p, p->d_lb_setparam.p, param_idx, (size_t)n_param * 8);
    gst::LbArgs a;
    lb_args(p, a);
    a.set_param = p->d_lb_setparam.p; a.sets = p->d_mm_models.p; a.eps = eps;
    HIP_TRY(gst::launch_lindblad_build(D, a, n_param, p->stream));
    std::vector<double> h((size_t)n_param * stride);
    {
        int rc2 = d2h_bytes(p, h.data(), p->d_mm_models.p, h.size() * 8);
        if (rc2) return rc2;
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int64_t m = 0; m < n_param; m++) {
        const double* s = h.data() + (size_t)m * stride;
        if (gates)
            for (int g = 0; g < p->hp.n_gates; g++)
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) gates[(size_t)m * ng + ((size_t)g * D + i) * D + j] = s[((size_t)g * D + j) * D + i];
        if (rhos) std::memcpy(rhos + (size_t)m * nr, s + ng, nr * 8);
        if (effects) std::memcpy(effects + (size_t)m * ne, s + ng + nr, ne * 8);
    }
    return GST_OK;
    });
}

}  // extern "C"
