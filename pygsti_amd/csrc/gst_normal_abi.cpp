// gst_normal_abi.cpp -- the objective's element-wise maps and the normal equations on a resident Jacobian
// (objectivefns.py:4573-4665, distlayout.py:1220-1359).
#include "gst_state.hpp"

using namespace gst_impl;

extern "C" {

int gst_fill_jtj_dev(gst_plan* p, double* d_J, int64_t n_rows, int64_t n_cols, int64_t ld, const double* d_row_scale,
                     double* d_jtj)
{
    return guarded([&]() -> int {
    if (!p || !d_J || !d_jtj || n_rows < 0 || n_cols < 0 || ld < n_cols || n_cols > 0x7fffffff) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    gst::track_touch(d_jtj, (size_t)n_cols * n_cols * 8);
    if (d_row_scale && n_rows > 0 && n_cols > 0) {
        // the in-place row scaling keeps an exact Jacobian's zeros zero -- unless a factor is not finite (0 * inf): the
        // claim's device word is cleared on the stream then, and the next exact fill stores everything (gst_track.cpp)
        // (the claim's word lives on its OWNER's stream: a scaling issued through another plan is not ordered against the
        //  owner's next contraction, so it simply ends the claim)
        bool several = false;
        uint64_t owner = 0;
        const size_t bytes = (size_t)((n_rows - 1) * ld + n_cols) * 8;
        uint32_t* w = gst::track_claim_overlapping(d_J, bytes, &several, &owner);
        if (several || (w && owner != p->uid)) gst::track_touch(d_J, bytes);
        else if (w) HIP_TRY(gst::launch_check_finite(d_row_scale, n_rows, w, p->stream));
    }
    TIME_REC(p, ev0);
    // Block sparsity (a row is exactly zero in the columns of gates its circuit never applies): one streaming pass marks,
    // per 16-row panel, the 128-column tiles that hold anything -- fused with the row scaling when there is one -- and
    // the product skips every (panel, tile pair) with an empty side.  Worth the pass from ~16 K rows x 4 tiles on.
    const bool sparse = p->jtj_sparse && n_rows >= 16384 && gst::jtj_mask_tiles((int)n_cols) >= 4 && gst::jtj_mask_tiles((int)n_cols) <= 32;
    const uint32_t* d_pmask = nullptr;
    if (sparse) {
        HIP_TRY(p->d_jtj_pmask.ensure((size_t)gst::jtj_mask_words(n_rows)));
        HIP_TRY(gst::launch_jtj_panel_masks(d_J, n_rows, (int)n_cols, ld, d_row_scale, p->d_jtj_pmask.p, p->stream));
        d_pmask = p->d_jtj_pmask.p;
    } else if (d_row_scale && n_rows > 0 && n_cols > 0) HIP_TRY(gst::launch_scale_rows(d_J, n_rows, n_cols, ld, d_row_scale, p->stream));
    if (n_cols > 0) {
        const int n_slabs = gst::jtj_num_slabs(n_rows, (int)n_cols, p->n_cus);
        HIP_TRY(p->d_jtj_part.ensure((size_t)n_slabs * n_cols * n_cols));
        TIME_REC(p, evk0);
        HIP_TRY(gst::launch_jtj(d_J, n_rows, (int)n_cols, ld, p->d_jtj_part.p, n_slabs, d_jtj, p->stream, d_pmask));
        TIME_REC(p, evk1);
    }
    TIME_REC(p, ev1);
    return GST_OK;
    });
}

int gst_fill_jtf_dev(gst_plan* p, const double* d_J, int64_t n_rows, int64_t n_cols, int64_t ld, const double* d_f,
                     double* d_jtf)
{
    return guarded([&]() -> int {
    if (!p || !d_J || !d_f || !d_jtf || n_rows < 0 || n_cols < 0 || ld < n_cols || n_cols > 0x7fffffff) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    if (n_cols == 0) return GST_OK;
    const int n_slabs = gst::jtf_num_slabs(n_rows);
    gst::track_touch(d_jtf, (size_t)n_cols * 8);
    HIP_TRY(p->d_jtf_part.ensure((size_t)n_slabs * n_cols));
    HIP_TRY(gst::launch_jtf(d_J, d_f, n_rows, (int)n_cols, ld, p->d_jtf_part.p, n_slabs, d_jtf, p->stream));
    return GST_OK;
    });
}

int gst_fill_normal_eqs_dev(gst_plan* p, const double* d_J, int64_t n_rows, int64_t n_cols, int64_t ld,
                            const double* d_row_scale, const double* d_f, double* d_jtj, double* d_jtf)
{
    return guarded([&]() -> int {
    if (!p || !d_J || n_rows < 0 || n_cols < 0 || ld < n_cols || n_cols > 0x7fffffff) return fail(GST_EINVAL, "bad argument");
    if (!d_jtj && !d_jtf) return fail(GST_EINVAL, "nothing to compute: d_jtj and d_jtf are both NULL");
    if (d_jtf && !d_f) return fail(GST_EINVAL, "d_jtf needs d_f");
    int rc = ensure_device(p);
    if (rc) return rc;
    if (n_cols == 0) return GST_OK;
    TIME_REC(p, ev0);
    bool jtf_done = false;
    if (d_jtj) {
        gst::track_touch(d_jtj, (size_t)n_cols * n_cols * 8);
        // d_J is only read: no claim on it changes (its resident zeros stay zeros whatever the weights are)
        const bool sparse = p->jtj_sparse && n_rows >= 16384 && gst::jtj_mask_tiles((int)n_cols) >= 4 && gst::jtj_mask_tiles((int)n_cols) <= 32;
        const uint32_t* d_pmask = nullptr;
        if (sparse) {
            HIP_TRY(p->d_jtj_pmask.ensure((size_t)gst::jtj_mask_words(n_rows)));
            if (d_jtf) {
                // one read of J for the masks and J_s^T f together (the sums are carried through ranges of panels: another
                // summation order than gst_fill_jtf_dev's, equally deterministic)
                const int n_ranges = gst::jtj_mask_jtf_ranges(n_rows, (int)n_cols);
                gst::track_touch(d_jtf, (size_t)n_cols * 8);
                HIP_TRY(p->d_jtf_part.ensure((size_t)n_ranges * n_cols));
                HIP_TRY(gst::launch_jtj_mask_jtf(d_J, n_rows, (int)n_cols, ld, d_row_scale, d_f, p->d_jtj_pmask.p, p->d_jtf_part.p, n_ranges,
                                                 d_jtf, p->stream));
                jtf_done = true;
            } else
                HIP_TRY(gst::launch_jtj_panel_masks(const_cast<double*>(d_J), n_rows, (int)n_cols, ld, d_row_scale, p->d_jtj_pmask.p, p->stream,
                                                    /*write_back=*/false));
            d_pmask = p->d_jtj_pmask.p;
        }
        const int n_slabs = gst::jtj_num_slabs(n_rows, (int)n_cols, p->n_cus);
        HIP_TRY(p->d_jtj_part.ensure((size_t)n_slabs * n_cols * n_cols));
        TIME_REC(p, evk0);
        HIP_TRY(gst::launch_jtj(d_J, n_rows, (int)n_cols, ld, p->d_jtj_part.p, n_slabs, d_jtj, p->stream, d_pmask, d_row_scale));
        TIME_REC(p, evk1);
    }
    if (d_jtf && !jtf_done) {
        const int n_slabs = gst::jtf_num_slabs(n_rows);
        gst::track_touch(d_jtf, (size_t)n_cols * 8);
        HIP_TRY(p->d_jtf_part.ensure((size_t)n_slabs * n_cols));
        HIP_TRY(gst::launch_jtf(d_J, d_f, n_rows, (int)n_cols, ld, p->d_jtf_part.p, n_slabs, d_jtf, p->stream, d_row_scale));
    }
    TIME_REC(p, ev1);
    return GST_OK;
    });
}

int gst_objective_rows_dev(gst_plan* p, const gst_objective_desc* d, double* d_probs, const double* d_counts,
                           const double* d_totals, int64_t n, double* d_lsvec, double* d_rowscale, double* d_terms,
                           double* sum_terms)
{
    return guarded([&]() -> int {
    if (!p || !d || !d_probs || !d_counts || !d_totals || !d_lsvec || !d_rowscale || n < 0) return fail(GST_EINVAL, "bad argument");
    if (d->kind != GST_OBJ_CHI2 && d->kind != GST_OBJ_POISSON_DLOGL) return fail(GST_EINVAL, "unknown objective kind");
    if (!(d->min_prob_clip > 0.0) || (d->kind == GST_OBJ_POISSON_DLOGL && !(d->radius > 0.0)))
        return fail(GST_EINVAL, "min_prob_clip and radius must be positive");
    int rc = ensure_device(p);
    if (rc) return rc;
    if (sum_terms) *sum_terms = 0.0;
    if (n == 0) return GST_OK;
    for (double* w : {d_probs, d_lsvec, d_rowscale, d_terms}) gst::track_touch(w, (size_t)n * 8);
    const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + 255) / 256));
    HIP_TRY(p->d_obj_part.ensure((size_t)n_blocks));
    HIP_TRY(gst::launch_objective_rows(d->kind, d_probs, d_counts, d_totals, n, d->min_prob_clip, d->radius, d->prob_clip_lo,
                                       d->prob_clip_hi, d_lsvec, d_rowscale, d_terms, p->d_obj_part.p, n_blocks, p->stream));
    if (sum_terms) {
        std::vector<double> part((size_t)n_blocks);
        if ((rc = d2h_bytes(p, part.data(), p->d_obj_part.p, part.size() * 8))) return rc;
        double s = 0.0;
        for (double x : part) s += x;
        *sum_terms = s;
    }
    return GST_OK;
    });
}


// One Levenberg-Marquardt evaluation in ONE call: model upload -> finite-difference Jacobian of all parameters -> objective rows
// -> J_s^T J_s and J_s^T lsvec, blocking, with sum(terms) returned -- what `Plan.lsq_step` composes from four calls.  For
// LAUNCH-BOUND plans (<= 65,536 states: the 1Q designs, where a fill is microseconds of work behind as many microseconds of
// launches) the whole sequence is captured into a HIP graph at the second call with the same arguments and replayed from then
// on: one graph launch per LM iteration instead of seven kernel launches, two memsets and two copies.  The graph reads the model
// from a page-locked buffer of its own that every call overwrites first; its kernels are the ordinary ones (bit-identical
// results).  Anything that changes what the sequence looks like -- another request, other pointers, a new parameter map --
// drops the graph; a capture the runtime refuses is not retried (the ordinary path serves the plan).
int gst_lm_step_dev(gst_plan* p, const gst_objective_desc* d, int64_t n_params, double eps, const double* d_counts,
                    const double* d_totals, double* d_J, int64_t ld, double* d_probs, double* d_lsvec, double* d_rowscale,
                    double* d_jtj, double* d_jtf, double* sum_terms)
{
    return guarded([&]() -> int {
    if (!p || !d || !d_counts || !d_totals || !d_J || !d_probs || !d_lsvec || !d_rowscale || !d_jtj || !d_jtf || n_params <= 0 || ld < n_params)
        return fail(GST_EINVAL, "bad argument");
    if (d->kind != GST_OBJ_CHI2 && d->kind != GST_OBJ_POISSON_DLOGL) return fail(GST_EINVAL, "unknown objective kind");
    if (!(d->min_prob_clip > 0.0) || (d->kind == GST_OBJ_POISSON_DLOGL && !(d->radius > 0.0)))
        return fail(GST_EINVAL, "min_prob_clip and radius must be positive");
    int rc = ensure_device(p);
    if (rc) return rc;
    if (!p->have_model) return fail(GST_ESTATE, "gst_set_model has not been called");
    if (!p->have_pmap || p->lb.set || p->cmp.set || p->derivs_set) return fail(GST_EUNSUPPORTED, "gst_lm_step_dev serves element-mapped (`full`, full TP) models");
    if (n_params != (int64_t)p->pkind.size()) return fail(GST_EINVAL, "n_params is not the parameter map's size");
    const int64_t nE = p->hp.n_elements;
    if (nE == 0) { if (sum_terms) *sum_terms = 0.0; return GST_OK; }
    std::vector<int64_t> pidx((size_t)n_params);
    for (int64_t q = 0; q < n_params; q++) pidx[(size_t)q] = q;
    const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (nE + 255) / 256));
    HIP_TRY(p->d_obj_part.ensure((size_t)n_blocks));
    gst_plan::LmGraph& G = p->lm_graph;
    if (!G.h_part) HIP_TRY(hipHostMalloc((void**)&G.h_part, 1024 * sizeof(double), hipHostMallocDefault));
    const size_t ng = p->h_gates.size(), nr = p->h_rhos.size(), ne = p->h_effects.size(), n_model = 2 * ng + nr + ne;
    // the sequence on p->stream; `from_graph_buffer`: the model comes from G.h_model (captured) instead of upload_model's pair
    auto sequence = [&](bool from_graph_buffer) -> int {
        int rcs;
        if (from_graph_buffer) {
            HIP_TRY(hipMemcpyAsync(p->d_model.p, G.h_model, n_model * 8, hipMemcpyHostToDevice, p->stream));
        } else if ((rcs = upload_model(p))) return rcs;
        if ((rcs = run_dprobs_fd(p, d_J, ld, pidx.data(), nullptr, n_params, eps, d_probs, nullptr, 0))) return rcs;
        HIP_TRY(gst::launch_objective_rows(d->kind, d_probs, d_counts, d_totals, nE, d->min_prob_clip, d->radius, d->prob_clip_lo,
                                           d->prob_clip_hi, d_lsvec, d_rowscale, nullptr, p->d_obj_part.p, n_blocks, p->stream));
        HIP_TRY(hipMemcpyAsync(G.h_part, p->d_obj_part.p, (size_t)n_blocks * 8, hipMemcpyDeviceToHost, p->stream));
        return gst_fill_normal_eqs_dev(p, d_J, nE, n_params, ld, d_rowscale, d_lsvec, d_jtj, d_jtf);
    };
    auto fill_graph_model = [&]() {
        if (ng) { std::memcpy(G.h_model, p->h_gates.data(), ng * 8); std::memcpy(G.h_model + ng, p->h_gates_t.data(), ng * 8); }
        std::memcpy(G.h_model + 2 * ng, p->h_rhos.data(), nr * 8);
        std::memcpy(G.h_model + 2 * ng + nr, p->h_effects.data(), ne * 8);
    };
    for (double* w : {d_J, d_probs, d_lsvec, d_rowscale, d_jtj, d_jtf}) gst::track_touch(w, 8);      // (claims on these destinations end: FD results)
    const uint64_t sig = (uint64_t)(uintptr_t)d_J ^ ((uint64_t)(uintptr_t)d_probs << 1) ^ ((uint64_t)(uintptr_t)d_lsvec << 2) ^ ((uint64_t)(uintptr_t)d_rowscale << 3) ^
                         ((uint64_t)(uintptr_t)d_jtj << 4) ^ ((uint64_t)(uintptr_t)d_jtf << 5) ^ ((uint64_t)(uintptr_t)d_counts << 6) ^ ((uint64_t)(uintptr_t)d_totals << 7) ^
                         ((uint64_t)ld * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)d->kind << 60) ^ (uint64_t)n_model;
    const bool small = p->lm_graph_enabled && p->hp.n_state_ids <= 65536 && p->hp.D <= 16 && p->comp_index < 0;
    const bool same = G.exec && G.sig == sig && G.eps == eps && std::memcmp(&G.desc, d, sizeof(*d)) == 0 && G.request_serial == p->fd_request_serial &&
                      p->request_cached(1, pidx.data(), nullptr, n_params);
    p->last_launches = 0;
    if (small && same) {
        fill_graph_model();
        p->model_dirty = true;                        // (the ordinary upload pair no longer mirrors the device)
        HIP_TRY(hipGraphLaunch(G.exec, p->stream));
        G.replays++;
    } else {
        if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
        const bool saved_timing = p->timing;
        bool captured = false;
        // capture at the second call with these arguments (the first one allocates, packs and uploads the request's tables)
        if (small && !G.failed && G.warm_sig == sig && p->request_cached(1, pidx.data(), nullptr, n_params) && p->d_model.p) {
            if (!G.h_model || G.h_model_n < n_model) {
                if (G.h_model) (void)hipHostFree(G.h_model);
                G.h_model = nullptr;
                HIP_TRY(hipHostMalloc((void**)&G.h_model, std::max<size_t>(n_model, 1) * 8, hipHostMallocDefault));
                G.h_model_n = n_model;
            }
            fill_graph_model();
            HIP_TRY(hipStreamSynchronize(p->stream));
            p->timing = false;                        // (no event records inside the graph)
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal);
            int rcs = GST_OK;
            if (e == hipSuccess) {
                rcs = sequence(true);
                e = hipStreamEndCapture(p->stream, &graph);
            }
            p->timing = saved_timing;
            if (e == hipSuccess && rcs == GST_OK && graph) {
                e = hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (e == hipSuccess) {
                    G.sig = sig; G.eps = eps; G.desc = *d; G.request_serial = p->fd_request_serial; G.n_blocks = n_blocks;
                    p->model_dirty = true;
                    HIP_TRY(hipGraphLaunch(G.exec, p->stream));
                    G.replays = 1;
                    captured = true;
                } else G.exec = nullptr;
            } else if (graph) (void)hipGraphDestroy(graph);
            if (!captured) { (void)hipGetLastError(); G.failed = true; p->model_dirty = true; }
        }
        if (!captured) {
            G.warm_sig = sig;
            if ((rc = sequence(false))) return rc;
        }
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (sum_terms) {
        double sacc = 0.0;
        for (int k = 0; k < n_blocks; k++) sacc += G.h_part[k];
        *sum_terms = sacc;
    }
    return GST_OK;
    });
}

}  // extern "C"
