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
    const int n_slabs = (int)std::max<int64_t>(1, std::min<int64_t>(256, (n_rows + 255) / 256));
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
        const int n_slabs = (int)std::max<int64_t>(1, std::min<int64_t>(256, (n_rows + 255) / 256));
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

}  // extern "C"
