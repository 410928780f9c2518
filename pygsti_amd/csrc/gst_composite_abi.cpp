// gst_composite_abi.cpp -- layer operations of implicit models on the device (gst_set_composite): the description, the
// base model per update of the leaves, FD Jacobians over device-built model sets and exact ones through device-built
// derivative matrices + the chain rule (opcreps.cpp:93-158, 242-276; embeddedop.py, composedop.py: deriv_wrt_params).
#include "gst_state.hpp"

using namespace gst_impl;

namespace gst_impl {

static size_t cmp_set_stride(const gst_plan* p)
{
    const int D = p->hp.D;
    return (size_t)p->hp.n_gates * D * D + (size_t)p->hp.n_rhos * D + (size_t)p->hp.n_effects * D;
}

static int cmp_upload(gst_plan* p)
{
    gst_plan::Composite& C = p->cmp;
    if (C.uploaded) return GST_OK;
    std::vector<int32_t> i32;
    i32.insert(i32.end(), C.leaf_dim.begin(), C.leaf_dim.end());
    i32.insert(i32.end(), C.gate_fptr.begin(), C.gate_fptr.end());
    i32.insert(i32.end(), C.factor_leaf.begin(), C.factor_leaf.end());
    i32.insert(i32.end(), C.factor_targets.begin(), C.factor_targets.end());
    i32.insert(i32.end(), C.leaf_np.begin(), C.leaf_np.end());
    int rc = upload_i32(p, p->d_cmp_i32, i32);
    if (rc) return rc;
    std::vector<int64_t> i64;
    i64.insert(i64.end(), C.leaf_off.begin(), C.leaf_off.end());
    i64.insert(i64.end(), C.leaf_param.begin(), C.leaf_param.end());
    i64.insert(i64.end(), C.leaf_plist_off.begin(), C.leaf_plist_off.end());
    i64.insert(i64.end(), C.leaf_deriv_off.begin(), C.leaf_deriv_off.end());
    i64.insert(i64.end(), C.leaf_fd_off.begin(), C.leaf_fd_off.end());
    i64.insert(i64.end(), C.leaf_plist.begin(), C.leaf_plist.end());
    HIP_TRY(p->d_cmp_i64.ensure(std::max<size_t>(i64.size(), 1)));
    H2D_TRY(p, p->d_cmp_i64.p, i64.data(), i64.size() * 8);
    HIP_TRY(p->d_cmp_values.ensure(std::max<size_t>(C.leaf_param.size(), 1)));
    HIP_TRY(p->d_cmp_gderiv.ensure((size_t)std::max<int64_t>(C.n_deriv_doubles, 1)));
    HIP_TRY(p->d_cmp_gfd.ensure((size_t)std::max<int64_t>(C.n_fd_doubles, 1)));
    HIP_TRY(p->d_cmp_spam.ensure(std::max<size_t>((size_t)(p->hp.n_rhos + p->hp.n_effects) * p->hp.D, 1)));
    HIP_TRY(p->d_cmp_base.ensure(cmp_set_stride(p)));
    HIP_TRY(p->d_cmp_gates_rm.ensure(std::max<size_t>((size_t)p->hp.n_gates * p->hp.D * p->hp.D, 1)));
    HIP_TRY(hipStreamSynchronize(p->stream));
    C.uploaded = true;
    return GST_OK;
}

static void cmp_args(gst_plan* p, gst::CompositeArgs& a)
{
    const gst_plan::Composite& C = p->cmp;
    const int D = p->hp.D;
    a = gst::CompositeArgs();
    a.D = D; a.nq = D == 4 ? 1 : D == 16 ? 2 : 3;
    a.n_gates = p->hp.n_gates; a.n_rhos = p->hp.n_rhos; a.n_effects = p->hp.n_effects;
    a.n_leaves = C.n_leaves; a.max_leaf_dim = C.max_leaf_dim;
    const int32_t* q = p->d_cmp_i32.p;
    a.leaf_dim = q; q += C.leaf_dim.size();
    a.gate_fptr = q; q += C.gate_fptr.size();
    a.factor_leaf = q; q += C.factor_leaf.size();
    a.factor_targets = q; q += C.factor_targets.size();
    a.leaf_off = p->d_cmp_i64.p;
    a.leaf_param = p->d_cmp_i64.p + C.leaf_off.size();
    if (C.any_general) {
        const int64_t* w = a.leaf_param + C.leaf_param.size();
        a.leaf_np = q;
        a.leaf_plist_off = w; w += C.leaf_plist_off.size();
        a.leaf_deriv_off = w; w += C.leaf_deriv_off.size();
        a.leaf_fd_off = w; w += C.leaf_fd_off.size();
        a.leaf_plist = w;
        a.leaf_deriv = p->d_cmp_gderiv.p;
        a.leaf_fd = C.have_general_fd ? p->d_cmp_gfd.p : nullptr;
    }
    a.leaf_values = p->d_cmp_values.p;
    a.rhos = p->d_cmp_spam.p;
    a.effects = p->d_cmp_spam.p + (size_t)p->hp.n_rhos * D;
    a.set_stride = (int64_t)cmp_set_stride(p);
}

// The plan's parameter map (SPAM columns) on the device for the builder; parameters of the leaves carry GST_KIND_NONE there.
static int cmp_upload_pmap(gst_plan* p, gst::CompositeArgs& a)
{
    if (!p->have_pmap || (int64_t)p->pkind.size() != p->cmp.n_params) return GST_OK;      // (refused earlier: cmp_require_pmap)
    std::vector<int32_t> v;
    v.insert(v.end(), p->pkind.begin(), p->pkind.end());
    v.insert(v.end(), p->pobj.begin(), p->pobj.end());
    v.insert(v.end(), p->pelem.begin(), p->pelem.end());
    int rc = upload_i32(p, p->d_cmp_pmap, v);
    if (rc) return rc;
    const size_t n = p->pkind.size();
    a.pkind = p->d_cmp_pmap.p; a.pobj = p->d_cmp_pmap.p + n; a.pelem = p->d_cmp_pmap.p + 2 * n;
    return GST_OK;
}

// The SPAM parameters of an implicit model are described by the plan's parameter map (GST_KIND_NONE for the leaves' own and
// for parameters of objects the atom never applies).  Without a map of the model's size the SPAM columns would silently come
// out as zeros: refuse instead.
static int cmp_require_pmap(const gst_plan* p)
{
    if (p->have_pmap && (int64_t)p->pkind.size() == p->cmp.n_params) return GST_OK;
    return fail(GST_ESTATE, "gst_set_composite is active: gst_set_param_map must describe the model's " + std::to_string(p->cmp.n_params) +
                            " parameters (SPAM elements; GST_KIND_NONE elsewhere) before a Jacobian is requested");
}

// FD Jacobian columns of an implicit model: the device builds the complete dense model after every parameter step and walks
// every (program, model set) pair (the whole-model walk of gst_fill_dprobs_models, fed from device memory).
int run_dprobs_composite(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param,
                         double eps, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int64_t nE = h.n_elements, nT = h.n_tasks();
    if (!p->cmp.have_values) return fail(GST_ESTATE, "gst_set_composite_values has not been called");
    if (int rcp = cmp_require_pmap(p)) return rcp;
    if (!(eps != 0.0)) return fail(GST_EINVAL, "eps must be non-zero");
    if (p->cmp.any_general && (!p->cmp.have_general_fd || p->cmp.general_fd_eps != eps))
        return fail(GST_ESTATE, "general leaves: gst_set_composite_general must supply the leaves' finite-difference values for this eps");
    for (int64_t c = 0; c < n_param; c++)
        if (param_idx[c] < 0 || param_idx[c] >= p->cmp.n_params) return fail(GST_EINVAL, "parameter index out of range");
    p->cached_kind = 0;
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    int rc = run_probs(p, d_base, false);
    if (rc || n_param == 0) return rc;
    const size_t stride = cmp_set_stride(p);
    int64_t chunk = std::max<int64_t>(1, (int64_t)(2.0e9 / (8.0 * (double)std::max<int64_t>(nE, 1))));
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, 0x7fffffffLL / std::max<int64_t>(nT, 1)));
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, (int64_t)(4.0e9 / (8.0 * (double)stride))));       // <= 4 GB of model sets
    chunk = std::min<int64_t>(chunk, n_param);
    HIP_TRY(p->d_mm_models.ensure((size_t)chunk * stride));
    HIP_TRY(p->d_mm_raw.ensure((size_t)chunk * (size_t)std::max<int64_t>(nE, 1)));
    HIP_TRY(p->d_cmp_setparam.ensure((size_t)n_param));
    H2D_TRY(p, p->d_cmp_setparam.p, param_idx, (size_t)n_param * 8);
    std::vector<int32_t> dest32;
    if (dest_idx) {
        dest32.resize((size_t)n_param);
        for (int64_t m = 0; m < n_param; m++) {
            if (dest_idx[m] < 0 || dest_idx[m] >= ld) return fail(GST_EINVAL, "destination column out of range");
            dest32[(size_t)m] = (int32_t)dest_idx[m];
        }
        if ((rc = upload_i32(p, p->d_mm_dest, dest32))) return rc;
    } else if (n_param > ld) return fail(GST_EINVAL, "more columns than the leading dimension");
    gst::CompositeArgs a;
    cmp_args(p, a);
    if ((rc = cmp_upload_pmap(p, a))) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));              // (param_idx / dest32 are the caller's / a local)
    TIME_REC(p, evk0);
    for (int64_t m0 = 0; m0 < n_param; m0 += chunk) {
        const int64_t nm = std::min<int64_t>(chunk, n_param - m0);
        a.set_param = p->d_cmp_setparam.p + m0;
        a.base_set = p->d_cmp_base.p;
        a.sets = p->d_mm_models.p;
        a.eps = eps;
        HIP_TRY(gst::launch_composite_build(a, nm, p->stream));
        p->last_launches++;
        if ((rc = run_models_chunk(p, nm, m0, d_base, d_out, ld, dest_idx ? p->d_mm_dest.p + m0 : nullptr, eps))) return rc;
    }
    TIME_REC(p, evk1);
    return GST_OK;
}

// GST_DERIV_ANALYTIC for an implicit model: every layer's d(dense)/d(parameter) matrix is computed ON THE DEVICE (product
// rule over its factors) into the buffers gst_set_derivs would have filled from the host's deriv_wrt_params(); the SPAM
// objects of the plan's parameter map get their selection matrices; then the ordinary chain rule runs.
int run_dprobs_composite_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                                  int64_t n_param, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    gst_plan::Composite& C = p->cmp;
    if (!C.have_values) return fail(GST_ESTATE, "gst_set_composite_values has not been called");
    if (int rcp = cmp_require_pmap(p)) return rcp;
    if (C.any_general && !C.have_general_derivs)
        return fail(GST_ESTATE, "general leaves: gst_set_composite_general must supply the leaves' derivative matrices");
    std::vector<int32_t> kind, obj, ncols;
    std::vector<int64_t> pidx;
    std::vector<double> spam_deriv;                       // the SPAM objects' matrices (host-built, tiny), in object order
    std::vector<int64_t> spam_off;                        // offset of each SPAM object's matrix in d_dv_deriv
    // SPAM objects first (parameters that are dense elements of a preparation / effect)
    int64_t doff = 0;
    if (p->have_pmap && (int64_t)p->pkind.size() == C.n_params) {
        for (int k : {GST_KIND_RHO, GST_KIND_EFFECT}) {
            const int nobj = k == GST_KIND_RHO ? h.n_rhos : h.n_effects;
            for (int o = 0; o < nobj; o++) {
                std::vector<int64_t> mine;
                for (int64_t q = 0; q < C.n_params; q++)
                    if (p->pkind[(size_t)q] == k && p->pobj[(size_t)q] == o) mine.push_back(q);
                if (mine.empty()) continue;
                kind.push_back(k); obj.push_back(o); ncols.push_back((int32_t)mine.size());
                spam_off.push_back(doff);
                const size_t base = spam_deriv.size();
                spam_deriv.resize(base + (size_t)D * mine.size(), 0.0);
                for (size_t c = 0; c < mine.size(); c++) {
                    pidx.push_back(mine[c]);
                    spam_deriv[base + (size_t)p->pelem[(size_t)mine[c]] * mine.size() + c] = 1.0;
                }
                doff += (int64_t)D * (int64_t)mine.size();
            }
        }
    }
    const size_t n_spam_objs = kind.size();
    // one object per layer that has parameters: its columns are the distinct parameters of its factors' leaves
    std::vector<int64_t> gate_doff((size_t)h.n_gates, 0);
    std::vector<int32_t> gate_ncols((size_t)h.n_gates, 0), item_gate, item_col;
    std::vector<int64_t> item_param;
    for (int g = 0; g < h.n_gates; g++) {
        const std::vector<int64_t>& qs = C.gate_params[(size_t)g];
        if (qs.empty()) continue;
        kind.push_back(GST_KIND_GATE); obj.push_back(g); ncols.push_back((int32_t)qs.size());
        gate_doff[(size_t)g] = doff; gate_ncols[(size_t)g] = (int32_t)qs.size();
        for (size_t c = 0; c < qs.size(); c++) {
            pidx.push_back(qs[c]);
            item_gate.push_back(g); item_col.push_back((int32_t)c); item_param.push_back(qs[c]);
        }
        doff += (int64_t)D * D * (int64_t)qs.size();
    }
    const int32_t n_objs = (int32_t)kind.size();
    p->dv_kind = kind; p->dv_obj = obj; p->dv_ncols = ncols; p->dv_param_idx = pidx;
    p->dv_off_cols.assign((size_t)n_objs + 1, 0); p->dv_off_deriv.assign((size_t)n_objs + 1, 0);
    for (int32_t o = 0; o < n_objs; o++) {
        p->dv_off_cols[(size_t)o + 1] = p->dv_off_cols[(size_t)o] + ncols[(size_t)o];
        p->dv_off_deriv[(size_t)o + 1] = p->dv_off_deriv[(size_t)o] + (int64_t)(kind[(size_t)o] == GST_KIND_GATE ? D * D : D) * ncols[(size_t)o];
    }
    p->dv_n_params = C.n_params;
    p->dv_deriv_h.clear();                 // (no host copy: exact Hessian blocks of implicit models still come through gst_set_derivs)
    p->dv2_set = false; p->dv2_off.clear();
    HIP_TRY(p->d_dv_deriv.ensure((size_t)std::max<int64_t>(doff, 1)));
    for (size_t o = 0, at = 0; o < n_spam_objs; o++) {
        const size_t n = (size_t)D * (size_t)ncols[o];
        H2D_TRY(p, p->d_dv_deriv.p + spam_off[o], spam_deriv.data() + at, n * 8);
        at += n;
    }
    // item tables: [item_gate | item_col | gate_ncols] (int32) and [item_param | gate_doff] (int64)
    std::vector<int32_t> t32;
    t32.insert(t32.end(), item_gate.begin(), item_gate.end());
    t32.insert(t32.end(), item_col.begin(), item_col.end());
    t32.insert(t32.end(), gate_ncols.begin(), gate_ncols.end());
    int rc = upload_i32(p, p->d_cmp_items32, t32);
    if (rc) return rc;
    std::vector<int64_t> t64;
    t64.insert(t64.end(), item_param.begin(), item_param.end());
    t64.insert(t64.end(), gate_doff.begin(), gate_doff.end());
    HIP_TRY(p->d_cmp_setparam.ensure(std::max<size_t>(t64.size(), 1)));
    H2D_TRY(p, p->d_cmp_setparam.p, t64.data(), t64.size() * 8);
    gst::CompositeArgs a;
    cmp_args(p, a);
    const size_t ni = item_gate.size();
    a.item_gate = p->d_cmp_items32.p; a.item_col = p->d_cmp_items32.p + ni; a.gate_ncols = p->d_cmp_items32.p + 2 * ni;
    a.item_param = p->d_cmp_setparam.p; a.gate_doff = p->d_cmp_setparam.p + ni;
    a.deriv_out = p->d_dv_deriv.p;
    HIP_TRY(gst::launch_composite_derivs(a, (int64_t)ni, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));          // (the host vectors above go out of scope)
    p->last_launches++;
    p->cached_kind = 0;
    return run_dprobs_general(p, d_out, ld, param_idx, dest_idx, n_param, d_probs_out);
}

}  // namespace gst_impl

extern "C" {

int gst_set_composite(gst_plan* p, int32_t n_params, const gst_composite_desc* d)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    gst_plan::Composite& C = p->cmp;
    if (!d || d->n_leaves == 0) { C = gst_plan::Composite(); p->cached_kind = 0; return GST_OK; }
    const int D = p->hp.D;
    const int nq = D == 4 ? 1 : D == 16 ? 2 : D == 64 ? 3 : 0;
    if (!nq || p->D_user) return fail(GST_EUNSUPPORTED, "composite layers exist for D = 4, 16 and 64 (registers of qubits)");
    if (n_params < 0 || d->n_leaves < 0 || !d->leaf_dim || !d->leaf_param || !d->gate_factor_ptr || !d->factor_leaf || !d->factor_targets)
        return fail(GST_EINVAL, "bad argument");
    gst_plan::Composite N;
    N.n_params = n_params; N.n_leaves = d->n_leaves;
    int64_t off = 0;
    for (int l = 0; l < d->n_leaves; l++) {
        const int dl = d->leaf_dim[l];
        if (dl != 4 && dl != 16 && dl != 64) return fail(GST_EINVAL, "leaf " + std::to_string(l) + ": dimension must be 4, 16 or 64");
        if (dl > D) return fail(GST_EINVAL, "leaf " + std::to_string(l) + " is larger than the register");
        N.leaf_dim.push_back(dl); N.leaf_off.push_back(off);
        N.max_leaf_dim = std::max(N.max_leaf_dim, dl);
        off += (int64_t)dl * dl;
    }
    N.leaf_param.assign(d->leaf_param, d->leaf_param + off);
    for (int64_t q : N.leaf_param)
        if (q < -1 || q >= n_params) return fail(GST_EINVAL, "a leaf element's parameter index is out of range");
    N.leaf_np.assign((size_t)d->n_leaves, 0);
    N.leaf_plist_off.assign((size_t)d->n_leaves, 0); N.leaf_deriv_off.assign((size_t)d->n_leaves, 0); N.leaf_fd_off.assign((size_t)d->n_leaves, 0);
    if (d->leaf_n_params) {
        int64_t lo = 0;
        for (int l = 0; l < d->n_leaves; l++) {
            const int np = d->leaf_n_params[l];
            if (np < 0) return fail(GST_EINVAL, "leaf " + std::to_string(l) + ": negative parameter count");
            if (np == 0) continue;
            if (!d->leaf_param_list) return fail(GST_EINVAL, "leaf_param_list is NULL");
            const int64_t dd = (int64_t)N.leaf_dim[(size_t)l] * N.leaf_dim[(size_t)l];
            for (int64_t e = 0; e < dd; e++)
                if (N.leaf_param[(size_t)(N.leaf_off[(size_t)l] + e)] != -1) return fail(GST_EINVAL, "leaf " + std::to_string(l) + ": a general leaf's elements carry no element parameters (leaf_param must be -1)");
            N.leaf_np[(size_t)l] = np; N.any_general = true;
            N.leaf_plist_off[(size_t)l] = lo; N.leaf_deriv_off[(size_t)l] = N.n_deriv_doubles; N.leaf_fd_off[(size_t)l] = N.n_fd_doubles;
            for (int c = 0; c < np; c++) {
                const int64_t q = d->leaf_param_list[lo + c];
                if (q < 0 || q >= n_params || (c > 0 && q <= d->leaf_param_list[lo + c - 1]))
                    return fail(GST_EINVAL, "leaf " + std::to_string(l) + ": parameter list must be ascending and in range");
                N.leaf_plist.push_back(q);
            }
            lo += np; N.n_deriv_doubles += dd * np; N.n_fd_doubles += dd * np;
        }
    }
    const int nG = p->hp.n_gates;
    N.gate_fptr.assign(d->gate_factor_ptr, d->gate_factor_ptr + nG + 1);
    if (N.gate_fptr[0] != 0) return fail(GST_EINVAL, "gate_factor_ptr[0] must be 0");
    for (int g = 0; g < nG; g++)
        if (N.gate_fptr[(size_t)g + 1] < N.gate_fptr[(size_t)g]) return fail(GST_EINVAL, "gate_factor_ptr must not decrease");
    const int nF = N.gate_fptr[(size_t)nG];
    N.factor_leaf.assign(d->factor_leaf, d->factor_leaf + nF);
    N.factor_targets.assign(d->factor_targets, d->factor_targets + (size_t)3 * nF);
    for (int f = 0; f < nF; f++) {
        const int l = N.factor_leaf[(size_t)f];
        if (l < 0 || l >= d->n_leaves) return fail(GST_EINVAL, "factor " + std::to_string(f) + ": leaf out of range");
        const int nt = N.leaf_dim[(size_t)l] == 4 ? 1 : N.leaf_dim[(size_t)l] == 16 ? 2 : 3;
        uint32_t seen = 0;
        for (int k = 0; k < 3; k++) {
            const int t = N.factor_targets[(size_t)3 * f + k];
            if (k < nt) {
                if (t < 0 || t >= nq || (seen >> t) & 1) return fail(GST_EINVAL, "factor " + std::to_string(f) + ": bad target qubits");
                seen |= 1u << t;
            } else if (t != -1) return fail(GST_EINVAL, "factor " + std::to_string(f) + ": more targets than the leaf has qubits");
        }
    }
    // the distinct parameters behind every layer, ascending (the columns of its derivative matrix)
    N.gate_params.assign((size_t)nG, {});
    for (int g = 0; g < nG; g++) {
        std::vector<int64_t>& qs = N.gate_params[(size_t)g];
        for (int f = N.gate_fptr[(size_t)g]; f < N.gate_fptr[(size_t)g + 1]; f++) {
            const int l = N.factor_leaf[(size_t)f];
            const int64_t o = N.leaf_off[(size_t)l], n = (int64_t)N.leaf_dim[(size_t)l] * N.leaf_dim[(size_t)l];
            for (int64_t e = 0; e < n; e++)
                if (N.leaf_param[(size_t)(o + e)] >= 0) qs.push_back(N.leaf_param[(size_t)(o + e)]);
            for (int c = 0; c < N.leaf_np[(size_t)l]; c++) qs.push_back(N.leaf_plist[(size_t)(N.leaf_plist_off[(size_t)l] + c)]);
        }
        std::sort(qs.begin(), qs.end());
        qs.erase(std::unique(qs.begin(), qs.end()), qs.end());
    }
    N.set = true;
    C = std::move(N);
    p->cached_kind = 0;
    return GST_OK;
    });
}

int gst_set_composite_values(gst_plan* p, const double* leaf_values, const double* rhos, const double* effects)
{
    return guarded([&]() -> int {
    if (!p || !leaf_values || !rhos || !effects) return fail(GST_EINVAL, "NULL argument");
    if (!p->cmp.set) return fail(GST_ESTATE, "gst_set_composite has not been called");
    int rc = ensure_device(p);
    if (rc) return rc;
    if ((rc = cmp_upload(p))) return rc;
    gst_plan::Composite& C = p->cmp;
    const int D = p->hp.D;
    const size_t ng = (size_t)p->hp.n_gates * D * D, nr = (size_t)p->hp.n_rhos * D, ne = (size_t)p->hp.n_effects * D;
    H2D_TRY(p, p->d_cmp_values.p, leaf_values, C.leaf_param.size() * 8);
    H2D_TRY(p, p->d_cmp_spam.p, rhos, nr * 8);
    H2D_TRY(p, p->d_cmp_spam.p + nr, effects, ne * 8);
    gst::CompositeArgs a;
    cmp_args(p, a);
    a.set_param = nullptr; a.base_set = nullptr; a.sets = p->d_cmp_base.p; a.gates_rowmajor = p->d_cmp_gates_rm.p; a.eps = 0.0;
    HIP_TRY(gst::launch_composite_build(a, 1, p->stream));
    // the base model also becomes the plan's model (what gst_set_model would have been given)
    std::vector<double> set(ng + nr + ne);
    p->h_gates.resize(ng);
    int rc2 = d2h_bytes(p, set.data(), p->d_cmp_base.p, set.size() * 8);
    if (!rc2 && ng) rc2 = d2h_bytes(p, p->h_gates.data(), p->d_cmp_gates_rm.p, ng * 8);
    if (rc2) return rc2;
    p->h_gates_t.assign(set.begin(), set.begin() + (long)ng);
    p->h_rhos.assign(set.begin() + (long)ng, set.begin() + (long)(ng + nr));
    p->h_effects.assign(set.begin() + (long)(ng + nr), set.end());
    p->have_model = true;
    p->model_dirty = true;
    C.have_values = true;
    C.have_general_derivs = false; C.have_general_fd = false;      // (they belong to the previous parameter vector)
    return GST_OK;
    });
}

int gst_set_composite_general(gst_plan* p, const double* leaf_derivs, const double* leaf_fd_values, double fd_eps)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    gst_plan::Composite& C = p->cmp;
    if (!C.set || !C.have_values) return fail(GST_ESTATE, "gst_set_composite / gst_set_composite_values have not been called");
    if (!C.any_general) return GST_OK;
    int rc = ensure_device(p);
    if (rc) return rc;
    if (leaf_derivs) {
        H2D_TRY(p, p->d_cmp_gderiv.p, leaf_derivs, (size_t)C.n_deriv_doubles * 8);
        C.have_general_derivs = true;
    }
    if (leaf_fd_values) {
        if (!(fd_eps != 0.0)) return fail(GST_EINVAL, "fd_eps must be non-zero");
        H2D_TRY(p, p->d_cmp_gfd.p, leaf_fd_values, (size_t)C.n_fd_doubles * 8);
        C.have_general_fd = true; C.general_fd_eps = fd_eps;
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->cached_kind = 0;
    return GST_OK;
    });
}

}  // extern "C"
