// gst_kernels_composite.hip -- the layer operations of IMPLICIT models built on the device.
//
// pyGSTi's implicit models (LocalNoiseModel, CloudNoiseModel: `create_crosstalk_free_model` ...) do not store a dense
// superoperator per circuit layer: a layer is a ComposedOp of EmbeddedOps, each embedding a small operation (one or two
// qubits) into the full register (opcreps.cpp:93-158 `OpCRep_Embedded::acton`, :242-276 `OpCRep_Composed::acton`;
// modelmembers/operations/embeddedop.py, composedop.py), and the SAME small operation usually sits behind several layers
// (`independent_gates=False`: one Gxpi2 for every qubit) -- its parameters are shared.  The reference's Map path steps a
// parameter with set_parameter_value and re-propagates through those reps factor by factor; a dense-gate device path had
// to ask the host for to_dense() of every layer after every step.
//
// Here the host describes the structure ONCE (gst_set_composite: leaves, factors = (leaf, target qubits), layers = ordered
// factor lists) and then only sends the leaves' few numbers per model update (gst_set_composite_values).  The device
//   * builds every dense layer  G = Emb(f_{n-1}) ... Emb(f_1) Emb(f_0)     (composite_build_kernel, base model),
//   * builds the complete dense model after EVERY finite-difference step (one model set per requested column: the leaf
//     element(s) of that parameter moved by eps, every layer that contains the leaf rebuilt, SPAM elements moved through
//     the plan's parameter map) straight into the layout the whole-model walk reads,
//   * and, for the exact route, every layer's d(dense)/d(parameter) by the product rule
//         dG/dq = sum over factors k whose leaf holds q of  Emb(f_{n-1}) .. Emb(dF_k) .. Emb(f_0)
//     into the buffers gst_set_derivs would have filled from the host's deriv_wrt_params() (composite_deriv_kernel).
//
// In the Pauli-product basis the register's state index is a base-4 number, one digit per qubit (qubit 0 most significant),
// and embedding on target qubits is a Kronecker product with identities on the others:
//     Emb(f)[i][j] = f[digits_T(i)][digits_T(j)]  if i and j agree outside T, else 0
// so a factor is never expanded: Emb(f) . X costs D * D * dim(f) multiply-adds.  One workgroup per (model set, layer),
// two D x D matrices ping-pong in LDS.  Not a throughput kernel (a model has a dozen layers); plain fp64, no FMA
// contraction (the library's build flag), agreement with the host's dense matrices to rounding.
#include "gst_kernels.hpp"

#include "../../include/gstfwd.h"

namespace gst {

namespace {

struct Digits {
    int nq, nt, t[3];
    __device__ __forceinline__ int leaf_index(int i) const
    {
        int li = 0;
        for (int k = 0; k < nt; k++) li = li * 4 + ((i >> (2 * (nq - 1 - t[k]))) & 3);
        return li;
    }
    // i with its target digits replaced by the digits of leaf index a
    __device__ __forceinline__ int with_leaf_index(int i, int a) const
    {
        for (int k = nt - 1; k >= 0; k--) {
            const int sh = 2 * (nq - 1 - t[k]);
            i = (i & ~(3 << sh)) | ((a & 3) << sh);
            a >>= 2;
        }
        return i;
    }
};

__device__ __forceinline__ Digits digits_of(const CompositeArgs& a, int factor)
{
    Digits d;
    d.nq = a.nq; d.nt = 0;
    for (int k = 0; k < 3; k++) { d.t[k] = a.factor_targets[factor * 3 + k]; if (d.t[k] >= 0) d.nt = k + 1; }
    return d;
}

// out = Emb(f) . in   (f: dl x dl row-major in LDS; in / out: D x D row-major in LDS; all threads)
__device__ __forceinline__ void apply_factor(const Digits& dg, const double* f, int dl, const double* in, double* out, int D, int t, int nt_threads)
{
    for (int idx = t; idx < D * D; idx += nt_threads) {
        const int i = idx / D, j = idx - i * D;
        const double* frow = f + dg.leaf_index(i) * dl;
        double s = 0.0;
        for (int b = 0; b < dl; b++) s += frow[b] * in[dg.with_leaf_index(i, b) * D + j];
        out[idx] = s;
    }
}

// GENERAL leaves (leaf_np[l] > 0: any member the host can densify -- exponentiated Lindblad generators, ...): position of
// parameter q in the leaf's ascending parameter list, or -1.  Wave-uniform.
__device__ __forceinline__ int general_leaf_col(const CompositeArgs& a, int l, int64_t q)
{
    const int np = a.leaf_np ? a.leaf_np[l] : 0;
    const int64_t* lst = a.leaf_plist + (np ? a.leaf_plist_off[l] : 0);
    int lo = 0, hi = np;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (lst[mid] < q) lo = mid + 1; else hi = mid; }
    return (lo < np && lst[lo] == q) ? lo : -1;
}

// Does leaf l move with parameter q?  (all threads; contains a barrier)
__device__ __forceinline__ bool leaf_holds(const CompositeArgs& a, int l, int64_t q, int t, int nt_threads)
{
    if (a.leaf_np && a.leaf_np[l] > 0) return general_leaf_col(a, l, q) >= 0;
    const int dl = a.leaf_dim[l];
    const int64_t off = a.leaf_off[l];
    int mine = 0;
    for (int e = t; e < dl * dl; e += nt_threads) mine |= (a.leaf_param[off + e] == q);
    return __syncthreads_or(mine) != 0;
}

// The (possibly stepped) values of leaf l into LDS.  Element leaves: f[e] = value[e] + (param[e] == q ? eps : 0).
// General leaves: the host's own perturbed leaf for q (leaf_fd: [np][dl*dl], the reference's to_dense() after
// set_parameter_value(q, theta_q + eps)), the base values otherwise.
__device__ __forceinline__ void load_leaf(const CompositeArgs& a, int l, int64_t q, double eps, double* f, int t, int nt_threads)
{
    const int dl = a.leaf_dim[l];
    const int64_t off = a.leaf_off[l];
    if (a.leaf_np && a.leaf_np[l] > 0) {
        const int c = q >= 0 ? general_leaf_col(a, l, q) : -1;
        const double* src = (c >= 0 && a.leaf_fd) ? a.leaf_fd + a.leaf_fd_off[l] + (int64_t)c * dl * dl : a.leaf_values + off;
        for (int e = t; e < dl * dl; e += nt_threads) f[e] = src[e];
        return;
    }
    for (int e = t; e < dl * dl; e += nt_threads) {
        double v = a.leaf_values[off + e];
        if (q >= 0 && a.leaf_param[off + e] == q) v = v + eps;
        f[e] = v;
    }
}

// The direction of leaf l for parameter q: element leaves dF[e] = 1 where param[e] == q, else 0; general leaves column q of
// the host's deriv_wrt_params() of the leaf (leaf_deriv: [dl*dl][np] row-major).
__device__ __forceinline__ void load_leaf_direction(const CompositeArgs& a, int l, int64_t q, double* f, int t, int nt_threads)
{
    const int dl = a.leaf_dim[l];
    const int64_t off = a.leaf_off[l];
    if (a.leaf_np && a.leaf_np[l] > 0) {
        const int np = a.leaf_np[l], c = general_leaf_col(a, l, q);
        const double* dm = a.leaf_deriv + a.leaf_deriv_off[l];
        for (int e = t; e < dl * dl; e += nt_threads) f[e] = c >= 0 ? dm[(int64_t)e * np + c] : 0.0;
        return;
    }
    for (int e = t; e < dl * dl; e += nt_threads) f[e] = (a.leaf_param[off + e] == q) ? 1.0 : 0.0;
}

__device__ __forceinline__ void set_identity(double* X, int D, int t, int nt_threads)
{
    for (int idx = t; idx < D * D; idx += nt_threads) X[idx] = (idx / D == idx % D) ? 1.0 : 0.0;
}

}  // namespace

// grid (n_sets, n_gates + 1): block (m, g) builds layer g of model set m (g == n_gates: the set's SPAM vectors).
__global__ __launch_bounds__(256) void composite_build_kernel(const CompositeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int D = a.D, t = threadIdx.x, NT = blockDim.x;
    const int64_t m = blockIdx.x;
    const int g = blockIdx.y;
    const int64_t q = a.set_param ? a.set_param[m] : -1;
    double* set = a.sets + m * a.set_stride;
    const size_t ng = (size_t)a.n_gates * D * D, nr = (size_t)a.n_rhos * D, ne = (size_t)a.n_effects * D;
    if (g == a.n_gates) {
        // SPAM: the base vectors, one element moved when q is a preparation / effect parameter of the plan's map
        for (size_t k = t; k < nr + ne; k += NT) {
            double v = k < nr ? a.rhos[k] : a.effects[k - nr];
            if (q >= 0 && a.pkind) {
                const int kd = a.pkind[q];
                if (kd == GST_KIND_RHO && k < nr && (size_t)a.pobj[q] * D + a.pelem[q] == k) v = v + a.eps;
                if (kd == GST_KIND_EFFECT && k >= nr && (size_t)a.pobj[q] * D + a.pelem[q] == k - nr) v = v + a.eps;
            }
            set[ng + k] = v;
        }
        return;
    }
    const int f0 = a.gate_fptr[g], f1 = a.gate_fptr[g + 1];
    // does this layer move with q?  (no: the base layer is copied)
    if (q >= 0 && a.base_set) {
        bool mine = false;
        for (int f = f0; f < f1; f++) mine = leaf_holds(a, a.factor_leaf[f], q, t, NT) || mine;
        if (!mine) {
            for (int idx = t; idx < D * D; idx += NT) set[(size_t)g * D * D + idx] = a.base_set[(size_t)g * D * D + idx];
            return;
        }
    }
    double* X = lds;
    double* Y = lds + D * D;
    double* F = lds + 2 * D * D;
    set_identity(X, D, t, NT);
    __syncthreads();
    for (int f = f0; f < f1; f++) {
        const int l = a.factor_leaf[f];
        load_leaf(a, l, q, a.eps, F, t, NT);
        __syncthreads();
        apply_factor(digits_of(a, f), F, a.leaf_dim[l], X, Y, D, t, NT);
        __syncthreads();
        double* tmp = X; X = Y; Y = tmp;
    }
    // the walk kernels read gates transposed: gates_t[g][j][i] = G[i][j]
    for (int idx = t; idx < D * D; idx += NT) {
        const int i = idx / D, j = idx - i * D;
        set[(size_t)g * D * D + (size_t)j * D + i] = X[idx];
        if (a.gates_rowmajor) a.gates_rowmajor[(size_t)g * D * D + idx] = X[idx];
    }
}

// grid (n_items): item = (layer g, column c of its derivative matrix, parameter q): column c of the row-major
// [D*D][ncols] matrix at deriv_out + gate_doff[g].
__global__ __launch_bounds__(256) void composite_deriv_kernel(const CompositeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int D = a.D, t = threadIdx.x, NT = blockDim.x;
    const int item = blockIdx.x;
    const int g = a.item_gate[item], c = a.item_col[item];
    const int64_t q = a.item_param[item];
    const int ncols = a.gate_ncols[g];
    double* out = a.deriv_out + a.gate_doff[g];
    const int f0 = a.gate_fptr[g], f1 = a.gate_fptr[g + 1];
    double* X = lds;
    double* Y = lds + D * D;
    double* F = lds + 2 * D * D;
    constexpr int MAXR = 16;                       // D * D / 256 at D = 64
    double acc[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) acc[r] = 0.0;
    for (int k = f0; k < f1; k++) {
        // does factor k's leaf hold q?
        if (!leaf_holds(a, a.factor_leaf[k], q, t, NT)) continue;
        set_identity(X, D, t, NT);
        __syncthreads();
        for (int f = f0; f < f1; f++) {
            const int l = a.factor_leaf[f];
            if (f == k) load_leaf_direction(a, l, q, F, t, NT);
            else load_leaf(a, l, -1, 0.0, F, t, NT);
            __syncthreads();
            apply_factor(digits_of(a, f), F, a.leaf_dim[l], X, Y, D, t, NT);
            __syncthreads();
            double* tmp = X; X = Y; Y = tmp;
        }
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            const int idx = t + r * NT;
            if (idx < D * D) acc[r] += X[idx];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < MAXR; r++) {
        const int idx = t + r * NT;
        if (idx < D * D) out[(size_t)idx * ncols + c] = acc[r];
    }
}

static size_t composite_lds_bytes(const CompositeArgs& a) { return (size_t)(2 * a.D * a.D + a.max_leaf_dim * a.max_leaf_dim) * sizeof(double); }

static hipError_t composite_allow_lds(const void* kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    if (bytes > 160 * 1024) return hipErrorInvalidValue;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_composite_build(const CompositeArgs& a, int64_t n_sets, hipStream_t stream)
{
    if (n_sets <= 0) return hipSuccess;
    if (n_sets > 0x7fffffffLL || a.n_gates + 1 > 65535 || a.D * a.D > 16 * 256) return hipErrorInvalidValue;
    (void)hipGetLastError();
    const size_t lds = composite_lds_bytes(a);
    hipError_t e = composite_allow_lds((const void*)composite_build_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(composite_build_kernel, dim3((unsigned)n_sets, (unsigned)(a.n_gates + 1)), dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_composite_derivs(const CompositeArgs& a, int64_t n_items, hipStream_t stream)
{
    if (n_items <= 0) return hipSuccess;
    if (n_items > 0x7fffffffLL || a.D * a.D > 16 * 256 || !a.item_gate || !a.deriv_out) return hipErrorInvalidValue;
    (void)hipGetLastError();
    const size_t lds = composite_lds_bytes(a);
    hipError_t e = composite_allow_lds((const void*)composite_deriv_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(composite_deriv_kernel, dim3((unsigned)n_items), dim3(256), lds, stream, a);
    return hipGetLastError();
}

}  // namespace gst
