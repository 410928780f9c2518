// gst_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the dense-matrix forward simulator.
//
// One kernel family, `walk_kernel<D,S,WPS>` (WPS = wavefronts per SIMD the register budget is cut for): ONE WAVEFRONT interprets ONE walk program (a task: a
// sub-trie of circuit prefixes, see gst_plan.hpp) and its 64 LANES are 64 different perturbed models
// (64 columns of the Jacobian / of one Hessian row).  The program is wave-uniform, so
//   * control flow never diverges,
//   * gate / effect / rho coefficients are wave-uniform and come through the SCALAR cache into
//     SGPRs (s_load_dwordx*), feeding v_mul_f64 / v_add_f64 as scalar operands -- no LDS or
//     vector-memory traffic in the inner loop at all,
//   * each lane keeps its whole state vector (D f64) in VGPRs, plus `S` "special rows": the one
//     row of the one gate (or rho / effect vector) its parameter perturbs.  For the reference's
//     `full` parameterisation an FD step changes exactly one dense element, so the perturbed model
//     differs from the base model in one row of one object; a lane recomputes just that row.
//
// Arithmetic contract (bitwise parity with the reference, opcreps.cpp:40-54 / effectcreps.cpp:39-45):
// every dot product starts at 0.0 and adds products in ascending index order with SEPARATE multiply
// and add.  This file MUST be compiled with -ffp-contract=off and without fast-math; the build
// script checks the ISA for v_fma_f64 in the walk kernels.
//
// What this replaces in the reference: dm_mapfill_probs (mapforwardsim_calc_densitymx.pyx:194-287)
// and the (nP+1)-pass loop of mapfill_dprobs_atom (:349-381) / _mapfill_hprobs_atom
// (mapforwardsim.py:394-438): all passes run concurrently, one lane each.
#include "gst_kernels.hpp"

#include "../../include/gstfwd.h"

namespace gst {

// Constant address space: loads through it are always "invariant", so a wave-uniform address turns
// into an s_load (scalar cache) instead of a vector load.
#define GST_CONST __attribute__((address_space(4)))
typedef const GST_CONST double* cdouble_p;
typedef const GST_CONST uint32_t* cu32_p;
typedef const GST_CONST int32_t* ci32_p;
typedef const GST_CONST int64_t* ci64_p;

template <typename T>
__device__ __forceinline__ const GST_CONST T* as_const(const T* p)
{
    return (const GST_CONST T*)(p);
}

template <int D>
struct State {
    double x[D];
};

// o = M v with M given TRANSPOSED (Mt[j*D+i] = M[i][j]): column sweep, o += M[:,j]*v[j].
// Every o[i] still accumulates its products in ascending j from +0.0 with separate mul/add (the
// reference's order, opcreps.cpp:45-51), but the D accumulators are independent, so the VALU never
// waits on a dependent add, and one contiguous scalar load brings a whole column.
// `NS` lane-private special rows ride along as extra accumulators: r[s] = sp[s] . v.
template <int D, int NS>
__device__ __forceinline__ void matvec_t(cdouble_p __restrict__ Mt, const double (&v)[D], double (&o)[D],
                                         const double (&sp)[NS > 0 ? NS : 1][D], double (&r)[NS > 0 ? NS : 1])
{
#pragma unroll
    for (int i = 0; i < D; i++) o[i] = 0.0;
#pragma unroll
    for (int s = 0; s < NS; s++) r[s] = 0.0;
    // Software pipeline over columns: the scalar loads of column j+1 are issued before column j's
    // 2*D VALU instructions, so their latency hides under arithmetic (the scheduling barriers keep
    // hipcc from sinking the loads back next to their uses, where every s_load would be waited for
    // immediately).
    double cur[D], nxt[D];
#pragma unroll
    for (int i = 0; i < D; i++) cur[i] = Mt[i];
#pragma unroll
    for (int j = 0; j < D; j++) {
        // column j has landed; SMEM returns out of order, so lgkmcnt(0) is the only usable wait and it
        // must sit BEFORE the next column's loads are issued (0xC07F = lgkmcnt(0), vmcnt/expcnt untouched)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < D) {
#pragma unroll
            for (int i = 0; i < D; i++) nxt[i] = Mt[(j + 1) * D + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double vj = v[j];
#pragma unroll
        for (int i = 0; i < D; i++) o[i] = o[i] + cur[i] * vj;
#pragma unroll
        for (int s = 0; s < NS; s++) r[s] = r[s] + sp[s][j] * vj;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; i++) cur[i] = nxt[i];
    }
}

template <int D>
__device__ __forceinline__ double dot_c(cdouble_p __restrict__ e, const double (&v)[D])
{
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) acc = acc + e[j] * v[j];
    return acc;
}

template <int D>
__device__ __forceinline__ double dot_r(const double (&r)[D], const double (&v)[D])
{
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) acc = acc + r[j] * v[j];
    return acc;
}

template <int D, int S, int WPS>
__global__ __launch_bounds__(64, WPS) void walk_kernel(const WalkArgs a)
{
    // Save slots (states kept while the children of a branching trie node are walked) live in LDS:
    // slot s, component j, lane l at lds[(s*D + j)*64 + l] -- lane-consecutive 8-byte words, conflict
    // free.  The plan compiler caps the number of slots (gst_options.max_slots) so that 4 wavefronts
    // per SIMD fit in the CU's 160 KB.
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t bid = blockIdx.x;
    const int32_t pw = (int32_t)(bid % a.n_pwaves);
    const int64_t task = bid / a.n_pwaves;
    const int64_t q = (int64_t)pw * 64 + lane;

    cdouble_p gates_t = as_const(a.gates_t);
    cdouble_p rhos = as_const(a.rhos);
    cdouble_p effects = as_const(a.effects);
    cu32_p prog = as_const(a.prog);
    ci32_p eff_ptr = as_const(a.eff_ptr);
    ci32_p eff_label = as_const(a.eff_label);
    ci32_p eff_dest = as_const(a.eff_dest);

    // ---- per-lane specials ---------------------------------------------------------------------
    constexpr int S1 = S > 0 ? S : 1;
    int kind[S1], obj[S1], row[S1];
    double sp[S1][D];
    int32_t col = (S == 0) ? (lane == 0 ? 0 : -1) : a.lanes.col[q];
    uint64_t gate_mask[S1];
    bool eff_any[S1], rho_any[S1];
#pragma unroll
    for (int s = 0; s < S1; s++) { kind[s] = GST_KIND_NONE; obj[s] = 0; row[s] = -1; gate_mask[s] = 0; eff_any[s] = false; rho_any[s] = false; }
    if (S > 0) {
        int el[S1];
#pragma unroll
        for (int s = 0; s < S; s++) {
            kind[s] = a.lanes.kind[s][q];
            obj[s] = a.lanes.obj[s][q];
            el[s] = a.lanes.elem[s][q];
            const double* base = a.gates;
            int b = -1;
            if (kind[s] == GST_KIND_GATE) { row[s] = el[s] / D; b = el[s] % D; base = a.gates + ((int64_t)obj[s] * D + row[s]) * D; }
            else if (kind[s] == GST_KIND_RHO) { row[s] = 0; b = el[s]; base = a.rhos + (int64_t)obj[s] * D; }
            else if (kind[s] == GST_KIND_EFFECT) { row[s] = 0; b = el[s]; base = a.effects + (int64_t)obj[s] * D; }
#pragma unroll
            for (int j = 0; j < D; j++) {
                const double x = base[j];
                sp[s][j] = (j == b) ? x + a.eps : x;      // theta_p + eps, as set_parameter_value does
            }
            el[s] = b;
        }
        if (S == 2) {
            // both perturbations in the same row of the same object: one special row carries both
            // ((orig + eps) + eps when it is the very same element, as from_vector then the FD step do)
            const bool same = kind[0] >= 0 && kind[0] == kind[S1 - 1] && obj[0] == obj[S1 - 1] && row[0] == row[S1 - 1];
            if (same) {
#pragma unroll
                for (int j = 0; j < D; j++) sp[0][j] = (j == el[S1 - 1]) ? sp[0][j] + a.eps : sp[0][j];
                kind[S1 - 1] = GST_KIND_NONE;
            }
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (a.n_gates <= 64) {
                for (int g = 0; g < a.n_gates; g++)
                    if (__ballot(kind[s] == GST_KIND_GATE && obj[s] == g)) gate_mask[s] |= (1ull << g);
            } else {
                gate_mask[s] = __ballot(kind[s] == GST_KIND_GATE) ? ~0ull : 0ull;
            }
            eff_any[s] = __ballot(kind[s] == GST_KIND_EFFECT) != 0;
            rho_any[s] = __ballot(kind[s] == GST_KIND_RHO) != 0;
        }
    }

    // ---- hessian addressing -----------------------------------------------------------------------
    int64_t hrow = 0, hrowidx = 0, hcolidx = 0;
    if (S == 2) {
        hrow = a.wave_row[pw];
        hrowidx = a.wave_rowidx[pw];
        hcolidx = a.lane_colidx[q];
    }

    double v[D];
    double sp0[1][D];   // placeholder operand of the no-special mat-vec (never read)
#pragma unroll
    for (int j = 0; j < D; j++) sp0[0][j] = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) v[j] = 0.0;

    int64_t pc = as_const(a.task_off)[task];
    uint32_t wnext = prog[pc];
    for (;;) {
        const uint32_t w = wnext;
        const uint32_t op = GST_OP(w);
        const uint32_t arg = GST_ARG(w);
        if (op == GST_OP_END) break;
        pc++;
        wnext = prog[pc];   // prefetch the next instruction word (scalar load) under this one's work

        if (op == GST_OP_APPLY) {
            double o[D];
            cdouble_p Mt = gates_t + (int64_t)arg * D * D;
            bool hit = false;
#pragma unroll
            for (int s = 0; s < S; s++)
                hit = hit || ((arg < 64) ? ((gate_mask[s] >> arg) & 1ull) : (gate_mask[s] != 0));
            if (S > 0 && hit) {      // wave-uniform: some lane's parameter lives in this gate
                double r[S1];
                matvec_t<D, S>(Mt, v, o, sp, r);
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const bool mine = (kind[s] == GST_KIND_GATE) && (obj[s] == (int)arg);
#pragma unroll
                    for (int i = 0; i < D; i++) o[i] = (mine && row[s] == i) ? r[s] : o[i];
                }
            } else {
                double r[1];
                matvec_t<D, 0>(Mt, v, o, sp0, r);
            }
#pragma unroll
            for (int j = 0; j < D; j++) v[j] = o[j];
        } else if (op == GST_OP_EMIT) {
            const int32_t x0 = eff_ptr[arg], x1 = eff_ptr[arg + 1];
            for (int32_t x = x0; x < x1; x++) {
                const int32_t e = eff_label[x];
                const int64_t dest = eff_dest[x];
                double p = dot_c<D>(effects + (int64_t)e * D, v);
#pragma unroll
                for (int s = 0; s < S; s++) {
                    if (eff_any[s]) {
                        const double r = dot_r<D>(sp[s], v);
                        p = (kind[s] == GST_KIND_EFFECT && obj[s] == e) ? r : p;
                    }
                }
                if (a.mode == EMIT_PROBS) {
                    if (col >= 0) a.out[dest] = p;
                } else if (a.mode == EMIT_FD) {
                    if (col >= 0) {
                        const double pb = as_const(a.pbase)[dest];
                        a.out[dest * a.ld + col] = (p - pb) / a.eps;
                        if (a.raw) a.raw[dest * a.ldraw + col] = p;
                    }
                } else {
                    if (col >= 0) {
                        const double pr = as_const(a.prow)[dest * a.ldrow + hrowidx];
                        const double d2 = (p - pr) / a.eps;                       // dprobs2 (mapforwardsim.py:431)
                        const double d1 = a.dcol[dest * a.lddcol + hcolidx];      // dprobs
                        a.out[(dest * a.ld + hrow) * a.ld2 + col] = (d2 - d1) / a.eps;
                    }
                }
            }
        } else if (op == GST_OP_SAVE) {
            double* sl = lds + (int64_t)arg * D * 64 + lane;
#pragma unroll
            for (int j = 0; j < D; j++) sl[j * 64] = v[j];
        } else if (op == GST_OP_LOAD) {
            const double* sl = lds + (int64_t)arg * D * 64 + lane;
#pragma unroll
            for (int j = 0; j < D; j++) v[j] = sl[j * 64];
        } else {  // GST_OP_RHO
            cdouble_p r0 = rhos + (int64_t)arg * D;
#pragma unroll
            for (int j = 0; j < D; j++) v[j] = r0[j];
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (rho_any[s]) {
                    const bool mine = (kind[s] == GST_KIND_RHO) && (obj[s] == (int)arg);
#pragma unroll
                    for (int j = 0; j < D; j++) v[j] = mine ? sp[s][j] : v[j];
                }
            }
        }
    }
}

template <int D, int S, int WPS>
static hipError_t launch_one(const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    const int64_t blocks = n_tasks * (int64_t)a.n_pwaves;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)n_slots * D * 64 * sizeof(double);
    if (lds_bytes > 64 * 1024) return hipErrorInvalidValue;
    (void)hipGetLastError();   // drop any stale sticky error (e.g. an event query) so that we report OUR launch
    hipLaunchKernelGGL((walk_kernel<D, S, WPS>), dim3((unsigned)blocks), dim3(64), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_walk(int D, int S, const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    if (D == 4) {
        if (S == 0) return launch_one<4, 0, 4>(a, n_tasks, n_slots, stream);
        if (S == 1) return launch_one<4, 1, 4>(a, n_tasks, n_slots, stream);
        if (S == 2) return launch_one<4, 2, 4>(a, n_tasks, n_slots, stream);
    } else if (D == 16) {
        if (S == 0) return launch_one<16, 0, 4>(a, n_tasks, n_slots, stream);
        if (S == 1) return launch_one<16, 1, 4>(a, n_tasks, n_slots, stream);
        if (S == 2) return launch_one<16, 2, 3>(a, n_tasks, n_slots, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace gst
