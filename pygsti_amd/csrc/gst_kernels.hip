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
#include "gst_chain.hpp"

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

// Wavefronts per SIMD the D = 16 Jacobian pass is compiled for.  The dispatcher-placed launch (whole designs: tens of
// pairs per SIMD) wants 4 (128 VGPRs, a few spills, best latency hiding: 25.3 vs 25.7 ms on the 2Q design); the persistent
// launch of small atoms has only 2-3 long walks per SIMD anyway, and with 3 (168 VGPRs, no spills) a 1/8 atom of the
// 2Q design takes 4.27 instead of 4.48 ms (interleaved repeats, tools/ab_libs.sh).  Measured and NOT kept: scalar-load
// stages of half a column (one s_load_dwordx16 per 16 VALU instructions): tools/ub_fd.hip shows a lone wavefront 15 %
// faster that way, the interpreter around the sweep does not (same time at 3 per SIMD, 6 % slower at 4: more spills).
#ifndef GST_FD_WPS
#define GST_FD_WPS 4
#endif
#ifndef GST_HESS_WPS
#define GST_HESS_WPS 3        // wavefronts per SIMD the two-perturbation (Hessian) pass is compiled for
#endif
#ifndef GST_FD_WPS_PERSIST
#define GST_FD_WPS_PERSIST 3
#endif

template <int D>
struct State {
    double x[D];
};

// Pin a set of accumulators at this point of the program: an empty volatile asm that "modifies" them.
// IR-level passes may otherwise move the (side-effect free) multiplies/adds of a column across the
// scheduling barriers, which empties the software pipeline below.
template <int D>
__device__ __forceinline__ void pin(double (&o)[D])
{
    if constexpr (D == 16) {
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]),
                          "+v"(o[8]), "+v"(o[9]), "+v"(o[10]), "+v"(o[11]), "+v"(o[12]), "+v"(o[13]), "+v"(o[14]), "+v"(o[15]));
    } else {
#pragma unroll
        for (int i = 0; i < D; i++) asm volatile("" : "+v"(o[i]));
    }
}

// o = M v with M given TRANSPOSED (Mt[j*D+i] = M[i][j]): column sweep, o += M[:,j]*v[j].
// Every o[i] still accumulates its products in ascending j from +0.0 with separate mul/add (the
// reference's order, opcreps.cpp:45-51), but the D accumulators are independent, so the VALU never
// waits on a dependent add, and one contiguous scalar load brings a whole column.
// `NS` lane-private special rows ride along as extra accumulators: r[s] = sp[s] . v.
//
// Software pipeline over columns: column j+1's scalar loads (2 x s_load_dwordx16 at D=16) are issued
// BEFORE column j's 2*D VALU instructions, so their latency hides under arithmetic.  SMEM returns out
// of order, so the only usable wait is lgkmcnt(0) and it must come before the next loads are issued.
// The order is enforced three ways: the wait and the pins are side-effecting (kept in program order),
// each column's load address passes through a volatile asm (the loads cannot be hoisted above it), and
// a scheduling barrier keeps the arithmetic behind the loads.
template <int N>
__device__ __forceinline__ void pin_n(double (&o)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(o[i]));
}

// RPW < D: the wavefront computes only rows [i0, i0 + RPW) (the state's rows are split over the D/RPW wavefronts of
// a workgroup, see walk_kernel's NW).  A pipeline stage then covers CPS = D/RPW columns, i.e. the same D
// coefficients (2*D VALU instructions) per scalar-load round trip as the full-row sweep.
template <int D, int NS, int RPW = D>
__device__ __forceinline__ void matvec_t(cdouble_p __restrict__ Mt, const int i0, const double (&v)[D], double (&o)[RPW],
                                         const double (&sp)[NS > 0 ? NS : 1][D], double (&r)[NS > 0 ? NS : 1])
{
    constexpr int CPS = D / RPW;
#pragma unroll
    for (int i = 0; i < RPW; i++) o[i] = 0.0;
#pragma unroll
    for (int s = 0; s < NS; s++) r[s] = 0.0;
    if constexpr (D == 4 && RPW == D) {
        // D = 4 walks are launch / latency bound: a column is only 10 VALU instructions, far less than the ~95 cycles one
        // scalar load takes, so the column pipeline degenerates into four exposed round trips per mat-vec.  The whole
        // 4 x 4 gate (16 coefficients, two s_load_dwordx16) is fetched with ONE wait instead.
        double c[D * D];
        {
            cdouble_p p = Mt;
            asm volatile("" : "+s"(p));
#pragma unroll
            for (int i = 0; i < D * D; i++) c[i] = p[i];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int j = 0; j < D; j++) {
            const double vj = v[j];
#pragma unroll
            for (int i = 0; i < D; i++) o[i] = o[i] + c[j * D + i] * vj;
#pragma unroll
            for (int s = 0; s < NS; s++) r[s] = r[s] + sp[s][j] * vj;
        }
        return;
    }
    double cur[CPS][RPW], nxt[CPS][RPW];
#pragma unroll
    for (int c = 0; c < CPS; c++) {
        cdouble_p p = Mt + c * D + i0;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < RPW; i++) cur[c][i] = p[i];
    }
#pragma unroll
    for (int j0 = 0; j0 < D; j0 += CPS) {
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this stage has landed (vmcnt/expcnt untouched)
        if (j0 + CPS < D) {
#pragma unroll
            for (int c = 0; c < CPS; c++) {
                cdouble_p p = Mt + (j0 + CPS + c) * D + i0;
                asm volatile("" : "+s"(p));
#pragma unroll
                for (int i = 0; i < RPW; i++) nxt[c][i] = p[i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CPS; c++) {
            const double vj = v[j0 + c];
#pragma unroll
            for (int i = 0; i < RPW; i++) o[i] = o[i] + cur[c][i] * vj;
#pragma unroll
            for (int s = 0; s < NS; s++) r[s] = r[s] + sp[s][j0 + c] * vj;
        }
        if constexpr (RPW == D) pin<D>(o); else pin_n<RPW>(o);
#pragma unroll
        for (int c = 0; c < CPS; c++)
#pragma unroll
            for (int i = 0; i < RPW; i++) cur[c][i] = nxt[c][i];
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

// NW > 1 (derivative passes of small atoms): a WORKGROUP of NW wavefronts walks one (task, 64 columns) pair, wavefront
// w computing rows [w*D/NW, (w+1)*D/NW) of every mat-vec; the rows are exchanged through LDS (one barrier per gate
// application).  Same arithmetic per row, so same bits; the pair's critical path gets ~NW/1.1 times shorter and the
// launch has NW times more, smaller pieces -- what a 1/8 atom (4.4 pairs per SIMD) needs to balance.
constexpr int XPAD = 2;      // exchange-buffer lane stride D + 2 doubles: 16-byte accesses stay bank-conflict free

// PERSIST: the workgroup is as many wavefronts as one CU holds and stays for the whole launch; each wavefront pops pairs
// from the queue of the SIMD it runs on (queues the host packed to equal estimated work, WalkArgs::bin_ptr), then from
// the other SIMDs' queues.  The hardware dispatcher places a new one-wavefront workgroup wherever a slot frees up,
// blind to how much work the slot's SIMD neighbours still hold; with ~4-9 pairs per SIMD (1/8, 1/4 atoms) that costs 2-3 %.
// COMP (Hessian pass of "full TP" models): the POVM's complement effect is identity - sum(others), recomputed by the
// reference after every parameter step (complementeffect.py:72-78), so a lane whose perturbations touch one of the
// others also carries the one or two changed components of the complement.
// OVL (persistent launch, D = 16): the base pass runs INSIDE this launch -- one wavefront of a workgroup walks a task's
// chain first (gst_chain.hpp, publishing states and probabilities with write-through stores) while the others already
// run finite-difference walks; a walk that reads the sentinel the destinations were pre-filled with waits for the
// value.  On a 1/8 atom of the 2Q design the base pass was 0.45 ms of a 4.4 ms step that nothing overlapped.
#ifndef GST_OVL_PRIO
#define GST_OVL_PRIO 0       // wavefront priority of a chain of the in-launch base pass: as the oldest wavefront of its SIMD it
                             // is served first anyway -- 0 / 1 / 2 / 3 measured 4.15 / 4.17 / 4.15 / 4.18 ms on a 1/8 atom
#endif
constexpr unsigned long long WAIT_LIMIT_TICKS = 10000000ull;     // bounded waits: 0.1 s of the 100 MHz wall clock

template <int D, int S, int WPS, int NW = 1, bool PERSIST = false, bool COMP = false, bool OVL = false>
__global__ __launch_bounds__(PERSIST ? 64 * WPS * 4 : 64 * NW, PERSIST ? 1 : WPS) void walk_kernel(const WalkArgs a)
{
    static_assert(!OVL || (PERSIST && D == 16), "the overlap form belongs to the persistent D = 16 launch");
    static_assert(!COMP || S == 2, "the single-perturbation pass leaves effect parameters to effect_fd_kernel");
    static_assert(NW == 1 || (S == 1 && D % NW == 0), "row splitting is implemented for the single-perturbation passes");
    static_assert(!PERSIST || (NW == 1 && S == 1), "the persistent form exists for the Jacobian pass");
    constexpr int RPW = D / NW;
    // Save slots (states kept while the children of a branching trie node are walked) live in LDS:
    // slot s, component j, lane l at lds[(s*D + j)*64 + l] -- lane-consecutive 8-byte words, conflict
    // free.  The plan compiler caps the number of slots (gst_options.max_slots) so that 4 wavefronts
    // per SIMD fit in the CU's 160 KB.
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wv = (NW > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int i0 = wv * RPW;                 // first state row this wavefront computes
    if constexpr (!PERSIST) {
        // stand-by launch behind an optimistic persistent one (WalkArgs::guard): nothing to do unless that one gave up
        if (a.guard && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) return;
    }
    if constexpr (OVL) {
        // ---- the chains of the base pass, inside this launch (see WalkArgs::ovl_n_tasks) -------------------------------
        const GST_CONST WalkArgs* c = (const GST_CONST WalkArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        const int n_waves = (int)(blockDim.x >> 6);
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        // chain k of the workgroup on wavefront k: the FIRST wavefronts.  The SIMD's arbiter serves its oldest wavefront
        // first, so a chain on the youngest wavefront of its SIMD starves behind two finite-difference walks (measured:
        // the step got 0.25 ms LONGER); raised priority on top, for the chain's duration only -- a chain is a latency
        // chain of dependent instructions, it leaves most issue slots to the walks anyway
        const int k = wave;
        const int64_t t = (int64_t)blockIdx.x + (int64_t)k * gridDim.x;
        if (t < (int64_t)c->ovl_n_tasks && !c->ovl_test_skip) {
            const int nG = c->n_gates, nE = c->n_effects;
            const int n_slots = c->lds_wave_doubles / (D * 64);
            double* const region = lds + (size_t)n_waves * c->lds_wave_doubles + (size_t)k * c->ovl_chain_doubles;
            double* const ldsE = region;
            double* const ldsG = ldsE + nE * D;
            double* const wlds = region + base_shared_doubles(D, nG, nE);
            uint32_t* const ldsP = (uint32_t*)((int32_t*)(wlds + (n_slots > 0 ? n_slots : 1) * 64 + BASE_ER * (D + 1)) + BASE_ER);
            const int64_t pc0 = as_const(c->task_off)[t];
            const int32_t n_words = (int32_t)(as_const(c->task_off)[t + 1] - pc0);
            const uint32_t* gprog = c->prog + pc0;
            const double* gt = c->gates_t;
            const double* ef = c->effects;
            stage_lds(ldsP, BASE_PW, lane, [&](int i) { return (i < n_words) ? gprog[i] : 0u; });
            stage_lds(ldsG, nG * D * D, lane, [&](int i) { return gt[i]; });
            stage_lds(ldsE, nE * D, lane, [&](int i) { return ef[i]; });
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            ChainArgs ca;
            ca.gprog = gprog; ca.n_words = n_words;
            ca.eff_ptr = c->eff_ptr; ca.eff_label = c->eff_label; ca.eff_dest = c->eff_dest;
            ca.rhos = c->rhos; ca.out = c->pbase_w; ca.cache = (double*)c->base_cache;
            ca.multi_start = 0; ca.start0 = 0;
            const unsigned long long tc0 = c->trace ? wall_clock64() : 0ull;
            __builtin_amdgcn_s_setprio(GST_OVL_PRIO);
            base_chain_walk<D, true>(ca, n_slots, ldsE, ldsG, wlds, lane);
            __builtin_amdgcn_s_setprio(0);
            if (c->trace && lane == 0) {           // development aid (GST_FD_TRACE): chains are records with bit 30 set
                unsigned long long* tr = c->trace;
                const unsigned long long kk = atomicAdd(tr, 1ull);
                tr[1 + 4 * kk] = 0x40000000ull | (unsigned long long)t; tr[2 + 4 * kk] = tc0; tr[3 + 4 * kk] = wall_clock64();
                tr[4 + 4 * kk] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4)) |
                                 ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);
            }
        }
    }
  for (;;) {                                 // PERSIST: one iteration per popped pair; otherwise exactly one
    int64_t bid;
    int32_t part = 0;                        // PERSIST: 0 whole pair, 1 first half (up to the split), 2 second half
    if constexpr (PERSIST) {
        const GST_CONST WalkArgs* c = (const GST_CONST WalkArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t* const head = c->bin_head;
        ci32_p bptr = as_const(c->bin_ptr);
        const int32_t nb = c->n_bins;
        // HW_REG_HW_ID (id 4) bits [5:4]: the SIMD this wavefront was placed on
        const int32_t home_bin = (int32_t)blockIdx.x * 4 + (int32_t)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3u);
        int64_t item = -1;
        {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&head[home_bin], 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            const int32_t b0 = bptr[home_bin], b1 = bptr[home_bin + 1];
            if ((int64_t)k < (int64_t)(b1 - b0)) item = (int64_t)c->bin_items[b0 + (int32_t)k];
        }
        for (int32_t base = 0; item < 0 && base < nb - 1; base += 64) {      // other queues: 64 inspected per step
            const int32_t off = base + lane;
            int32_t vb = home_bin + 1 + off;
            vb = vb >= nb ? vb - nb : vb;
            bool cand = false;
            if (off < nb - 1) {
                const uint32_t h = __hip_atomic_load(&head[vb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cand = (int64_t)h < (int64_t)(bptr[vb + 1] - bptr[vb]);
            }
            uint64_t m = __ballot(cand);
            while (m != 0 && item < 0) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int32_t vv = __builtin_amdgcn_readlane(vb, l);
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&head[vv], 1u);
                k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                const int32_t b0 = bptr[vv], b1 = bptr[vv + 1];
                if ((int64_t)k < (int64_t)(b1 - b0)) item = (int64_t)c->bin_items[b0 + (int32_t)k];
            }
        }
        if (item < 0) break;
        // somebody's bounded wait ran out (WalkArgs::abort_flag): the stand-by launches behind this one redo everything
        if (c->abort_flag && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0) return;
        const uint32_t raw = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        part = (int32_t)(raw >> 30);
        bid = (int64_t)(raw & 0x3fffffffu);
    } else {
        bid = a.block_order ? (int64_t)a.block_order[blockIdx.x] : (int64_t)blockIdx.x;
    }
    const unsigned long long t_begin = (S == 1 && a.trace) ? wall_clock64() : 0ull;
    const int32_t pw = (int32_t)(bid % a.n_pwaves);
    const int64_t task = bid / a.n_pwaves;
    const int64_t q = (int64_t)pw * 64 + lane;

    cdouble_p gates_t = as_const(a.gates_t);
    // Cold kernel arguments (everything EMIT / RHO / NODE need) are NOT kept live in SGPRs across the
    // interpreter loop: `cold()` re-reads them from the kernarg segment where they are used.  The hot loop
    // is SGPR-bound (two 32-SGPR coefficient buffers); every argument that stays live gets spilled into
    // VGPR lanes and comes back through v_readlane in the middle of the arithmetic.
    auto cold = [&]() -> const GST_CONST WalkArgs* {
        const GST_CONST WalkArgs* p = (const GST_CONST WalkArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));      // opaque: keeps LICM from hoisting the loads out of the loop
        return p;
    };

    // ---- per-lane specials ---------------------------------------------------------------------
    constexpr int S1 = S > 0 ? S : 1;
    int kind[S1], obj[S1], row[S1];
    double sp[S1][D];
    int32_t col = (S == 0) ? (lane == 0 ? 0 : -1) : a.lanes.col[q];
    uint64_t gate_mask[S1];
    uint64_t own_mask[S1];            // NW > 1: gates for which a lane's perturbed row is one of THIS wavefront's rows
    bool eff_any[S1], rho_any[S1];
    int cc[2] = {-1, -1};             // COMP: components of the complement effect this lane's perturbations change
    double cval[2] = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < S1; s++) { kind[s] = GST_KIND_NONE; obj[s] = 0; row[s] = -1; gate_mask[s] = 0; own_mask[s] = 0; eff_any[s] = false; rho_any[s] = false; }
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
            else if (kind[s] == GST_KIND_RHO) {
                row[s] = 0; b = el[s]; base = a.rhos + (int64_t)obj[s] * D;
                if (a.rho_models) { base = a.rho_models + (int64_t)el[s] * a.rho_model_stride; b = -1; }
            }
            else if (kind[s] == GST_KIND_EFFECT) { row[s] = 0; b = el[s]; base = a.effects + (int64_t)obj[s] * D; }
#pragma unroll
            for (int j = 0; j < D; j++) {
                const double x = base[j];
                sp[s][j] = (j == b) ? x + a.eps : x;      // theta_p + eps, as set_parameter_value does
            }
            el[s] = b;
        }
        if constexpr (COMP) {
            // changed components of the complement: cc[k] (or -1) with value cval[k] = identity - sum over the others
            // in the reference's order, each other effect's component stepped once per perturbation that hits it
            // (the same element twice: (theta + eps) + eps, like the special row below)
            const GST_CONST WalkArgs* c = cold();
            ci32_p others = as_const(c->comp_others);
            bool hits[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                hits[s] = false;
                if (kind[s] == GST_KIND_EFFECT)
                    for (int k = 0; k < c->n_others; k++) hits[s] = hits[s] || (others[k] == obj[s]);
                cc[s] = hits[s] ? el[s] : -1;
            }
            if (cc[0] >= 0 && cc[0] == cc[1]) cc[1] = -1;            // one component carries both steps
#pragma unroll
            for (int s = 0; s < 2; s++) {
                double sum = 0.0;
                const int comp = cc[s] >= 0 ? cc[s] : 0;
                for (int k = 0; k < c->n_others; k++) {
                    const int o = others[k];
                    double x = as_const(c->effects)[(int64_t)o * D + comp];
                    if (hits[0] && obj[0] == o && el[0] == comp) x = x + c->eps;
                    if (hits[1] && obj[1] == o && el[1] == comp) x = x + c->eps;
                    sum = sum + x;
                }
                cval[s] = as_const(c->comp_identity)[comp] - sum;
            }
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
            const bool own_row = (NW == 1) || (row[s] >= i0 && row[s] < i0 + RPW);
            if (a.n_gates <= 64) {
                for (int g = 0; g < a.n_gates; g++) {
                    if (__ballot(kind[s] == GST_KIND_GATE && obj[s] == g)) gate_mask[s] |= (1ull << g);
                    if (__ballot(kind[s] == GST_KIND_GATE && obj[s] == g && own_row)) own_mask[s] |= (1ull << g);
                }
            } else {
                gate_mask[s] = __ballot(kind[s] == GST_KIND_GATE) ? ~0ull : 0ull;
                own_mask[s] = __ballot(kind[s] == GST_KIND_GATE && own_row) ? ~0ull : 0ull;
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

    // LDS: [exchange buffers: 2 x 64 lanes x (D + XPAD), NW > 1 only] [save slots]
    constexpr int XB = (NW > 1) ? 2 * 64 * (D + XPAD) : 0;
    double* const slots = PERSIST ? lds + (int64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * a.lds_wave_doubles
                                  : lds + XB;
    int xsel = 0;

    double v[D];
    double sp0[1][D];   // placeholder operand of the no-special mat-vec (never read)
#pragma unroll
    for (int j = 0; j < D; j++) sp0[0][j] = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) v[j] = 0.0;

    // ---- clean/dirty tracking (derivative passes) ---------------------------------------------------------
    // A state whose path contains no gate that any lane of this wavefront perturbs (and that did not start
    // from a perturbed rho) is BIT-IDENTICAL to the base pass's state.  Such "clean" states are not computed:
    // the wavefront only follows the NODE markers; when a perturbed gate is finally applied the state is
    // fetched from the base-state cache written by the S=0 pass.  Clean EMITs write exact zeros
    // ((p - p)/eps == +0.0), unless effect parameters are perturbed, in which case the dot runs on the
    // cached state.  All of this is wave-uniform scalar bookkeeping.
    constexpr int MAXSLOT = 4;    // the launcher refuses plans with more slots
    uint64_t wave_gates = 0;
    bool wave_rho = false, wave_eff = false;
#pragma unroll
    for (int s = 0; s < S; s++) { wave_gates |= gate_mask[s]; wave_rho = wave_rho || rho_any[s]; wave_eff = wave_eff || eff_any[s]; }
    if (S == 1 && a.fused) wave_rho = true;     // fused base lane: every path starts from the preparation, nothing is "clean"
    bool dirty = (S == 0);
    int32_t cur_id = 0;
    int32_t slot_tag[MAXSLOT];          // >= 0: the slot "holds" clean state id; -1: real data in LDS
#pragma unroll
    for (int i = 0; i < MAXSLOT; i++) slot_tag[i] = -1;
    cdouble_p cache_c = as_const(a.base_cache);
    // OVL: a value that still shows the sentinel has not been produced yet (or sits in a stale line of this XCD's L2 /
    // scalar cache): poll it with system-scope loads, which bypass the non-coherent caches.  Returns false when the wait
    // ran out (the abort flag is then raised).
    bool ovl_dead = false;
    auto ovl_poll = [&](const double* g) -> double {
        const unsigned long long t_wait = wall_clock64();
        for (;;) {
            // (every lane reads the same address; the value is made wave-uniform explicitly so that control flow stays scalar)
            const long long xb = __double_as_longlong(__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            const int xlo = __builtin_amdgcn_readfirstlane((int)(xb & 0xffffffffLL)), xhi = __builtin_amdgcn_readfirstlane((int)(xb >> 32));
            const double x = __longlong_as_double(((long long)xhi << 32) | (unsigned int)xlo);
            if ((unsigned long long)__double_as_longlong(x) != OVL_SENTINEL64) return x;
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t_wait > WAIT_LIMIT_TICKS) {
                const GST_CONST WalkArgs* c = (const GST_CONST WalkArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                // (all lanes store the same word: a lane-0 branch here is a divergent region the register allocator cannot
                //  keep the interpreter's scalar state across -- "illegal VGPR to SGPR copy")
                if (c->abort_flag) __hip_atomic_store(c->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ovl_dead = true;
                return x;
            }
        }
    };
    // (a macro, not a lambda: an array captured by reference from inside the interpreter loop ends up in scratch memory)
#define GST_FETCH_STATE(id_)                                                                              \
    do {                                                                                                  \
        cdouble_p b_ = cache_c + (int64_t)(id_) * D;                                                      \
        uint32_t late_ = 0;                  /* wave-uniform: bit j = component j still shows the sentinel */ \
        _Pragma("unroll") for (int j = 0; j < D; j++) {                                                   \
            const double x_ = b_[j];                                                                      \
            if constexpr (OVL) late_ |= ((unsigned long long)__double_as_longlong(x_) == OVL_SENTINEL64) ? (1u << j) : 0u; \
            v[j] = x_;                                                                                    \
        }                                                                                                 \
        if constexpr (OVL) {                                                                              \
            if (__builtin_expect(late_ != 0, 0)) {                                                        \
                const double* g_ = ((const GST_CONST WalkArgs*)__builtin_amdgcn_kernarg_segment_ptr())->base_cache + (int64_t)(id_) * D; \
                _Pragma("unroll") for (int j = 0; j < D; j++)                                             \
                    if (!ovl_dead && ((late_ >> j) & 1u)) v[j] = ovl_poll(g_ + j);                        \
                if (ovl_dead) return;                                                                     \
            }                                                                                             \
        }                                                                                                 \
    } while (0)

    // Instruction fetch: the program is read 64 words at a time with ONE coalesced vector load (lane l holds
    // word base+l) and the current word is picked with v_readlane; the next window is already in flight.
    const int64_t pc0 = as_const(a.task_off)[task];
    const int64_t pc_end = as_const(a.task_off)[task + 1];
    const uint32_t* gprog = a.prog + pc0;
    const int32_t n_words = (int32_t)(pc_end - pc0);
    int32_t wbase = 0;                  // first word of the current window (relative to pc0)
    uint32_t win_cur = (lane < n_words) ? gprog[lane] : 0u;
    uint32_t win_nxt = (64 + lane < n_words) ? gprog[64 + lane] : 0u;
    int32_t pc = 0;
    uint32_t op, arg;
#define GST_FETCH()                                                                                   \
    do {                                                                                              \
        if (pc - wbase == 64) {                                                                       \
            wbase += 64;                                                                              \
            win_cur = win_nxt;                                                                        \
            win_nxt = (wbase + 64 + lane < n_words) ? gprog[wbase + 64 + lane] : 0u;                  \
        }                                                                                             \
        const uint32_t w_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, (int)(pc - wbase));     \
        op = GST_OP(w_); arg = GST_ARG(w_); pc++;                                                     \
    } while (0)
#define GST_HIT(g, out)                                                                               \
    do {                                                                                              \
        out = false;                                                                                  \
        _Pragma("unroll") for (int s_ = 0; s_ < S; s_++)                                              \
            out = out || (((g) < 64) ? ((gate_mask[s_] >> (g)) & 1ull) : (gate_mask[s_] != 0));       \
    } while (0)
#define GST_HIT_OWN(g, out)                                                                           \
    do {                                                                                              \
        out = false;                                                                                  \
        _Pragma("unroll") for (int s_ = 0; s_ < S; s_++)                                              \
            out = out || (((g) < 64) ? ((own_mask[s_] >> (g)) & 1ull) : (own_mask[s_] != 0));         \
    } while (0)

    int32_t stop_pc = -1, ho = -1;
    if constexpr (PERSIST) {
        if (part != 0) {
            const GST_CONST WalkArgs* c = cold();
            ho = as_const(c->ho_index)[bid];
            stop_pc = as_const(c->ho_pc)[ho];
        }
        if (part == 2) {
            // second half: the first half (on another SIMD, possibly another XCD with its own L2) stores the lane states
            // and then raises the flag
            const GST_CONST WalkArgs* c = cold();
            uint32_t* const flag = c->ho_flag + ho;
            // (every access to the hand-over buffers is a system-scope relaxed atomic, i.e. a load / store that bypasses
            //  the non-coherent caches: no fence, hence no write-back or invalidation of a whole L2, is needed)
            // The wait is BOUNDED: the first half runs in another workgroup, which is only certain to make progress if
            // it is resident -- not guaranteed when the device is shared (two ranks on one GPU, other kernels holding
            // CUs).  On expiry: raise the abort flag and leave; the stand-by launches redo the Jacobian without hand-overs.
            {
                const unsigned long long t_wait = wall_clock64();
                while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == 0) {
                    __builtin_amdgcn_s_sleep(64);
                    if (wall_clock64() - t_wait > WAIT_LIMIT_TICKS) {
                        if (lane == 0 && c->abort_flag) __hip_atomic_store(c->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return;
                    }
                }
            }
            // (the loads below are issued after the flag has been seen -- a wavefront does not speculate -- and the
            //  compiler may not move them above the loop either)
            asm volatile("" ::: "memory");
            const int32_t id = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c->ho_id + ho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            double* const hs0 = c->ho_state + (int64_t)ho * c->ho_blocks * (D * 64) + lane;
            if (id >= 0) { dirty = false; cur_id = id; }
            else {
#pragma unroll
                for (int j = 0; j < D; j++) v[j] = __hip_atomic_load(hs0 + j * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dirty = true;
            }
            // save slots that are live at the cut travel with the walk: a tag (the id of a clean state) or the data
            const uint32_t lm = c->ho_live ? as_const(c->ho_live)[ho] : 0u;
#pragma unroll
            for (int i = 0; i < MAXSLOT; i++) {
                if ((lm >> i) & 1u) {
                    const int32_t tg = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c->ho_tag + ho * MAXSLOT + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                    slot_tag[i] = tg;
                    if (tg < 0) {
                        const double* hs = hs0 + (int64_t)(1 + i) * (D * 64);
                        double* sl = slots + (int64_t)i * D * 64 + lane;
#pragma unroll
                        for (int j = 0; j < D; j++) sl[j * 64] = __hip_atomic_load(hs + j * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            pc = stop_pc;
            wbase = stop_pc & ~63;
            win_cur = (wbase + lane < n_words) ? gprog[wbase + lane] : 0u;
            win_nxt = (wbase + 64 + lane < n_words) ? gprog[wbase + 64 + lane] : 0u;
        }
    }
    GST_FETCH();
    for (;;) {
        if (op == GST_OP_END) break;
        if constexpr (PERSIST) {
            if (part == 1 && pc - 1 == stop_pc) {
                // first part done: hand the walk over -- the lane states (or the id of a clean state) and the save slots
                // that are live at this position
                const GST_CONST WalkArgs* c = cold();
                double* const hs0 = c->ho_state + (int64_t)ho * c->ho_blocks * (D * 64) + lane;
                if (dirty) {
#pragma unroll
                    for (int j = 0; j < D; j++) __hip_atomic_store(hs0 + j * 64, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                const uint32_t lm = c->ho_live ? as_const(c->ho_live)[ho] : 0u;
#pragma unroll
                for (int i = 0; i < MAXSLOT; i++) {
                    if ((lm >> i) & 1u) {
                        if (lane == 0) __hip_atomic_store(c->ho_tag + ho * MAXSLOT + i, slot_tag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (slot_tag[i] < 0) {
                            double* hs = hs0 + (int64_t)(1 + i) * (D * 64);
                            const double* sl = slots + (int64_t)i * D * 64 + lane;
#pragma unroll
                            for (int j = 0; j < D; j++) __hip_atomic_store(hs + j * 64, sl[j * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
                if (lane == 0) __hip_atomic_store(c->ho_id + ho, dirty ? -1 : cur_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_s_waitcnt(0);           // the write-through stores above have been acknowledged
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) __hip_atomic_store(c->ho_flag + ho, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        if (op == GST_OP_APPLY) {
            // Runs of (APPLY, NODE) pairs -- the chains of the trie -- are handled in two tight inner loops so
            // that the state vector stays in place (one loop-carried definition) instead of being shuffled
            // through the big dispatch loop's join blocks.
            if (S > 0 && !dirty) {
                bool hit;
                GST_HIT(arg, hit);
                while (!hit) {                           // clean run: nothing to compute, follow the markers
                    GST_FETCH();                         // NODE
                    cur_id = (int32_t)arg;
                    GST_FETCH();
                    if (op != GST_OP_APPLY) break;
                    if (PERSIST && pc - 1 == stop_pc) break;          // (a cut inside a chain: handed over at the top of the loop)
                    GST_HIT(arg, hit);
                }
                if (op != GST_OP_APPLY || !hit) continue;
                GST_FETCH_STATE(cur_id);                          // first perturbed gate on this path: start from the cache
                dirty = true;
            }
            do {
                bool hit;
                if constexpr (NW > 1) GST_HIT_OWN(arg, hit); else GST_HIT(arg, hit);
                double o[RPW];
                cdouble_p Mt = gates_t + (int64_t)arg * D * D;
                if (S > 0 && hit) {      // wave-uniform: some lane's parameter lives in this gate (in one of our rows)
                    double r[S1];
                    matvec_t<D, S, RPW>(Mt, i0, v, o, sp, r);
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const bool mine = (kind[s] == GST_KIND_GATE) && (obj[s] == (int)arg);
#pragma unroll
                        for (int i = 0; i < RPW; i++) o[i] = (mine && row[s] == i0 + i) ? r[s] : o[i];
                    }
                } else {
                    double r[1];
                    matvec_t<D, 0, RPW>(Mt, i0, v, o, sp0, r);
                }
                if constexpr (NW == 1) {
#pragma unroll
                    for (int j = 0; j < D; j++) v[j] = o[j];
                } else {
                    // row exchange: every wavefront publishes its RPW rows, then reads the whole state back
                    typedef double d2_t __attribute__((ext_vector_type(2)));
                    double* xb = lds + (xsel * 64 + lane) * (D + XPAD);
                    d2_t* xw = (d2_t*)__builtin_assume_aligned(xb + i0, 16);
#pragma unroll
                    for (int i = 0; i < RPW; i += 2) xw[i / 2] = (d2_t){o[i], o[i + 1]};
                    __syncthreads();
                    const d2_t* xr = (const d2_t*)__builtin_assume_aligned(xb, 16);
#pragma unroll
                    for (int j = 0; j < D; j += 2) { const d2_t t = xr[j / 2]; v[j] = t.x; v[j + 1] = t.y; }
                    xsel ^= 1;
                }
                GST_FETCH();                             // the NODE marker of the state just produced
                cur_id = (int32_t)arg;
                if (S == 0 && cold()->base_cache_w) {
                    // all lanes hold the same state: lane j stores component j (one 8*D-byte line per state)
                    double x = v[0];
#pragma unroll
                    for (int j = 1; j < D; j++) x = (lane == j) ? v[j] : x;
                    if (lane < D) cold()->base_cache_w[(int64_t)arg * D + lane] = x;
                }
                GST_FETCH();
            } while (op == GST_OP_APPLY && !(PERSIST && pc - 1 == stop_pc));
            continue;
        }
        if (op == GST_OP_NODE) {                         // marker after a RHO
            cur_id = (int32_t)arg;
            if (S == 0 && cold()->base_cache_w) {
                double x = v[0];
#pragma unroll
                for (int j = 1; j < D; j++) x = (lane == j) ? v[j] : x;
                if (lane < D) cold()->base_cache_w[(int64_t)arg * D + lane] = x;
            }
        } else if (op == GST_OP_EMIT) {
            const GST_CONST WalkArgs* c = cold();
            ci32_p eff_ptr = as_const(c->eff_ptr);
            ci32_p eff_label = as_const(c->eff_label);
            ci32_p eff_dest = as_const(c->eff_dest);
            cdouble_p effects = as_const(c->effects);
            const int32_t x0 = eff_ptr[arg], x1 = eff_ptr[arg + 1];
            const bool zero = (S > 0) && !dirty && !wave_eff;
            if (S > 0 && !dirty && wave_eff) {           // effect parameters: real dots on the (clean) cached state
                GST_FETCH_STATE(cur_id);
            }
            for (int32_t x = x0; x < x1; x++) {
                if (NW > 1 && ((x - x0) % NW) != wv) continue;       // the workgroup's wavefronts share out the outcomes
                const int32_t e = eff_label[x];
                const int64_t dest = eff_dest[x];
                if (zero) {                              // (p - p)/eps: exact zeros, no arithmetic
                    if (col >= 0) {
                        if (c->mode == EMIT_FD) {
                            c->out[dest * c->ld + col] = 0.0;
                            if (c->raw) c->raw[dest * c->ldraw + col] = as_const(c->pbase)[dest];
                        } else {
                            const double p0 = as_const(c->pbase)[dest];
                            const double pr = as_const(c->prow)[dest * c->ldrow + hrowidx];
                            const double d2 = (p0 - pr) / c->eps;
                            const double d1 = c->dcol[dest * c->lddcol + hcolidx];
                            c->out[(dest * c->ld + hrow) * c->ld2 + col] = (d2 - d1) / c->eps;
                        }
                    }
                    continue;
                }
                double p = dot_c<D>(effects + (int64_t)e * D, v);
#pragma unroll
                for (int s = 0; s < S; s++) {
                    if (eff_any[s]) {
                        const double r = dot_r<D>(sp[s], v);
                        p = (kind[s] == GST_KIND_EFFECT && obj[s] == e) ? r : p;
                    }
                }
                if constexpr (COMP) {
                    if ((int32_t)e == c->comp_index && __ballot(cc[0] >= 0 || cc[1] >= 0) != 0) {
                        double acc = 0.0;
#pragma unroll
                        for (int j = 0; j < D; j++) {
                            double ev = effects[(int64_t)e * D + j];
                            ev = (j == cc[0]) ? cval[0] : ev;
                            ev = (j == cc[1]) ? cval[1] : ev;
                            acc = acc + ev * v[j];
                        }
                        p = (cc[0] >= 0 || cc[1] >= 0) ? acc : p;
                    }
                }
                if (c->mode == EMIT_PROBS) {
                    if (col >= 0) c->out[dest] = p;
                } else if (c->mode == EMIT_FD) {
                    if (S == 1 && c->fused) {
                        // the base model's probability sits in lane 63 of this very wavefront
                        const long long pbits = __double_as_longlong(p);
                        const int plo = __builtin_amdgcn_readlane((int)(pbits & 0xffffffffLL), 63);
                        const int phi = __builtin_amdgcn_readlane((int)(pbits >> 32), 63);
                        const double pb = __longlong_as_double(((long long)phi << 32) | (unsigned int)plo);
                        if (col >= 0) c->out[dest * c->ld + col] = (p - pb) / c->eps;
                        if (lane == 63 && pw == 0 && c->probs_out) c->probs_out[dest] = p;
                    } else if (OVL || col >= 0) {
                        double pb = as_const(c->pbase)[dest];
                        if constexpr (OVL) {
                            if (__builtin_expect((unsigned long long)__double_as_longlong(pb) == OVL_SENTINEL64, 0)) {
                                pb = ovl_poll(c->pbase + dest);      // the chain has not flushed this circuit yet
                                if (ovl_dead) return;
                            }
                        }
                        if (col >= 0) c->out[dest * c->ld + col] = (p - pb) / c->eps;
                        if (col >= 0 && c->raw) c->raw[dest * c->ldraw + col] = p;
                    }
                } else {
                    if (col >= 0) {
                        const double pr = as_const(c->prow)[dest * c->ldrow + hrowidx];
                        const double d2 = (p - pr) / c->eps;                       // dprobs2 (mapforwardsim.py:431)
                        const double d1 = c->dcol[dest * c->lddcol + hcolidx];      // dprobs
                        c->out[(dest * c->ld + hrow) * c->ld2 + col] = (d2 - d1) / c->eps;
                    }
                }
            }
        } else if (op == GST_OP_SAVE) {
            if (S > 0 && !dirty) {
#pragma unroll
                for (int i = 0; i < MAXSLOT; i++) if (arg == (uint32_t)i) slot_tag[i] = cur_id;
            } else {
#pragma unroll
                for (int i = 0; i < MAXSLOT; i++) if (arg == (uint32_t)i) slot_tag[i] = -1;
                double* sl = slots + (int64_t)arg * D * 64 + lane;   // (NW > 1: every wavefront writes the same values)
#pragma unroll
                for (int j = 0; j < D; j++) sl[j * 64] = v[j];
            }
        } else if (op == GST_OP_LOAD) {
            int32_t tag = -1;
#pragma unroll
            for (int i = 0; i < MAXSLOT; i++) if (arg == (uint32_t)i) tag = slot_tag[i];
            if (S > 0 && tag >= 0) {
                dirty = false; cur_id = tag;
            } else {
                const double* sl = slots + (int64_t)arg * D * 64 + lane;
#pragma unroll
                for (int j = 0; j < D; j++) v[j] = sl[j * 64];
                dirty = true;
            }
        } else {  // GST_OP_RHO
            if (S > 0 && !wave_rho) {
                dirty = false;                           // unperturbed preparation: clean
            } else {
                cdouble_p r0 = as_const(cold()->rhos) + (int64_t)arg * D;
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
                dirty = true;
            }
        }
        GST_FETCH();
    }
    if (S == 1) {
        unsigned long long* tr = ((const GST_CONST WalkArgs*)__builtin_amdgcn_kernarg_segment_ptr())->trace;
        if (tr && lane == 0) {
            const unsigned long long k = atomicAdd(tr, 1ull);
            tr[1 + 4 * k] = (unsigned long long)bid; tr[2 + 4 * k] = t_begin; tr[3 + 4 * k] = wall_clock64();
            tr[4 + 4 * k] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4)) |
                            ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);
        }
    }
    if constexpr (!PERSIST) break;
  }
#undef GST_FETCH
#undef GST_HIT
#undef GST_FETCH_STATE
}

// ---- small helpers for the normal equations (gst_fill_jtj_dev) ------------------------------------------------
__global__ void scale_rows_kernel(double* J, int64_t n_rows, int64_t n_cols, int64_t ld, const double* w)
{
    const int64_t total = n_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_cols, c = i - r * n_cols;
        J[r * ld + c] *= w[r];
    }
}
__global__ void symmetrize_kernel(double* C, int64_t n)   // copy the computed triangle onto the other one
{
    const int64_t total = n * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n, c = i - r * n;
        if (c > r) C[c * n + r] = C[r * n + c];   // dsyrk(lower, column-major) filled the row-major UPPER triangle
    }
}
hipError_t launch_scale_rows(double* J, int64_t n_rows, int64_t n_cols, int64_t ld, const double* w, hipStream_t s)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(scale_rows_kernel, dim3(2048), dim3(256), 0, s, J, n_rows, n_cols, ld, w);
    return hipGetLastError();
}
hipError_t launch_symmetrize(double* C, int64_t n, hipStream_t s)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(symmetrize_kernel, dim3(1024), dim3(256), 0, s, C, n);
    return hipGetLastError();
}

template <int D, int S, int WPS, int NW = 1, bool COMP = false>
static hipError_t launch_one(const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    const int64_t blocks = n_tasks * (int64_t)a.n_pwaves;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds_bytes = ((size_t)n_slots * D * 64 + (NW > 1 ? 2 * 64 * (D + XPAD) : 0)) * sizeof(double);
    if (lds_bytes > 64 * 1024 || n_slots > 4) return hipErrorInvalidValue;   // MAXSLOT tags in the kernel
    (void)hipGetLastError();   // drop any stale sticky error (e.g. an event query) so that we report OUR launch
    hipLaunchKernelGGL((walk_kernel<D, S, WPS, NW, false, COMP>), dim3((unsigned)blocks), dim3(64 * NW), lds_bytes, stream, a);
    return hipGetLastError();
}

int persistent_waves(int D) { return (D == 16) ? 4 * GST_FD_WPS_PERSIST : 16; }

hipError_t launch_walk_persistent(int D, const WalkArgs& a, int n_wg, int n_slots, hipStream_t stream)
{
    const int waves = persistent_waves(D);
    if (n_wg <= 0 || n_slots > 4 || (D != 4 && D != 16)) return hipErrorInvalidValue;
    size_t lds_bytes = (size_t)waves * a.lds_wave_doubles * sizeof(double);
    const bool ovl = a.ovl_n_tasks > 0;
    if (ovl) {
        // the chains' LDS regions sit behind the wavefronts' save slots; wavefront (waves - 1 - k) walks chain k of its workgroup
        const int chains = (a.ovl_n_tasks + n_wg - 1) / n_wg;
        if (D != 16 || chains > waves || a.ovl_chain_doubles <= 0 || !a.pbase_w || !a.base_cache) return hipErrorInvalidValue;
        lds_bytes += (size_t)chains * (size_t)a.ovl_chain_doubles * sizeof(double);
    }
    if ((size_t)a.lds_wave_doubles < (size_t)n_slots * D * 64 || lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    (void)hipGetLastError();
    if (D == 16 && ovl) {
        auto k = walk_kernel<16, 1, GST_FD_WPS_PERSIST, 1, true, false, true>;
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, dim3((unsigned)n_wg), dim3(64 * waves), lds_bytes, stream, a);
    } else if (D == 16) {
        auto k = walk_kernel<16, 1, GST_FD_WPS_PERSIST, 1, true>;
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, dim3((unsigned)n_wg), dim3(64 * waves), lds_bytes, stream, a);
    } else {
        auto k = walk_kernel<4, 1, 4, 1, true>;
        hipLaunchKernelGGL(k, dim3((unsigned)n_wg), dim3(64 * waves), lds_bytes, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_walk(int D, int S, const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream, int split, bool complement)
{
    if (complement) {
        if (S == 2 && D == 4) return launch_one<4, 2, 4, 1, true>(a, n_tasks, n_slots, stream);
        if (S == 2 && D == 16) return launch_one<16, 2, GST_HESS_WPS, 1, true>(a, n_tasks, n_slots, stream);
        return hipErrorInvalidValue;
    }
    if (D == 16 && S == 1 && split == 4) return launch_one<16, 1, 4, 4>(a, n_tasks, n_slots, stream);
    if (D == 16 && S == 1 && split == 2) return launch_one<16, 1, 4, 2>(a, n_tasks, n_slots, stream);
    if (D == 4) {
        if (S == 0) return launch_one<4, 0, 4>(a, n_tasks, n_slots, stream);
        if (S == 1) return launch_one<4, 1, 4>(a, n_tasks, n_slots, stream);
        if (S == 2) return launch_one<4, 2, 4>(a, n_tasks, n_slots, stream);
    } else if (D == 16) {
        if (S == 0) return launch_one<16, 0, 4>(a, n_tasks, n_slots, stream);
        if (S == 1) return launch_one<16, 1, GST_FD_WPS>(a, n_tasks, n_slots, stream);
        if (S == 2) return launch_one<16, 2, GST_HESS_WPS>(a, n_tasks, n_slots, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace gst
