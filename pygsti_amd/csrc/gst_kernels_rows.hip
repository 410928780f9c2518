// gst_kernels_rows.hip -- `walk_rows_kernel<D>`: ONE wavefront = ONE model, lane i = ROW i of the state.
//
// Complements the lane-per-model kernel of gst_kernels.hip for the cases where there are few models and
// latency matters, or where a whole state does not fit in one lane's registers:
//   * the single-pass probabilities / base pass of every derivative call (S = 0): 16 (D=16) or 4 (D=4)
//     active lanes, one mat-vec = D x (2 v_readlane + v_mul_f64 + v_add_f64) instead of 2*D*D VALU
//     instructions per wavefront -> the latency-bound pass gets ~10x shorter; the base-state cache is
//     written with one coalesced D*8-byte store per state (lane j holds component j);
//   * D = 64 (three qubits) in all modes: state = one f64 per lane, coefficients of the lane's row
//     stream through L2 (gates_t[g][j][lane], coalesced 512-byte lines), one perturbed model per
//     wavefront (the perturbed element lives in exactly one lane's coefficient register).
// Arithmetic contract unchanged: out_i = (((0.0 + G[i][0]*v_0) + G[i][1]*v_1) + ...), separate multiply and
// add, ascending j (opcreps.cpp:40-54); effect dots accumulate ascending i from 0.0 (effectcreps.cpp:39-45).
// The walk-program interpreter (RHO/APPLY/NODE/SAVE/LOAD/EMIT), the clean/dirty tracking against the
// base-state cache and the three EMIT modes are the same as in gst_kernels.hip.
#include "gst_kernels.hpp"
#include "gst_chain.hpp"

#include "../../include/gstfwd.h"

namespace gst {

#define GST_CONST __attribute__((address_space(4)))
typedef const GST_CONST double* cdouble_p;
typedef const GST_CONST int32_t* ci32_p;
typedef const GST_CONST int64_t* ci64_p;
template <typename T>
__device__ __forceinline__ const GST_CONST T* as_const(const T* p) { return (const GST_CONST T*)(p); }

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int D, bool LDSG>
__global__ __launch_bounds__(64) void walk_rows_kernel(const WalkArgs a_, const int n_slots)
{
    extern __shared__ double lds[];          // save slots: lds[slot*D + row]; then (LDSG) all gates, transposed
    const int lane = threadIdx.x;
    const int li = lane < D ? lane : 0;      // idle lanes (D < 64) shadow row 0 and never store
    const bool act = lane < D;
    WalkArgs a = a_;
    const int64_t bid = a.block_order ? (int64_t)a.block_order[blockIdx.x] : (int64_t)blockIdx.x;
    const int32_t pw = (int32_t)(bid % a.n_pwaves);
    const int64_t task = bid / a.n_pwaves;
    const int S = a.rows_S;
    if (a.n_models > 0) {                    // whole-model mode: wavefront pw of a task carries model set pw
        a.gates_t += (int64_t)pw * a.model_stride; a.rhos += (int64_t)pw * a.model_stride;
        a.effects += (int64_t)pw * a.model_stride; a.out += (int64_t)pw * a.out_model_stride;
    }

    // ---- the (up to two) perturbations of this wavefront's model: wave-uniform -------------------------
    int kind[2] = {GST_KIND_NONE, GST_KIND_NONE}, obj[2] = {0, 0}, row[2] = {-1, -1}, cb[2] = {-1, -1};
    int32_t col = 0;
    uint64_t wave_gates = 0;
    bool wave_rho = false, wave_eff = false;
    if (S > 0) {
        col = as_const(a.lanes.col)[pw];
        for (int s = 0; s < S; s++) {
            kind[s] = as_const(a.lanes.kind[s])[pw];
            obj[s] = as_const(a.lanes.obj[s])[pw];
            const int el = as_const(a.lanes.elem[s])[pw];
            if (kind[s] == GST_KIND_GATE) { row[s] = el / D; cb[s] = el % D; wave_gates |= (obj[s] < 64) ? (1ull << obj[s]) : ~0ull; }
            else if (kind[s] == GST_KIND_RHO) { row[s] = el; wave_rho = true; }
            else if (kind[s] == GST_KIND_EFFECT) { row[s] = el; wave_eff = true; }
        }
    }
    int64_t hrow = 0, hrowidx = 0, hcolidx = 0;
    if (S == 2) { hrow = as_const(a.wave_row)[pw]; hrowidx = as_const(a.wave_rowidx)[pw]; hcolidx = as_const(a.lane_colidx)[pw]; }

    // Small state dimension: stage every (transposed) gate in LDS once -- 2 KB per gate at D = 16 -- so that a
    // mat-vec's D coefficient reads cost an LDS round trip instead of an L2 one on this latency-bound path.
    double* ldsG = lds + (n_slots > 0 ? n_slots : 1) * D;
    if (LDSG) {
        const int total = a.n_gates * D * D;
        for (int k = lane; k < total; k += 64) ldsG[k] = a.gates_t[k];
        __builtin_amdgcn_s_waitcnt(0);      // single-wavefront block: no barrier needed, just drain
        __builtin_amdgcn_wave_barrier();
    }

    double v = 0.0;
    bool dirty = (S == 0);
    int32_t cur_id = 0;
    int32_t tags = -1;                       // lane s: clean-state id "held" by save slot s (-1: real data in LDS)
    // S = 0 with a.multi_start > 0 (backward states of the analytic mode): wavefront pw of a task walks from
    // rhos[a.start0 + pw] whatever RHO's argument and stores its states at base_cache_w[id][a.start0 + pw][D]
    const bool multi = (S == 0) && a.multi_start > 0;
    const int32_t my_start = a.start0 + pw;
    const int64_t cstride = multi ? (int64_t)a.multi_start * D : (int64_t)D;
    const int64_t coff = multi ? (int64_t)my_start * D : 0;

    const int64_t pc0 = as_const(a.task_off)[task];
    const int32_t n_words = (int32_t)(as_const(a.task_off)[task + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    int32_t wbase = 0, pc = 0;
    uint32_t win_cur = (lane < n_words) ? gprog[lane] : 0u;
    uint32_t win_nxt = (64 + lane < n_words) ? gprog[64 + lane] : 0u;
    uint32_t op, arg;
#define ROWS_FETCH()                                                                                  \
    do {                                                                                              \
        if (pc - wbase == 64) {                                                                       \
            wbase += 64;                                                                              \
            win_cur = win_nxt;                                                                        \
            win_nxt = (wbase + 64 + lane < n_words) ? gprog[wbase + 64 + lane] : 0u;                  \
        }                                                                                             \
        const uint32_t w_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, (int)(pc - wbase));     \
        op = GST_OP(w_); arg = GST_ARG(w_); pc++;                                                     \
    } while (0)

    ROWS_FETCH();
    for (;;) {
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            bool hit = (S > 0) && ((arg < 64) ? ((wave_gates >> arg) & 1ull) : (wave_gates != 0));
            if (S > 0 && !dirty) {
                if (!hit) { ROWS_FETCH(); continue; }                 // base state: nothing to compute
                v = a.base_cache[(int64_t)cur_id * D + li];           // first perturbed gate: start from the cache
                dirty = true;
            }
            // Chains of (APPLY, NODE) pairs run in this tight loop; the coefficients of the NEXT mat-vec (row
            // `lane` of its gate: c[j] = G[lane][j] = gates_t[g][j][lane], coalesced across lanes) are requested
            // before the current one is computed, so their LDS / L2 latency hides under the arithmetic.
            double c[D], cn[D];
#define ROWS_LOADC(dst, g_)                                                                           \
            do {                                                                                      \
                if (LDSG) {                                                                           \
                    const double* Gt_ = ldsG + (int)(g_) * D * D + li;                                \
                    _Pragma("unroll") for (int j = 0; j < D; j++) dst[j] = Gt_[j * D];                \
                } else {                                                                              \
                    const double* Gt_ = a.gates_t + (int64_t)(g_) * D * D + li;                       \
                    _Pragma("unroll") for (int j = 0; j < D; j++) dst[j] = Gt_[j * D];                \
                }                                                                                     \
            } while (0)
            ROWS_LOADC(c, arg);
            for (;;) {
                const uint32_t g = arg;
                ROWS_FETCH();                                          // the NODE marker of the state being produced
                const int32_t node_id = (int32_t)arg;
                ROWS_FETCH();                                          // what follows
                const bool more = (op == GST_OP_APPLY);
                if (more) ROWS_LOADC(cn, arg);
                if (hit) {
                    for (int s = 0; s < S; s++) {
                        const bool mine = kind[s] == GST_KIND_GATE && obj[s] == (int)g && lane == row[s];
#pragma unroll
                        for (int j = 0; j < D; j++) c[j] = (mine && j == cb[s]) ? c[j] + a.eps : c[j];   // theta + eps
                    }
                }
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < D; j++) acc = acc + c[j] * readlane_f64(v, j);
                v = acc;
                cur_id = node_id;
                if (S == 0 && a.base_cache_w && act) a.base_cache_w[(int64_t)node_id * cstride + coff + lane] = v;
                if (!more) break;
#pragma unroll
                for (int j = 0; j < D; j++) c[j] = cn[j];
                hit = (S > 0) && ((arg < 64) ? ((wave_gates >> arg) & 1ull) : (wave_gates != 0));
            }
#undef ROWS_LOADC
            continue;                                                  // `op` already holds the next instruction
        } else if (op == GST_OP_NODE) {
            cur_id = (int32_t)arg;
            if (S == 0 && a.base_cache_w && act) a.base_cache_w[(int64_t)arg * cstride + coff + lane] = v;
        } else if (op == GST_OP_EMIT) {
            const int32_t x0 = as_const(a.eff_ptr)[arg], x1 = as_const(a.eff_ptr)[arg + 1];
            const bool zero = (S > 0) && !dirty && !wave_eff;
            if (S > 0 && !dirty && wave_eff) v = a.base_cache[(int64_t)cur_id * D + li];
            for (int32_t x = x0; x < x1; x++) {
                const int32_t e = as_const(a.eff_label)[x];
                const int64_t dest = as_const(a.eff_dest)[x];
                double p = 0.0;
                if (!zero) {
                    double ei = a.effects[(int64_t)e * D + li];
                    for (int s = 0; s < S; s++)
                        if (kind[s] == GST_KIND_EFFECT && obj[s] == e && lane == row[s]) ei = ei + a.eps;
                    const double prod = ei * v;
#pragma unroll
                    for (int i = 0; i < D; i++) p = p + readlane_f64(prod, i);
                }
                if (lane == 0) {
                    if (a.mode == EMIT_PROBS) {
                        a.out[dest] = p;
                    } else if (a.mode == EMIT_FD) {
                        const double pb = a.pbase[dest];
                        a.out[dest * a.ld + col] = zero ? 0.0 : (p - pb) / a.eps;
                        if (a.raw) a.raw[dest * a.ldraw + col] = zero ? pb : p;
                    } else {
                        if (zero) p = a.pbase[dest];
                        const double d2 = (p - a.prow[dest * a.ldrow + hrowidx]) / a.eps;
                        const double d1 = a.dcol[dest * a.lddcol + hcolidx];
                        a.out[(dest * a.ld + hrow) * a.ld2 + col] = (d2 - d1) / a.eps;
                    }
                }
            }
        } else if (op == GST_OP_SAVE) {
            if (S > 0 && !dirty) {
                tags = (lane == (int)arg) ? cur_id : tags;
            } else {
                tags = (lane == (int)arg) ? -1 : tags;
                if (act) lds[arg * D + lane] = v;
            }
        } else if (op == GST_OP_LOAD) {
            const int32_t tag = __builtin_amdgcn_readlane(tags, (int)arg);
            if (S > 0 && tag >= 0) { dirty = false; cur_id = tag; }
            else { v = lds[arg * D + li]; dirty = true; }
        } else {  // GST_OP_RHO
            if (S > 0 && !wave_rho) {
                dirty = false;
            } else {
                v = a.rhos[(int64_t)(multi ? (uint32_t)my_start : arg) * D + li];
                for (int s = 0; s < S; s++)
                    if (kind[s] == GST_KIND_RHO && obj[s] == (int)arg && lane == row[s]) v = v + a.eps;
                dirty = true;
            }
        }
        ROWS_FETCH();
    }
#undef ROWS_FETCH
}


// ---------------------------------------------------------------------------------------------------------------
// walk_rows_shared_kernel<D, NWB>: the derivative passes at D = 64 (S = 1 finite differences, S = 2 Hessian blocks).
// With one model per wavefront every wavefront used to pull its own copy of each gate (32 KB) through L2 for every
// gate application: 18 TB/s of L2 traffic and 4.6 TFLOP/s of arithmetic on the 3-qubit configuration.  Here NWB
// wavefronts -- NWB perturbed models, consecutive parameter columns of ONE task -- form a workgroup that walks the
// task in lockstep: a gate application stages the transposed gate ONCE in LDS (the next gate's tile is already in
// flight in registers during the current mat-vec) and every wavefront streams its coefficients from there.
// Lockstep needs workgroup-uniform control flow, so the clean/dirty bookkeeping runs on the UNION of the wavefronts'
// perturbed objects: a wavefront may compute a state it could have taken from the base cache, with the unperturbed
// coefficients and the same operation order -- the same bits.  Arithmetic contract unchanged.
template <int D, int NWB>
__global__ __launch_bounds__(64 * NWB) void walk_rows_shared_kernel(const WalkArgs a, const int n_slots)
{
    extern __shared__ double lds[];          // gate tile [D][D] (transposed) | save slots [NWB][n_slots][D] | NWB*3 masks
    constexpr int NT = 64 * NWB;
    constexpr int TPT = D * D / NT;          // tile elements per thread
    static_assert(D == 64 && (D * D) % NT == 0, "row-per-lane kernel for three qubits");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bid = a.block_order ? (int64_t)a.block_order[blockIdx.x] : (int64_t)blockIdx.x;
    const int32_t n_pblocks = (a.n_pwaves + NWB - 1) / NWB;
    const int32_t pb = (int32_t)(bid % n_pblocks);
    const int64_t task = bid / n_pblocks;
    const bool live = pb * NWB + wv < a.n_pwaves;            // spare wavefronts of the last group shadow its last model
    const int32_t pw = live ? pb * NWB + wv : a.n_pwaves - 1;
    const int S = a.rows_S;
    double* const ldsG = lds;
    double* const slots = lds + D * D + (size_t)wv * (n_slots > 0 ? n_slots : 1) * D;
    uint64_t* const msk = (uint64_t*)(lds + D * D + (size_t)NWB * (n_slots > 0 ? n_slots : 1) * D);

    int kind[2] = {GST_KIND_NONE, GST_KIND_NONE}, obj[2] = {0, 0}, row[2] = {-1, -1}, cb[2] = {-1, -1};
    uint64_t wave_gates = 0;
    bool wave_rho = false, wave_eff = false;
    const int32_t col = as_const(a.lanes.col)[pw];
    for (int s = 0; s < S; s++) {
        kind[s] = as_const(a.lanes.kind[s])[pw];
        obj[s] = as_const(a.lanes.obj[s])[pw];
        const int el = as_const(a.lanes.elem[s])[pw];
        if (kind[s] == GST_KIND_GATE) { row[s] = el / D; cb[s] = el % D; wave_gates |= (obj[s] < 64) ? (1ull << obj[s]) : ~0ull; }
        else if (kind[s] == GST_KIND_RHO) { row[s] = el; wave_rho = true; }
        else if (kind[s] == GST_KIND_EFFECT) { row[s] = el; wave_eff = true; }
    }
    int64_t hrow = 0, hrowidx = 0, hcolidx = 0;
    if (S == 2) { hrow = as_const(a.wave_row)[pw]; hrowidx = as_const(a.wave_rowidx)[pw]; hcolidx = as_const(a.lane_colidx)[pw]; }
    // workgroup-wide union of the perturbed objects
    if (lane == 0) { msk[wv] = wave_gates; msk[NWB + wv] = wave_rho ? 1 : 0; msk[2 * NWB + wv] = wave_eff ? 1 : 0; }
    __syncthreads();
    uint64_t blk_gates = 0, blk_rho_m = 0, blk_eff_m = 0;
    for (int w = 0; w < NWB; w++) { blk_gates |= msk[w]; blk_rho_m |= msk[NWB + w]; blk_eff_m |= msk[2 * NWB + w]; }
    blk_gates = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(blk_gates >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)blk_gates);
    const bool blk_rho = __builtin_amdgcn_readfirstlane((int)blk_rho_m) != 0;
    const bool blk_eff = __builtin_amdgcn_readfirstlane((int)blk_eff_m) != 0;

    double v = 0.0;
    bool dirty = false;
    int32_t cur_id = 0;
    int32_t tags = -1;                       // lane s: clean-state id "held" by save slot s (-1: real data in LDS)

    const int64_t pc0 = as_const(a.task_off)[task];
    const int32_t n_words = (int32_t)(as_const(a.task_off)[task + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    int32_t wbase = 0, pc = 0;
    uint32_t win_cur = (lane < n_words) ? gprog[lane] : 0u;
    uint32_t win_nxt = (64 + lane < n_words) ? gprog[64 + lane] : 0u;
    uint32_t op, arg;
#define SH_FETCH()                                                                                    \
    do {                                                                                              \
        if (pc - wbase == 64) {                                                                       \
            wbase += 64;                                                                              \
            win_cur = win_nxt;                                                                        \
            win_nxt = (wbase + 64 + lane < n_words) ? gprog[wbase + 64 + lane] : 0u;                  \
        }                                                                                             \
        const uint32_t w_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, (int)(pc - wbase));     \
        op = GST_OP(w_); arg = GST_ARG(w_); pc++;                                                     \
    } while (0)
#define SH_BLKHIT(g_) (((g_) < 64) ? ((blk_gates >> (g_)) & 1ull) != 0 : (blk_gates != 0))

    SH_FETCH();
    for (;;) {
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            if (!dirty) {
                if (!SH_BLKHIT(arg)) { SH_FETCH(); continue; }             // base state: nothing to compute
                v = a.base_cache[(int64_t)cur_id * D + lane];              // first perturbed gate: start from the cache
                dirty = true;
            }
            double pre[TPT];
            {
                const double* Gt_ = a.gates_t + (int64_t)arg * D * D + tid;
#pragma unroll
                for (int k = 0; k < TPT; k++) pre[k] = Gt_[k * NT];
            }
            for (;;) {
                const uint32_t g = arg;
                __syncthreads();                                           // nobody reads the previous tile any more
#pragma unroll
                for (int k = 0; k < TPT; k++) ldsG[tid + k * NT] = pre[k];
                __syncthreads();
                SH_FETCH();                                                // the NODE marker of the state being produced
                const int32_t node_id = (int32_t)arg;
                SH_FETCH();                                                // what follows
                const bool more = (op == GST_OP_APPLY);
                if (more) {                                                // next tile: in flight during this mat-vec
                    const double* Gt_ = a.gates_t + (int64_t)arg * D * D + tid;
#pragma unroll
                    for (int k = 0; k < TPT; k++) pre[k] = Gt_[k * NT];
                }
                bool m0 = false, m1 = false;
                if (S > 0) m0 = kind[0] == GST_KIND_GATE && obj[0] == (int)g && lane == row[0];
                if (S > 1) m1 = kind[1] == GST_KIND_GATE && obj[1] == (int)g && lane == row[1];
                const bool any_mine = __builtin_amdgcn_readfirstlane((int)(__ballot(m0 || m1) != 0)) != 0;
                double acc = 0.0;
                if (any_mine) {
#pragma unroll 16
                    for (int j = 0; j < D; j++) {
                        double c = ldsG[j * D + lane];
                        c = (m0 && j == cb[0]) ? c + a.eps : c;            // theta + eps
                        c = (m1 && j == cb[1]) ? c + a.eps : c;
                        acc = acc + c * readlane_f64(v, j);
                    }
                } else {
#pragma unroll 16
                    for (int j = 0; j < D; j++) acc = acc + ldsG[j * D + lane] * readlane_f64(v, j);
                }
                v = acc;
                cur_id = node_id;
                if (!more) break;
            }
            continue;                                                      // `op` already holds the next instruction
        } else if (op == GST_OP_NODE) {
            cur_id = (int32_t)arg;
        } else if (op == GST_OP_EMIT) {
            const int32_t x0 = as_const(a.eff_ptr)[arg], x1 = as_const(a.eff_ptr)[arg + 1];
            const bool zero = !dirty && !blk_eff;
            if (!dirty && blk_eff) v = a.base_cache[(int64_t)cur_id * D + lane];
            for (int32_t x = x0; x < x1; x++) {
                const int32_t e = as_const(a.eff_label)[x];
                const int64_t dest = as_const(a.eff_dest)[x];
                double p = 0.0;
                if (!zero) {
                    double ei = a.effects[(int64_t)e * D + lane];
                    for (int s = 0; s < S; s++)
                        if (kind[s] == GST_KIND_EFFECT && obj[s] == e && lane == row[s]) ei = ei + a.eps;
                    const double prod = ei * v;
#pragma unroll 16
                    for (int i = 0; i < D; i++) p = p + readlane_f64(prod, i);
                }
                if (lane == 0 && live) {
                    if (a.mode == EMIT_FD) {
                        const double pbv = a.pbase[dest];
                        a.out[dest * a.ld + col] = zero ? 0.0 : (p - pbv) / a.eps;
                        if (a.raw) a.raw[dest * a.ldraw + col] = zero ? pbv : p;
                    } else {
                        if (zero) p = a.pbase[dest];
                        const double d2 = (p - a.prow[dest * a.ldrow + hrowidx]) / a.eps;
                        const double d1 = a.dcol[dest * a.lddcol + hcolidx];
                        a.out[(dest * a.ld + hrow) * a.ld2 + col] = (d2 - d1) / a.eps;
                    }
                }
            }
        } else if (op == GST_OP_SAVE) {
            if (!dirty) {
                tags = (lane == (int)arg) ? cur_id : tags;
            } else {
                tags = (lane == (int)arg) ? -1 : tags;
                slots[arg * D + lane] = v;
            }
        } else if (op == GST_OP_LOAD) {
            const int32_t tag = __builtin_amdgcn_readlane(tags, (int)arg);
            if (tag >= 0) { dirty = false; cur_id = tag; }
            else { v = slots[arg * D + lane]; dirty = true; }
        } else {  // GST_OP_RHO
            if (!blk_rho) {
                dirty = false;
            } else {
                v = a.rhos[(int64_t)arg * D + lane];
                for (int s = 0; s < S; s++)
                    if (kind[s] == GST_KIND_RHO && obj[s] == (int)arg && lane == row[s]) v = v + a.eps;
                dirty = true;
            }
        }
        SH_FETCH();
    }
#undef SH_FETCH
#undef SH_BLKHIT
}

// ---------------------------------------------------------------------------------------------------------------
// walk_quad64_kernel<K, NWB>: the derivative passes at D = 64, register-blocked.  A wavefront carries 4*K perturbed
// models: DPP row m (16 lanes) of "set" k is model 4k + m, and lane i of that row holds the four state components
// i, 16+i, 32+i, 48+i of it (v[k][0..3]) and computes the same four rows of every mat-vec.  Then
//   * v_j reaches the 16 lanes of its model with ONE v_mov_b64_dpp row_newbcast (no v_readlane pairs, no SGPR hazard);
//   * a coefficient G[16c+i][j] read from the LDS tile (one 32-byte read per lane and column: the tile is stored
//     [j][i][c]) serves all 4*K models of the wavefront, and the tile itself (32 KB, staged once per gate application,
//     next one prefetched in registers) serves the NWB wavefronts of the workgroup: 9 VALU instructions per 4 rows x 4
//     models and column instead of 16 x (2 v_readlane + mul + add);
//   * a model's ONE perturbed row (the reference's `full` parameterisation: one element of one dense object) is redone
//     on its own after the sweep, with the perturbed coefficient in place, in the same ascending-j order (this is the
//     "special row" of gst_kernels.hip), only for applications of the perturbed gate;
//   * save slots live in registers (4 components per lane and model); at most QUAD_MAXSLOT of them.
// Lockstep across the workgroup (the tile barriers) runs on the union of the perturbed objects, as in
// walk_rows_shared_kernel.  Arithmetic contract unchanged: acc = 0.0; acc = acc + G[r][j] * v[j], ascending j.
constexpr int QUAD_MAXSLOT = 2;
constexpr int QUAD_K = 4;
constexpr int QUAD_NWB = 4;

template <int J>
__device__ __forceinline__ double bcast16(double x)      // element J of every 16-lane row to the whole row
{
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xf, 0xf, true);
}

template <int K, int J>
struct QuadOps {
    // column J of the sweep: 4 rows x 4*K models
    static __device__ __forceinline__ void sweep(const double* __restrict__ tile_lane, const double (&v)[K][4], double (&acc)[K][4])
    {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t* tp = (const d2_t*)__builtin_assume_aligned(tile_lane + J * 64, 16);
        const d2_t c01 = tp[0], c23 = tp[1];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const double b = bcast16<J & 15>(v[k][J >> 4]);
            acc[k][0] = acc[k][0] + c01.x * b;
            acc[k][1] = acc[k][1] + c01.y * b;
            acc[k][2] = acc[k][2] + c23.x * b;
            acc[k][3] = acc[k][3] + c23.y * b;
        }
        if constexpr (J + 1 < 64) QuadOps<K, J + 1>::sweep(tile_lane, v, acc);
    }
    // one row redone with up to two perturbed coefficients: r = sum_j (G[row][j] (+eps at j == cb0) (+eps at j == cb1)) v_j
    static __device__ __forceinline__ double special(const double* __restrict__ row_ptr, const double (&vk)[4], int cb0, int cb1,
                                                     double eps, double r)
    {
        double c = row_ptr[J * 64];
        c = (J == cb0) ? c + eps : c;
        c = (J == cb1) ? c + eps : c;
        r = r + c * bcast16<J & 15>(vk[J >> 4]);
        if constexpr (J + 1 < 64) return QuadOps<K, J + 1>::special(row_ptr, vk, cb0, cb1, eps, r);
        else return r;
    }
    // ordered sum of the 64 products of an effect dot (component 16q + i in prod[q] of lane i)
    static __device__ __forceinline__ double dot(const double (&prod)[4], double p)
    {
        p = p + bcast16<J & 15>(prod[J >> 4]);
        if constexpr (J + 1 < 64) return QuadOps<K, J + 1>::dot(prod, p);
        else return p;
    }
};

template <int K, int NWB>
__global__ __launch_bounds__(64 * NWB, 2) void walk_quad64_kernel(const WalkArgs a)
{
    constexpr int D = 64, NT = 64 * NWB, TPT = D * D / NT, MPW = 4 * K, MPB = MPW * NWB;
    extern __shared__ __attribute__((aligned(16))) double lds[];   // tile [j][i][c] = G[16c+i][j] | 3*NWB masks
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lm = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bid = a.block_order ? (int64_t)a.block_order[blockIdx.x] : (int64_t)blockIdx.x;
    const int32_t n_models = a.n_pwaves;                       // the lane tables hold one entry per model
    const int32_t n_pblocks = (n_models + MPB - 1) / MPB;
    const int32_t pb = (int32_t)(bid % n_pblocks);
    const int64_t task = bid / n_pblocks;
    const int S = a.rows_S;
    double* const tile = lds;
    const double* const tile_lane = tile + li * 4;
    uint64_t* const msk = (uint64_t*)(lds + D * D);

    // ---- per lane: the model of each set, its perturbation(s) -----------------------------------------------------
    bool live[K];
    int32_t col[K];
    int kind[K][2], obj[K][2], prow[K][2], pcb[K][2];
    int64_t hrow[K], hrowidx[K], hcolidx[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int32_t M_ = (pb * NWB + wv) * MPW + k * 4 + lm;
        live[k] = M_ < n_models;
        const int32_t M = live[k] ? M_ : n_models - 1;         // spare models shadow the last one, never store
        col[k] = a.lanes.col[M];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            kind[k][s] = GST_KIND_NONE; obj[k][s] = 0; prow[k][s] = -1; pcb[k][s] = -1;
            if (s < S) {
                kind[k][s] = a.lanes.kind[s][M];
                obj[k][s] = a.lanes.obj[s][M];
                const int el = a.lanes.elem[s][M];
                if (kind[k][s] == GST_KIND_GATE) { prow[k][s] = el / D; pcb[k][s] = el % D; }
                else if (kind[k][s] == GST_KIND_RHO || kind[k][s] == GST_KIND_EFFECT) prow[k][s] = el;
            }
        }
        hrow[k] = hrowidx[k] = hcolidx[k] = 0;
        if (S == 2) { hrow[k] = a.wave_row[M]; hrowidx[k] = a.wave_rowidx[M]; hcolidx[k] = a.lane_colidx[M]; }
    }
    uint64_t wave_gates = 0;
    bool wave_rho = false, wave_eff = false;
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (a.n_gates <= 64) {
                for (int g = 0; g < a.n_gates; g++)
                    if (__ballot(kind[k][s] == GST_KIND_GATE && obj[k][s] == g)) wave_gates |= (1ull << g);
            } else if (__ballot(kind[k][s] == GST_KIND_GATE)) wave_gates = ~0ull;
            wave_rho = wave_rho || __ballot(kind[k][s] == GST_KIND_RHO) != 0;
            wave_eff = wave_eff || __ballot(kind[k][s] == GST_KIND_EFFECT) != 0;
        }
    if (lane == 0) { msk[wv] = wave_gates; msk[NWB + wv] = wave_rho ? 1 : 0; msk[2 * NWB + wv] = wave_eff ? 1 : 0; }
    __syncthreads();
    uint64_t blk_gates = 0, blk_rho_m = 0, blk_eff_m = 0;
    for (int w = 0; w < NWB; w++) { blk_gates |= msk[w]; blk_rho_m |= msk[NWB + w]; blk_eff_m |= msk[2 * NWB + w]; }
    blk_gates = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(blk_gates >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)blk_gates);
    const bool blk_rho = __builtin_amdgcn_readfirstlane((int)blk_rho_m) != 0;
    const bool blk_eff = __builtin_amdgcn_readfirstlane((int)blk_eff_m) != 0;

    double v[K][4];
    double sv[QUAD_MAXSLOT][K][4];
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int q = 0; q < 4; q++) { v[k][q] = 0.0; sv[0][k][q] = 0.0; sv[1][k][q] = 0.0; }
    int32_t slot_tag[QUAD_MAXSLOT] = {-1, -1};
    bool dirty = false;
    int32_t cur_id = 0;

    const int64_t pc0 = as_const(a.task_off)[task];
    const int32_t n_words = (int32_t)(as_const(a.task_off)[task + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    int32_t wbase = 0, pc = 0;
    uint32_t win_cur = (lane < n_words) ? gprog[lane] : 0u;
    uint32_t win_nxt = (64 + lane < n_words) ? gprog[64 + lane] : 0u;
    uint32_t op, arg;
#define Q_FETCH()                                                                                     \
    do {                                                                                              \
        if (pc - wbase == 64) {                                                                       \
            wbase += 64;                                                                              \
            win_cur = win_nxt;                                                                        \
            win_nxt = (wbase + 64 + lane < n_words) ? gprog[wbase + 64 + lane] : 0u;                  \
        }                                                                                             \
        const uint32_t w_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, (int)(pc - wbase));     \
        op = GST_OP(w_); arg = GST_ARG(w_); pc++;                                                     \
    } while (0)
#define Q_BLKHIT(g_) (((g_) < 64) ? ((blk_gates >> (g_)) & 1ull) != 0 : (blk_gates != 0))
#define Q_FROM_CACHE()                                                                                \
    do {                                                                                              \
        const double* b_ = a.base_cache + (int64_t)cur_id * D + li;                                   \
        const double b0_ = b_[0], b1_ = b_[16], b2_ = b_[32], b3_ = b_[48];                           \
        _Pragma("unroll") for (int k = 0; k < K; k++) { v[k][0] = b0_; v[k][1] = b1_; v[k][2] = b2_; v[k][3] = b3_; } \
    } while (0)
    // tile element e (row-major in gates_t: e = j*64 + r, value G[r][j]) goes to [j][r & 15][r >> 4]
#define Q_TILE_POS(e_) (((e_) & ~63) | ((((e_) & 15)) << 2) | (((e_) >> 4) & 3))

    Q_FETCH();
    for (;;) {
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            if (!dirty) {
                if (!Q_BLKHIT(arg)) { Q_FETCH(); continue; }               // base state: nothing to compute
                Q_FROM_CACHE();                                            // first perturbed gate: start from the cache
                dirty = true;
            }
            double pre[TPT];
            {
                const double* Gt_ = a.gates_t + (int64_t)arg * D * D + tid;
#pragma unroll
                for (int t = 0; t < TPT; t++) pre[t] = Gt_[t * NT];
            }
            for (;;) {
                const uint32_t g = arg;
                __syncthreads();                                           // nobody reads the previous tile any more
#pragma unroll
                for (int t = 0; t < TPT; t++) { const int e = tid + t * NT; tile[Q_TILE_POS(e)] = pre[t]; }
                __syncthreads();
                Q_FETCH();                                                 // the NODE marker of the state being produced
                const int32_t node_id = (int32_t)arg;
                Q_FETCH();                                                 // what follows
                const bool more = (op == GST_OP_APPLY);
                if (more) {                                                // next tile: in flight during this mat-vec
                    const double* Gt_ = a.gates_t + (int64_t)arg * D * D + tid;
#pragma unroll
                    for (int t = 0; t < TPT; t++) pre[t] = Gt_[t * NT];
                }
                double acc[K][4];
#pragma unroll
                for (int k = 0; k < K; k++) { acc[k][0] = 0.0; acc[k][1] = 0.0; acc[k][2] = 0.0; acc[k][3] = 0.0; }
                QuadOps<K, 0>::sweep(tile_lane, v, acc);
                // special rows: the perturbed row of every model whose perturbed gate this is
                const bool wave_hit = (g < 64) ? ((wave_gates >> g) & 1ull) != 0 : (wave_gates != 0);
                if (wave_hit) {
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        for (int s = 0; s < S; s++) {
                            const bool tgt = kind[k][s] == GST_KIND_GATE && obj[k][s] == (int)g;
                            if (!__builtin_amdgcn_readfirstlane((int)(__ballot(tgt) != 0))) continue;
                            const int r_ = tgt ? prow[k][s] : 0;           // every lane of the model's row group helps
                            // all perturbations of this model that sit in the same row of this gate, in order
                            const int cb0 = (kind[k][0] == GST_KIND_GATE && obj[k][0] == (int)g && prow[k][0] == r_) ? pcb[k][0] : -1;
                            const int cb1 = (S > 1 && kind[k][1] == GST_KIND_GATE && obj[k][1] == (int)g && prow[k][1] == r_) ? pcb[k][1] : -1;
                            const double* row_ptr = tile + (r_ & 15) * 4 + (r_ >> 4);      // + j*64: G[r_][j]
                            const double r = QuadOps<K, 0>::special(row_ptr, v[k], cb0, cb1, a.eps, 0.0);
                            const bool put = tgt && li == (r_ & 15);
#pragma unroll
                            for (int c = 0; c < 4; c++) acc[k][c] = (put && c == (r_ >> 4)) ? r : acc[k][c];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < K; k++) { v[k][0] = acc[k][0]; v[k][1] = acc[k][1]; v[k][2] = acc[k][2]; v[k][3] = acc[k][3]; }
                cur_id = node_id;
                if (!more) break;
            }
            continue;                                                      // `op` already holds the next instruction
        } else if (op == GST_OP_NODE) {
            cur_id = (int32_t)arg;
        } else if (op == GST_OP_EMIT) {
            const int32_t x0 = as_const(a.eff_ptr)[arg], x1 = as_const(a.eff_ptr)[arg + 1];
            const bool zero = !dirty && !blk_eff;
            if (!dirty && blk_eff) Q_FROM_CACHE();
            for (int32_t x = x0; x < x1; x++) {
                const int32_t e = as_const(a.eff_label)[x];
                const int64_t dest = as_const(a.eff_dest)[x];
                double Ev[4];
                if (!zero) {
                    const double* E_ = a.effects + (int64_t)e * D + li;
                    Ev[0] = E_[0]; Ev[1] = E_[16]; Ev[2] = E_[32]; Ev[3] = E_[48];
                }
                const double pbv = a.pbase[dest];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    double p = 0.0;
                    if (!zero) {
                        double prod[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            double ei = Ev[q];
                            for (int s = 0; s < S; s++)
                                if (kind[k][s] == GST_KIND_EFFECT && obj[k][s] == e && prow[k][s] == 16 * q + li) ei = ei + a.eps;
                            prod[q] = ei * v[k][q];
                        }
                        p = QuadOps<K, 0>::dot(prod, 0.0);
                    }
                    if (li == 0 && live[k]) {
                        if (a.mode == EMIT_FD) {
                            a.out[dest * a.ld + col[k]] = zero ? 0.0 : (p - pbv) / a.eps;
                            if (a.raw) a.raw[dest * a.ldraw + col[k]] = zero ? pbv : p;
                        } else {
                            if (zero) p = pbv;
                            const double d2 = (p - a.prow[dest * a.ldrow + hrowidx[k]]) / a.eps;
                            const double d1 = a.dcol[dest * a.lddcol + hcolidx[k]];
                            a.out[(dest * a.ld + hrow[k]) * a.ld2 + col[k]] = (d2 - d1) / a.eps;
                        }
                    }
                }
            }
        } else if (op == GST_OP_SAVE) {
#pragma unroll
            for (int i = 0; i < QUAD_MAXSLOT; i++) {
                if (arg == (uint32_t)i) {
                    if (!dirty) slot_tag[i] = cur_id;
                    else {
                        slot_tag[i] = -1;
#pragma unroll
                        for (int k = 0; k < K; k++) { sv[i][k][0] = v[k][0]; sv[i][k][1] = v[k][1]; sv[i][k][2] = v[k][2]; sv[i][k][3] = v[k][3]; }
                    }
                }
            }
        } else if (op == GST_OP_LOAD) {
#pragma unroll
            for (int i = 0; i < QUAD_MAXSLOT; i++) {
                if (arg == (uint32_t)i) {
                    if (slot_tag[i] >= 0) { dirty = false; cur_id = slot_tag[i]; }
                    else {
#pragma unroll
                        for (int k = 0; k < K; k++) { v[k][0] = sv[i][k][0]; v[k][1] = sv[i][k][1]; v[k][2] = sv[i][k][2]; v[k][3] = sv[i][k][3]; }
                        dirty = true;
                    }
                }
            }
        } else {  // GST_OP_RHO
            if (!blk_rho) {
                dirty = false;
            } else {
                const double* r_ = a.rhos + (int64_t)arg * D + li;
                const double r0 = r_[0], r1 = r_[16], r2 = r_[32], r3 = r_[48];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    v[k][0] = r0; v[k][1] = r1; v[k][2] = r2; v[k][3] = r3;
                    for (int s = 0; s < S; s++) {
                        if (kind[k][s] == GST_KIND_RHO && obj[k][s] == (int)arg && (prow[k][s] & 15) == li) {
#pragma unroll
                            for (int q = 0; q < 4; q++) v[k][q] = ((prow[k][s] >> 4) == q) ? v[k][q] + a.eps : v[k][q];
                        }
                    }
                }
                dirty = true;
            }
        }
        Q_FETCH();
    }
#undef Q_TILE_POS
#undef Q_FROM_CACHE
#undef Q_BLKHIT
#undef Q_FETCH
}

constexpr int ROWS_SHARED_NWB = 4;           // wavefronts (models) per workgroup of walk_rows_shared_kernel
int rows_group(int D, int n_slots)
{
    if (D != 64) return 1;
    if (n_slots <= QUAD_MAXSLOT) return 4 * QUAD_K * QUAD_NWB;          // walk_quad64_kernel: models per workgroup
    const size_t bytes = ((size_t)D * D + (size_t)ROWS_SHARED_NWB * (n_slots > 0 ? n_slots : 1) * D) * sizeof(double) +
                         3 * ROWS_SHARED_NWB * sizeof(uint64_t);
    return bytes <= 64 * 1024 ? ROWS_SHARED_NWB : 1;
}
// ---------------------------------------------------------------------------------------------------------------
// walk_base_kernel<D> (D = 4, 16): the S = 0 pass -- probabilities and the base-state cache -- on its own.  The chain
// walk itself lives in gst_chain.hpp (the persistent FD kernel of small atoms runs the same walk inside its own launch).
constexpr int BASE_MAXW = 8;                 // wavefronts (tasks) per workgroup of walk_base_kernel, at most

// Does walk_base_kernel (the only D <= 16 kernel with multi-start walks) fit its tables into LDS for this model?
bool chain_kernel_fits(int D, int n_gates, int n_effects, int n_slots)
{
    if (D > 16 || n_gates <= 0) return false;
    const size_t gate_bytes = (size_t)n_gates * D * D * sizeof(double);
    return gate_bytes <= 128 * 1024 && (base_shared_doubles(D, n_gates, n_effects) + base_wave_doubles(D, n_slots)) * sizeof(double) <= 156 * 1024;
}
size_t chain_lds_doubles(int D, int n_gates, int n_effects, int n_slots)
{
    return base_shared_doubles(D, n_gates, n_effects) + base_wave_doubles(D, n_slots);
}

// `wpb` wavefronts per workgroup, one task each (task = blockIdx.x * wpb + wavefront), sharing ONE staged copy of the
// gates and effects: a pass is a latency chain per wavefront, so what it needs is residency -- every task of the
// pass (and of the pass running beside it on the other stream, analytic mode) on the chip at once -- and the 12 KB
// of 2Q gate tables per single-wavefront workgroup were what limited that (6 tasks per CU).
// `guard` (may be NULL): the launch is a stand-by repeat of work an optimistic launch before it may have abandoned
// (WalkArgs::abort_flag); it leaves at once unless that flag is raised.
template <int D>
__global__ __launch_bounds__(64 * BASE_MAXW) void walk_base_kernel(const WalkArgs a_, const int n_slots, const int wpb, const int64_t n_tasks,
                                                                   const uint32_t* guard)
{
    if (guard && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    constexpr int W = BASE_PW;
    extern __shared__ double lds[];          // effects | gates_t | per wavefront: save slots | emit ring | ring circuits | program window
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WalkArgs a = a_;
    int64_t task = (int64_t)blockIdx.x * wpb + wv;
    const bool have_task = task < n_tasks;
    if (!have_task) task = 0;                // (stages the shared tables with the others, then leaves)
    if (a.n_models > 0) {                    // whole-model mode (wpb = 1): block = (model set, task)
        const int64_t m = task / a.mm_tasks;
        task -= m * a.mm_tasks;
        a.gates_t += m * a.model_stride; a.rhos += m * a.model_stride; a.effects += m * a.model_stride;
        a.out += m * a.out_model_stride;
    }
    double* const ldsE = lds;
    double* const ldsG = ldsE + a.n_effects * D;
    double* const wlds = lds + base_shared_doubles(D, a.n_gates, a.n_effects) + (size_t)wv * base_wave_doubles(D, n_slots);
    uint32_t* const ldsP = (uint32_t*)((int32_t*)(wlds + (n_slots > 0 ? n_slots : 1) * 64 + BASE_ER * (D + 1)) + BASE_ER);

    const int64_t pc0 = as_const(a.task_off)[task];
    const int32_t n_words = (int32_t)(as_const(a.task_off)[task + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    stage_lds(ldsP, W, lane, [&](int k) { return (k < n_words) ? gprog[k] : 0u; });
    {   // the shared tables: every wavefront of the workgroup copies its share
        const int nG2 = a.n_gates * D * D, nE2 = a.n_effects * D;
        const int per = ((nG2 + wpb - 1) / wpb + 63) / 64 * 64, g0 = wv * per;
        stage_lds(ldsG + g0, (g0 < nG2) ? ((nG2 - g0 < per) ? nG2 - g0 : per) : 0, lane, [&](int k) { return a.gates_t[g0 + k]; });
        if (wv == 0) stage_lds(ldsE, nE2, lane, [&](int k) { return a.effects[k]; });
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (wpb > 1) __syncthreads();            // (one wavefront: the drain is enough)
    else __builtin_amdgcn_wave_barrier();
    if (!have_task) return;
    ChainArgs ca;
    ca.gprog = gprog; ca.n_words = n_words;
    ca.eff_ptr = a.eff_ptr; ca.eff_label = a.eff_label; ca.eff_dest = a.eff_dest;
    ca.rhos = a.rhos; ca.out = a.out; ca.cache = a.base_cache_w;
    ca.multi_start = a.multi_start; ca.start0 = a.start0;
    base_chain_walk<D, false>(ca, n_slots, ldsE, ldsG, wlds, lane);
}

template <int D>
static hipError_t launch_rows(const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    const int64_t blocks = n_tasks * (int64_t)a.n_pwaves;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL || n_slots > 64) return hipErrorInvalidValue;
    size_t lds_bytes = (size_t)(n_slots > 0 ? n_slots : 1) * D * sizeof(double);
    const size_t gate_bytes = (size_t)a.n_gates * D * D * sizeof(double);
    // gates in LDS: up to 128 KB of the CU's 160 KB (64 gates at D = 16); above 64 KB the kernel needs its dynamic-LDS
    // limit raised, per device and per kernel (cheap, so done on every such launch)
    const bool ldsg = (D <= 16) && gate_bytes > 0 && gate_bytes <= 128 * 1024;
    (void)hipGetLastError();
    auto raise_lds = [](const void* fn, size_t bytes) -> hipError_t {
        return bytes > 64 * 1024 ? hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) : hipSuccess;
    };
    if constexpr (D == 64) {
        constexpr int NWB = ROWS_SHARED_NWB;
        const size_t sh_bytes = ((size_t)D * D + (size_t)NWB * (n_slots > 0 ? n_slots : 1) * D) * sizeof(double) + 3 * NWB * sizeof(uint64_t);
        if (a.rows_S > 0 && a.mode != EMIT_PROBS && n_slots <= QUAD_MAXSLOT) {
            const size_t q_bytes = (size_t)D * D * sizeof(double) + 3 * QUAD_NWB * sizeof(uint64_t);
            const int64_t groups = n_tasks * (int64_t)((a.n_pwaves + 4 * QUAD_K * QUAD_NWB - 1) / (4 * QUAD_K * QUAD_NWB));
            hipLaunchKernelGGL((walk_quad64_kernel<QUAD_K, QUAD_NWB>), dim3((unsigned)groups), dim3(64 * QUAD_NWB), q_bytes, stream, a);
            return hipGetLastError();
        }
        if (a.rows_S > 0 && a.mode != EMIT_PROBS && rows_group(D, n_slots) > 1) {
            const int64_t groups = n_tasks * (int64_t)((a.n_pwaves + NWB - 1) / NWB);
            hipLaunchKernelGGL((walk_rows_shared_kernel<D, NWB>), dim3((unsigned)groups), dim3(64 * NWB), sh_bytes, stream, a, n_slots);
            return hipGetLastError();
        }
    }
    if constexpr (D <= 16) {
        if (ldsg && a.rows_S == 0 && a.n_pwaves == 1 && a.mode == EMIT_PROBS) {
            const size_t sh = base_shared_doubles(D, a.n_gates, a.n_effects) * sizeof(double);
            const size_t pw = base_wave_doubles(D, n_slots) * sizeof(double);
            if (sh + pw <= 156 * 1024) {
                // wavefronts per workgroup: the fewest with which every task of the launch is resident at once (a.chain_share
                // passes run side by side, each gets its share of the chip); a chain alone on its SIMD steps fastest
                static const int n_cus = []() {
                    int dev = 0, n = 0;
                    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
                    return 256;
                }();
                const int64_t want = blocks * (int64_t)(a.chain_share > 1 ? a.chain_share : 1);
                int wpb = 1;
                for (; wpb < BASE_MAXW; wpb *= 2) {
                    const int64_t per_cu = std::min<int64_t>((int64_t)(160 * 1024 / (sh + wpb * pw)) * wpb, 20);      // (96 VGPRs: 5 wavefronts per SIMD)
                    if (per_cu * n_cus >= want) break;
                    if (sh + 2 * wpb * pw > 156 * 1024) break;
                }
                const size_t bytes = sh + wpb * pw;
                if (hipError_t ea = raise_lds((const void*)walk_base_kernel<D>, bytes)) return ea;
                hipLaunchKernelGGL((walk_base_kernel<D>), dim3((unsigned)((blocks + wpb - 1) / wpb)), dim3(64 * wpb), bytes,
                                   stream, a, n_slots, wpb, blocks, a.guard);
                return hipGetLastError();
            }
        }
        if (ldsg && a.rows_S == 0 && a.n_models > 0 && a.mode == EMIT_PROBS && a.n_pwaves == a.n_models) {
            // whole-model mode on the chain kernel: one single-wavefront block per (model set, task)
            const size_t bytes = (base_shared_doubles(D, a.n_gates, a.n_effects) + base_wave_doubles(D, n_slots)) * sizeof(double);
            if (bytes <= 156 * 1024) {
                if (hipError_t ea = raise_lds((const void*)walk_base_kernel<D>, bytes)) return ea;
                WalkArgs b = a;
                b.mm_tasks = (int32_t)n_tasks; b.n_pwaves = 1;
                hipLaunchKernelGGL((walk_base_kernel<D>), dim3((unsigned)blocks), dim3(64), bytes, stream, b, n_slots, 1, blocks, b.guard);
                return hipGetLastError();
            }
        }
        if (a.multi_start > 0) return hipErrorInvalidValue;        // D <= 16: only the chain kernel implements multi-start walks
    }
    if (ldsg) {
        if (hipError_t ea = raise_lds((const void*)walk_rows_kernel<D, true>, lds_bytes + gate_bytes)) return ea;
        hipLaunchKernelGGL((walk_rows_kernel<D, true>), dim3((unsigned)blocks), dim3(64), lds_bytes + gate_bytes, stream, a, n_slots);
    }
    else
        hipLaunchKernelGGL((walk_rows_kernel<D, false>), dim3((unsigned)blocks), dim3(64), lds_bytes, stream, a, n_slots);
    return hipGetLastError();
}


// Finite-difference quotients of whole perturbed models (gst_fill_dprobs_models): raw holds one probability vector per
// model set, [model][element]; the Jacobian wants [element][column].  A 64 x 64 tile goes through LDS so that both
// the reads (along elements) and the writes (along columns) are coalesced.  (p_m - p) / eps with the correctly rounded
// fp64 division the FD kernels use (mapforwardsim_calc_densitymx.pyx:378).
__global__ __launch_bounds__(256) void fd_from_models_kernel(const double* __restrict__ raw, int64_t raw_stride,
                                                             const double* __restrict__ pbase, int64_t nE, int32_t n_models,
                                                             const int32_t* __restrict__ dest, int32_t m0, double eps,
                                                             double* __restrict__ out, int64_t ld)
{
    __shared__ double tile[64][65];
    const int64_t e0 = (int64_t)blockIdx.x * 64;
    const int32_t c0 = (int32_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;         // 4 rows of the tile per pass
    for (int j = ty; j < 64; j += 4) {
        const int32_t m = c0 + j;
        const int64_t e = e0 + tx;
        tile[j][tx] = (m < n_models && e < nE) ? raw[(int64_t)m * raw_stride + e] : 0.0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t e = e0 + i;
        const int32_t m = c0 + tx;
        if (e < nE && m < n_models) {
            const int64_t col = dest ? (int64_t)dest[m] : (int64_t)(m0 + m);
            out[e * ld + col] = (tile[tx][i] - pbase[e]) / eps;
        }
    }
}

hipError_t launch_fd_from_models(const double* raw, int64_t raw_stride, const double* pbase, int64_t nE, int32_t n_models,
                                 const int32_t* dest, int32_t m0, double eps, double* out, int64_t ld, hipStream_t stream)
{
    if (nE <= 0 || n_models <= 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fd_from_models_kernel, dim3((unsigned)((nE + 63) / 64), (unsigned)((n_models + 63) / 64)), dim3(256), 0, stream,
                       raw, raw_stride, pbase, nE, n_models, dest, m0, eps, out, ld);
    return hipGetLastError();
}

// One row of a COMPOSED FD-of-FD Hessian block (mapforwardsim.py:432-436: `(dprobs2 - dprobs) / eps` with dprobs2 the FD
// Jacobian of the model stepped in the row's parameter): H[e][row][dest2[j]] = (J_i[e][j] - J_0[e][j]) / eps.
__global__ __launch_bounds__(256) void hess_compose_kernel(const double* __restrict__ Ji, const double* __restrict__ J0, int64_t nE, int32_t n2,
                                                           double eps, double* __restrict__ H, int64_t ld1, int64_t ld2, int64_t row,
                                                           const int32_t* __restrict__ dest2)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nE * (int64_t)n2) return;
    const int64_t e = k / n2;
    const int32_t j = (int32_t)(k - e * n2);
    const int64_t col = dest2 ? (int64_t)dest2[j] : (int64_t)j;
    H[(e * ld1 + row) * ld2 + col] = (Ji[k] - J0[k]) / eps;
}

hipError_t launch_hess_compose(const double* Ji, const double* J0, int64_t nE, int32_t n2, double eps, double* H, int64_t ld1, int64_t ld2,
                               int64_t row, const int32_t* dest2, hipStream_t stream)
{
    if (nE <= 0 || n2 <= 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(hess_compose_kernel, dim3((unsigned)((nE * n2 + 255) / 256)), dim3(256), 0, stream, Ji, J0, nE, n2, eps, H, ld1, ld2, row, dest2);
    return hipGetLastError();
}

// FD columns of effect parameters, from the cached final states.  Perturbing an element of an effect vector changes no
// propagated state, so the column is (E'.F_n - p)/eps on the circuits' final states -- E' being the perturbed effect for
// outcomes of that effect, the RECOMPUTED complement for the outcome of a TP POVM's complement effect
// (complementeffect.py:72-78: identity - sum(others); the host passes the one changed component), and the unperturbed
// effect -- an exact zero -- otherwise.  Same dot product as the walk kernels' EMIT (ascending j, separate mul / add).
// One thread per (circuit, column).
__global__ __launch_bounds__(256) void effect_fd_kernel(EffectFDArgs a)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_circuits * (int64_t)a.n_cols) return;
    const int64_t c = t / a.n_cols;
    const int k = (int)(t - c * a.n_cols);
    const int D = a.D;
    const int e = a.col_obj[k], i = a.col_elem[k];
    const int64_t col = a.col_dest[k];
    const double own = a.col_own[k], cmp = a.col_comp[k];
    const bool touches_comp = a.col_touches_comp[k] != 0;
    const double* F = a.base_cache + (int64_t)a.circ_leaf[c] * D;
    for (int32_t x = a.eff_ptr[c]; x < a.eff_ptr[c + 1]; x++) {
        const int l = a.eff_label[x];
        const int64_t dest = a.eff_dest[x];
        const bool is_own = (l == e), is_comp = (l == a.comp_index) && touches_comp;
        double d = 0.0, praw = a.pbase[dest];
        if (is_own || is_comp) {
            const double sub = is_own ? own : cmp;
            const double* E = a.effects + (int64_t)l * D;
            double acc = 0.0;
            for (int j = 0; j < D; j++) {
                const double ev = (j == i) ? sub : E[j];
                acc = acc + ev * F[j];
            }
            praw = acc;
            d = (acc - a.pbase[dest]) / a.eps;
        }
        a.out[dest * a.ld + col] = d;
        if (a.raw) a.raw[dest * a.ldraw + col] = praw;
    }
}

hipError_t launch_effect_fd(const EffectFDArgs& a, hipStream_t stream)
{
    const int64_t n = a.n_circuits * (int64_t)a.n_cols;
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(effect_fd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_walk_rows(int D, const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    if (D == 4) return launch_rows<4>(a, n_tasks, n_slots, stream);
    if (D == 16) return launch_rows<16>(a, n_tasks, n_slots, stream);
    if (D == 64) return launch_rows<64>(a, n_tasks, n_slots, stream);
    return hipErrorInvalidValue;
}

}  // namespace gst
