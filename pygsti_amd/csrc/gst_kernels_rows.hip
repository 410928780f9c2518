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
__global__ __launch_bounds__(64) void walk_rows_kernel(const WalkArgs a, const int n_slots)
{
    extern __shared__ double lds[];          // save slots: lds[slot*D + row]; then (LDSG) all gates, transposed
    const int lane = threadIdx.x;
    const int li = lane < D ? lane : 0;      // idle lanes (D < 64) shadow row 0 and never store
    const bool act = lane < D;
    const int64_t bid = blockIdx.x;
    const int32_t pw = (int32_t)(bid % a.n_pwaves);
    const int64_t task = bid / a.n_pwaves;
    const int S = a.rows_S;

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
                if (S == 0 && a.base_cache_w && act) a.base_cache_w[(int64_t)node_id * D + lane] = v;
                if (!more) break;
#pragma unroll
                for (int j = 0; j < D; j++) c[j] = cn[j];
                hit = (S > 0) && ((arg < 64) ? ((wave_gates >> arg) & 1ull) : (wave_gates != 0));
            }
#undef ROWS_LOADC
            continue;                                                  // `op` already holds the next instruction
        } else if (op == GST_OP_NODE) {
            cur_id = (int32_t)arg;
            if (S == 0 && a.base_cache_w && act) a.base_cache_w[(int64_t)arg * D + lane] = v;
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
                v = a.rhos[(int64_t)arg * D + li];
                for (int s = 0; s < S; s++)
                    if (kind[s] == GST_KIND_RHO && obj[s] == (int)arg && lane == row[s]) v = v + a.eps;
                dirty = true;
            }
        }
        ROWS_FETCH();
    }
#undef ROWS_FETCH
}

template <int D>
static hipError_t launch_rows(const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    const int64_t blocks = n_tasks * (int64_t)a.n_pwaves;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL || n_slots > 64) return hipErrorInvalidValue;
    size_t lds_bytes = (size_t)(n_slots > 0 ? n_slots : 1) * D * sizeof(double);
    const size_t gate_bytes = (size_t)a.n_gates * D * D * sizeof(double);
    const bool ldsg = (D <= 16) && gate_bytes > 0 && gate_bytes <= 48 * 1024;
    (void)hipGetLastError();
    if (ldsg)
        hipLaunchKernelGGL((walk_rows_kernel<D, true>), dim3((unsigned)blocks), dim3(64), lds_bytes + gate_bytes, stream, a, n_slots);
    else
        hipLaunchKernelGGL((walk_rows_kernel<D, false>), dim3((unsigned)blocks), dim3(64), lds_bytes, stream, a, n_slots);
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
