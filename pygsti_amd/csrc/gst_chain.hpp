// gst_chain.hpp -- the S = 0 "chain" walk (probabilities + base-state cache) as a device function, shared by
//   * walk_base_kernel<D> (gst_kernels_rows.hip): the pass on its own, one wavefront per task;
//   * the persistent FD kernel with overlap (gst_kernels.hip, walk_kernel<..., OVL>): on a small atom the base pass is a
//     pure latency chain (~0.45 ms of a ~4.4 ms step at 1/8 of the 2Q design) that nothing overlapped; there, one
//     wavefront of a workgroup walks a task's chain FIRST, publishing every state and probability with write-through
//     stores, while the other wavefronts already run finite-difference walks that consume them (see `PUB`).
//
// The pass is a pure latency chain (one wavefront per task, ~1,150 dependent mat-vecs at 2Q L<=1024), so the code
// is organised around the length of one chain step (tools/ub_chain.hip measures the pieces):
//   * every D-lane group of the wavefront carries the SAME state (lane l holds component l % D; identical
//     arithmetic, so identical bits): v_j reaches all lanes with ONE v_mov_b64_dpp row_newbcast (D = 16) instead of
//     two v_readlane + an SGPR hazard;
//   * gates (transposed), effects AND the task's walk program live in LDS (the program in a window that is
//     refilled half by half); a chain step reads its two program words one step ahead, and the coefficients of the
//     NEXT mat-vec stream into the registers the current one has just consumed -- the only vector-memory operation of
//     a step is the 128-byte store of the produced state into the base-state cache, which nothing ever waits for;
//   * EMITs only park (state, circuit) in an LDS ring; the parked circuits are evaluated at once, ONE LANE PER
//     CIRCUIT (each lane walks its circuit's effects and runs the D-term dot product by itself, ascending index from
//     0.0 like effectcreps.cpp:39-45), so the element-table lookups overlap instead of costing two
//     dependent L2 round trips per circuit.
// Arithmetic contract as everywhere: acc = 0.0; acc = acc + G[i][j] * v[j], ascending j, separate multiply and add
// (opcreps.cpp:40-54).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/gstfwd.h"
#include "gst_kernels.hpp"

namespace gst {

template <int D, int J>
__device__ __forceinline__ double grp_bcast(double x)
{
    if constexpr (D == 16) {
        return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xf, 0xf, true);           // row_newbcast:J
    } else {
        constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);                          // quad_perm [J,J,J,J]
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), ctrl, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), ctrl, 0xf, 0xf, true);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
}
// One mat-vec of the chain, phase by phase so that a lone wavefront never issues an instruction that depends on
// the one just before it except in the final (inherently serial) sum: D broadcasts of v, D products, the D
// coefficient reads of the NEXT gate into the registers the products just freed, then the ordered sum.
template <int D, int J>
__device__ __forceinline__ void bcast_all(double (&bv)[D], const double v)
{
    bv[J] = grp_bcast<D, J>(v);
    if constexpr (J + 1 < D) bcast_all<D, J + 1>(bv, v);
}
template <int D>
__device__ __forceinline__ double matvec_stream(double (&c)[D], const double v, const double* Gn)
{
    double bv[D];
    bcast_all<D, 0>(bv, v);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < D; j++) bv[j] = c[j] * bv[j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < D; j++) c[j] = Gn[j * D];
    __builtin_amdgcn_sched_barrier(0);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) acc = acc + bv[j];
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

#ifndef GST_BASE_PW
#define GST_BASE_PW 512
#endif
#ifndef GST_BASE_ER
#define GST_BASE_ER 32
#endif
constexpr int BASE_PW = GST_BASE_PW;         // program window (words) in LDS
constexpr int BASE_ER = GST_BASE_ER;         // emit ring: one lane per parked circuit at evaluation time

// LDS of a chain: the model tables (shared by the wavefronts of a walk_base_kernel workgroup), then one private region
// per wavefront: save slots | emit ring | ring circuits | program window
__host__ __device__ inline size_t base_shared_doubles(int D, int n_gates, int n_effects) { return (size_t)n_effects * D + (size_t)n_gates * D * D; }
__host__ __device__ inline size_t base_wave_doubles(int D, int n_slots)
{
    return (size_t)(n_slots > 0 ? n_slots : 1) * 64 + (size_t)BASE_ER * (D + 1) + (BASE_ER + BASE_PW) / 2;
}

// global -> LDS copy by one wavefront, 8 loads in flight per lane (a load-wait-store loop costs one L2 round trip per
// 64 elements, which at 12 KB of gates + 4 KB of program is tens of microseconds of pure latency per task)
template <typename T, typename F>
__device__ __forceinline__ void stage_lds(T* dst, const int total, const int lane, F&& src)
{
    for (int base = 0; base < total; base += 64 * 8) {
        T t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = base + u * 64 + lane;
            t[u] = (k < total) ? src(k) : T(0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = base + u * 64 + lane;
            if (k < total) dst[k] = t[u];
        }
    }
}

// What one chain needs (a subset of WalkArgs, so that the fused FD kernel does not copy its whole argument block).
struct ChainArgs {
    const uint32_t* gprog;       // the task's walk program
    int32_t n_words;
    const int32_t* eff_ptr;
    const int32_t* eff_label;
    const int32_t* eff_dest;
    const double* rhos;
    double* out;                 // probabilities [nE]
    double* cache;               // base-state cache (may be NULL)
    int32_t multi_start, start0; // see WalkArgs
};

// PUB: results are PUBLISHED while other wavefronts of the same launch -- on any XCD, each with its own L2 -- consume
// them: states and probabilities are stored with system-scope (write-through) stores, and the consumer tells a value
// that has not arrived from one that has by the sentinel the destination was pre-filled with.  8-byte stores are
// single-copy atomic, every word is written exactly once, so no flag, fence or ordering is needed on either side and
// the chain never waits.
template <bool PUB>
__device__ __forceinline__ void chain_store(double* p, double x)
{
    if constexpr (PUB) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = x;
}

// The walk itself.  ldsE / ldsG: effects and transposed gates, already staged; wlds: this wavefront's private region
// (its program window must already hold the first BASE_PW words).  The caller has drained its staging stores.
template <int D, bool PUB>
__device__ __forceinline__ void base_chain_walk(const ChainArgs& a, const int n_slots, const double* ldsE, const double* ldsG,
                                                double* wlds, const int lane)
{
    static_assert(D == 4 || D == 16, "group broadcasts are DPP quad / row operations");
    constexpr int W = BASE_PW, ER = BASE_ER;
    constexpr int ES = D + 1;                // padded state stride of the emit ring (bank spread for lane-per-circuit reads)
    const int li = lane % D, grp = lane / D;
    double* const ering = wlds + (n_slots > 0 ? n_slots : 1) * 64;      // (a save slot holds one state per lane group)
    int32_t* const ering_circ = (int32_t*)(ering + ER * ES);
    uint32_t* const ldsP = (uint32_t*)(ering_circ + ER);
    const uint32_t* gprog = a.gprog;
    const int32_t n_words = a.n_words;

    // State-cache addressing.  Normal pass: every lane group holds the same state, group 0 stores it at
    // cache[id][D].  Multi-start pass (a.multi_start = number of start vectors, used for the backward states of the
    // analytic mode): lane group q walks from start vector a.start0 + q (RHO loads rhos[a.start0 + q] whatever its
    // argument) and stores at cache[id][component][a.start0 + q] -- the walk program, the gates and their order are the same,
    // so one pass propagates 64/D start vectors.
    double* const cache = a.cache;
    const bool multi = a.multi_start > 0;
    const int my_start = a.start0 + grp;
    const bool store_on = multi ? (my_start < a.multi_start) : (grp == 0);
    const int64_t node_stride = multi ? (int64_t)a.multi_start * D : (int64_t)D;
    // (multi-start states are stored [state][component][start]: the consumer reads all starts of a component at once)
    const int lane_off = multi ? (store_on ? li * a.multi_start + my_start : 0) : lane;
    double* const slot_lane = wlds + grp * D + li;           // save slot s of this lane group: + s * 64
    int32_t lo = 0;                          // words [lo, lo + W) are resident, word i at ldsP[i % W]
    int32_t pc = 0;                          // index of the current word
    int n_er = 0;                            // parked EMITs (wave-uniform)

#define BASE_WORD(i_) ldsP[(i_) & (W - 1)]
#define BASE_REFILL()                                                                                 \
    do {                                                                                              \
        if (pc - lo >= W / 2 + 8) {          /* the half behind pc is dead: bring in the next W/2 words */ \
            uint32_t* const half_ = ldsP + (lo & (W - 1));   /* lo is a multiple of W/2 */             \
            const int32_t g0_ = lo + W;                                                               \
            stage_lds(half_, W / 2, lane, [&](int k_) { return (g0_ + k_ < n_words) ? gprog[g0_ + k_] : 0u; }); \
            lo += W / 2;                                                                              \
        }                                                                                             \
    } while (0)
#define BASE_STORE_STATE(id_)                                                                         \
    do {                                                                                              \
        if (cache && store_on) chain_store<PUB>(&cache[(int64_t)(id_) * node_stride + lane_off], v);  \
    } while (0)
#define BASE_FLUSH_EMITS()                                                                            \
    do {                                                                                              \
        if (lane < n_er) {                                                                            \
            const int32_t circ_ = ering_circ[lane];                                                   \
            const int32_t x0_ = a.eff_ptr[circ_], x1_ = a.eff_ptr[circ_ + 1];                         \
            const double* st_ = ering + lane * ES;                                                    \
            for (int32_t x_ = x0_; x_ < x1_; x_++) {                                                  \
                const double* E_ = ldsE + a.eff_label[x_] * D;                                        \
                const int64_t dest_ = a.eff_dest[x_];                                                 \
                double p_ = 0.0;                                                                      \
                _Pragma("unroll") for (int i = 0; i < D; i++) p_ = p_ + E_[i] * st_[i];               \
                chain_store<PUB>(&a.out[dest_], p_);                                                  \
            }                                                                                         \
        }                                                                                             \
        n_er = 0;                                                                                     \
    } while (0)

    double v = 0.0;
    uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)BASE_WORD(0));
    for (;;) {
        const uint32_t op = GST_OP(w), arg = GST_ARG(w);
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            double c[D];
            {
                const double* G0 = ldsG + (int)arg * D * D + li;
#pragma unroll
                for (int j = 0; j < D; j++) c[j] = G0[j * D];
            }
            uint32_t g = arg;
            uint32_t p0 = BASE_WORD(pc + 1), p1 = BASE_WORD(pc + 2);   // this step's NODE marker and what follows
            for (;;) {
                const int32_t node_id = (int32_t)GST_ARG((uint32_t)__builtin_amdgcn_readfirstlane((int)p0));
                const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)p1);
                const bool more = (GST_OP(x) == GST_OP_APPLY);
                pc += 2;                                                // at `x`
                p0 = BASE_WORD(pc + 1); p1 = BASE_WORD(pc + 2);         // the next step's pair, one step ahead
                const double* Gn = ldsG + (int)(more ? GST_ARG(x) : g) * D * D + li;
                v = matvec_stream<D>(c, v, Gn);
                BASE_STORE_STATE(node_id);
                w = x;
                if (!more) break;
                g = GST_ARG(x);
                BASE_REFILL();
            }
            continue;                                                  // `w` holds the next instruction, at pc
        } else if (op == GST_OP_NODE) {
            BASE_STORE_STATE(arg);
        } else if (op == GST_OP_EMIT) {
            if (grp == 0) ering[n_er * ES + lane] = v;
            if (lane == 0) ering_circ[n_er] = (int32_t)arg;
            if (++n_er == ER) BASE_FLUSH_EMITS();
        } else if (op == GST_OP_SAVE) {
            slot_lane[arg * 64] = v;
        } else if (op == GST_OP_LOAD) {
            v = slot_lane[arg * 64];
        } else {  // GST_OP_RHO
            v = multi ? (store_on ? a.rhos[(int64_t)my_start * D + li] : 0.0) : a.rhos[(int64_t)arg * D + li];
        }
        pc++;
        BASE_REFILL();
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)BASE_WORD(pc));
    }
    BASE_FLUSH_EMITS();
#undef BASE_FLUSH_EMITS
#undef BASE_STORE_STATE
#undef BASE_REFILL
#undef BASE_WORD
}

}  // namespace gst
