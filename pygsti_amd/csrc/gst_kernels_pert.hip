// gst_kernels_pert.hip -- `walk_pert_kernel<D>`: finite-difference walks for perturbations that change a WHOLE object.
//
// The lane-per-model kernel of gst_kernels.hip rests on the reference's `full` parameterisation: a parameter step changes
// one dense element, so a lane carries one "special row".  A Lindblad parameter (CPTPLND, GLND, H+S: gst_set_lindblad)
// moves every element of its member's dense matrix -- exp(L(theta + eps e_p)) . U -- so the perturbed model of column p
// differs from the base model in one whole object, which gst_kernels_lindblad.hip has built on the device.  Walking
// every (program, perturbed model) pair independently (gst_fill_dprobs_models' way) costs 580 ms for the 1,920 columns
// of the 2Q design; this kernel brings back what made the `full` Jacobian cheap:
//   * CLEAN / DIRTY sharing with the base pass: a state whose path has not yet applied the perturbed gate (or does not
//     start from the perturbed preparation) is the base pass's state -- not computed, only its id is tracked; at the
//     first application of the perturbed gate the state comes from the base-state cache.  POVM parameters change no state
//     at all: their columns are dots of the cached states with the perturbed effects.  Clean EMITs of state-changing
//     columns are exact zeros ((p - p) / eps);
//   * 64/D MODELS PER WAVEFRONT: the chain kernel's D-lane groups (gst_chain.hpp: lane l holds component l % D, v_j
//     reaches its group with one DPP broadcast) each carry a DIFFERENT column -- D = 16: four parameters of the same
//     member -- so control flow stays wave-uniform (same perturbed object, same dirty flag) while the perturbed
//     coefficients differ per group.  A lane keeps its row of its column's perturbed gate in REGISTERS (D doubles): no
//     LDS copy per wavefront, ~4 KB of LDS per wavefront in all, many wavefronts per CU to hide the chain's latency.
// Arithmetic: as everywhere acc = 0.0; acc = acc + G[i][j] * v[j], ascending j, separate multiply and add; effect dots
// ascending from 0.0 -- the same operation order as the base pass, so a perturbation of exactly zero gives exactly the
// base probabilities and the quotient's noise is that of the perturbed members alone.
#include "gst_kernels.hpp"
#include "gst_chain.hpp"

#include "../../include/gstfwd.h"

namespace gst {

namespace {

#define GST_CONST __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const GST_CONST T* as_const(const T* p) { return (const GST_CONST T*)(p); }

constexpr int PERT_PW = 256;                 // program window (words) in LDS
constexpr int PERT_MAXW = 8;                 // wavefronts per workgroup at most (they share the staged base tables)
constexpr int PERT_WPB = 4;                  // ... and as launched: 156 VGPRs allow 3 wavefronts per SIMD = 12 per CU = 3 workgroups of 4
constexpr int PERT_MAXSLOT = 4;

template <int D> struct PertGeom {
    static constexpr int G = 64 / D;         // models (columns) per wavefront
    static constexpr int ER = (D == 16) ? 8 : 4;      // parked circuits per flush: ER * G <= 64 lanes
    static constexpr int ES = D + 1;         // padded state stride of the ring
};

}  // namespace

__host__ __device__ inline size_t pert_shared_doubles(int D, int n_gates, int n_effects) { return (size_t)n_effects * D + (size_t)n_gates * D * D; }
__host__ __device__ inline size_t pert_wave_doubles(int D, int n_slots)
{
    const int G = 64 / D, ER = (D == 16) ? 8 : 4;
    return (size_t)(n_slots > 0 ? n_slots : 1) * 64 + (size_t)ER * G * (D + 1) + (size_t)(ER + PERT_PW + 1) / 2;
}

namespace {

template <int D>
__global__ __launch_bounds__(64 * PERT_MAXW) void walk_pert_kernel(const PertArgs a, const int n_slots, const int wpb, const int64_t n_blocks)
{
    constexpr int G = PertGeom<D>::G, ER = PertGeom<D>::ER, ES = PertGeom<D>::ES, W = PERT_PW;
    extern __shared__ double lds[];          // effects | gates_t | per wavefront: save slots | ring states | ring circuits | program window
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane % D, grp = lane / D;
    int64_t blk = (int64_t)blockIdx.x * wpb + wv;
    const bool have = blk < n_blocks;
    if (!have) blk = 0;
    // work item = (dirty program, parameter wavefront): the program of (task, class of the wavefront's member)
    const uint32_t item_prog = a.item_prog[blk];
    const int32_t pw = a.item_pw[blk];
    double* const ldsE = lds;
    double* const ldsG = ldsE + a.n_effects * D;
    double* const wlds = lds + pert_shared_doubles(D, a.n_gates, a.n_effects) + (size_t)wv * pert_wave_doubles(D, n_slots);
    double* const ering = wlds + (n_slots > 0 ? n_slots : 1) * 64;
    int32_t* const ering_circ = (int32_t*)(ering + ER * G * ES);
    uint32_t* const ldsP = (uint32_t*)(ering_circ + ER);

    const int64_t pc0 = as_const(a.prog_off)[item_prog];
    const int32_t n_words = (int32_t)(as_const(a.prog_off)[item_prog + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    stage_lds(ldsP, W, lane, [&](int k) { return (k < n_words) ? gprog[k] : 0u; });
    {   // the shared base tables: every wavefront of the workgroup copies its share
        const int nG2 = a.n_gates * D * D, nE2 = a.n_effects * D;
        const int per = ((nG2 + wpb - 1) / wpb + 63) / 64 * 64, g0 = wv * per;
        stage_lds(ldsG + g0, (g0 < nG2) ? ((nG2 - g0 < per) ? nG2 - g0 : per) : 0, lane, [&](int k) { return a.gates_t[g0 + k]; });
        if (wv == 0) stage_lds(ldsE, nE2, lane, [&](int k) { return a.effects[k]; });
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (wpb > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
    if (!have) return;

    // ---- this wavefront's columns: up to G consecutive columns of ONE member ----------------------------------------------
    const int32_t kind = a.wave_kind[pw], obj = a.wave_obj[pw];
    const int32_t col0 = a.wave_col0[pw], ncols = a.wave_ncols[pw];
    const double* const my_pert = a.pert + (int64_t)(col0 + (grp < ncols ? grp : 0)) * a.pert_stride;   // (idle groups shadow column 0)
    double cp[D];                            // kind GATE: row li of my column's perturbed gate, cp[j] = G'[li][j]
#pragma unroll
    for (int j = 0; j < D; j++) cp[j] = 0.0;
    if (kind == GST_KIND_GATE) {
#pragma unroll
        for (int j = 0; j < D; j++) cp[j] = my_pert[j * D + li];      // (stored transposed: [j][i])
    }
    const double rp = (kind == GST_KIND_RHO) ? my_pert[li] : 0.0;
    const int32_t pgate = (kind == GST_KIND_GATE) ? obj : -1;

    double* const slot_lane = wlds + grp * D + li;
    int32_t lo = 0, pc = 0;
    int n_er = 0;
    double v = 0.0;
    const double* const cache = a.base_cache;

#define PERT_WORD(i_) ldsP[(i_) & (W - 1)]
#define PERT_REFILL()                                                                                 \
    do {                                                                                              \
        if (pc - lo >= W / 2 + 8) {                                                                   \
            uint32_t* const half_ = ldsP + (lo & (W - 1));                                            \
            const int32_t g0_ = lo + W;                                                               \
            stage_lds(half_, W / 2, lane, [&](int k_) { return (g0_ + k_ < n_words) ? gprog[g0_ + k_] : 0u; }); \
            lo += W / 2;                                                                              \
        }                                                                                             \
    } while (0)
    // Parked EMITs: at flush lane (e, q) = e * G + q evaluates circuit e for column q -- every outcome's dot product in
    // the reference's order -- and writes (p - p_base) / eps.
#define PERT_FLUSH()                                                                                  \
    do {                                                                                              \
        const int e_ = lane / G, q_ = lane % G;                                                       \
        if (e_ < n_er && q_ < ncols) {                                                                \
            const int32_t circ_ = ering_circ[e_];                                                     \
            const int32_t x0_ = a.eff_ptr[circ_], x1_ = a.eff_ptr[circ_ + 1];                         \
            const double* st_ = ering + (e_ * G + q_) * ES;                                           \
            const int64_t dcol_ = a.col_dest[col0 + q_];                                              \
            for (int32_t x_ = x0_; x_ < x1_; x_++) {                                                  \
                const int64_t dest_ = a.eff_dest[x_];                                                 \
                const double* E_ = ldsE + a.eff_label[x_] * D;                                        \
                double p_ = 0.0;                                                                      \
                _Pragma("unroll") for (int i = 0; i < D; i++) p_ = p_ + E_[i] * st_[i];               \
                a.out[dest_ * a.ld + dcol_] = (p_ - a.pbase[dest_]) / a.eps;                          \
            }                                                                                         \
        }                                                                                             \
        n_er = 0;                                                                                     \
    } while (0)

    uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)PERT_WORD(0));
    for (;;) {
        const uint32_t op = GST_OP(w), arg = GST_ARG(w);
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            // The perturbed gate's coefficients live in registers (cp), the others stream in from LDS one step ahead (c).
            // Two multiply paths behind a wave-uniform branch: copying cp into c, which a select between them becomes,
            // costs 32 extra VALU moves per step -- two thirds of the step's arithmetic again.
            double c[D];
            bool hit = (int32_t)arg == pgate;
            if (!hit) {
                const double* G0 = ldsG + (int)arg * D * D + li;
#pragma unroll
                for (int j = 0; j < D; j++) c[j] = G0[j * D];
            } else {
#pragma unroll
                for (int j = 0; j < D; j++) c[j] = 0.0;
            }
            uint32_t g = arg;
            uint32_t p1 = PERT_WORD(pc + 1);
            for (;;) {
                const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)p1);
                const bool more = (GST_OP(x) == GST_OP_APPLY);
                pc += 1;
                p1 = PERT_WORD(pc + 1);                                 // the word after the next one, one step ahead
                const uint32_t gn = more ? GST_ARG(x) : g;
                const bool hitn = (int32_t)gn == pgate;
                double bv[D];
                bcast_all<D, 0>(bv, v);
                __builtin_amdgcn_sched_barrier(0);
                if (hit) {
#pragma unroll
                    for (int j = 0; j < D; j++) bv[j] = cp[j] * bv[j];
                    asm volatile("" ::: "memory");
                } else {
#pragma unroll
                    for (int j = 0; j < D; j++) bv[j] = c[j] * bv[j];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!hitn) {
                    const double* Gn = ldsG + (int)gn * D * D + li;
#pragma unroll
                    for (int j = 0; j < D; j++) c[j] = Gn[j * D];
                }
                __builtin_amdgcn_sched_barrier(0);
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < D; j++) acc = acc + bv[j];
                __builtin_amdgcn_sched_barrier(0);
                v = acc;
                w = x;
                if (!more) break;
                g = gn;
                hit = hitn;
                PERT_REFILL();
            }
            continue;                                                  // `w` holds the next instruction, at pc
        } else if (op == GST_OP_CACHE) {
            v = cache[(int64_t)arg * D + li];                           // a perturbed excursion starts from the base pass's state
        } else if (op == GST_OP_EMIT) {
            ering[(n_er * G + grp) * ES + li] = v;
            if (lane == 0) ering_circ[n_er] = (int32_t)arg;
            if (++n_er == ER) PERT_FLUSH();
        } else if (op == GST_OP_SAVE) {
            slot_lane[arg * 64] = v;
        } else if (op == GST_OP_LOAD) {
            v = slot_lane[arg * 64];
        } else {  // GST_OP_RHO: the perturbed preparation (a program of a rho class holds no other)
            v = rp;
        }
        pc++;
        PERT_REFILL();
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)PERT_WORD(pc));
    }
    PERT_FLUSH();
#undef PERT_FLUSH
#undef PERT_REFILL
#undef PERT_WORD
}

// out[e][col_dest[c]] = 0 for every element e and every listed column c: the entries the walks never touch are exact
// zeros ((p - p) / eps), written once at streaming rate instead of one EMIT at a time.
__global__ __launch_bounds__(256) void zero_columns_kernel(double* out, int64_t ld, int64_t nE, const int32_t* col_dest, int32_t n_cols, int32_t dense0)
{
    const int64_t total = nE * (int64_t)n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / n_cols;
        const int32_t c = (int32_t)(i - e * n_cols);
        out[e * ld + (dense0 >= 0 ? dense0 + c : col_dest[c])] = 0.0;
    }
}

// Columns of POVM parameters: an effect perturbation changes no propagated state, so column c is, for every circuit,
// (E'_c . F - p) / eps on the circuit's final base state F for the outcomes of the perturbed POVM's effects and an exact
// zero for any other outcome.  One thread per (circuit, column); same ascending dot product as every EMIT.
template <int D>
__global__ __launch_bounds__(256) void effect_columns_kernel(const PertArgs a, const int32_t* circ_leaf, int64_t n_circuits, int32_t obj,
                                                             int32_t n_eff, int32_t col0, int32_t ncols)
{
    // one wavefront per circuit, its lanes take the POVM member's columns col0 .. col0 + ncols
    const int64_t circ = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (circ >= n_circuits) return;
    const int lane = threadIdx.x & 63;
    const double* F = a.base_cache + (int64_t)circ_leaf[circ] * D;
    double f[D];
#pragma unroll
    for (int i = 0; i < D; i++) f[i] = F[i];
    const int32_t x0 = a.eff_ptr[circ], x1 = a.eff_ptr[circ + 1];
    for (int32_t q = lane; q < ncols; q += 64) {
        const double* pe = a.pert + (int64_t)(col0 + q) * a.pert_stride;
        const int64_t dcol = a.col_dest[col0 + q];
        for (int32_t x = x0; x < x1; x++) {
            const int32_t lbl = a.eff_label[x];
            const int64_t dest = a.eff_dest[x];
            if (lbl < obj || lbl >= obj + n_eff) { a.out[dest * a.ld + dcol] = 0.0; continue; }
            const double* E = pe + (lbl - obj) * D;
            double p = 0.0;
#pragma unroll
            for (int i = 0; i < D; i++) p = p + E[i] * f[i];
            a.out[dest * a.ld + dcol] = (p - a.pbase[dest]) / a.eps;
        }
    }
}

}  // namespace

bool pert_kernel_fits(int D, int n_gates, int n_effects, int n_slots)
{
    if ((D != 4 && D != 16) || n_slots > PERT_MAXSLOT) return false;
    return (pert_shared_doubles(D, n_gates, n_effects) + pert_wave_doubles(D, n_slots)) * sizeof(double) <= 150 * 1024;
}

hipError_t launch_walk_pert(int D, const PertArgs& a, int64_t n_items, int n_slots, hipStream_t stream)
{
    const int64_t blocks = n_items;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL || !pert_kernel_fits(D, a.n_gates, a.n_effects, n_slots)) return hipErrorInvalidValue;
    const size_t sh = pert_shared_doubles(D, a.n_gates, a.n_effects) * sizeof(double);
    const size_t pw = pert_wave_doubles(D, n_slots) * sizeof(double);
    int wpb = PERT_WPB;
    while (wpb > 1 && sh + wpb * pw > 64 * 1024) wpb /= 2;          // stay within the default dynamic-LDS limit: several workgroups per CU
    const size_t bytes = sh + wpb * pw;
    (void)hipGetLastError();
    if (bytes > 64 * 1024) {
        const void* fn = D == 16 ? (const void*)walk_pert_kernel<16> : (const void*)walk_pert_kernel<4>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (unsigned)((blocks + wpb - 1) / wpb);
    if (D == 16) hipLaunchKernelGGL((walk_pert_kernel<16>), dim3(grid), dim3(64 * wpb), bytes, stream, a, n_slots, wpb, blocks);
    else hipLaunchKernelGGL((walk_pert_kernel<4>), dim3(grid), dim3(64 * wpb), bytes, stream, a, n_slots, wpb, blocks);
    return hipGetLastError();
}

hipError_t launch_zero_columns(double* out, int64_t ld, int64_t nE, const int32_t* col_dest, int32_t n_cols, int32_t dense0, hipStream_t stream)
{
    if (nE <= 0 || n_cols <= 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(zero_columns_kernel, dim3(4096), dim3(256), 0, stream, out, ld, nE, col_dest, n_cols, dense0);
    return hipGetLastError();
}

hipError_t launch_effect_columns(int D, const PertArgs& a, const int32_t* circ_leaf, int64_t n_circuits, int32_t obj, int32_t n_eff,
                                 int32_t col0, int32_t ncols, hipStream_t stream)
{
    if (n_circuits <= 0 || ncols <= 0) return hipSuccess;
    (void)hipGetLastError();
    const unsigned grid = (unsigned)((n_circuits + 3) / 4);
    if (D == 16) hipLaunchKernelGGL((effect_columns_kernel<16>), dim3(grid), dim3(256), 0, stream, a, circ_leaf, n_circuits, obj, n_eff, col0, ncols);
    else if (D == 4) hipLaunchKernelGGL((effect_columns_kernel<4>), dim3(grid), dim3(256), 0, stream, a, circ_leaf, n_circuits, obj, n_eff, col0, ncols);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace gst
