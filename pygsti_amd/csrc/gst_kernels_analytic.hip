// gst_kernels_analytic.hip -- exact (analytic) Jacobian of the outcome probabilities.
//
// What MatrixForwardSimulator computes (matrixforwardsim.py:1059-1140: dp = E.dG.rho + E.G.drho + dE.G.rho
// over an eval tree of process-matrix products), restated for the state-vector picture:
//     p = E^T G_L ... G_1 rho
//     dp/dG_g[a][b] = sum_{k: g_k = g}  B_k[a] * F_{k-1}[b]        F_k = G_k ... G_1 rho   (forward states)
//     dp/drho[b]    = B_0[b]                                        B_k^T = E^T G_L ... G_{k+1} (backward vectors)
//     dp/dE[a]      = F_L[a]
// The forward states of EVERY circuit prefix are already in HBM: the base pass writes them to the
// base-state cache, indexed by state id, and the plan knows each state's parent id and gate
// (HostPlan::node_parent / node_sym / circ_leaf).  So one wavefront takes one circuit, walks the state ids from
// the circuit's last state up to its rho state, and on the way
//   * propagates the backward vector of up to 64/D outcomes at once (lane = (outcome group, index a)):
//     B'[a'] = sum_a G[a][a'] B[a]  -- D DPP row-broadcasts + D FMAs, the lane's column of G read from LDS;
//   * accumulates the rank-1 update acc[g_k][b] += B[a] * F_{k-1}[b] -- D FMAs whose F operand is a
//     wave-uniform SGPR pair loaded straight from the cache (s_load_dwordx16);
// and finally writes the D x D block of every gate plus the SPAM columns of its Jacobian rows: lane (.., a)
// holds row a of each block, i.e. D consecutive columns.  fp64 FMA throughout (this mode is judged against an
// analytic reference to <= 1e-8; the bit-parity contract of the FD mode does not apply).
// Roofline: the Jacobian write (8*nE*nP bytes) against ~5 FMA-issue cycles per element and gate application.
#include "gst_kernels.hpp"

#include <type_traits>

#include "../../include/gstfwd.h"

namespace gst {

#define GST_CONST __attribute__((address_space(4)))
typedef const GST_CONST double* cdouble_p;
typedef const GST_CONST int32_t* ci32_p;
template <typename T>
__device__ __forceinline__ const GST_CONST T* as_const(const T* p) { return (const GST_CONST T*)(p); }

// Broadcast element SRC of every D-lane group to the whole group: D = 16 -> DPP row_newbcast (gfx90a+),
// D = 4 -> DPP quad_perm [SRC,SRC,SRC,SRC].  One v_mov_b32_dpp per half of the double, no LDS.
template <int D, int SRC>
__device__ __forceinline__ double row_bcast(double x)
{
    if constexpr (D == 16) {
        return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + SRC, 0xf, 0xf, true);     // one v_mov_b64_dpp row_newbcast
    } else {
        constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), ctrl, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), ctrl, 0xf, 0xf, true);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
}

// One backward step for the D-lane group of an outcome.  Lane b holds B[b] and F_{k-1}[b]:
//   Bn[b]      = sum_a G[a][b] * B[a]          (column b of G from LDS, B[a] by DPP broadcast)
//   acc[g][a] += B[a] * F[b]                   (entry (a, b) of gate g's block: lane b keeps column b)
// The D broadcasts serve both updates.
template <int D, int J>
struct BackStep {
    static __device__ __forceinline__ void run(const double* col, double B, double F, double& Bn, double (&acc)[D])
    {
        const double Ba = row_bcast<D, J>(B);
        Bn = __builtin_fma(col[J], Ba, Bn);
        acc[J] = __builtin_fma(Ba, F, acc[J]);
        BackStep<D, J + 1>::run(col, B, F, Bn, acc);
    }
    static __device__ __forceinline__ void run_noacc(const double* col, double B, double& Bn)
    {
        Bn = __builtin_fma(col[J], row_bcast<D, J>(B), Bn);
        BackStep<D, J + 1>::run_noacc(col, B, Bn);
    }
};
template <int D>
struct BackStep<D, D> {
    static __device__ __forceinline__ void run(const double*, double, double, double&, double (&)[D]) {}
    static __device__ __forceinline__ void run_noacc(const double*, double, double&) {}
};

constexpr int ANA_PAD = 2;   // LDS row padding (doubles): row stride D+2 keeps the wide reads conflict-free
constexpr int ANA_P = 4;     // steps per chunk; the next chunk's F / gate symbols are in flight during the current one

template <int D, int NG, int WPS>
__global__ __launch_bounds__(256, WPS) void analytic_dprobs_kernel(const AnaArgs a)
{
    static_assert(D == 16 || D == 4, "one DPP row (D = 16) or quad (D = 4) per outcome");
    extern __shared__ __attribute__((aligned(16))) double lds[];   // gates_t, padded: lds[(g*D + b)*(D+PAD) + a] = G[a][b]
    constexpr int RS = D + ANA_PAD;
    constexpr int P = ANA_P;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_in_block = tid >> 6;
    const int eg = lane / D;                 // outcome group within the wavefront
    const int bi = lane % D;                 // this lane's index b (column of every D x D block it accumulates)
    constexpr int EG = 64 / D;

    for (int k = tid; k < a.n_gates * D * D; k += 256) {
        const int g = k / (D * D), r = (k / D) % D, c = k % D;
        lds[(g * D + r) * RS + c] = a.gates_t[k];
    }
    __syncthreads();

    const int64_t total_waves = (int64_t)gridDim.x * 4;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave_in_block; c < a.n_circuits; c += total_waves) {
        const int32_t x0 = as_const(a.eff_ptr)[c], x1 = as_const(a.eff_ptr)[c + 1];
        const int32_t leaf = as_const(a.circ_leaf)[c];
        for (int32_t xb = x0; xb < x1; xb += EG) {
            const int32_t x = xb + eg;
            const bool valid = x < x1;
            const int32_t e = a.eff_label[valid ? x : x0];
            const int64_t dest = a.eff_dest[valid ? x : x0];
            for (int g0 = 0; g0 < a.n_gates; g0 += NG) {            // gate groups (one pass when n_gates <= NG)
                double acc[NG][D];
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int j = 0; j < D; j++) acc[g][j] = 0.0;
                double B = a.effects[(int64_t)e * D + bi];
                int32_t id = leaf;
                if (g0 == 0 && valid) {      // dp/dE[a] = F_L[a]; the other effects' columns are exact zeros
                    const double FL = a.base_cache[(int64_t)id * D + bi];
                    for (int e2 = 0; e2 < a.n_effects; e2++) {
                        const int32_t ce = a.colmap_eff[e2 * D + bi];
                        if (ce >= 0) a.out[dest * a.ld + ce] = (e2 == e) ? FL : 0.0;
                    }
                }
                // Walk up in RUNS of consecutive state ids (chains of the trie get consecutive ids, so the
                // parent of id-t is id-t-1 for t < run): inside a run every address is known up front and the
                // next chunk's forward states and gate symbols are prefetched while the current chunk computes.
                for (;;) {
                    int32_t run = as_const(a.node_run)[id];          // >= 1 steps with consecutive parents, 0: see below
                    int32_t par0 = id - 1;
                    if (run == 0) {
                        par0 = as_const(a.node_parent)[id];
                        if (par0 < 0) break;                         // `id` is the rho state
                        run = 1;                                     // a jump: one step with an explicit parent
                    }
                    // step t of the run: node id-t (t = 0) / its consecutive ancestors, parent par0-t
                    double Fc[P], Fn[P];
                    int32_t symc, symn;
#define ANA_LOAD(F_, sym_, t0_)                                                                           \
                    do {                                                                                  \
                        _Pragma("unroll") for (int u = 0; u < P; u++) {                                   \
                            const int32_t t_ = (t0_) + u < run ? (t0_) + u : run - 1;                     \
                            F_[u] = a.base_cache[(int64_t)(par0 - t_) * D + bi];                          \
                        }                                                                                 \
                        const int32_t tl_ = (t0_) + (lane & (P - 1)) < run ? (t0_) + (lane & (P - 1)) : run - 1; \
                        sym_ = a.node_sym[id - tl_];                                                      \
                    } while (0)
                    ANA_LOAD(Fc, symc, 0);
                    for (int32_t t0 = 0; t0 < run; t0 += P) {
                        if (t0 + P < run) ANA_LOAD(Fn, symn, t0 + P);
#pragma unroll
                        for (int u = 0; u < P; u++) {
                            if (t0 + u < run) {
                                const int32_t sym = __builtin_amdgcn_readlane(symc, u);
                                // column b of G, G[.][b]: 16-byte reads at a 144-byte lane stride are bank-conflict free
                                typedef double d2_t __attribute__((ext_vector_type(2)));
                                const d2_t* colp = (const d2_t*)__builtin_assume_aligned(lds + (sym * D + bi) * RS, 16);
                                double col[D];
#pragma unroll
                                for (int j = 0; j < D; j += 2) { const d2_t t = colp[j / 2]; col[j] = t.x; col[j + 1] = t.y; }
                                double Bn = 0.0;
                                const int gl = sym - g0;
                                bool done = false;
#pragma unroll
                                for (int g = 0; g < NG; g++)
                                    if (g == gl) { BackStep<D, 0>::run(col, B, Fc[u], Bn, acc[g]); done = true; }
                                if (!done) BackStep<D, 0>::run_noacc(col, B, Bn);
                                B = Bn;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < P; u++) Fc[u] = Fn[u];
                        symc = symn;
                    }
#undef ANA_LOAD
                    id = par0 - (run - 1);
                }
                // `id` is the rho state: dp/drho[b] = B_0[b]; other preparations' columns are zeros
                if (valid) {
                    if (g0 == 0) {
                        const int32_t rsym = as_const(a.node_sym)[id];
                        for (int r2 = 0; r2 < a.n_rhos; r2++) {
                            const int32_t cr = a.colmap_rho[r2 * D + bi];
                            if (cr >= 0) a.out[dest * a.ld + cr] = (r2 == rsym) ? B : 0.0;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        if (g0 + g >= a.n_gates) break;
                        const int32_t c0 = as_const(a.gate_col0)[g0 + g];
                        if (c0 >= 0) {           // D*D consecutive columns: entry (j, b) at c0 + j*D + b -> lanes b coalesce
                            double* o = a.out + dest * a.ld + c0 + bi;
#pragma unroll
                            for (int j = 0; j < D; j++) o[j * D] = acc[g][j];
                        } else if (c0 == -1) {   // arbitrary subset / order: per-element column map
                            const int32_t* cm = a.colmap_gate + (int64_t)(g0 + g) * D * D + bi;
#pragma unroll
                            for (int j = 0; j < D; j++) { const int32_t cc = cm[j * D]; if (cc >= 0) a.out[dest * a.ld + cc] = acc[g][j]; }
                        }                        // c0 == -2: no parameter of this gate was requested
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// analytic_mfma_kernel (D = 16): exact Jacobian rows from BOTH state caches on the matrix cores.
//   dp/dG_g[a][b] = sum over the applications k of gate g in the circuit of  B_k[a] * F_{k-1}[b]
// with F the forward states (base pass over the prefix trie) and B_k = (G_n ... G_{k+1})^T E the backward states
// (the same chain kernel run over the SUFFIX trie of the reversed circuits with the transposed gates, one lane group
// per effect).  Nothing is propagated here: for a circuit, an outcome and a gate the 16 x 16 block is a
// [16 x n_k] x [n_k x 16] product over that gate's applications, i.e. n_k / 4 v_mfma_f64_16x16x4_f64 whose operands
// are 128-byte state vectors gathered by the host-built pair tables (pair_f / pair_r, grouped by circuit and gate):
// the A operand wants lane l to hold B_{k(l>>4)}[l&15], the B operand F_{k(l>>4)-1}[l&15] -- both are straight
// coalesced reads of cache lines, no LDS, no transposition.  One F gather feeds the blocks of 4 outcomes.
// A wavefront takes circuits from a shared counter (depths range from 1 to 1030) and writes whole 16-column row
// segments of the Jacobian (D-matrix layout: lane l, register r -> row (l>>4) + 4r, column l&15).
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int ANA_MFMA_M = 1;      // chunks of 4 gate applications per pipeline block
// Three wavefronts per SIMD: with the gathers a block ahead of their MFMAs the fourth wavefront buys nothing, and the
// two register sets of the pipeline do not fit 128 VGPRs next to the 8 accumulator tiles of a two-circuit item.
#ifndef GST_ANA_WPS
#define GST_ANA_WPS 3
#endif
constexpr int ANA_MFMA_WPS = GST_ANA_WPS;
#ifndef GST_ANA_STREAM_M
#define GST_ANA_STREAM_M 1
#endif
constexpr int ANA_STREAM_M = GST_ANA_STREAM_M;

// WIDE: a state cache of 4 GB or more -- 64-bit per-lane byte offsets instead of 32-bit ones (AnaArgs::wide)
#if GST_ANA_TIMING
__device__ unsigned long long ana_dbg[16];     // development build only (tools/ana_phases.py): per-phase wavefront cycles
#define AT_NOW() __builtin_amdgcn_s_memtime()
#endif
template <bool WIDE>
__global__ __launch_bounds__(256, ANA_MFMA_WPS) void analytic_mfma_kernel(const AnaArgs a)
{
#if GST_ANA_TIMING
    unsigned long long at[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long at_start = AT_NOW();
#endif
    constexpr int D = 16, NX = 4;
    using off_t = typename std::conditional<WIDE, uint64_t, uint32_t>::type;
    const int lane = threadIdx.x & 63;
    const int kk = lane >> 4, i = lane & 15;
    const int nE = a.n_effects, nG = a.n_gates;
    // structural zeros the destination still holds are not stored again; a tracked destination's claim carries a device
    // word that a row scaling with a non-finite factor (0 * inf) has cleared (gst_track.cpp)
    const bool zeros_resident = a.zeros_resident && (!a.zeros_ok || *(const volatile uint32_t*)a.zeros_ok != 0u);
    const uint32_t fstride = a.fwd_stride ? a.fwd_stride : (uint32_t)(D * 8);
    const uint32_t rstride = a.rev_stride ? a.rev_stride : (uint32_t)nE * (D * 8);
    // uniform 64-bit bases + 32-bit per-lane byte offsets (both caches are < 4 GB, checked on the host): one offset
    // register serves the F load / the NX backward loads of a gathered application
    const char* const fb = (const char*)a.base_cache;
    const char* const rb = (const char*)a.rev_cache;
    const uint32_t lane_b = (uint32_t)i * 8u;
    const uint32_t lane_r = (uint32_t)i * (uint32_t)nE * 8u;       // backward cache is [state][component][effect]
    typedef double d2_t __attribute__((ext_vector_type(2)));

    // outcome labels / destinations of one circuit's block of (up to) NX outcomes, made wave-uniform
    auto outcomes = [&](int32_t x0, int32_t xb, int nx, int32_t& e_l, int64_t& dest_l, int32_t (&e_u)[NX], int64_t (&dest_u)[NX]) {
        const bool on = kk < nx;
        e_l = a.eff_label[on ? xb + kk : x0];
        dest_l = a.eff_dest[on ? xb + kk : x0];
#pragma unroll
        for (int x = 0; x < NX; x++) {
            e_u[x] = __builtin_amdgcn_readlane(e_l, 16 * x);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(dest_l & 0xffffffffLL), 16 * x);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(dest_l >> 32), 16 * x);
            dest_u[x] = (int64_t)(((uint64_t)hi << 32) | lo);
        }
    };
    // dp/dE[a] = F_n[a] for the outcome's own effect, exact zeros for the others; dp/drho[b] = B_0[b]
    // (Hessian rows: the same with derivative states, one of the two families zeroed, the second launch adding)
    auto spam_columns = [&](int64_t c, int32_t e_l, int64_t dest_l) {
        const int32_t fleaf = as_const(a.circ_leaf)[c], rleaf = as_const(a.rev_leaf)[c], rsym = as_const(a.circ_rho)[c];
        const double FL = a.eff_zero ? 0.0 : *(const double*)((const char*)a.base_cache + (int64_t)fleaf * fstride + i * 8);
        for (int e2 = 0; e2 < nE; e2++) {
            const int32_t ce = a.colmap_eff[e2 * D + i];
            if (ce >= 0) {
                double* o = a.out + dest_l * a.ld + ce;
                const double val = (e2 == e_l) ? FL : 0.0;
                *o = a.accumulate ? *o + val : val;
            }
        }
        const double B0 = a.rho_zero ? 0.0 : *(const double*)((const char*)a.rev_cache + (int64_t)rleaf * rstride + ((int64_t)i * nE + e_l) * 8);
        for (int r2 = 0; r2 < a.n_rhos; r2++) {
            const int32_t cr = a.colmap_rho[r2 * D + i];
            if (cr >= 0) {
                double* o = a.out + dest_l * a.ld + cr;
                const double val = (r2 == rsym) ? B0 : 0.0;
                *o = a.accumulate ? *o + val : val;
            }
        }
    };
    // acc[x] += sum over the applications [p0, p1) of the pair tables of B_k (outcome x) (x) F_{k-1}, as a three-stage
    // pipeline over blocks of M chunks (4*M applications): while the MFMAs of block b run, the state vectors of block
    // b + 1 and the pair indices of block b + 2 are in flight (two register sets; the loop is unrolled by two so that
    // nothing is copied).  Two things make the overlap real: no load sits under a branch (the compiler counts outstanding
    // loads exactly only on straight-line code; blocks past the end re-read the last pair, always a valid address, and
    // multiply by a zeroed F), and scheduling fences keep the three stages in program order -- left alone, the
    // scheduler hoists a stage's index loads in front of the gathers that consume them and waits for them at once.
#define DB_FENCE() __builtin_amdgcn_sched_barrier(0)
    auto sweep = [&](auto m_tag, auto fast_tag, int64_t p0, int64_t p1, const int32_t (&e_u)[NX], d4_t (&acc)[NX]) {
        constexpr int M = decltype(m_tag)::value;
        constexpr bool FAST4 = decltype(fast_tag)::value;
        if (p1 <= p0) return;
        const int32_t n = (int32_t)(p1 - p0);
        const int32_t nb = (n + 4 * M - 1) / (4 * M);
        const int32_t* const pf = a.pair_f + p0;
        const int32_t* const pr = a.pair_r + p0;
        auto idx_at = [&](int32_t blk, int32_t (&f)[M], int32_t (&r)[M]) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const int32_t pi = (blk * M + m) * 4 + kk;
                const int32_t pc = pi < n ? pi : n - 1;
                f[m] = pf[pc]; r[m] = pr[pc];
            }
        };
        auto gather = [&](const int32_t (&f)[M], const int32_t (&r)[M], double (&Fv)[M], double (&Bv)[M][NX]) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const off_t fo = (off_t)(uint32_t)f[m] * fstride + lane_b;
                const off_t ro = (off_t)(uint32_t)r[m] * rstride + lane_r;
                Fv[m] = *(const double*)(fb + fo);
                if constexpr (FAST4) {
                    const d2_t* q2 = (const d2_t*)__builtin_assume_aligned(rb + ro, 16);
                    const d2_t t0 = q2[0], t1 = q2[1];
                    Bv[m][0] = t0.x; Bv[m][1] = t0.y; Bv[m][2] = t1.x; Bv[m][3] = t1.y;
                } else {
#pragma unroll
                    for (int x = 0; x < NX; x++) Bv[m][x] = *(const double*)(rb + ro + (uint32_t)e_u[x] * 8u);
                }
            }
        };
        auto mma = [&](int32_t blk, const double (&Fv)[M], const double (&Bv)[M][NX]) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const double Fz = ((blk * M + m) * 4 + kk < n) ? Fv[m] : 0.0;
#pragma unroll
                for (int x = 0; x < NX; x++) acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bv[m][x], Fz, acc[x], 0, 0, 0);
            }
        };
        int32_t fA[M], rA[M], fB[M], rB[M];
        double F0[M], B0[M][NX], F1[M], B1[M][NX];
        idx_at(0, fA, rA);
        idx_at(1, fB, rB);
        gather(fA, rA, F0, B0);
        for (int32_t b = 0; b < nb; b += 2) {
            DB_FENCE();
            idx_at(b + 2, fA, rA);
            DB_FENCE();
            gather(fB, rB, F1, B1);
            DB_FENCE();
            mma(b, F0, B0);
            DB_FENCE();
            idx_at(b + 3, fB, rB);
            DB_FENCE();
            gather(fA, rA, F0, B0);
            DB_FENCE();
            mma(b + 1, F1, B1);
            DB_FENCE();
        }
    };
    // a circuit's 16 x 16 block of gate g for (up to) NX outcomes -> Jacobian rows (D-matrix layout of the MFMA)
    auto store_block = [&](int g, int32_t c0, int nx, const int64_t (&dest_u)[NX], const d4_t (&acc)[NX]) {
#pragma unroll
        for (int x = 0; x < NX; x++) {
            if (x >= nx) break;
            if (c0 >= 0) {           // D*D consecutive columns: entry (row, col) at c0 + row*D + col
                double* o = a.out + dest_u[x] * a.ld + c0 + kk * D + i;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (a.accumulate) o[4 * r * D] = o[4 * r * D] + acc[x][r];
                    else __builtin_nontemporal_store(acc[x][r], &o[4 * r * D]);     // written once, never re-read here: do not displace the state caches in L2
                }
            } else {                 // arbitrary subset / order: per-element column map
                const int32_t* cm = a.colmap_gate + (int64_t)g * D * D + kk * D + i;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int32_t cc = cm[4 * r * D];
                    if (cc >= 0) { double* o = a.out + dest_u[x] * a.ld + cc; *o = a.accumulate ? *o + acc[x][r] : acc[x][r]; }
                }
            }
        }
    };

    // Work distribution: the suffix-ordered list of work items (a circuit, or two circuits that end with the same
    // gates) is cut into 8 contiguous ranges of equal work, one per XCD (workgroup b runs on XCD b % 8 -- observed
    // dispatch rule, used for locality only): neighbours in that order share their backward chains and, over a longer
    // stretch, the forward chains of their germ, so both stay in that XCD's 4 MB L2 instead of being pulled from HBM
    // by all eight.  A wavefront that finds its range empty helps the next.
    // Group fetch (a.group_fetch): the four wavefronts of a workgroup take four CONSECUTIVE items together.  The kernel is
    // bound by the L1's outstanding misses (TCP_PENDING_STALL_CYCLES: 69 % of the launch; 107 M line requests at 613
    // cycles each, profiles/r02_analytic_mix.json), not by the latency one wavefront sees -- prefetching the gathers a
    // block ahead changes nothing -- and neighbours of the item order read the same backward vectors: started together
    // on one CU, part of their requests merge in that CU's L1 instead of each holding a miss slot (-11 % on the launch).
    const int xcd = blockIdx.x & 7;
    __shared__ uint32_t s_fetch[2];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int probe = 0;;) {
        int64_t ci;
#if GST_ANA_TIMING
        const unsigned long long t_f0 = AT_NOW();
#endif
        if (a.group_fetch) {
            __syncthreads();                   // (everybody has read the previous group's s_fetch)
            if (threadIdx.x == 0) {
                uint32_t base = 0xffffffffu, end = 0;
                for (; probe < 8; probe++) {
                    const int rg = (xcd + probe) & 7;
                    const uint32_t rb = a.range_begin[rg], re = a.range_begin[rg + 1];
                    const uint32_t cu = atomicAdd(a.work_counter + rg, 4u);
                    if ((uint64_t)rb + cu < (uint64_t)re) { base = rb + cu; end = re; break; }
                }
                s_fetch[0] = base; s_fetch[1] = end;
            }
            __syncthreads();
            const uint32_t g_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_fetch[0]);
            const uint32_t g_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_fetch[1]);
            if (g_base == 0xffffffffu) break;
            ci = (int64_t)g_base + wv;
            if (ci >= (int64_t)g_end) continue;
        } else {
            if (probe >= 8) break;
            const int rg = (xcd + probe) & 7;
            const uint32_t r_begin = as_const(a.range_begin)[rg], r_end = as_const(a.range_begin)[rg + 1];
            uint32_t cu = 0;
            if (lane == 0) cu = atomicAdd(a.work_counter + rg, 1u);
            ci = (int64_t)r_begin + (uint32_t)__builtin_amdgcn_readfirstlane((int)cu);
            if (ci >= (int64_t)r_end) { probe++; continue; }
        }
#if GST_ANA_TIMING
        const unsigned long long t_f1 = AT_NOW();
        at[0] += t_f1 - t_f0; at[9] += 1;
#endif
        const int64_t c = as_const(a.circ_order)[ci];
        const int64_t c2 = a.circ_partner ? (int64_t)as_const(a.circ_partner)[ci] : -1;
        if (c2 >= 0) {
            // Two circuits whose last applications coincide (same germ power and measurement fiducial, different
            // preparation fiducial): over that common tail the backward states are THE SAME vectors, so they are
            // gathered once and multiplied with both circuits' forward states -- 128 + 128 + 512 bytes per application
            // pair instead of 2 x 640.  (The host pairs only circuits with the plain 4-outcome block.)
            int32_t e_l, e_l2, e_u[NX], e_u2[NX];
            int64_t dest_l, dest_l2, dest_u[NX], dest_u2[NX];
            const int32_t x0 = as_const(a.eff_ptr)[c], x02 = as_const(a.eff_ptr)[c2];
            outcomes(x0, x0, NX, e_l, dest_l, e_u, dest_u);
            outcomes(x02, x02, NX, e_l2, dest_l2, e_u2, dest_u2);
            spam_columns(c, e_l, dest_l);
            spam_columns(c2, e_l2, dest_l2);
#if GST_ANA_TIMING
            const unsigned long long t_s1 = AT_NOW();
            at[1] += t_s1 - t_f1;
#endif
            if (a.blk_ptr) {
                // The item as ONE stream of 4-slot blocks (host-built, AnaArgs::blk_*): the three-stage pipeline of `sweep`
                // runs through all gates and segments of the item -- indices two blocks ahead, state vectors one block
                // ahead, 8 MFMAs per block (both circuits against the shared backward vectors; a circuit without an
                // application in a slot contributes a zeroed operand) -- and a gate's two 16 x 16 x 4-outcome blocks are
                // stored when its last block has been multiplied, while the next gate's vectors are already in flight.
                const int32_t bp_l = a.blk_ptr[ci * nG + (lane <= nG ? lane : nG)];     // lane g: first block of gate g (lane nG: end)
                const int32_t c0_l = lane < nG ? a.gate_col0[lane] : -2;
                struct Cur { int g; int32_t pos, end; };
                const int32_t item_b0 = __builtin_amdgcn_readlane(bp_l, 0);
                auto seek = [&](int g) {
                    for (; g < nG; g++) {
                        const int32_t b0 = __builtin_amdgcn_readlane(bp_l, g), b1 = __builtin_amdgcn_readlane(bp_l, g + 1);
                        if (__builtin_amdgcn_readlane(c0_l, g) != -2 && b1 > b0) return Cur{g, b0, b1};
                    }
                    return Cur{nG, item_b0, item_b0};          // done: a valid block to keep the prefetches in bounds
                };
                auto advance = [&](const Cur& q) {
                    if (q.g >= nG) return q;
                    if (q.pos + 1 < q.end) return Cur{q.g, q.pos + 1, q.end};
                    return seek(q.g + 1);
                };
                // gates the item never applies have no blocks, but their columns must be written all the same -- as zeros
                // (written out on its own, with no accumulator array handed to `store_block` from inside a run-time loop:
                //  that puts the arrays -- the live MFMA accumulators included -- into scratch memory)
                auto zero_gates = [&](int g_from, int g_to) {
                    if (a.accumulate || zeros_resident) return;      // (adding zeros; or zeros the destination already holds)
                    for (int g = g_from; g < g_to; g++) {
                        const int32_t c0 = __builtin_amdgcn_readlane(c0_l, g);
                        if (c0 == -2) continue;
#pragma unroll
                        for (int x = 0; x < 2 * NX; x++) {
                            const int64_t dst = x < NX ? dest_u[x & (NX - 1)] : dest_u2[x & (NX - 1)];
                            if (c0 >= 0) {
                                double* o = a.out + dst * a.ld + c0 + kk * D + i;
#pragma unroll
                                for (int r = 0; r < 4; r++) __builtin_nontemporal_store(0.0, &o[4 * r * D]);
                            } else {
                                const int32_t* cm = a.colmap_gate + (int64_t)g * D * D + kk * D + i;
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const int32_t cc = cm[4 * r * D];
                                    if (cc >= 0) a.out[dst * a.ld + cc] = 0.0;
                                }
                            }
                        }
                    }
                };
#if GST_ANA_TIMING
                const unsigned long long t_l0 = AT_NOW();
                at[4] += t_l0 - t_s1;
#endif
                Cur cur0 = seek(0);
                zero_gates(0, cur0.g);
                if (cur0.g < nG) {
                    d4_t acc[NX], acc2[NX];
#pragma unroll
                    for (int x = 0; x < NX; x++) { acc[x] = (d4_t){0.0, 0.0, 0.0, 0.0}; acc2[x] = (d4_t){0.0, 0.0, 0.0, 0.0}; }
                    constexpr int SM = ANA_STREAM_M;       // chunks of 4 slots per pipeline block (host pads gates to 4 * SM slots)
                    auto idx_load = [&](const Cur& q, int32_t (&f1)[SM], int32_t (&f2)[SM], int32_t (&rr)[SM]) {
#pragma unroll
                        for (int m = 0; m < SM; m++) {
                            const int64_t o = ((int64_t)q.pos * SM + m) * 4 + kk;
                            f1[m] = a.blk_f1[o]; f2[m] = a.blk_f2[o]; rr[m] = a.blk_r[o];
                        }
                    };
                    auto gather = [&](const bool live, const int32_t (&f1)[SM], const int32_t (&f2)[SM], const int32_t (&rr)[SM],
                                      double (&F1)[SM], double (&F2)[SM], double (&Bv)[SM][NX], bool (&ok1)[SM], bool (&ok2)[SM]) {
#pragma unroll
                        for (int m = 0; m < SM; m++) {
                            ok1[m] = live && f1[m] >= 0; ok2[m] = live && f2[m] >= 0;
                            F1[m] = *(const double*)(fb + (off_t)(uint32_t)(f1[m] < 0 ? 0 : f1[m]) * fstride + lane_b);
                            F2[m] = *(const double*)(fb + (off_t)(uint32_t)(f2[m] < 0 ? 0 : f2[m]) * fstride + lane_b);
                            const d2_t* q2 = (const d2_t*)__builtin_assume_aligned(rb + (off_t)(uint32_t)rr[m] * rstride + lane_r, 16);
                            const d2_t t0 = q2[0], t1 = q2[1];
                            Bv[m][0] = t0.x; Bv[m][1] = t0.y; Bv[m][2] = t1.x; Bv[m][3] = t1.y;
                        }
                    };
                    auto mma = [&](const double (&F1)[SM], const double (&F2)[SM], const double (&Bv)[SM][NX], const bool (&ok1)[SM], const bool (&ok2)[SM]) {
#if GST_ANA_TIMING
                        at[7] += 1;
#endif
#pragma unroll
                        for (int m = 0; m < SM; m++) {
                            const double Fa = ok1[m] ? F1[m] : 0.0, Fb2 = ok2[m] ? F2[m] : 0.0;
#pragma unroll
                            for (int x = 0; x < NX; x++) {
                                acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bv[m][x], Fa, acc[x], 0, 0, 0);
                                acc2[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bv[m][x], Fb2, acc2[x], 0, 0, 0);
                            }
                        }
                    };
                    auto flush_if_last = [&](const Cur& q, const Cur& next) {
                        if (q.pos + 1 != q.end) return;
#if GST_ANA_TIMING
                        const unsigned long long t_x0 = AT_NOW();
#endif
                        const int32_t c0 = __builtin_amdgcn_readlane(c0_l, q.g);
                        store_block(q.g, c0, NX, dest_u, acc);
                        store_block(q.g, c0, NX, dest_u2, acc2);
#pragma unroll
                        for (int x = 0; x < NX; x++) { acc[x] = (d4_t){0.0, 0.0, 0.0, 0.0}; acc2[x] = (d4_t){0.0, 0.0, 0.0, 0.0}; }
                        zero_gates(q.g + 1, next.g);          // (the gates between this one and the next block's)
#if GST_ANA_TIMING
                        at[3] += AT_NOW() - t_x0; at[8] += 1;
#endif
                    };
                    Cur cur1 = advance(cur0), cur2 = advance(cur1);
                    int32_t f1A[SM], f2A[SM], rrA[SM], f1B[SM], f2B[SM], rrB[SM];
                    double F10[SM], F20[SM], B0[SM][NX], F11[SM], F21[SM], B1[SM][NX];
                    bool ok10[SM], ok20[SM], ok11[SM], ok21[SM];
                    idx_load(cur0, f1A, f2A, rrA);
                    idx_load(cur1, f1B, f2B, rrB);
                    gather(true, f1A, f2A, rrA, F10, F20, B0, ok10, ok20);
                    for (;;) {
                        DB_FENCE();
                        idx_load(cur2, f1A, f2A, rrA);
                        DB_FENCE();
                        gather(cur1.g < nG, f1B, f2B, rrB, F11, F21, B1, ok11, ok21);
                        DB_FENCE();
                        mma(F10, F20, B0, ok10, ok20);
                        DB_FENCE();
                        flush_if_last(cur0, cur1);
                        cur0 = cur1; cur1 = cur2; cur2 = advance(cur2);
                        if (cur0.g >= nG) break;
                        DB_FENCE();
                        idx_load(cur2, f1B, f2B, rrB);
                        DB_FENCE();
                        gather(cur1.g < nG, f1A, f2A, rrA, F10, F20, B0, ok10, ok20);
                        DB_FENCE();
                        mma(F11, F21, B1, ok11, ok21);
                        DB_FENCE();
                        flush_if_last(cur0, cur1);
                        cur0 = cur1; cur1 = cur2; cur2 = advance(cur2);
                        if (cur0.g >= nG) break;
                    }
                }
#if GST_ANA_TIMING
                at[2] += AT_NOW() - t_l0;      // (stream loop incl. the flushes: subtract at[3])
#endif
                continue;
            }
            for (int g = 0; g < nG; g++) {
                const int32_t c0 = as_const(a.gate_col0)[g];
                if (c0 == -2) continue;
                const int64_t p0 = as_const(a.pos_ptr)[c * nG + g], p1 = as_const(a.pos_ptr)[c * nG + g + 1];
                const int64_t q0 = as_const(a.pos_ptr)[c2 * nG + g], q1 = as_const(a.pos_ptr)[c2 * nG + g + 1];
                const int64_t cg = (int64_t)as_const(a.pair_common)[ci * nG + g];
                if (zeros_resident && p1 == p0 && q1 == q0) continue;      // neither circuit applies g: zeros the destination already holds
                d4_t acc[NX], acc2[NX];
#pragma unroll
                for (int x = 0; x < NX; x++) { acc[x] = (d4_t){0.0, 0.0, 0.0, 0.0}; acc2[x] = (d4_t){0.0, 0.0, 0.0, 0.0}; }
                // common tail: applications p1 - cg .. p1 - 1 of c and q1 - cg .. q1 - 1 of c2 have equal backward ids; the
                // same three-stage pipeline as `sweep`, one backward gather feeding both circuits' MFMAs
                constexpr int M2 = ANA_MFMA_M;
                if (cg > 0) {
                    const int32_t n = (int32_t)cg;
                    const int32_t nb = (n + 4 * M2 - 1) / (4 * M2);
                    const int32_t* const pf1 = a.pair_f + (p1 - cg);
                    const int32_t* const pf2 = a.pair_f + (q1 - cg);
                    const int32_t* const prr = a.pair_r + (p1 - cg);
                    auto idx_at = [&](int32_t blk, int32_t (&f1)[M2], int32_t (&f2)[M2], int32_t (&rr)[M2]) {
#pragma unroll
                        for (int m = 0; m < M2; m++) {
                            const int32_t off = (blk * M2 + m) * 4 + kk;
                            const int32_t oc = off < n ? off : n - 1;
                            f1[m] = pf1[oc]; f2[m] = pf2[oc]; rr[m] = prr[oc];
                        }
                    };
                    auto gather = [&](const int32_t (&f1)[M2], const int32_t (&f2)[M2], const int32_t (&rr)[M2],
                                      double (&F1)[M2], double (&F2)[M2], double (&Bv)[M2][NX]) {
#pragma unroll
                        for (int m = 0; m < M2; m++) {
                            F1[m] = *(const double*)(fb + (off_t)(uint32_t)f1[m] * fstride + lane_b);
                            F2[m] = *(const double*)(fb + (off_t)(uint32_t)f2[m] * fstride + lane_b);
                            const d2_t* q2 = (const d2_t*)__builtin_assume_aligned(rb + (off_t)(uint32_t)rr[m] * rstride + lane_r, 16);
                            const d2_t t0 = q2[0], t1 = q2[1];
                            Bv[m][0] = t0.x; Bv[m][1] = t0.y; Bv[m][2] = t1.x; Bv[m][3] = t1.y;
                        }
                    };
                    auto mma = [&](int32_t blk, const double (&F1)[M2], const double (&F2)[M2], const double (&Bv)[M2][NX]) {
#pragma unroll
                        for (int m = 0; m < M2; m++) {
                            const bool ok = (blk * M2 + m) * 4 + kk < n;
                            const double Fa = ok ? F1[m] : 0.0, Fb2 = ok ? F2[m] : 0.0;
#pragma unroll
                            for (int x = 0; x < NX; x++) {
                                acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bv[m][x], Fa, acc[x], 0, 0, 0);
                                acc2[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bv[m][x], Fb2, acc2[x], 0, 0, 0);
                            }
                        }
                    };
                    int32_t f1A[M2], f2A[M2], rrA[M2], f1B[M2], f2B[M2], rrB[M2];
                    double F10[M2], F20[M2], B0[M2][NX], F11[M2], F21[M2], B1[M2][NX];
                    idx_at(0, f1A, f2A, rrA);
                    idx_at(1, f1B, f2B, rrB);
                    gather(f1A, f2A, rrA, F10, F20, B0);
                    for (int32_t b = 0; b < nb; b += 2) {
                        DB_FENCE();
                        idx_at(b + 2, f1A, f2A, rrA);
                        DB_FENCE();
                        gather(f1B, f2B, rrB, F11, F21, B1);
                        DB_FENCE();
                        mma(b, F10, F20, B0);
                        DB_FENCE();
                        idx_at(b + 3, f1B, f2B, rrB);
                        DB_FENCE();
                        gather(f1A, f2A, rrA, F10, F20, B0);
                        DB_FENCE();
                        mma(b + 1, F11, F21, B1);
                        DB_FENCE();
                    }
                }
                // the applications before the common tail, each circuit on its own
                sweep(std::integral_constant<int, ANA_MFMA_M>{}, std::true_type{}, p0, p1 - cg, e_u, acc);
                sweep(std::integral_constant<int, ANA_MFMA_M>{}, std::true_type{}, q0, q1 - cg, e_u2, acc2);
                store_block(g, c0, NX, dest_u, acc);
                store_block(g, c0, NX, dest_u2, acc2);
            }
            continue;
        }
        const int32_t x0 = as_const(a.eff_ptr)[c], x1 = as_const(a.eff_ptr)[c + 1];
        for (int32_t xb = x0; xb < x1; xb += NX) {
            const int nx = (x1 - xb < NX) ? (x1 - xb) : NX;
            // lane group kk looks after outcome xb + kk for the SPAM columns; the gate blocks need all NX labels uniform
            int32_t e_l, e_u[NX];
            int64_t dest_l, dest_u[NX];
            outcomes(x0, xb, nx, e_l, dest_l, e_u, dest_u);
            const bool fast4 = nE == 4 && e_u[0] == 0 && e_u[1] == 1 && e_u[2] == 2 && e_u[3] == 3;
            if (kk < nx) spam_columns(c, e_l, dest_l);
            for (int g = 0; g < nG; g++) {
                const int32_t c0 = as_const(a.gate_col0)[g];
                if (c0 == -2) continue;                                // no parameter of this gate was requested
                const int64_t p0 = as_const(a.pos_ptr)[c * nG + g], p1 = as_const(a.pos_ptr)[c * nG + g + 1];
                if (zeros_resident && p1 == p0) continue;                  // g never applied: zeros the destination already holds
                d4_t acc[NX];
#pragma unroll
                for (int x = 0; x < NX; x++) acc[x] = (d4_t){0.0, 0.0, 0.0, 0.0};
                if (fast4) sweep(std::integral_constant<int, ANA_MFMA_M>{}, std::true_type{}, p0, p1, e_u, acc);
                else sweep(std::integral_constant<int, ANA_MFMA_M>{}, std::false_type{}, p0, p1, e_u, acc);
                store_block(g, c0, nx, dest_u, acc);
            }
        }
    }
#if GST_ANA_TIMING
    at[5] = AT_NOW() - at_start;
    if (lane == 0)
        for (int q = 0; q < 10; q++) atomicAdd(&ana_dbg[q], at[q]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// dwalk_kernel (D = 16): derivative states for the analytic Hessian, four parameters per wavefront (lane group q <->
// theta_q; see DWalkArgs).  A plain interpreter of the walk programs: the passes are short next to the contractions
// they feed, so nothing is pipelined here -- the base-state element of an injection is simply read from the cache.
template <int D>
__global__ __launch_bounds__(64) void dwalk_kernel(const DWalkArgs a, const int n_slots)
{
    static_assert(D == 4 || D == 16, "lane groups of D lanes, four of them in use");
    extern __shared__ double lds[];           // save slots [n_slots][64] | gate tiles [nG][D][D]
    const int lane = threadIdx.x, li = lane % D, grp = lane / D;
    const int32_t n_tg = (a.n_theta + 3) / 4;
    const int64_t task = blockIdx.x / n_tg;
    const int32_t tg = (int32_t)(blockIdx.x % n_tg);
    const int32_t th = tg * 4 + grp;
    const bool live = grp < 4 && th < a.n_theta;
    const int32_t ig = live ? a.inj_gate[th] : -1, idst = live ? a.inj_dst[th] : 0, isrc = live ? a.inj_src[th] : 0;
    const int32_t sobj = live ? a.start_obj[th] : -2, sidx = live ? a.start_idx[th] : 0;
    double* const slot_lane = lds + lane;
    double* const tile = lds + (n_slots > 0 ? n_slots : 1) * 64;
    for (int k = lane; k < a.n_gates * D * D; k += 64) tile[k] = a.tile[k];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    const int64_t pc0 = as_const(a.task_off)[task];
    const int32_t n_words = (int32_t)(as_const(a.task_off)[task + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    double v = 0.0;
    int32_t cur_id = 0;
    int32_t slot_id[4] = {0, 0, 0, 0};        // the state id that goes with each saved state (D <= 16 plans use <= 4 slots)
    for (int32_t pc = 0; pc < n_words; pc++) {
        const uint32_t w = as_const(gprog)[pc];
        const uint32_t op = GST_OP(w), arg = GST_ARG(w);
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            const bool hit = (int32_t)arg == ig;
            double inj = 0.0;
            if (hit) inj = a.base[(int64_t)cur_id * a.bstride + (int64_t)isrc * a.bmul + a.boff];     // S_{k-1}[src]
            const double* T = tile + (int)arg * D * D + li;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < D; j++) {
                const double vj = __shfl(v, (lane - li) + j, 64);
                acc = __builtin_fma(T[j * D], vj, acc);
            }
            v = (hit && li == idst) ? acc + inj : acc;
        } else if (op == GST_OP_NODE) {
            cur_id = (int32_t)arg;
            if (live) a.out[((int64_t)arg * 4 + grp) * a.ostride + (int64_t)li * a.omul + a.ooff] = v;
        } else if (op == GST_OP_SAVE) {
            slot_lane[arg * 64] = v;
#pragma unroll
            for (int t = 0; t < 4; t++) slot_id[t] = (arg == (uint32_t)t) ? cur_id : slot_id[t];
        } else if (op == GST_OP_LOAD) {
            v = slot_lane[arg * 64];
#pragma unroll
            for (int t = 0; t < 4; t++) cur_id = (arg == (uint32_t)t) ? slot_id[t] : cur_id;
        } else if (op == GST_OP_RHO) {
            v = ((sobj == -1 || sobj == (int32_t)arg) && li == sidx) ? 1.0 : 0.0;
        }                                             // EMIT: nothing to do
    }
}

// D = 64: one parameter per wavefront (lane = state row, coefficients streamed from L2 like walk_rows_kernel<64>);
// blockIdx -> (task, theta); the theta slot q of DWalkArgs' output layout is theta % 4.
__global__ __launch_bounds__(64) void dwalk64_kernel(const DWalkArgs a, const int n_slots)
{
    constexpr int D = 64;
    extern __shared__ double lds[];           // save slots [n_slots][64], then their state ids
    const int lane = threadIdx.x;
    const int64_t task = blockIdx.x / a.n_theta;
    const int32_t th = (int32_t)(blockIdx.x % a.n_theta);
    const int32_t ig = as_const(a.inj_gate)[th], idst = as_const(a.inj_dst)[th], isrc = as_const(a.inj_src)[th];
    const int32_t sobj = as_const(a.start_obj)[th], sidx = as_const(a.start_idx)[th];
    int32_t* const slot_id = (int32_t*)(lds + (n_slots > 0 ? n_slots : 1) * 64);
    const int64_t pc0 = as_const(a.task_off)[task];
    const int32_t n_words = (int32_t)(as_const(a.task_off)[task + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    double v = 0.0;
    int32_t cur_id = 0;
    for (int32_t pc = 0; pc < n_words; pc++) {
        const uint32_t w = as_const(gprog)[pc];
        const uint32_t op = GST_OP(w), arg = GST_ARG(w);
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            const bool hit = (int32_t)arg == ig;
            const double inj = hit ? a.base[(int64_t)cur_id * a.bstride + (int64_t)isrc * a.bmul + a.boff] : 0.0;
            const double* T = a.tile + (int64_t)arg * D * D + lane;
            double acc = 0.0;
#pragma unroll 16
            for (int j = 0; j < D; j++) acc = __builtin_fma(T[j * D], __shfl(v, j, 64), acc);
            v = (hit && lane == idst) ? acc + inj : acc;
        } else if (op == GST_OP_NODE) {
            cur_id = (int32_t)arg;
            a.out[((int64_t)arg * 4 + (th & 3)) * a.ostride + (int64_t)lane * a.omul + a.ooff] = v;
        } else if (op == GST_OP_SAVE) {
            lds[arg * 64 + lane] = v;
            if (lane == 0) slot_id[arg] = cur_id;
        } else if (op == GST_OP_LOAD) {
            v = lds[arg * 64 + lane];
            cur_id = slot_id[arg];
        } else if (op == GST_OP_RHO) {
            v = ((sobj == -1 || sobj == (int32_t)arg) && lane == sidx) ? 1.0 : 0.0;
        }
    }
}

hipError_t launch_dwalk(int D, const DWalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    if (n_tasks <= 0 || a.n_theta <= 0) return hipSuccess;
    (void)hipGetLastError();
    if (D == 64) {
        const int64_t blocks = n_tasks * a.n_theta;
        const size_t lds_bytes = (size_t)(n_slots > 0 ? n_slots : 1) * (64 * sizeof(double) + sizeof(int32_t));
        if (blocks > 0x7fffffffLL || lds_bytes > 64 * 1024 || a.n_theta > 4) return hipErrorInvalidValue;
        hipLaunchKernelGGL(dwalk64_kernel, dim3((unsigned)blocks), dim3(64), lds_bytes, stream, a, n_slots);
        return hipGetLastError();
    }
    if (D != 16 && D != 4) return hipErrorInvalidValue;
    const int64_t blocks = n_tasks * ((a.n_theta + 3) / 4);
    const size_t lds_bytes = ((size_t)(n_slots > 0 ? n_slots : 1) * 64 + (size_t)a.n_gates * D * D) * sizeof(double);
    if (blocks > 0x7fffffffLL || lds_bytes > 64 * 1024 || n_slots > 4) return hipErrorInvalidValue;
    if (D == 16) hipLaunchKernelGGL(dwalk_kernel<16>, dim3((unsigned)blocks), dim3(64), lds_bytes, stream, a, n_slots);
    else hipLaunchKernelGGL(dwalk_kernel<4>, dim3((unsigned)blocks), dim3(64), lds_bytes, stream, a, n_slots);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// analytic_mfma64_kernel (D = 64, three qubits): the same contraction with 64 x 64 blocks.  One wavefront owns one
// circuit outcome: for every gate it holds the 4 x 4 grid of 16 x 16 MFMA tiles of that gate's block (64 fp64
// accumulators per lane) and, per 4 applications of the gate, gathers 4 + 4 operand registers (the four 16-row
// slices of B_k and of F_{k-1}) for 16 v_mfma_f64_16x16x4_f64 -- two MFMAs per gathered register.
template <bool WIDE>
__global__ __launch_bounds__(256, 2) void analytic_mfma64_kernel(const AnaArgs a)
{
    using off_t = typename std::conditional<WIDE, uint64_t, uint32_t>::type;
    constexpr int D = 64;
    const int lane = threadIdx.x & 63;
    const int kk = lane >> 4, i = lane & 15;
    const int nE = a.n_effects, nG = a.n_gates;
    for (;;) {
        uint32_t cu = 0;
        if (lane == 0) cu = atomicAdd(a.work_counter, 1u);
        const int64_t item = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)cu);     // (ordered circuit, outcome slot)
        if (item >= a.n_circuits * (int64_t)nE) break;
        const int64_t c = as_const(a.circ_order)[item / nE];
        const int32_t x0 = as_const(a.eff_ptr)[c], x1 = as_const(a.eff_ptr)[c + 1];
        const int32_t x = x0 + (int32_t)(item % nE);
        if (x >= x1) continue;                                         // (a circuit with fewer outcomes than effects)
        const int32_t e = as_const(a.eff_label)[x];
        const int64_t dest = as_const(a.eff_dest)[x];
        const int32_t fleaf = as_const(a.circ_leaf)[c], rleaf = as_const(a.rev_leaf)[c], rsym = as_const(a.circ_rho)[c];
        double* const orow = a.out + dest * a.ld;
        const uint32_t fstride = a.fwd_stride ? a.fwd_stride : (uint32_t)(D * 8);
        const uint32_t rstride = a.rev_stride ? a.rev_stride : (uint32_t)nE * (D * 8);
        const bool accum = a.accumulate != 0;
        // SPAM columns: dp/dE[a] = F_n[a] (own effect; zeros for the others), dp/drho[b] = B_0[b]
        // (Hessian rows: derivative states in one of the two caches, the other family zeroed, the second launch adding)
        {
            const double FL = a.eff_zero ? 0.0 : *(const double*)((const char*)a.base_cache + (int64_t)fleaf * fstride + lane * 8);
            for (int e2 = 0; e2 < nE; e2++) {
                const int32_t ce = a.colmap_eff[e2 * D + lane];
                if (ce >= 0) { const double val = (e2 == e) ? FL : 0.0; orow[ce] = accum ? orow[ce] + val : val; }
            }
            const double B0 = a.rho_zero ? 0.0 : *(const double*)((const char*)a.rev_cache + (int64_t)rleaf * rstride + ((int64_t)e * D + lane) * 8);
            for (int r2 = 0; r2 < a.n_rhos; r2++) {
                const int32_t cr = a.colmap_rho[r2 * D + lane];
                if (cr >= 0) { const double val = (r2 == rsym) ? B0 : 0.0; orow[cr] = accum ? orow[cr] + val : val; }
            }
        }
        const char* const fb = (const char*)a.base_cache;
        const char* const rb = (const char*)a.rev_cache + (uint32_t)e * (D * 8);
        const uint32_t lane_b = (uint32_t)i * 8u;
        // The gates of the item one after the other.  A gate's block is 32 KB of output behind a few MFMA rounds: left to
        // themselves, "gather - MFMA - store" run one after the other in a wavefront, and the gathers of every wavefront
        // of the CU queue behind the stores of the others in the CU's one memory pipeline (measured on 4,000 circuits:
        // 3.09 ms, against 1.84 ms for the stores alone and 1.48 ms without them).  So the operands of the NEXT gate's
        // first PF rounds are requested before the current block is stored -- their state ids a block earlier still --
        // and arrive while the wavefront's own 64 stores are being accepted.
        constexpr int PF = 4;
        double pF[PF][4], pB[PF][4];
        int32_t idF[PF], idR[PF];
        int64_t np0 = 0, np1 = 0;                       // the positions of the gate the prefetch registers belong to
        int gn = 0;
        while (gn < nG && as_const(a.gate_col0)[gn] == -2) gn++;
        auto fetch_ids = [&](int g) {                   // (per lane: the position q + kk of round `it`, clamped to the gate's last)
            np0 = as_const(a.pos_ptr)[c * nG + g]; np1 = as_const(a.pos_ptr)[c * nG + g + 1];
#pragma unroll
            for (int it = 0; it < PF; it++) {
                const int64_t pi = np0 + 4 * it + kk;
                const int64_t pc = pi < np1 ? pi : (np1 > np0 ? np1 - 1 : np0);
                const bool any = np0 + 4 * it < np1;
                idF[it] = any ? a.pair_f[pc] : 0;
                idR[it] = any ? a.pair_r[pc] : 0;
            }
        };
        auto fetch_vectors = [&]() {
#pragma unroll
            for (int it = 0; it < PF; it++) {
                if (np0 + 4 * it < np1) {
                    const off_t fo = (off_t)(uint32_t)idF[it] * fstride + lane_b;
                    const off_t ro = (off_t)(uint32_t)idR[it] * rstride + lane_b;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        pF[it][t] = *(const double*)(fb + fo + t * 128);
                        pB[it][t] = *(const double*)(rb + ro + t * 128);
                    }
                }
            }
        };
        if (gn < nG) { fetch_ids(gn); fetch_vectors(); }
        while (gn < nG) {
            const int g = gn;
            const int32_t c0 = as_const(a.gate_col0)[g];
            const int64_t p0 = np0, p1 = np1;
            gn = g + 1;
            while (gn < nG && as_const(a.gate_col0)[gn] == -2) gn++;
            d4_t acc[4][4];
#pragma unroll
            for (int tr = 0; tr < 4; tr++)
#pragma unroll
                for (int tc = 0; tc < 4; tc++) acc[tr][tc] = (d4_t){0.0, 0.0, 0.0, 0.0};
            const int64_t last = p1 - 1;
            // rounds 0 .. PF-1 from the prefetched registers
#pragma unroll
            for (int it = 0; it < PF; it++) {
                if (p0 + 4 * it < p1) {
                    const bool ok = p0 + 4 * it + kk <= last;
                    double Fv[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) Fv[t] = ok ? pF[it][t] : 0.0;
#pragma unroll
                    for (int tr = 0; tr < 4; tr++)
#pragma unroll
                        for (int tc = 0; tc < 4; tc++)
                            acc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(pB[it][tr], Fv[tc], acc[tr][tc], 0, 0, 0);
                }
            }
            // the next gate's state ids: on their way during the remaining rounds
            if (gn < nG) fetch_ids(gn);
            for (int64_t q = p0 + 4 * PF; q < p1; q += 4) {
                const int64_t pi = q + kk;
                const bool ok = pi <= last;
                const int64_t pc = ok ? pi : last;
                const off_t fo = (off_t)(uint32_t)a.pair_f[pc] * fstride + lane_b;
                const off_t ro = (off_t)(uint32_t)a.pair_r[pc] * rstride + lane_b;
                double Fv[4], Bv[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    Fv[t] = *(const double*)(fb + fo + t * 128);
                    Bv[t] = *(const double*)(rb + ro + t * 128);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) Fv[t] = ok ? Fv[t] : 0.0;
#pragma unroll
                for (int tr = 0; tr < 4; tr++)
#pragma unroll
                    for (int tc = 0; tc < 4; tc++)
                        acc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bv[tr], Fv[tc], acc[tr][tc], 0, 0, 0);
            }
            // the next gate's operands go out BEFORE this block's stores
            if (gn < nG) fetch_vectors();
            // tile (tr, tc), lane l, register r -> block entry (16 tr + (l>>4) + 4r, 16 tc + (l&15))
            if (c0 >= 0) {
#pragma unroll
                for (int tr = 0; tr < 4; tr++)
#pragma unroll
                    for (int tc = 0; tc < 4; tc++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            double* o = orow + c0 + (16 * tr + kk + 4 * r) * D + 16 * tc + i;
                            if (accum) *o = *o + acc[tr][tc][r];
                            else __builtin_nontemporal_store(acc[tr][tc][r], o);      // written once, never re-read here: keep the state caches in L2
                        }
            } else {
                const int32_t* cm = a.colmap_gate + (int64_t)g * D * D;
#pragma unroll
                for (int tr = 0; tr < 4; tr++)
#pragma unroll
                    for (int tc = 0; tc < 4; tc++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int32_t cc = cm[(16 * tr + kk + 4 * r) * D + 16 * tc + i];
                            if (cc >= 0) orow[cc] = accum ? orow[cc] + acc[tr][tc][r] : acc[tr][tc][r];
                        }
            }
        }
    }
}

hipError_t launch_analytic_mfma64(const AnaArgs& a, hipStream_t stream)
{
    if (a.n_circuits <= 0) return hipSuccess;
    int64_t blocks = (a.n_circuits * (int64_t)a.n_effects + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;       // persistent wavefronts pulling (circuit, outcome) items
    (void)hipGetLastError();
    if (a.wide) hipLaunchKernelGGL(analytic_mfma64_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(analytic_mfma64_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// analytic_small_kernel (D = 4, one qubit): the same contraction from the two state caches without matrix
// instructions -- a 4 x 4 block per (outcome, gate) is too small for an MFMA tile.  One wavefront per circuit; lane
// (x, a, b) = 4 outcomes x 16 block entries accumulates sum_k B^x_k[a] F_{k-1}[b] over the gate's applications (two
// 8-byte gathers and one FMA per application).  Same options as the MFMA kernels (strides, zeroed families, accumulate).
__global__ __launch_bounds__(256) void analytic_small_kernel(const AnaArgs a)
{
    constexpr int D = 4, NX = 4;
    const int lane = threadIdx.x & 63;
    const int xg = lane >> 4, ea = (lane >> 2) & 3, eb = lane & 3;       // outcome slot, block row a, block column b
    const int nE = a.n_effects, nG = a.n_gates;
    const uint32_t fstride = a.fwd_stride ? a.fwd_stride : (uint32_t)(D * 8);
    const uint32_t rstride = a.rev_stride ? a.rev_stride : (uint32_t)nE * (D * 8);
    const bool accum = a.accumulate != 0;
    const int64_t total_waves = (int64_t)gridDim.x * 4;
    for (int64_t ci = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); ci < a.n_circuits; ci += total_waves) {
        const int64_t c = as_const(a.circ_order)[ci];
        const int32_t x0 = as_const(a.eff_ptr)[c], x1 = as_const(a.eff_ptr)[c + 1];
        const int32_t fleaf = as_const(a.circ_leaf)[c], rleaf = as_const(a.rev_leaf)[c], rsym = as_const(a.circ_rho)[c];
        for (int32_t xb = x0; xb < x1; xb += NX) {
            const bool on = xb + xg < x1;
            const int32_t e = a.eff_label[on ? xb + xg : x0];
            const int64_t dest = a.eff_dest[on ? xb + xg : x0];
            double* const orow = a.out + dest * a.ld;
            if (on && ea == 0) {             // SPAM columns, component eb: dp/dE = F_n (own effect), dp/drho = B_0
                const double FL = a.eff_zero ? 0.0 : *(const double*)((const char*)a.base_cache + (int64_t)fleaf * fstride + eb * 8);
                for (int e2 = 0; e2 < nE; e2++) {
                    const int32_t ce = a.colmap_eff[e2 * D + eb];
                    if (ce >= 0) { const double val = (e2 == e) ? FL : 0.0; orow[ce] = accum ? orow[ce] + val : val; }
                }
                const double B0 = a.rho_zero ? 0.0 : *(const double*)((const char*)a.rev_cache + (int64_t)rleaf * rstride + ((int64_t)eb * nE + e) * 8);
                for (int r2 = 0; r2 < a.n_rhos; r2++) {
                    const int32_t cr = a.colmap_rho[r2 * D + eb];
                    if (cr >= 0) { const double val = (r2 == rsym) ? B0 : 0.0; orow[cr] = accum ? orow[cr] + val : val; }
                }
            }
            for (int g = 0; g < nG; g++) {
                const int32_t c0 = as_const(a.gate_col0)[g];
                if (c0 == -2) continue;
                const int64_t p0 = as_const(a.pos_ptr)[c * nG + g], p1 = as_const(a.pos_ptr)[c * nG + g + 1];
                double acc = 0.0;
                for (int64_t q = p0; q < p1; q++) {
                    const int32_t fid = as_const(a.pair_f)[q], rid = as_const(a.pair_r)[q];
                    const double Fv = *(const double*)((const char*)a.base_cache + (int64_t)fid * fstride + eb * 8);
                    const double Bv = *(const double*)((const char*)a.rev_cache + (int64_t)rid * rstride + ((int64_t)ea * nE + e) * 8);
                    acc = __builtin_fma(Bv, Fv, acc);
                }
                if (on) {
                    const int32_t cc = (c0 >= 0) ? c0 + ea * D + eb : a.colmap_gate[(int64_t)g * D * D + ea * D + eb];
                    if (cc >= 0) orow[cc] = accum ? orow[cc] + acc : acc;
                }
            }
        }
    }
}

hipError_t launch_analytic_small(const AnaArgs& a, hipStream_t stream)
{
    if (a.n_circuits <= 0) return hipSuccess;
    int64_t blocks = (a.n_circuits + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    (void)hipGetLastError();
    hipLaunchKernelGGL(analytic_small_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

int analytic_stream_chunks() { return ANA_STREAM_M; }

hipError_t launch_analytic_mfma(const AnaArgs& a, hipStream_t stream)
{
    if (a.n_circuits <= 0) return hipSuccess;
    int64_t blocks = (a.n_circuits + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;       // persistent wavefronts pulling circuits from a.work_counter[0..7]
    blocks = (blocks + 7) / 8 * 8;
    (void)hipGetLastError();
    if (a.wide) hipLaunchKernelGGL(analytic_mfma_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(analytic_mfma_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_analytic(int D, const AnaArgs& a, hipStream_t stream)
{
    if (D != 16 && D != 4) return hipErrorInvalidValue;
    if (a.n_circuits <= 0) return hipSuccess;
    const size_t lds_bytes = (size_t)a.n_gates * D * (D + ANA_PAD) * sizeof(double);
    if (lds_bytes > 64 * 1024) return hipErrorInvalidValue;
    int64_t blocks = (a.n_circuits + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;     // persistent: every wavefront strides over the circuits
    (void)hipGetLastError();
    if (D == 16) hipLaunchKernelGGL((analytic_dprobs_kernel<16, 6, 2>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a);
    else hipLaunchKernelGGL((analytic_dprobs_kernel<4, 8, 4>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a);
    return hipGetLastError();
}

#if GST_ANA_TIMING
extern "C" __attribute__((visibility("default"))) int gst_debug_ana_phases(unsigned long long* out, int reset)
{
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(ana_dbg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ana_dbg), 16 * sizeof(unsigned long long));
}
#endif
}  // namespace gst
