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

}  // namespace gst
