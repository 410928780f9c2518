// gst_kernels_levels.hip -- log-depth evaluation of the state tries on the matrix cores (gst_levels.hpp).
//
// level_pass_kernel: one workgroup (4 wavefronts) per task.  The task's level program is a sequence of stages; a stage is
// a list of independent TILES, each  OUT[16 rows] = IN[16 rows] x M  with M a 16 x 16 matrix in "row form" (gate table,
// identity, or one of the task's scratch matrices: the germ's product and its squarings).  A tile is four
// v_mfma_f64_16x16x4_f64 (k = 0..15 in steps of 4):
//     A operand  lane l <- IN[row l & 15][k = 4 s + (l >> 4)]        (a row = `nv`-strided components of a cached state)
//     B operand  lane l <- M[k = 4 s + (l >> 4)][col l & 15]         (row-major M: one coalesced 512-byte read per s)
//     D          lane l -> OUT[row (l >> 4) + 4 r][col l & 15], r = 0..3: sixteen lanes store one whole 128-byte state
// Wavefront w of the workgroup takes tiles w, w + 4, ... of a stage; a workgroup-scope fence and a barrier end it.
// Every location -- trie state, scratch matrix -- is written exactly ONCE per launch and read only in later stages, so
// no cache can hold a stale copy of anything.
//
// Only the modes without an ordering contract use this (exact derivatives, the opt-in fast probabilities): the sums
// here are fused and re-associated (matrix powers), <= 1e-13 from the sequential walk on the deepest GST circuits.
#include "gst_kernels.hpp"
#include "../../include/gstfwd.h"

namespace gst {

typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void level_pass_kernel(const LevelArgs a)
{
    constexpr int D = 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int64_t task = blockIdx.x;
    const int32_t* __restrict__ w = a.words + a.task_off[task];
    const int32_t* __restrict__ ids = a.ids;
    int64_t ids_bias = 0;                  // subtracted from the tile words' GLOBAL id-pool offsets
    double* __restrict__ mats = a.mats + task * (int64_t)a.max_mats * (D * D);
    if (a.stage_lds) {
        // the task's program and its id lists into LDS, once: every stage otherwise starts with two dependent trips to
        // memory (tile words, then ids) before the first operand can be requested
        extern __shared__ int32_t lds[];
        const int nw = (int)(a.task_off[task + 1] - a.task_off[task]);
        const int64_t i0 = a.task_ids_off[task];
        const int ni = (int)(a.task_ids_off[task + 1] - i0);
        for (int k = tid; k < nw; k += 256) lds[k] = w[k];
        for (int k = tid; k < ni; k += 256) lds[nw + k] = a.ids[i0 + k];
        __syncthreads();
        w = lds;
        ids = lds + nw;                    // (an LDS pointer must not be biased out of its aperture: the offsets are rebased instead)
        ids_bias = i0;
    }
    const int n_stages = w[0];
    w += 1;
    for (int s = 0; s < n_stages; s++) {
        const int nt = w[0];
        const int32_t* tw = w + 1;
        for (int t = wv; t < nt; t += 4) {
            const int32_t w0 = tw[4 * t], mref = tw[4 * t + 1], wa = tw[4 * t + 2], wb = tw[4 * t + 3];
            const int kind = w0 & 255, n_nodes = w0 >> 8;
            // ---- A operand: row i of the input, components 4 s4 + kk ----
            const double* sp;
            int64_t cs = 1;
            bool valid = true;
            if (kind == LV_KIND_MAT_) {
                sp = (wa >= 0 ? a.bmats + (int64_t)wa * (D * D) : mats + (int64_t)(-(wa + 2)) * (D * D)) + i * D;
            } else {
                const int node = i / NV, v = i % NV;
                valid = node < n_nodes;
                const int32_t id = valid ? ids[wa - ids_bias + node] : 0;
                if (id >= 0) { sp = a.cache + (int64_t)id * (D * NV) + v; cs = NV; }
                else sp = a.starts + ((int64_t)(-(id + 1)) * NV + v) * D;
            }
            double av[4], bv[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) av[s4] = valid ? sp[(int64_t)(4 * s4 + kk) * cs] : 0.0;
            // ---- B operand: M[4 s4 + kk][i] ----
            if (mref == LV_BMAT_IDENT_) {
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) bv[s4] = (4 * s4 + kk == i) ? 1.0 : 0.0;
            } else {
                const double* bp = mref >= 0 ? a.bmats + (int64_t)mref * (D * D) : mats + (int64_t)(-(mref + 2)) * (D * D);
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) bv[s4] = bp[(4 * s4 + kk) * D + i];
            }
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
            // ---- D: out[row kk + 4 r][component i] ----
            if (kind == LV_KIND_MAT_) {
                double* dp = mats + (int64_t)wb * (D * D);
#pragma unroll
                for (int r = 0; r < 4; r++) dp[(kk + 4 * r) * D + i] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = kk + 4 * r, node = row / NV, v = row % NV;
                    if (node < n_nodes) {
                        const int32_t id = ids[wb - ids_bias + node];
                        a.cache[(int64_t)id * (D * NV) + (int64_t)i * NV + v] = acc[r];
                    }
                }
            }
        }
        w = tw + 4 * nt;
        // A workgroup lives on one CU of one XCD: workgroup scope is all the ordering the next stage needs (s_waitcnt
        // vmcnt(0) + s_barrier).  A device-scope fence here writes back the XCD's whole L2 (the XCDs' L2s are not coherent
        // with each other): measured, the two passes took 8.7 ms instead of well under one.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

hipError_t launch_level_pass(const LevelArgs& a, int64_t n_tasks, hipStream_t stream)
{
    if (n_tasks <= 0) return hipSuccess;
    (void)hipGetLastError();
    const dim3 grid((unsigned)n_tasks), block(256);
    const size_t lds = a.stage_lds ? (size_t)a.lds_ints * 4 : 0;
    switch (a.nv) {
    case 1: hipLaunchKernelGGL(level_pass_kernel<1>, grid, block, lds, stream, a); break;
    case 2: hipLaunchKernelGGL(level_pass_kernel<2>, grid, block, lds, stream, a); break;
    case 4: hipLaunchKernelGGL(level_pass_kernel<4>, grid, block, lds, stream, a); break;
    case 8: hipLaunchKernelGGL(level_pass_kernel<8>, grid, block, lds, stream, a); break;
    case 16: hipLaunchKernelGGL(level_pass_kernel<16>, grid, block, lds, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Probabilities of every circuit from its cached final state: out[eff_dest[x]] = effects[eff_label[x]] . cache[leaf(c)].
// Sixteen lanes per circuit (lane = state component: the 128-byte state is ONE coalesced read; a thread per circuit made
// every load instruction touch 64 different lines -- 90 us for the bench design instead of ~15), the dot product summed
// over the lane group by four butterfly steps.  (The level pass emits nothing itself.)
template <int D>
__global__ __launch_bounds__(256) void probs_from_cache_kernel(const double* __restrict__ cache, const int32_t* __restrict__ circ_leaf,
                                                               const int32_t* __restrict__ eff_ptr, const int32_t* __restrict__ eff_label,
                                                               const int32_t* __restrict__ eff_dest, const double* __restrict__ effects,
                                                               int64_t n_circuits, double* __restrict__ out)
{
    constexpr int PER = 64 / D;                                    // circuits per wavefront
    const int lane = threadIdx.x & 63, k = lane % D;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t c = wave * PER + lane / D;
    const bool on = c < n_circuits;
    const int64_t cc = on ? c : n_circuits - 1;
    const double f = cache[(int64_t)circ_leaf[cc] * D + k];
    const int32_t x0 = eff_ptr[cc], x1 = eff_ptr[cc + 1];
    int32_t nx = x1 - x0;
    // (every lane group of the wavefront runs the longest group's trip count: the shuffles below need all lanes)
    int32_t nmax = nx;
#pragma unroll
    for (int o = D; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    for (int32_t j = 0; j < nmax; j++) {
        const bool live = j < nx;
        const int32_t x = x0 + j;
        double part = 0.0;
        if (live) part = effects[(int64_t)eff_label[x] * D + k] * f;
#pragma unroll
        for (int o = D / 2; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
        if (on && live && k == 0) out[eff_dest[x]] = part;
    }
}

hipError_t launch_probs_from_cache(const double* cache, const int32_t* circ_leaf, const int32_t* eff_ptr, const int32_t* eff_label,
                                   const int32_t* eff_dest, const double* effects, int64_t n_circuits, int D, double* out, hipStream_t stream)
{
    if (n_circuits <= 0) return hipSuccess;
    (void)hipGetLastError();
    if (D != 16 && D != 4) return hipErrorInvalidValue;
    const int64_t per_block = 4 * (64 / D);
    const dim3 grid((unsigned)((n_circuits + per_block - 1) / per_block)), block(256);
    if (D == 16) hipLaunchKernelGGL(probs_from_cache_kernel<16>, grid, block, 0, stream, cache, circ_leaf, eff_ptr, eff_label, eff_dest, effects, n_circuits, out);
    else hipLaunchKernelGGL(probs_from_cache_kernel<4>, grid, block, 0, stream, cache, circ_leaf, eff_ptr, eff_label, eff_dest, effects, n_circuits, out);
    return hipGetLastError();
}

}  // namespace gst
