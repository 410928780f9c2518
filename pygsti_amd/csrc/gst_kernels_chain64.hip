// gst_kernels_chain64.hip -- the state walk of a three-qubit plan (D = 64) on the matrix cores.
//
// For the modes WITHOUT an ordering contract only (GST_DERIV_ANALYTIC's two chain passes, gst_fill_probs* under
// GST_OPT_FAST_PROBS): the finite-difference modes keep walk_rows_kernel<64>, whose ascending-j, un-fused sums ARE their
// bit-parity contract.  There a wavefront owns a state (lane = row), pulls 64 coefficients per lane through L2 for every
// gate and spends 64 x (2 v_readlane + v_mul + v_add) on it: 1.8 us per dependent step, one wavefront per (task, start
// vector) -- 0.45 ms forward and 0.96 ms backward (8 effects) for circuits of depth 256, the whole cost of a 3-qubit exact
// Jacobian (the contraction itself takes 0.1 ms).
//
// Here ONE workgroup (4 wavefronts) walks a task with ALL its start vectors at once: the state is a [nv <= 16][64] row
// block X in LDS (nv = 1 forward, = the number of effects backward), a gate application is X' = X M with M the gate in row
// form (gates_t forward, gates backward: exactly the arrays the row kernel is handed), wavefront w produces columns
// 16 w .. 16 w + 15 as 16 chained v_mfma_f64_16x16x4_f64:
//      A operand (lane l, k-step s)  X[l & 15][4 s + (l >> 4)]                 one ds_read_b64 each (row stride 68: conflict-free)
//      B operand                     M[4 s + (l >> 4)][16 w + (l & 15)]        128-byte row segments, the NEXT gate's in flight
//      D         (lane l, reg r)     X'[(l >> 4) + 4 r][16 w + (l & 15)]       -> LDS (other buffer) and the state cache
// One barrier per step (the row blocks ping-pong).  The walk program (RHO / APPLY / NODE / SAVE / LOAD / EMIT) and the
// cache layout ([state][start vector][64]) are the row kernel's; results differ from it by re-association only.
#include "gst_kernels.hpp"

#include "../../include/gstfwd.h"

namespace gst {

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int C64_D = 64;
constexpr int C64_XS = 68;              // row stride of X in LDS (doubles): rows 4 doubles apart modulo 32 banks-of-8-bytes

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence as well: it drains vmcnt --
// the state-cache stores just issued (nobody in this launch reads them) and the next gate's prefetch -- on every step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// VEC: a single start vector (the forward pass).  A 16-row MFMA tile would carry one live row -- and MI355X runs fp64 MFMA at
// the VALU's rate, so the padding is not free (1,024 matrix-pipe cycles per step and wavefront).  Instead thread
// (column 16 w + (l & 15), k-quarter l >> 4) sums its 16 products and the four quarters meet in two shuffles: 32 VALU
// instructions per step, the same coalesced 128-byte row segments of M.  (Measured: 0.48 -> 0.29 ms for depth-256 circuits.
// What a step costs now is its own dependent chain -- LDS read, 16 ordered adds, two cross-lane steps, LDS write, barrier:
// ~0.5 us -- plus what of the next gate's trip to L2 the step does not cover; operands two gates ahead in rotating
// register sets changed nothing.)
template <bool VEC>
__global__ __launch_bounds__(256) void chain64_mfma_kernel(const WalkArgs a, const int n_slots)
{
    constexpr int D = C64_D, XS = C64_XS;
    extern __shared__ double lds[];                        // X[2][16][XS] | slots[n_slots][nv][D]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int64_t task = blockIdx.x;
    const bool multi = a.multi_start > 0;
    const int nv = multi ? ((a.multi_start - a.start0) < 16 ? (a.multi_start - a.start0) : 16) : 1;
    const int64_t cstride = multi ? (int64_t)a.multi_start * D : (int64_t)D;
    const int64_t coff = multi ? (int64_t)a.start0 * D : 0;
    double* X = lds;                                       // the current row block; Y = the other one
    double* Y = lds + 16 * XS;
    double* const slots = lds + 2 * 16 * XS;

    for (int k = tid; k < 2 * 16 * XS; k += 256) lds[k] = 0.0;          // rows >= nv stay zero for the whole walk
    __syncthreads();

    const int64_t pc0 = a.task_off[task];
    const int32_t n_words = (int32_t)(a.task_off[task + 1] - pc0);
    const uint32_t* gprog = a.prog + pc0;
    int32_t wbase = 0, pc = 0;
    uint32_t win_cur = (lane < n_words) ? gprog[lane] : 0u;
    uint32_t win_nxt = (64 + lane < n_words) ? gprog[64 + lane] : 0u;
    uint32_t op, arg;
#define C64_FETCH()                                                                                   \
    do {                                                                                              \
        if (pc - wbase == 64) {                                                                       \
            wbase += 64;                                                                              \
            win_cur = win_nxt;                                                                        \
            win_nxt = (wbase + 64 + lane < n_words) ? gprog[wbase + 64 + lane] : 0u;                  \
        }                                                                                             \
        const uint32_t w_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, (int)(pc - wbase));     \
        op = GST_OP(w_); arg = GST_ARG(w_); pc++;                                                     \
    } while (0)
    // B operand of gate g for this wavefront's 16 columns: b[s] = M[4 s + lk][16 w + lr]
    // (VEC: b[s] = M[16 lk + s][16 w + lr])
#define C64_LOADB(dst, g_)                                                                            \
    do {                                                                                              \
        const double* M_ = a.gates_t + (int64_t)(g_) * D * D + (VEC ? 16 * lk : lk) * D + 16 * w + lr; \
        _Pragma("unroll") for (int s = 0; s < 16; s++) dst[s] = M_[s * (VEC ? 1 : 4) * D];           \
    } while (0)

    C64_FETCH();
    for (;;) {
        if (op == GST_OP_END) break;
        if (op == GST_OP_APPLY) {
            double b[16], bn[16];
            C64_LOADB(b, arg);
            for (;;) {
                C64_FETCH();                                           // the NODE marker of the state being produced
                const int32_t node_id = (int32_t)arg;
                C64_FETCH();                                           // what follows
                const bool more = (op == GST_OP_APPLY);
                if (more) C64_LOADB(bn, arg);
                if constexpr (VEC) {
                    const double* xq = X + 16 * lk;                    // (row 0; the same address in all 16 lanes of a quarter)
                    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;     // four independent partial sums: shorter dependent chain
#pragma unroll
                    for (int s = 0; s < 16; s += 4) { p0 += xq[s] * b[s]; p1 += xq[s + 1] * b[s + 1]; p2 += xq[s + 2] * b[s + 2]; p3 += xq[s + 3] * b[s + 3]; }
                    double part = (p0 + p1) + (p2 + p3);
                    part += __shfl_xor(part, 16, 64);
                    part += __shfl_xor(part, 32, 64);
                    if (lk == 0) {
                        Y[16 * w + lr] = part;
                        if (a.base_cache_w) a.base_cache_w[(int64_t)node_id * cstride + coff + 16 * w + lr] = part;
                    }
                    lds_barrier();
                    { double* t = X; X = Y; Y = t; }
                    if (!more) break;
#pragma unroll
                    for (int s = 0; s < 16; s++) b[s] = bn[s];
                    continue;
                }
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
                const double* xa = X + lr * XS + lk;
                double xr[16];                                         // all 16 LDS reads in flight before the first MFMA
#pragma unroll
                for (int s = 0; s < 16; s++) xr[s] = xa[4 * s];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[s], b[s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = lk + 4 * r;
                    if (row < nv) {
                        Y[row * XS + 16 * w + lr] = acc[r];
                        if (a.base_cache_w) a.base_cache_w[(int64_t)node_id * cstride + coff + row * D + 16 * w + lr] = acc[r];
                    }
                }
                lds_barrier();
                { double* t = X; X = Y; Y = t; }
                if (!more) break;
#pragma unroll
                for (int s = 0; s < 16; s++) b[s] = bn[s];
            }
            continue;                                                  // `op` already holds the next instruction
        } else if (op == GST_OP_NODE) {
            if (a.base_cache_w)
                for (int k = tid; k < nv * D; k += 256) a.base_cache_w[(int64_t)arg * cstride + coff + k] = X[(k >> 6) * XS + (k & 63)];
        } else if (op == GST_OP_EMIT) {
            const int32_t x0 = a.eff_ptr[arg], x1 = a.eff_ptr[arg + 1];
            if (!multi && a.out) {
                const double xv = X[lane];
                for (int32_t x = x0 + w; x < x1; x += 4) {
                    const double p = wave_sum(a.effects[(int64_t)a.eff_label[x] * D + lane] * xv);
                    if (lane == 0) a.out[a.eff_dest[x]] = p;
                }
            }
        } else if (op == GST_OP_SAVE) {
            double* s_ = slots + (size_t)arg * nv * D;
            for (int k = tid; k < nv * D; k += 256) s_[k] = X[(k >> 6) * XS + (k & 63)];
            lds_barrier();
        } else if (op == GST_OP_LOAD) {
            const double* s_ = slots + (size_t)arg * nv * D;
            for (int k = tid; k < nv * D; k += 256) Y[(k >> 6) * XS + (k & 63)] = s_[k];
            lds_barrier();
            { double* t = X; X = Y; Y = t; }
        } else {                                                        // GST_OP_RHO
            for (int k = tid; k < nv * D; k += 256)
                Y[(k >> 6) * XS + (k & 63)] = a.rhos[(int64_t)(multi ? (uint32_t)(a.start0 + (k >> 6)) : arg) * D + (k & 63)];
            lds_barrier();
            { double* t = X; X = Y; Y = t; }
        }
        C64_FETCH();
    }
#undef C64_FETCH
#undef C64_LOADB
}

}  // namespace

// LDS the walk needs for `nv` start vectors and `n_slots` save slots; plans beyond a CU's LDS keep the row kernel.
bool chain64_fits(int nv, int n_slots)
{
    return ((size_t)2 * 16 * C64_XS + (size_t)(n_slots > 0 ? n_slots : 1) * nv * C64_D) * sizeof(double) <= 156 * 1024;
}

// The S = 0 walk of a D = 64 plan on the matrix cores: same arguments as launch_walk_rows (rows_S = 0, EMIT_PROBS); with
// a.multi_start > 0 one launch covers start vectors a.start0 .. min(a.start0 + 16, a.multi_start) - 1.
hipError_t launch_chain64(const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream)
{
    if (n_tasks <= 0) return hipSuccess;
    if (n_tasks > 0x7fffffffLL || a.rows_S != 0 || a.mode != EMIT_PROBS || a.n_models > 0) return hipErrorInvalidValue;
    const int nv = a.multi_start > 0 ? std::min(16, a.multi_start - a.start0) : 1;
    const size_t lds_bytes = ((size_t)2 * 16 * C64_XS + (size_t)(n_slots > 0 ? n_slots : 1) * nv * C64_D) * sizeof(double);
    (void)hipGetLastError();
    if (lds_bytes > 64 * 1024) {
        if (lds_bytes > 156 * 1024) return hipErrorInvalidValue;
        hipError_t e = nv == 1 ? hipFuncSetAttribute((const void*)chain64_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
                               : hipFuncSetAttribute((const void*)chain64_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (nv == 1) hipLaunchKernelGGL(chain64_mfma_kernel<true>, dim3((unsigned)n_tasks), dim3(256), lds_bytes, stream, a, n_slots);
    else hipLaunchKernelGGL(chain64_mfma_kernel<false>, dim3((unsigned)n_tasks), dim3(256), lds_bytes, stream, a, n_slots);
    return hipGetLastError();
}

}  // namespace gst
