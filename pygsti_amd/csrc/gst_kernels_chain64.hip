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
//      A operand (lane l, k-step s)  X[l & 15][4 s + (l >> 4)]                 LDS reads (row stride 65: conflict-free as ds_read2_b64)
//      B operand                     M[4 s + (l >> 4)][16 w + (l & 15)]        128-byte row segments, the NEXT gate's in flight
//      D         (lane l, reg r)     X'[(l >> 4) + 4 r][16 w + (l & 15)]       -> LDS (other buffer) and the state cache
// One barrier per step (the row blocks ping-pong).  The walk program (RHO / APPLY / NODE / SAVE / LOAD / EMIT) and the
// cache layout ([state][start vector][64]) are the row kernel's; results differ from it by re-association only.
#include "gst_kernels.hpp"

#include "../../include/gstfwd.h"

namespace gst {

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) uint32_t* c64_prog_t;     // (walk programs are read through the scalar cache)
constexpr int C64_D = 64;
// Row stride of X in LDS (doubles), per kernel: what matters is how the sixteen A-operand reads X[lr][4 s + lk] of a lane
// (dword address 2 (XS lr + 4 s + lk)) fall on the banks, and the LDS serves each read instruction in its own lane groups
// (MI355X_MICROARCH.md, LDS): the compiler pairs chain64_mfma_kernel's reads into ds_read2_b64 -- 16 consecutive lanes
// (lr = 0..15) at a time over 32 banks: XS = 65 gives bank 2 lr, conflict-free (round 4's 68 gave 8 lr mod 32: 4-way, 1,024
// LDS cycles per step and workgroup -- as much as the step's 16 MFMAs); chain64_resident_kernel issues ds_read_b64 itself --
// 32 lanes (lr = 0..15, lk = 0..1) at a time over 64 banks: XS = 66 gives bank 4 lr + 2 lk, all 64 banks once.
constexpr int C64_XS = 65;
constexpr int C64R_XS = 66;

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

    // The walk program through the SCALAR cache (constant address space: s_load_dword).  A 64-word window in vector
    // registers read with v_readlane, refilled by a load under a rare branch, made the compiler put `s_waitcnt vmcnt(0)` in
    // front of every v_readlane -- i.e. every step waited for the state-cache stores of the step before and for the next
    // gate's operands it had just requested (round 6; the prefetch then only covered the step's own MFMAs).
    const c64_prog_t gprog = (c64_prog_t)(a.prog + a.task_off[task]);
    int32_t pc = 0;
    uint32_t op, arg;
#define C64_FETCH()                                                                                   \
    do {                                                                                              \
        const uint32_t w_ = gprog[pc++];                                                              \
        op = GST_OP(w_); arg = GST_ARG(w_);                                                           \
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
                    for (int s = 0; s < 16; s += 4) {                   // (fused: this walk has no ordering contract)
                        p0 = __builtin_fma(xq[s], b[s], p0); p1 = __builtin_fma(xq[s + 1], b[s + 1], p1);
                        p2 = __builtin_fma(xq[s + 2], b[s + 2], p2); p3 = __builtin_fma(xq[s + 3], b[s + 3], p3);
                    }
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


// ---------------------------------------------------------------------------------------------------------------------------
// The same walk with the model's gates RESIDENT in registers (the backward pass of the exact Jacobian on random circuits).
//
// chain64_mfma_kernel pulls the 32 KB of every gate it applies through L2 (16 GB per backward pass over 4,000 random circuits
// of depth <= 256, 514 k applications) and hides that behind four workgroups per CU.  A three-qubit model has about ten
// gates: wavefront w's B operands of ALL of them -- M_g[4 s + lk][16 w + lr], 16 doubles per gate and lane -- are 320 of a
// lane's 512 registers (256 architectural + 256 accumulation: MFMA takes its A and B operands from either file on gfx950).
// One workgroup per CU (one wavefront per SIMD: that is what a 512-register wavefront leaves), persistent: it loads the
// fragments once and pops tasks, longest first, from a counter.  No global load is left on the dependent path of a step.
//
// With ONE wavefront per SIMD nothing hides anything else, so:
//  * the walk program is read through the SCALAR cache (constant address space: s_load_dword -- no vector registers, no
//    v_readlane window, no vmcnt, which counts the state-cache stores of the step before as well); lane offsets are formed
//    once; the state-cache address is a scalar base + a 32-bit lane offset; the row blocks ping-pong by one scalar offset;
//  * a step's LDS reads and MFMAs are ONE asm statement (below);
//  * the workgroup walks TWO tasks at a time: between the last MFMA of a step and the first of the task's next step lie the
//    result's trip through LDS to the other three wavefronts, the barrier and the program decode -- about as long as the
//    sixteen MFMAs themselves (measured: 0.87 us per step against 0.48 us of MFMAs).  The second task's MFMAs run there:
//    both tasks' A operands are read behind ONE barrier, task 1's reads issue between task 0's MFMAs, task 0's result is
//    written while task 1's MFMAs are in the pipe.
// The asm statements of a step.  The B operands stay where they were loaded -- architectural registers for the first
// C64_VGPR_GATES gates ("v"), accumulation registers for the rest ("a").  Left to the register allocator, the 320 operand
// registers are spilled to the accumulation file and every MFMA gets a two-instruction reload into ONE temporary pair in
// front of it, which the MFMA before it is still reading (measured: 142 instead of 64 cycles per MFMA).  Every MFMA waits
// for its own A operand with a counted lgkmcnt (LDS reads return in order; a scalar load still in flight only makes the
// count conservative); the s_nop tail covers the wait states between the last MFMA's write and the stores that read it.
constexpr int C64_VGPR_GATES = 3;
// one task: 16 reads of X[xa32 ...], 16 MFMAs
#define C64_STEP(CON, B)                                                                              \
    asm volatile(                                                                                     \
                 "ds_read_b64 %1, %17\n\t"                                                                           \
                 "ds_read_b64 %2, %17 offset:32\n\t"                                                                 \
                 "ds_read_b64 %3, %17 offset:64\n\t"                                                                 \
                 "ds_read_b64 %4, %17 offset:96\n\t"                                                                 \
                 "ds_read_b64 %5, %17 offset:128\n\t"                                                                \
                 "ds_read_b64 %6, %17 offset:160\n\t"                                                                \
                 "ds_read_b64 %7, %17 offset:192\n\t"                                                                \
                 "ds_read_b64 %8, %17 offset:224\n\t"                                                                \
                 "ds_read_b64 %9, %17 offset:256\n\t"                                                                \
                 "ds_read_b64 %10, %17 offset:288\n\t"                                                               \
                 "ds_read_b64 %11, %17 offset:320\n\t"                                                               \
                 "ds_read_b64 %12, %17 offset:352\n\t"                                                               \
                 "ds_read_b64 %13, %17 offset:384\n\t"                                                               \
                 "ds_read_b64 %14, %17 offset:416\n\t"                                                               \
                 "ds_read_b64 %15, %17 offset:448\n\t"                                                               \
                 "ds_read_b64 %16, %17 offset:480\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %1, %18, 0\n\t"                                \
                 "s_waitcnt lgkmcnt(14)\n\tv_mfma_f64_16x16x4_f64 %0, %2, %19, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(13)\n\tv_mfma_f64_16x16x4_f64 %0, %3, %20, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(12)\n\tv_mfma_f64_16x16x4_f64 %0, %4, %21, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(11)\n\tv_mfma_f64_16x16x4_f64 %0, %5, %22, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(10)\n\tv_mfma_f64_16x16x4_f64 %0, %6, %23, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(9)\n\tv_mfma_f64_16x16x4_f64 %0, %7, %24, %0\n\t"                                \
                 "s_waitcnt lgkmcnt(8)\n\tv_mfma_f64_16x16x4_f64 %0, %8, %25, %0\n\t"                                \
                 "s_waitcnt lgkmcnt(7)\n\tv_mfma_f64_16x16x4_f64 %0, %9, %26, %0\n\t"                                \
                 "s_waitcnt lgkmcnt(6)\n\tv_mfma_f64_16x16x4_f64 %0, %10, %27, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(5)\n\tv_mfma_f64_16x16x4_f64 %0, %11, %28, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(4)\n\tv_mfma_f64_16x16x4_f64 %0, %12, %29, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(3)\n\tv_mfma_f64_16x16x4_f64 %0, %13, %30, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(2)\n\tv_mfma_f64_16x16x4_f64 %0, %14, %31, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(1)\n\tv_mfma_f64_16x16x4_f64 %0, %15, %32, %0\n\t"                               \
                 "s_waitcnt lgkmcnt(0)\n\tv_mfma_f64_16x16x4_f64 %0, %16, %33, %0\n\t"                               \
                 "s_nop 15\n\ts_nop 7"                                                                               \
                 : "=&v"(acc), "=&v"(xr[0]), "=&v"(xr[1]), "=&v"(xr[2]), "=&v"(xr[3]), "=&v"(xr[4]), "=&v"(xr[5]), "=&v"(xr[6]), "=&v"(xr[7]), "=&v"(xr[8]), "=&v"(xr[9]), "=&v"(xr[10]), "=&v"(xr[11]), "=&v"(xr[12]), "=&v"(xr[13]), "=&v"(xr[14]), "=&v"(xr[15]) \
                 : "v"(xa32), CON(B[0]), CON(B[1]), CON(B[2]), CON(B[3]), CON(B[4]), CON(B[5]), CON(B[6]), CON(B[7]), CON(B[8]), CON(B[9]), CON(B[10]), CON(B[11]), CON(B[12]), CON(B[13]), CON(B[14]), CON(B[15]) \
                 : "memory")
// two tasks, first half: task 0's reads, then its MFMAs with task 1's reads between them (the read in front of MFMA s has
// exactly 15 younger LDS operations behind it at that point: 15 - s of task 0 and s of task 1)
#define C64_DUAL_A(CON, B)                                                                            \
    asm volatile(                                                                                     \
                 "ds_read_b64 %1, %33\n\t"                                                                           \
                 "ds_read_b64 %2, %33 offset:32\n\t"                                                                 \
                 "ds_read_b64 %3, %33 offset:64\n\t"                                                                 \
                 "ds_read_b64 %4, %33 offset:96\n\t"                                                                 \
                 "ds_read_b64 %5, %33 offset:128\n\t"                                                                \
                 "ds_read_b64 %6, %33 offset:160\n\t"                                                                \
                 "ds_read_b64 %7, %33 offset:192\n\t"                                                                \
                 "ds_read_b64 %8, %33 offset:224\n\t"                                                                \
                 "ds_read_b64 %9, %33 offset:256\n\t"                                                                \
                 "ds_read_b64 %10, %33 offset:288\n\t"                                                               \
                 "ds_read_b64 %11, %33 offset:320\n\t"                                                               \
                 "ds_read_b64 %12, %33 offset:352\n\t"                                                               \
                 "ds_read_b64 %13, %33 offset:384\n\t"                                                               \
                 "ds_read_b64 %14, %33 offset:416\n\t"                                                               \
                 "ds_read_b64 %15, %33 offset:448\n\t"                                                               \
                 "ds_read_b64 %16, %33 offset:480\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %1, %35, 0\n\t"                                \
                 "ds_read_b64 %17, %34\n\t"                                                                          \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %2, %36, %0\n\t"                               \
                 "ds_read_b64 %18, %34 offset:32\n\t"                                                                \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %3, %37, %0\n\t"                               \
                 "ds_read_b64 %19, %34 offset:64\n\t"                                                                \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %4, %38, %0\n\t"                               \
                 "ds_read_b64 %20, %34 offset:96\n\t"                                                                \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %5, %39, %0\n\t"                               \
                 "ds_read_b64 %21, %34 offset:128\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %6, %40, %0\n\t"                               \
                 "ds_read_b64 %22, %34 offset:160\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %7, %41, %0\n\t"                               \
                 "ds_read_b64 %23, %34 offset:192\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %8, %42, %0\n\t"                               \
                 "ds_read_b64 %24, %34 offset:224\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %9, %43, %0\n\t"                               \
                 "ds_read_b64 %25, %34 offset:256\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %10, %44, %0\n\t"                              \
                 "ds_read_b64 %26, %34 offset:288\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %11, %45, %0\n\t"                              \
                 "ds_read_b64 %27, %34 offset:320\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %12, %46, %0\n\t"                              \
                 "ds_read_b64 %28, %34 offset:352\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %13, %47, %0\n\t"                              \
                 "ds_read_b64 %29, %34 offset:384\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %14, %48, %0\n\t"                              \
                 "ds_read_b64 %30, %34 offset:416\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %15, %49, %0\n\t"                              \
                 "ds_read_b64 %31, %34 offset:448\n\t"                                                               \
                 "s_waitcnt lgkmcnt(15)\n\tv_mfma_f64_16x16x4_f64 %0, %16, %50, %0\n\t"                              \
                 "ds_read_b64 %32, %34 offset:480\n\t"                                                               \
                 "s_waitcnt lgkmcnt(0)"                                                                              \
                 : "=&v"(acc), "=&v"(xr[0]), "=&v"(xr[1]), "=&v"(xr[2]), "=&v"(xr[3]), "=&v"(xr[4]), "=&v"(xr[5]), "=&v"(xr[6]), "=&v"(xr[7]), "=&v"(xr[8]), "=&v"(xr[9]), "=&v"(xr[10]), "=&v"(xr[11]), "=&v"(xr[12]), "=&v"(xr[13]), "=&v"(xr[14]), "=&v"(xr[15]), "=&v"(xr1[0]), "=&v"(xr1[1]), "=&v"(xr1[2]), "=&v"(xr1[3]), "=&v"(xr1[4]), "=&v"(xr1[5]), "=&v"(xr1[6]), "=&v"(xr1[7]), "=&v"(xr1[8]), "=&v"(xr1[9]), "=&v"(xr1[10]), "=&v"(xr1[11]), "=&v"(xr1[12]), "=&v"(xr1[13]), "=&v"(xr1[14]), "=&v"(xr1[15]) \
                 : "v"(xa32), "v"(xb32), CON(B[0]), CON(B[1]), CON(B[2]), CON(B[3]), CON(B[4]), CON(B[5]), CON(B[6]), CON(B[7]), CON(B[8]), CON(B[9]), CON(B[10]), CON(B[11]), CON(B[12]), CON(B[13]), CON(B[14]), CON(B[15]) \
                 : "memory")
// two tasks, second half: task 1's MFMAs
#define C64_DUAL_B(CON, B)                                                                            \
    asm volatile(                                                                                     \
                 "v_mfma_f64_16x16x4_f64 %0, %1, %17, 0\n\t"                                                         \
                 "v_mfma_f64_16x16x4_f64 %0, %2, %18, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %3, %19, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %4, %20, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %5, %21, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %6, %22, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %7, %23, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %8, %24, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %9, %25, %0\n\t"                                                        \
                 "v_mfma_f64_16x16x4_f64 %0, %10, %26, %0\n\t"                                                       \
                 "v_mfma_f64_16x16x4_f64 %0, %11, %27, %0\n\t"                                                       \
                 "v_mfma_f64_16x16x4_f64 %0, %12, %28, %0\n\t"                                                       \
                 "v_mfma_f64_16x16x4_f64 %0, %13, %29, %0\n\t"                                                       \
                 "v_mfma_f64_16x16x4_f64 %0, %14, %30, %0\n\t"                                                       \
                 "v_mfma_f64_16x16x4_f64 %0, %15, %31, %0\n\t"                                                       \
                 "v_mfma_f64_16x16x4_f64 %0, %16, %32, %0\n\t"                                                       \
                 "s_nop 15\n\ts_nop 7"                                                                               \
                 : "=&v"(acc1)                                                                      \
                 : "v"(xr1[0]), "v"(xr1[1]), "v"(xr1[2]), "v"(xr1[3]), "v"(xr1[4]), "v"(xr1[5]), "v"(xr1[6]), "v"(xr1[7]), "v"(xr1[8]), "v"(xr1[9]), "v"(xr1[10]), "v"(xr1[11]), "v"(xr1[12]), "v"(xr1[13]), "v"(xr1[14]), "v"(xr1[15]), CON(B[0]), CON(B[1]), CON(B[2]), CON(B[3]), CON(B[4]), CON(B[5]), CON(B[6]), CON(B[7]), CON(B[8]), CON(B[9]), CON(B[10]), CON(B[11]), CON(B[12]), CON(B[13]), CON(B[14]), CON(B[15]))

#define C64_CASE(G, WHAT)                                                                             \
    case G:                                                                                           \
        if constexpr (G < NG) {                                                                       \
            if constexpr (G < C64_VGPR_GATES) WHAT("v", bfr[G < NG ? G : 0]);                         \
            else WHAT("a", bfr[G < NG ? G : 0]);                                                      \
        }                                                                                             \
        break;
#define C64_SWITCH(g, WHAT)                                                                           \
    switch (g) {                                                                                      \
        C64_CASE(0, WHAT) C64_CASE(1, WHAT) C64_CASE(2, WHAT) C64_CASE(3, WHAT) C64_CASE(4, WHAT) C64_CASE(5, WHAT) \
        C64_CASE(6, WHAT) C64_CASE(7, WHAT) C64_CASE(8, WHAT) C64_CASE(9, WHAT)                                       \
        default: break;                                                                               \
    }

struct C64Walk {               // one of the two walks of a workgroup (wave-uniform)
    c64_prog_t prog;
    int32_t pc;
    uint32_t word;             // the instruction the walk stands at
    uint32_t cur;              // its current row block is region[cur ...], the other region[BUF - cur ...]
    int32_t live;
};

template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void chain64_resident_kernel(const WalkArgs a, const int n_slots, const int n_tasks)
{
    constexpr int D = C64_D, XS = C64R_XS, BUF = 16 * XS;
    extern __shared__ double lds[];                        // per walk: X[2][16][XS] | slots[n_slots][nv][D]; then the pop word
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int nv = (a.multi_start - a.start0) < 16 ? (a.multi_start - a.start0) : 16;
    const int64_t cstride = (int64_t)a.multi_start * D;
    const int64_t coff = (int64_t)a.start0 * D;
    const int REG = 2 * BUF + (n_slots > 0 ? n_slots : 1) * nv * D;       // doubles per walk
    volatile int32_t* const next_task = (volatile int32_t*)(lds + 2 * REG);
    const uint32_t rd_off = (uint32_t)(lr * XS + lk);                   // A operand s of this lane: X[rd_off + 4 s]
    const uint32_t wr_off = (uint32_t)(lk * XS + 16 * w + lr);          // D register r of this lane: X'[wr_off + 4 r XS]
    const uint32_t g_off = (uint32_t)(lk * D + 16 * w + lr);            //   and state cache [g_off + 4 r D]
    const uint32_t lds32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    const bool store = a.base_cache_w != nullptr;

    double bfr[NG][16];                                    // this wavefront's B operands of every gate
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const double* M_ = a.gates_t + (int64_t)(g < a.n_gates ? g : 0) * D * D + lk * D + 16 * w + lr;
#pragma unroll
        for (int s = 0; s < 16; s++) bfr[g][s] = M_[s * 4 * D];
    }
    // Pin every fragment in its register file HERE (an empty asm that reads and redefines it): the compiler's wait for the
    // load then stands in this prologue.  Otherwise it stands in front of the fragment's first use, inside the walk loop --
    // `s_waitcnt vmcnt(0)` for the gate loaded last, which in steady state waits for the state-cache stores of the step before.
#pragma unroll
    for (int g = 0; g < NG; g++) {
#pragma unroll
        for (int s = 0; s < 16; s++) {
            if (g < C64_VGPR_GATES) asm volatile("" : "+v"(bfr[g][s]));
            else asm volatile("" : "+a"(bfr[g][s]));
        }
    }
    for (int k = tid; k < 2 * REG; k += 256) lds[k] = 0.0;              // rows >= nv of the row blocks stay zero for the whole launch
    __syncthreads();

    bool exhausted = false;
    // Bring a walk to its next APPLY: pop a task when it has none, execute everything that is not a gate application.
    auto prepare = [&](C64Walk& S, double* const region) {
        for (;;) {
            if (!S.live) {
                if (exhausted) return;
                if (tid == 0) *next_task = (int32_t)atomicAdd(a.bin_head, 1u);
                __syncthreads();
                const int32_t item = __builtin_amdgcn_readfirstlane(*next_task);
                __syncthreads();
                if (item >= n_tasks) { exhausted = true; return; }
                const int64_t task = __builtin_amdgcn_readfirstlane(a.block_order ? (int32_t)a.block_order[item] : item);
                S.prog = (c64_prog_t)(a.prog + a.task_off[task]);
                S.pc = 1; S.cur = 0; S.live = 1;
                S.word = S.prog[0];
            }
            const uint32_t op = GST_OP(S.word), arg = GST_ARG(S.word);
            if (op == GST_OP_APPLY) return;
            if (op == GST_OP_END) { S.live = 0; continue; }
            double* const X = region + S.cur;
            double* const Y = region + (BUF - S.cur);
            double* const slots = region + 2 * BUF;
            if (op == GST_OP_NODE) {
                if (a.base_cache_w)
                    for (int k = tid; k < nv * D; k += 256) a.base_cache_w[(int64_t)arg * cstride + coff + k] = X[(k >> 6) * XS + (k & 63)];
            } else if (op == GST_OP_SAVE) {
                double* s_ = slots + (size_t)arg * nv * D;
                for (int k = tid; k < nv * D; k += 256) s_[k] = X[(k >> 6) * XS + (k & 63)];
                lds_barrier();
            } else if (op == GST_OP_LOAD) {
                const double* s_ = slots + (size_t)arg * nv * D;
                for (int k = tid; k < nv * D; k += 256) Y[(k >> 6) * XS + (k & 63)] = s_[k];
                lds_barrier();
                S.cur = BUF - S.cur;
            } else if (op == GST_OP_RHO) {
                for (int k = tid; k < nv * D; k += 256)
                    Y[(k >> 6) * XS + (k & 63)] = a.rhos[(int64_t)(a.start0 + (k >> 6)) * D + (k & 63)];
                lds_barrier();
                S.cur = BUF - S.cur;
            }                                                          // (GST_OP_EMIT: nothing to do for a multi-start walk)
            S.word = S.prog[S.pc++];
        }
    };
    // The result of a step into the walk's other row block and the state cache.
    auto put = [&](const d4_t& acc, double* const ya, const uint32_t node_w) {
        double* const gb = a.base_cache_w + ((int64_t)GST_ARG(node_w) * cstride + coff);
        if (nv == 8) {                                                  // rows lk and lk + 4 of every lane
            ya[0] = acc[0]; ya[4 * XS] = acc[1];
            if (store) { gb[g_off] = acc[0]; gb[g_off + 4 * D] = acc[1]; }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (lk + 4 * r < nv) {
                    ya[4 * r * XS] = acc[r];
                    if (store) gb[g_off + 4 * r * D] = acc[r];
                }
        }
    };
    auto single = [&](C64Walk& S, const uint32_t region_off) {
        // this step's NODE marker and the instruction after it: in flight under the LDS reads and the MFMAs
        const uint32_t node_w = S.prog[S.pc], next_w = S.prog[S.pc + 1];
        S.pc += 2;
        d4_t acc = {0.0, 0.0, 0.0, 0.0};                                // (overwritten by the step's first MFMA: C = 0)
        const uint32_t xa32 = lds32 + 8u * (region_off + S.cur + rd_off);   // LDS byte address of this lane's first A operand
        double xr[16];
        C64_SWITCH(GST_ARG(S.word), C64_STEP)
        put(acc, lds + region_off + (BUF - S.cur) + wr_off, node_w);
        lds_barrier();
        S.cur = BUF - S.cur;
        S.word = next_w;
    };

    C64Walk S0, S1;
    S0.live = 0; S1.live = 0; S0.pc = S1.pc = 0; S0.word = S1.word = 0; S0.cur = S1.cur = 0; S0.prog = S1.prog = (c64_prog_t)a.prog;
    for (;;) {
        prepare(S0, lds);
        prepare(S1, lds + REG);
        if (S0.live && S1.live) {
            const uint32_t node0 = S0.prog[S0.pc], next0 = S0.prog[S0.pc + 1];
            const uint32_t node1 = S1.prog[S1.pc], next1 = S1.prog[S1.pc + 1];
            S0.pc += 2; S1.pc += 2;
            d4_t acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            const uint32_t xa32 = lds32 + 8u * (S0.cur + rd_off), xb32 = lds32 + 8u * (REG + S1.cur + rd_off);
            double xr[16], xr1[16];
            C64_SWITCH(GST_ARG(S0.word), C64_DUAL_A)
            C64_SWITCH(GST_ARG(S1.word), C64_DUAL_B)
            put(acc, lds + (BUF - S0.cur) + wr_off, node0);
            put(acc1, lds + REG + (BUF - S1.cur) + wr_off, node1);
            lds_barrier();
            S0.cur = BUF - S0.cur; S1.cur = BUF - S1.cur;
            S0.word = next0; S1.word = next1;
        } else if (S0.live) {
            single(S0, 0u);
        } else if (S1.live) {
            single(S1, (uint32_t)REG);
        } else {
            break;
        }
    }
}
#undef C64_STEP
#undef C64_DUAL_A
#undef C64_DUAL_B
#undef C64_CASE
#undef C64_SWITCH

}  // namespace

bool chain64_resident_fits(int nv, int n_slots, int n_gates)
{
    return nv > 1 && n_gates <= 10 &&
           (2 * ((size_t)2 * 16 * C64R_XS + (size_t)(n_slots > 0 ? n_slots : 1) * nv * C64_D) + 2) * sizeof(double) <= 156 * 1024;
}

// The multi-start walk (a.multi_start > 0) with resident gates: `n_cus` persistent workgroups; a.bin_head = a zeroed
// counter, a.block_order = the tasks longest first (or NULL).
hipError_t launch_chain64_resident(const WalkArgs& a, int64_t n_tasks, int n_slots, int n_cus, hipStream_t stream)
{
    if (n_tasks <= 0) return hipSuccess;
    if (n_tasks > 0x7fffffffLL || a.rows_S != 0 || a.mode != EMIT_PROBS || a.n_models > 0 || a.multi_start <= 0 || !a.bin_head) return hipErrorInvalidValue;
    const int nv = std::min(16, a.multi_start - a.start0);
    if (!chain64_resident_fits(nv, n_slots, a.n_gates)) return hipErrorInvalidValue;
    const size_t lds_bytes = (2 * ((size_t)2 * 16 * C64R_XS + (size_t)(n_slots > 0 ? n_slots : 1) * nv * C64_D) + 2) * sizeof(double);
    const unsigned grid = (unsigned)std::min<int64_t>(n_tasks, n_cus > 0 ? n_cus : 256);
    (void)hipGetLastError();
    auto go = [&](auto kern) -> hipError_t {
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, stream, a, n_slots, (int)n_tasks);
        return hipGetLastError();
    };
    return go(chain64_resident_kernel<10>);
}

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
