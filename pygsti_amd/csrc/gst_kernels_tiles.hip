// gst_kernels_tiles.hip -- the exact Jacobian's contraction as a TILED product (D = 16, four outcomes per circuit).
//
// analytic_mfma_kernel (gst_kernels_analytic.hip) gives every work item -- one or two circuits -- to one wavefront, which
// gathers 128 + 128 + 512 bytes of state vectors per pair of gate applications.  But a GST design is a product: the circuits
// prep_i . W . meas_m of one middle string W (a germ power) read the SAME forward states F_i[k] whatever m is and the SAME
// backward states B_m[k] whatever i is -- 176 circuits of the 2Q design share 16 + 11 chains, and the item kernel pulls every
// vector through the L1s 8.8 times (its wavefronts wait on the L1 miss queue, TCP_PENDING_STALL 70 %).  Here a WORKGROUP owns a
// tile of up to 8 rows (forward chains) x 4 columns (backward chains) = 32 circuits; the tile's middle segment is a stream of
// 4-slot blocks whose 8 + 4 x 4 state vectors are staged ONCE per block in LDS (12 KB, double-buffered, indices two blocks
// and vectors one block ahead of the MFMAs) and every wavefront -- 4 rows x 1 column x 4 outcomes = 16 accumulator tiles --
// reads its 4 + 4 operands from there: 12 KB of L2 traffic per 128 v_mfma_f64_16x16x4_f64 instead of 48 KB.
// What a circuit has outside the middle segment (its fiducials: a handful of applications whose backward / forward states
// are its own) are "remnants": gathered straight from the caches by the wavefront that owns the circuit, issued before the
// gate's segment stream and multiplied after it.  The host (gst_fill_analytic.cpp: build_tiles) finds the tiles from the two
// tries alone -- nothing about fiducials or germs is assumed; circuits that fit no tile stay with the item kernel.
// Output layout, SPAM columns, structural zeros: exactly analytic_mfma_kernel's.
#include "gst_kernels.hpp"

#include "../../include/gstfwd.h"

namespace gst {

namespace {

typedef double t_d4 __attribute__((ext_vector_type(4)));
typedef double t_d2 __attribute__((ext_vector_type(2)));
constexpr int TD = 16, TNX = 4;
constexpr int TR = TILE_ROWS, TC = TILE_COLS;          // 8 x 4

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

__global__ __launch_bounds__(512) void analytic_tile_kernel(const TileArgs t)
{
    const AnaArgs& a = t.a;
    constexpr int D = TD, NX = TNX;
    __shared__ __attribute__((aligned(16))) double Fs[2][4][TR][D];            // [buffer][slot][row][component b]
    __shared__ __attribute__((aligned(16))) double Bs[2][4][TC][NX][D];        // [buffer][slot][column][outcome x][component a]
    __shared__ int32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int ig = w & 1, mc = w >> 1;                 // this wavefront: rows 4 ig .. 4 ig + 3, column mc
    const int nG = a.n_gates, nE = a.n_effects;
    const bool zeros_resident = a.zeros_resident && (!a.zeros_ok || *(const volatile uint32_t*)a.zeros_ok != 0u);
    const char* const fb = (const char*)a.base_cache;
    const char* const rb = (const char*)a.rev_cache;
    const uint32_t fstride = (uint32_t)(D * 8), rstride = (uint32_t)nE * (D * 8);
    // staging roles of this thread: one double of a forward vector, two doubles of a backward vector
    const int sF_v = tid >> 4, sF_c = tid & 15;        // vector (slot = v >> 3, row = v & 7), component
    const int sB_v = tid >> 5, sB_q = tid & 31;        // vector (slot = v >> 2, column = v & 3), doubles 2 q, 2 q + 1 of [a][x]
    const int sB_a = sB_q >> 1, sB_x = (sB_q & 1) * 2;

    for (;;) {
        __syncthreads();
        if (tid == 0) s_tile = (int32_t)atomicAdd(t.counter, 1u);
        __syncthreads();
        const int32_t ti = __builtin_amdgcn_readfirstlane(s_tile);
        if (ti >= t.n_tiles) break;
        const int32_t tile = __builtin_amdgcn_readfirstlane(t.tile_order[ti]);
        // ---- this wavefront's four circuits, their rows of the Jacobian, their SPAM columns -------------------------------
        const int32_t cid_l = t.tile_cid[(int64_t)tile * (TR * TC) + (4 * ig + (lane & 3)) * TC + mc];
        int32_t c[4];
        int64_t dest[4][NX];
#pragma unroll
        for (int li = 0; li < 4; li++) {
            c[li] = __builtin_amdgcn_readlane(cid_l, li);
            const int32_t x0 = c[li] >= 0 ? a.eff_ptr[c[li]] : 0;
            const int64_t dl = c[li] >= 0 ? (int64_t)a.eff_dest[x0 + lk] : 0;      // lane group lk <-> outcome lk
            const int32_t el = c[li] >= 0 ? a.eff_label[x0 + lk] : 0;
#pragma unroll
            for (int x = 0; x < NX; x++) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(dl & 0xffffffffLL), 16 * x);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(dl >> 32), 16 * x);
                dest[li][x] = (int64_t)(((uint64_t)hi << 32) | lo);
            }
            if (c[li] >= 0) {       // dp/dE[a] = F_n[a] for the outcome's own effect, zeros for the others; dp/drho[b] = B_0[b]
                const int32_t fleaf = a.circ_leaf[c[li]], rleaf = a.rev_leaf[c[li]], rsym = a.circ_rho[c[li]];
                const double FL = *(const double*)(fb + (int64_t)fleaf * fstride + lr * 8);
                for (int e2 = 0; e2 < nE; e2++) {
                    const int32_t ce = a.colmap_eff[e2 * D + lr];
                    if (ce >= 0) a.out[dl * a.ld + ce] = (e2 == el) ? FL : 0.0;
                }
                const double B0 = *(const double*)(rb + (int64_t)rleaf * rstride + ((int64_t)lr * nE + el) * 8);
                for (int r2 = 0; r2 < a.n_rhos; r2++) {
                    const int32_t cr = a.colmap_rho[r2 * D + lr];
                    if (cr >= 0) a.out[dl * a.ld + cr] = (r2 == rsym) ? B0 : 0.0;
                }
            }
        }
        const int32_t* const tblk = t.tile_blk + (int64_t)tile * (nG + 1);
        for (int g = 0; g < nG; g++) {
            const int32_t c0 = __builtin_amdgcn_readfirstlane(a.gate_col0[g]);
            if (c0 == -2) continue;                                        // no parameter of this gate was requested
            const int32_t b0 = __builtin_amdgcn_readfirstlane(tblk[g]), b1 = __builtin_amdgcn_readfirstlane(tblk[g + 1]);
            const int32_t* const rp = t.rem_ptr + ((int64_t)tile * nG + g) * (TR * TC + 1);
            int32_t r0[4], r1[4];
            bool any_rem = false;
#pragma unroll
            for (int li = 0; li < 4; li++) {
                const int slot = (4 * ig + li) * TC + mc;
                r0[li] = __builtin_amdgcn_readfirstlane(rp[slot]);
                r1[li] = c[li] >= 0 ? __builtin_amdgcn_readfirstlane(rp[slot + 1]) : r0[li];
                any_rem = any_rem || r1[li] > r0[li];
            }
            t_d4 acc[4][NX];
#pragma unroll
            for (int li = 0; li < 4; li++)
#pragma unroll
                for (int x = 0; x < NX; x++) acc[li][x] = (t_d4){0.0, 0.0, 0.0, 0.0};
            // ---- the tile's segment of gate g: blocks b0 .. b1 - 1 through LDS ------------------------------------------------
            if (b1 > b0 && !(t.debug & 4)) {          // (workgroup-uniform)
                // (no load sits under a branch -- the compiler counts the loads in flight exactly only on straight-line code, and
                //  otherwise waits for ALL of them before every use: blocks past the end re-read the last one and come out dead)
                // (and nothing is DONE with a loaded index before its ring slot comes round: a select right behind the load would
                //  make every iteration wait for the load it has just issued -- dead blocks are zeroed when they are staged)
                auto ids = [&](int32_t b, int32_t& fi, int32_t& ri) {
                    const int32_t bc = b < b1 ? b : b1 - 1;
                    fi = t.tsf[(int64_t)bc * (4 * TR) + sF_v];
                    ri = t.tsr[(int64_t)bc * (4 * TC) + sB_v];
                };
                auto fetch = [&](int32_t fi, int32_t ri, double& fv, t_d2& bv) {
                    fv = *(const double*)(fb + (uint64_t)(uint32_t)(fi < 0 ? 0 : fi) * fstride + sF_c * 8);
                    bv = *(const t_d2*)__builtin_assume_aligned(rb + (uint64_t)(uint32_t)(ri < 0 ? 0 : ri) * rstride + sB_q * 16, 16);
                };
                auto stage = [&](int buf, bool live, int32_t fi, int32_t ri, double fv, t_d2 bv) {
                    (&Fs[buf][0][0][0])[sF_v * D + sF_c] = (!live || fi < 0) ? 0.0 : fv;
                    double* bd = &Bs[buf][0][0][0][0] + sB_v * (NX * D);
                    bd[sB_x * D + sB_a] = (!live || ri < 0) ? 0.0 : bv.x;
                    bd[(sB_x + 1) * D + sB_a] = (!live || ri < 0) ? 0.0 : bv.y;
                };
                // Register ring: the vectors of block b + PF and the indices of block b + 2 PF are requested while block b is
                // multiplied -- with two wavefronts per SIMD nothing else hides the L2 / HBM latency of the gathers.
                constexpr int PF = 4;
                int32_t fi[PF], ri[PF], fin[PF], rin[PF];
                double fvq[PF]; t_d2 bvq[PF];
#pragma unroll
                for (int j = 0; j < PF; j++) ids(b0 + j, fi[j], ri[j]);
#pragma unroll
                for (int j = 0; j < PF; j++) ids(b0 + PF + j, fin[j], rin[j]);
#pragma unroll
                for (int j = 0; j < PF; j++) fetch(fi[j], ri[j], fvq[j], bvq[j]);
                __syncthreads();                                           // (nobody still reads either buffer)
                stage(0, true, fi[0], ri[0], fvq[0], bvq[0]);
                fi[0] = fin[0]; ri[0] = rin[0];
                fetch(fi[0], ri[0], fvq[0], bvq[0]);                       // block b0 + PF into the freed ring slot
                ids(b0 + 2 * PF, fin[0], rin[0]);
                lds_barrier();
                int buf = 0;
                for (int32_t bg = b0; bg < b1; bg += PF) {
#pragma unroll
                    for (int j = 0; j < PF; j++) {
                        const int32_t b = bg + j;                          // (b >= b1 in the last group: a dead block, all zeros)
                        double Fo[4], Ao[NX];
#pragma unroll
                        for (int li = 0; li < 4; li++) Fo[li] = Fs[buf][lk][4 * ig + li][lr];
#pragma unroll
                        for (int x = 0; x < NX; x++) Ao[x] = Bs[buf][lk][mc][x][lr];
#pragma unroll
                        for (int li = 0; li < 4; li++)
#pragma unroll
                            for (int x = 0; x < NX; x++) acc[li][x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ao[x], Fo[li], acc[li][x], 0, 0, 0);
                        {                                                  // block b + 1: ring slot -> the other buffer
                            const int q = (j + 1) % PF;                    // ring slot of block b + 1 (a constant once unrolled)
                            stage(buf ^ 1, b + 1 < b1, fi[q], ri[q], fvq[q], bvq[q]);
                            fi[q] = fin[q]; ri[q] = rin[q];
                            fetch(fi[q], ri[q], fvq[q], bvq[q]);           // block b + 1 + PF
                            ids(b + 1 + 2 * PF, fin[q], rin[q]);
                        }
                        lds_barrier();
                        buf ^= 1;
                    }
                }
            } else if (!any_rem) {
                // nobody in this wavefront applies g here: exact zeros (which a tracked destination already holds)
                if (!zeros_resident) {
#pragma unroll
                    for (int li = 0; li < 4; li++) {
                        if (c[li] < 0) continue;
#pragma unroll
                        for (int x = 0; x < NX; x++) {
                            if (c0 >= 0) {
                                double* o = a.out + dest[li][x] * a.ld + c0 + lk * D + lr;
#pragma unroll
                                for (int r = 0; r < 4; r++) __builtin_nontemporal_store(0.0, &o[4 * r * D]);
                            } else {
                                const int32_t* cm = a.colmap_gate + (int64_t)g * D * D + lk * D + lr;
#pragma unroll
                                for (int r = 0; r < 4; r++) { const int32_t cc = cm[4 * r * D]; if (cc >= 0) a.out[dest[li][x] * a.ld + cc] = 0.0; }
                            }
                        }
                    }
                }
                continue;
            }
            // ---- remnants: what each of this wavefront's circuits applies of g outside the tile's segment -- its own state ids,
            //      gathered straight from the caches: all four circuits' blocks of one round are in flight together ------------------
            {
                int32_t nrb = 0;
#pragma unroll
                for (int li = 0; li < 4; li++) nrb = (r1[li] - r0[li]) > nrb ? (r1[li] - r0[li]) : nrb;
                if (t.debug & 2) nrb = 0;
                for (int32_t q = 0; q < nrb; q++) {
                    int32_t rf[4], rr[4];
#pragma unroll
                    for (int li = 0; li < 4; li++) {
                        const bool on = r0[li] + q < r1[li];
                        const int64_t o = (int64_t)(on ? r0[li] + q : 0) * 4 + lk;
                        rf[li] = on ? t.rem_f[o] : -1; rr[li] = on ? t.rem_r[o] : 0;
                    }
                    double RF[4]; t_d2 RB0[4], RB1[4];
#pragma unroll
                    for (int li = 0; li < 4; li++) {
                        RF[li] = *(const double*)(fb + (uint64_t)(uint32_t)(rf[li] < 0 ? 0 : rf[li]) * fstride + lr * 8);
                        const t_d2* q2 = (const t_d2*)__builtin_assume_aligned(rb + (uint64_t)(uint32_t)rr[li] * rstride + lr * 32, 16);
                        RB0[li] = q2[0]; RB1[li] = q2[1];
                    }
#pragma unroll
                    for (int li = 0; li < 4; li++) {
                        if (r0[li] + q >= r1[li]) continue;                    // (uniform)
                        const double Fz = rf[li] >= 0 ? RF[li] : 0.0;
                        acc[li][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(RB0[li].x, Fz, acc[li][0], 0, 0, 0);
                        acc[li][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(RB0[li].y, Fz, acc[li][1], 0, 0, 0);
                        acc[li][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(RB1[li].x, Fz, acc[li][2], 0, 0, 0);
                        acc[li][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(RB1[li].y, Fz, acc[li][3], 0, 0, 0);
                    }
                }
            }
            // ---- the 16 x 16 blocks of gate g: rows (circuit, outcome), D-matrix layout of the MFMA ---------------------------------
#pragma unroll
            for (int li = 0; li < 4; li++) {
                if (c[li] < 0 || (t.debug & 1)) continue;
                if (zeros_resident && b1 == b0 && r1[li] == r0[li]) continue;     // this circuit never applies g: zeros already there
#pragma unroll
                for (int x = 0; x < NX; x++) {
                    if (c0 >= 0) {
                        double* o = a.out + dest[li][x] * a.ld + c0 + lk * D + lr;
#pragma unroll
                        for (int r = 0; r < 4; r++) __builtin_nontemporal_store(acc[li][x][r], &o[4 * r * D]);
                    } else {
                        const int32_t* cm = a.colmap_gate + (int64_t)g * D * D + lk * D + lr;
#pragma unroll
                        for (int r = 0; r < 4; r++) { const int32_t cc = cm[4 * r * D]; if (cc >= 0) a.out[dest[li][x] * a.ld + cc] = acc[li][x][r]; }
                    }
                }
            }
        }
    }
}

hipError_t launch_analytic_tiles(const TileArgs& t, int n_cus, hipStream_t stream)
{
    if (t.n_tiles <= 0) return hipSuccess;
    if (t.a.n_effects != 4 || t.a.accumulate || t.a.rho_zero || t.a.eff_zero || t.a.wide || t.a.fwd_stride || t.a.rev_stride) return hipErrorInvalidValue;
    int blocks = n_cus > 0 ? n_cus : 256;                 // one 512-thread workgroup per CU, persistent
    if (blocks > t.n_tiles) blocks = t.n_tiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL(analytic_tile_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, t);
    return hipGetLastError();
}

}  // namespace gst
