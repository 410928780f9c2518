// gst_kernels_normal.hip -- normal equations of the least-squares fit on the device-resident Jacobian:
//   JtJ[c1][c2] = sum_k J[k][c1] * J[k][c2]      (n_cols x n_cols, symmetric)      -- MFMA fp64
//   Jtf[c]      = sum_k J[k][c]  * f[k]                                              -- streaming
// This is kernel K10 / "next" row f1 of SURVEY.md (layout.fill_jtj / fill_jtf, pygsti/layouts/distlayout.py:1220-1359),
// the place where a dense contraction DOES warrant the matrix cores: J is tall and skinny (545,100 x 1,616 for the
// 2Q L<=1024 design), so J^T J is a 1.4 TFLOP fp64 SYRK whose K dimension is the long one -- the shape vendor GEMMs
// handle badly (rocBLAS dsyrk: 299 ms = 4.8 TFLOP/s on this box).
//
// Tiling (gfx950, wave64, v_mfma_f64_16x16x4_f64):
//   * The A operand of the MFMA wants lane l to hold A[i = l&15][k = l>>4], the B operand B[k = l>>4][j = l&15].
//     With A[i][k] = J[k][i0+i] and B[k][j] = J[k][j0+j] BOTH are plain 4-row x 16-column patches of the row-major
//     Jacobian, 128 contiguous bytes per row: no transposition anywhere; a workgroup stages its 8-row panel in LDS.
//   * One wavefront owns a 64 x 64 block of JtJ (4 x 4 MFMA tiles, 64 fp64 accumulators per lane); a 256-thread
//     workgroup owns a 128 x 128 tile over ONE SLAB of rows (split-K: the row dimension is cut into slabs so that
//     tiles x slabs >> 256 CUs); per 4 rows a wavefront reads 4 + 4 patches and issues 16 MFMAs.
//   * Only tiles on or above the diagonal are computed; partial tiles of every slab go to a scratch buffer and a
//     second kernel sums the slabs in a fixed order (deterministic, no atomics) and mirrors the triangle.
#include "gst_internal.hpp"
#include "gst_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

namespace gst {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int JTJ_TILE = 128;     // workgroup tile of JtJ
constexpr int JTJ_WT = 64;        // wavefront tile

// The row panel is staged through LDS: the four wavefronts of a workgroup need the same 8 rows x (128 + 128) columns,
// so the workgroup fetches them ONCE (64 bytes per thread, coalesced 1 KB runs) and every wavefront reads its 4 + 4
// operand patches from LDS.  That halves the L2 -> CU traffic of the first version, in which each wavefront loaded its
// own patches straight into the MFMA layout (33.7 -> 29.2 ms on the 2Q design, 42 -> 49 TFLOP/s).  Two panels in LDS:
// the next one is in flight (global -> registers) while the current one feeds 64 MFMAs per wavefront; one barrier per
// JTJ_PANEL = 16 rows.  (Three workgroups per CU instead of two: 168 VGPRs, spills, 35.5 ms.)
constexpr int JTJ_LDS_STRIDE = 2 * JTJ_TILE + 8;        // doubles per staged row (+8: the 4 rows of a patch start in different banks)

constexpr int JTJ_PANEL = 16;
#if GST_JTJ_TIMING
__device__ unsigned long long jtj_dbg[8];      // development build only: per-phase wavefront cycles of the generic loop
#endif                           // rows staged per barrier

// SPARSITY (round 3): a GST Jacobian is block sparse -- row (circuit, outcome) is exactly zero in the columns of every
// gate the circuit never applies, 30 % of all 16 x 16 blocks on the 2Q design.  `pmask[k / JTJ_PANEL]` has bit t set
// when the 16-row panel starting at row k holds ANY non-zero in the 128 columns of tile t (jtj_panel_mask_kernel, one
// streaming pass over J, fused into the row scaling when there is one); a panel contributes to tile pair (ti, tj) only if
// both bits are set, and the others are skipped outright -- no fetch, no staging, no MFMAs: exact, since what is
// skipped is a product with zeros.  On the 2Q design a third of the (panel, tile pair) products go.
// WEIGHTED (round 5): the product of diag(w) J is taken without forming it -- a staged row is multiplied by its weight on
// the way into LDS (fl(J[r][c] * w[r]), the very number the in-place scaling would have stored, so both routes give the
// same bits), and J stays as the caller filled it.
// FAST (round 5): n_cols a multiple of 8, even ld and a 16-byte aligned matrix -- every staging load is an unconditional 16-byte load (rows
// past the slab and columns past the matrix re-read a valid address and are zeroed on the way into LDS), so the compiler
// can count the loads in flight exactly and the panel TWO steps ahead is fetched into a second register set while the
// panel one step ahead is still landing: each fetch has two MFMA blocks (~7 us) to arrive instead of one.
template <bool WEIGHTED, bool FAST>
__global__ __launch_bounds__(256, 2) void jtj_mfma_lds_kernel(const double* __restrict__ J, int64_t n_rows, int n_cols,
                                                              int64_t ld, int64_t slab_rows, int n_tiles,
                                                              double* __restrict__ part, const uint32_t* __restrict__ pmask,
                                                              const double* __restrict__ w, const uint32_t* __restrict__ pair_order)
{
    extern __shared__ __attribute__((aligned(16))) double panel_mem[];      // [2][JTJ_PANEL * JTJ_LDS_STRIDE]
    double* const panel0 = panel_mem;
    double* const panel1 = panel_mem + JTJ_PANEL * JTJ_LDS_STRIDE;
    const int n_pairs = n_tiles * (n_tiles + 1) / 2;
    const int xcd = blockIdx.x % 8, q = blockIdx.x / 8;
    // (pair_order: the tile pairs with the most live panels first -- with a block-sparse Jacobian a pair's work varies 3 x,
    //  and whatever is dispatched last should be short)
    const int p = pair_order ? (int)pair_order[q % n_pairs] : q % n_pairs;
    const int64_t s = (int64_t)(q / n_pairs) * 8 + xcd;
    int ti = 0, rem = p;
    while (rem >= n_tiles - ti) { rem -= n_tiles - ti; ti++; }
    const int tj = ti + rem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = (wave >> 1) * JTJ_WT, wj = (wave & 1) * JTJ_WT;       // this wavefront's 64 x 64 block inside the tile
    const int lk = lane >> 4, lc = lane & 15;

    d4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};

    const int64_t k_begin = s * slab_rows;
    const int64_t k_end = (k_begin + slab_rows < n_rows) ? k_begin + slab_rows : n_rows;
    // staging role of this thread: rows sr and sr + 8 of the panel, segment (A columns of tile ti / B columns of
    // tile tj), 8 doubles
    const int sr = threadIdx.x >> 5, sseg = (threadIdx.x >> 4) & 1, sch = threadIdx.x & 15;
    const int scol = (sseg ? tj : ti) * JTJ_TILE + sch * 8;
    const int soff = sr * JTJ_LDS_STRIDE + sseg * JTJ_TILE + sch * 8;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    d2_t g[JTJ_PANEL / 8][4];
    double gw[JTJ_PANEL / 8];
    auto fetch = [&](int64_t k) {
#pragma unroll
        for (int u = 0; u < JTJ_PANEL / 8; u++) {
            const int64_t r = k + sr + 8 * u;
            const bool rv = r < k_end;
            if (WEIGHTED) gw[u] = rv ? w[r] : 0.0;
            const double* src = J + (rv ? r : 0) * ld + scol;
            if (rv && scol + 8 <= n_cols && (((uintptr_t)src & 15) == 0)) {
#pragma unroll
                for (int t = 0; t < 4; t++) g[u][t] = *(const d2_t*)(src + 2 * t);
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    g[u][t].x = (rv && scol + 2 * t < n_cols) ? src[2 * t] : 0.0;
                    g[u][t].y = (rv && scol + 2 * t + 1 < n_cols) ? src[2 * t + 1] : 0.0;
                }
            }
        }
    };
    auto stash = [&](double* pan) {
#pragma unroll
        for (int u = 0; u < JTJ_PANEL / 8; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                d2_t x = g[u][t];
                if (WEIGHTED) { x.x *= gw[u]; x.y *= gw[u]; }
                *(d2_t*)(pan + soff + 8 * u * JTJ_LDS_STRIDE + 2 * t) = x;
            }
    };
    // the next panel at or after k that contributes to this tile pair (workgroup-uniform: scalar loads of the masks)
    auto next_live = [&](int64_t k) -> int64_t {
        if (pmask) {
            const uint32_t need = (1u << ti) | (1u << tj);
            while (k < k_end && (pmask[k / JTJ_PANEL] & need) != need) k += JTJ_PANEL;
        }
        return k;
    };
    auto mma_panel = [&](const double* pc) {
#pragma unroll
        for (int h = 0; h < JTJ_PANEL / 4; h++) {                // 4-row steps of the panel
            const double* row = pc + (4 * h + lk) * JTJ_LDS_STRIDE;
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { a[t] = row[wi + 16 * t + lc]; b[t] = row[JTJ_TILE + wj + 16 * t + lc]; }
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 4; y++) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
        }
    };
    if constexpr (FAST) {
        // addresses = uniform base of the panel (scalar registers) + one 32-bit byte offset per staged row; a chunk of 8
        // columns lies wholly inside or wholly outside the matrix (n_cols % 8 == 0 on this path)
        const bool cvalid = scol + 8 <= n_cols;
        const uint32_t cbytes = (uint32_t)(cvalid ? scol : 0) * 8u;
        const uint32_t ldb = (uint32_t)ld * 8u;
        auto fetch2 = [&](int64_t k, d2_t (&G)[JTJ_PANEL / 8][4], double (&GW)[JTJ_PANEL / 8]) {
            const char* const base = (const char*)(J + k * ld);
            const uint32_t last = (uint32_t)(k_end - 1 - k);                      // (k < k_end)
#pragma unroll
            for (int u = 0; u < JTJ_PANEL / 8; u++) {
                const uint32_t rr = (uint32_t)(sr + 8 * u) < last ? (uint32_t)(sr + 8 * u) : last;
                const uint32_t off = rr * ldb + cbytes;
                if (WEIGHTED) GW[u] = w[k + rr];
#pragma unroll
                for (int t = 0; t < 4; t++) G[u][t] = *(const d2_t*)__builtin_assume_aligned(base + off + 16 * t, 16);
            }
        };
        // (under the sibling wavefront's MFMA stream a VALU instruction issues every ~16 cycles -- tools/jtj_phases.py -- so the
        //  zeroing selects are skipped for the panels that need none, a workgroup-uniform decision)
        const bool tiles_full = (ti + 1) * JTJ_TILE <= n_cols && (tj + 1) * JTJ_TILE <= n_cols;
        auto stash2 = [&](double* pan, int64_t k, const d2_t (&G)[JTJ_PANEL / 8][4], const double (&GW)[JTJ_PANEL / 8]) {
            if (tiles_full && k + JTJ_PANEL <= k_end) {
#pragma unroll
                for (int u = 0; u < JTJ_PANEL / 8; u++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        d2_t x = G[u][t];
                        if (WEIGHTED) { x.x *= GW[u]; x.y *= GW[u]; }
                        *(d2_t*)(pan + soff + 8 * u * JTJ_LDS_STRIDE + 2 * t) = x;
                    }
                return;
            }
#pragma unroll
            for (int u = 0; u < JTJ_PANEL / 8; u++) {
                const bool ok = cvalid && (k + sr + 8 * u < k_end);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    d2_t x = G[u][t];
                    if (WEIGHTED) { x.x *= GW[u]; x.y *= GW[u]; }
                    if (!ok) x = (d2_t){0.0, 0.0};
                    *(d2_t*)(pan + soff + 8 * u * JTJ_LDS_STRIDE + 2 * t) = x;
                }
            }
        };
        // The live panels of this tile pair as a bitmap in scalar registers: 64 mask words per vector load (one per lane,
        // the next window's already in flight), a ballot, and "next live panel" is a shift and a count -- the scalar-memory
        // probe per panel of the generic loop costs ~1,400 cycles of every panel here (s_load round trip + SALU at the
        // sibling's mercy).
        const uint32_t need = (1u << ti) | (1u << tj);
        const int64_t p_begin = k_begin / JTJ_PANEL, p_end = (k_end + JTJ_PANEL - 1) / JTJ_PANEL;
        auto load_word = [&](int64_t w0) -> uint32_t {
            const int64_t pi = w0 + lane;
            const int64_t pv = pi < p_end ? pi : p_end - 1;
            const uint32_t m = (pmask && pv >= 0) ? pmask[pv] : need;
            return pi < p_end ? m : 0u;
        };
        int64_t win = p_begin;
        uint64_t bits = __ballot((load_word(win) & need) == need);
        uint32_t nextw = load_word(win + 64);
        auto next_live2 = [&](int64_t kq) -> int64_t {
            int64_t pi = kq / JTJ_PANEL;
            for (;;) {
                if (pi >= p_end) return k_end;
                if (pi >= win + 64) {
                    win += 64;
                    bits = __ballot((nextw & need) == need);
                    nextw = load_word(win + 64);
                    continue;
                }
                const uint64_t rest = bits >> (int)(pi - win);
                if (rest) return (pi + __builtin_ctzll(rest)) * JTJ_PANEL;
                pi = win + 64;
            }
        };
        d2_t g1[JTJ_PANEL / 8][4];
        double gw1[JTJ_PANEL / 8];
        int64_t k = next_live2(k_begin);
        if (k < k_end) {                       // (workgroup-uniform)
            fetch2(k, g, gw);
            stash2(panel0, k, g, gw);
            int64_t kn = next_live2(k + JTJ_PANEL);
            fetch2(kn < k_end ? kn : k_begin, g, gw);                            // one ahead (a dummy re-read when there is none)
            __syncthreads();
#if GST_JTJ_TIMING
            unsigned long long fph[6] = {0, 0, 0, 0, 0, 0};
#endif
            while (k < k_end) {                 // (one exit: the accumulators keep their registers through both halves)
                // panel0 holds k, `g` is landing with kn: fetch two ahead into g1
#if GST_JTJ_TIMING
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
                int64_t kn2 = kn < k_end ? next_live2(kn + JTJ_PANEL) : k_end;
#if GST_JTJ_TIMING
                const unsigned long long t0b = __builtin_amdgcn_s_memtime();
#endif
                fetch2(kn2 < k_end ? kn2 : k_begin, g1, gw1);
#if GST_JTJ_TIMING
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
                mma_panel(panel0);
#if GST_JTJ_TIMING
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
                if (kn < k_end) stash2(panel1, kn, g, gw);
#if GST_JTJ_TIMING
                __builtin_amdgcn_s_waitcnt(0xC07F);
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();
#endif
                __syncthreads();
#if GST_JTJ_TIMING
                const unsigned long long t5 = __builtin_amdgcn_s_memtime();
                fph[0] += t1 - t0b; fph[1] += t2 - t1; fph[2] += t0b - t0; fph[3] += t4 - t2; fph[4] += t5 - t4; fph[5] += 1;
#endif
                k = kn; kn = kn2;
                // panel1 holds k (if any), `g1` is landing with kn: fetch two ahead into g
                kn2 = kn < k_end ? next_live2(kn + JTJ_PANEL) : k_end;
                fetch2(kn2 < k_end ? kn2 : k_begin, g, gw);
                if (k < k_end) mma_panel(panel1);
                if (kn < k_end) stash2(panel0, kn, g1, gw1);
                __syncthreads();
                k = kn; kn = kn2;
            }
#if GST_JTJ_TIMING
            if (lane == 0)
                for (int i = 0; i < 6; i++) atomicAdd(&jtj_dbg[i], fph[i]);
#endif
        }
    } else {
    int64_t k = next_live(k_begin);
    if (k < k_end) {
        fetch(k);
        stash(panel0);
    }
    __syncthreads();
    int cur = 0;
#if GST_JTJ_TIMING
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
#define JT_NOW() __builtin_amdgcn_s_memtime()
#endif
    while (k < k_end) {
#if GST_JTJ_TIMING
        const unsigned long long t0 = JT_NOW();
#endif
        const int64_t kn = next_live(k + JTJ_PANEL);
        const bool more = kn < k_end;
        if (more) fetch(kn);                                     // next live panel: global -> registers, lands during the MFMAs
#if GST_JTJ_TIMING
        const unsigned long long t1 = JT_NOW();
#endif
        mma_panel(cur ? panel1 : panel0);
#if GST_JTJ_TIMING
        const unsigned long long t2 = JT_NOW();
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
        const unsigned long long t3 = JT_NOW();
#endif
        if (more) stash(cur ? panel0 : panel1);
#if GST_JTJ_TIMING
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
        const unsigned long long t4 = JT_NOW();
#endif
        __syncthreads();
#if GST_JTJ_TIMING
        const unsigned long long t5 = JT_NOW();
        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += 1;
#endif
        cur ^= 1;
        k = kn;
    }
#if GST_JTJ_TIMING
    if (lane == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&jtj_dbg[i], ph[i]);
#endif
    }
    const int i0 = ti * JTJ_TILE + wi, j0 = tj * JTJ_TILE + wj;
    double* out = part + (int64_t)s * n_cols * n_cols;
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i0 + 16 * x + lk + 4 * r, col = j0 + 16 * y + lc;
                if (row < n_cols && col < n_cols) out[(int64_t)row * n_cols + col] = acc[x][y][r];
            }
}

// Panel masks (and, with `w`, the row scaling J <- diag(w) J in the same pass): one wavefront per (16-row panel, 128-column
// tile) -- 32 doubles per lane, coalesced 1 KB row segments -- sets bit `tile` of pmask[panel] when anything there is
// non-zero (after scaling: a zero weight annihilates its row).  pmask must be zeroed before the launch.
// write_back = false: the weights only decide what counts as non-zero (J is read, never written).
__global__ __launch_bounds__(256) void jtj_panel_mask_kernel(double* __restrict__ J, int64_t n_rows, int n_cols, int64_t ld,
                                                             const double* __restrict__ w, int n_tiles, uint32_t* __restrict__ pmask,
                                                             bool write_back)
{
    const int64_t n_panels = (n_rows + JTJ_PANEL - 1) / JTJ_PANEL;
    const int lane = threadIdx.x & 63;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < n_panels * n_tiles; item += (int64_t)gridDim.x * 4) {
        const int64_t panel = item / n_tiles;
        const int tile = (int)(item - panel * n_tiles);
        const int c0 = tile * JTJ_TILE + 2 * lane;
        bool any = false;
#pragma unroll 4
        for (int r = 0; r < JTJ_PANEL; r++) {
            const int64_t row = panel * JTJ_PANEL + r;
            if (row >= n_rows) break;
            double* p = J + row * ld + c0;
            const double ws = w ? w[row] : 1.0;
            if (c0 + 1 < n_cols && (((uintptr_t)p & 15) == 0)) {
                typedef double d2_t __attribute__((ext_vector_type(2)));
                d2_t x = *(d2_t*)p;
                if (w) { x.x *= ws; x.y *= ws; if (write_back) *(d2_t*)p = x; }
                any = any || x.x != 0.0 || x.y != 0.0;
            } else {
                for (int q = 0; q < 2; q++)
                    if (c0 + q < n_cols) {
                        double x = p[q];
                        if (w) { x *= ws; if (write_back) p[q] = x; }
                        any = any || x != 0.0;
                    }
            }
        }
        if (__ballot(any) != 0 && lane == 0) atomicOr(&pmask[panel], 1u << tile);
    }
}

// Live panels per tile pair (from the panel masks) and the pairs ranked by them, heaviest first: two tiny launches
// (~10 us) that let jtj_mfma_lds_kernel dispatch every slab's long pairs before its short ones.
constexpr int JTJ_MAX_PAIRS = 32 * 33 / 2;
__global__ __launch_bounds__(256) void jtj_pair_count_kernel(const uint32_t* __restrict__ pmask, int64_t n_panels, int n_tiles,
                                                             uint32_t* __restrict__ counts)
{
    __shared__ uint32_t loc[JTJ_MAX_PAIRS];
    const int n_pairs = n_tiles * (n_tiles + 1) / 2;
    for (int i = threadIdx.x; i < n_pairs; i += 256) loc[i] = 0;
    __syncthreads();
    for (int64_t pi = (int64_t)blockIdx.x * 256 + threadIdx.x; pi < n_panels; pi += (int64_t)gridDim.x * 256) {
        const uint32_t m = pmask[pi];
        for (uint32_t a = m; a; a &= a - 1) {
            const int i = __builtin_ctz(a);
            const int base = i * n_tiles - i * (i - 1) / 2 - i;              // index of pair (i, j) = base + j
            for (uint32_t b = a; b; b &= b - 1) atomicAdd(&loc[base + __builtin_ctz(b)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_pairs; i += 256)
        if (loc[i]) atomicAdd(&counts[i], loc[i]);
}
__global__ __launch_bounds__(1024) void jtj_pair_rank_kernel(const uint32_t* __restrict__ counts, int n_pairs, uint32_t* __restrict__ order)
{
    const int p = threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t c = counts[p];
    int rank = 0;
    for (int q = 0; q < n_pairs; q++) {
        const uint32_t d = counts[q];
        rank += (d > c || (d == c && q < p)) ? 1 : 0;
    }
    order[rank] = (uint32_t)p;
}

// The panel masks of diag(w) J AND (diag(w) J)^T f in ONE streaming pass over J (round 5; J is only read): a wavefront owns a
// 128-column tile over a contiguous range of panels -- lane l the columns 2l, 2l + 1 -- marks the panels of its range that
// hold anything in the tile and carries the tile's 128 partial sums of J_s^T f through the range in registers (rows in
// ascending order, one fused multiply-add per element on the weighted, rounded element: deterministic); the ranges are
// summed by jtf_reduce_kernel.  Saves the second 7 GB read of the 2Q design's Jacobian (jtf_kernel: 1.2 ms).
// pmask must be zeroed before the launch; part is [n_ranges][n_cols].
__global__ __launch_bounds__(256) void jtj_mask_jtf_kernel(const double* __restrict__ J, int64_t n_rows, int n_cols, int64_t ld,
                                                           const double* __restrict__ w, const double* __restrict__ f, int n_tiles,
                                                           uint32_t* __restrict__ pmask, int n_ranges, int64_t panels_per_range,
                                                           double* __restrict__ part)
{
    const int64_t n_panels = (n_rows + JTJ_PANEL - 1) / JTJ_PANEL;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: w, f by scalar loads)
    if (item >= (int64_t)n_ranges * n_tiles) return;
    const int64_t range = item / n_tiles;
    const int tile = (int)(item - range * n_tiles);
    const int c0 = tile * JTJ_TILE + 2 * lane;
    const int64_t p0 = range * panels_per_range, p1 = (p0 + panels_per_range < n_panels) ? p0 + panels_per_range : n_panels;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    double a0 = 0.0, a1 = 0.0;
    const bool pairable = c0 + 1 < n_cols && (ld % 2 == 0) && ((((uintptr_t)J) & 15) == 0);      // 16-byte loads legal on every row
    for (int64_t panel = p0; panel < p1; panel++) {
        bool any = false;
        const int64_t r0 = panel * JTJ_PANEL;
        const int nr = (int)((r0 + JTJ_PANEL <= n_rows) ? JTJ_PANEL : n_rows - r0);
        if (pairable && nr == JTJ_PANEL) {
            d2_t x[JTJ_PANEL];
#pragma unroll
            for (int r = 0; r < JTJ_PANEL; r++) x[r] = *(const d2_t*)(J + (r0 + r) * ld + c0);
#pragma unroll
            for (int r = 0; r < JTJ_PANEL; r++) {
                const double ws = w ? w[r0 + r] : 1.0, fs = f[r0 + r];
                d2_t y = x[r];
                if (w) { y.x *= ws; y.y *= ws; }
                any = any || y.x != 0.0 || y.y != 0.0;
                a0 = __builtin_fma(y.x, fs, a0);
                a1 = __builtin_fma(y.y, fs, a1);
            }
        } else {
            for (int r = 0; r < nr; r++) {
                const double* p = J + (r0 + r) * ld + c0;
                const double ws = w ? w[r0 + r] : 1.0, fs = f[r0 + r];
                double y0 = (c0 < n_cols) ? p[0] : 0.0, y1 = (c0 + 1 < n_cols) ? p[1] : 0.0;
                if (w) { y0 *= ws; y1 *= ws; }
                any = any || y0 != 0.0 || y1 != 0.0;
                a0 = __builtin_fma(y0, fs, a0);
                a1 = __builtin_fma(y1, fs, a1);
            }
        }
        if (__ballot(any) != 0 && lane == 0) atomicOr(&pmask[panel], 1u << tile);
    }
    if (c0 < n_cols) part[range * n_cols + c0] = a0;
    if (c0 + 1 < n_cols) part[range * n_cols + c0 + 1] = a1;
}

// JtJ = sum over slabs of the partial tiles (fixed order: deterministic), mirrored.  One workgroup per 32 x 32 block (br <= bc)
// of the result: every entry of such a block was computed directly (its 128-tile row index is <= its column index; the
// diagonal tiles are computed in full, both halves from the same products in the same order, so they are symmetric to the
// bit), the sums are read and written in coalesced 256-byte rows, and the mirrored block goes out through an LDS
// transpose.  (Rounds 2-4 read the lower triangle through transposed addresses: 0.95 ms for 2.6 M sums; this form 0.3.)
__global__ __launch_bounds__(256) void jtj_reduce_kernel(const double* __restrict__ part, int n_slabs, int n_cols, double* __restrict__ C)
{
    __shared__ double tile[32][33];
    const int nb = (n_cols + 31) / 32;
    int br = 0, rem = (int)blockIdx.x;
    while (rem >= nb - br) { rem -= nb - br; br++; }
    const int bc = br + rem;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t total = (int64_t)n_cols * n_cols;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int r = br * 32 + j, c = bc * 32 + tx;
        double sum = 0.0;
        if (r < n_cols && c < n_cols) {
            const double* src = part + (int64_t)r * n_cols + c;
            // eight loads in flight, added in slab order (a small matrix has four workgroups here, and forty dependent loads
            // were 40 us of a 1Q LM step): the same sum, bit for bit
            int sl = 0;
            for (; sl + 8 <= n_slabs; sl += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = src[(int64_t)(sl + u) * total];
#pragma unroll
                for (int u = 0; u < 8; u++) sum += v[u];
            }
            for (; sl < n_slabs; sl++) sum += src[(int64_t)sl * total];
            C[(int64_t)r * n_cols + c] = sum;
        }
        tile[j][tx] = sum;
    }
    __syncthreads();
    if (br != bc) {
#pragma unroll
        for (int j = ty; j < 32; j += 8) {
            const int r = bc * 32 + j, c = br * 32 + tx;
            if (r < n_cols && c < n_cols) C[(int64_t)r * n_cols + c] = tile[tx][j];
        }
    }
}

// w != nullptr: (diag(w) J)^T f with the weighted element rounded first, as the in-place scaling stores it
__global__ void jtf_kernel(const double* __restrict__ J, const double* __restrict__ f, int64_t n_rows, int n_cols,
                           int64_t ld, int64_t slab_rows, double* __restrict__ part /* [n_slabs][n_cols] */,
                           const double* __restrict__ w)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int64_t s = blockIdx.y;
    const int64_t k0 = s * slab_rows, k1 = (k0 + slab_rows < n_rows) ? k0 + slab_rows : n_rows;
    if (c >= n_cols) return;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int64_t k = k0;
    if (w) {
        for (; k + 3 < k1; k += 4) {
            acc0 = __builtin_fma(J[k * ld + c] * w[k], f[k], acc0);
            acc1 = __builtin_fma(J[(k + 1) * ld + c] * w[k + 1], f[k + 1], acc1);
            acc2 = __builtin_fma(J[(k + 2) * ld + c] * w[k + 2], f[k + 2], acc2);
            acc3 = __builtin_fma(J[(k + 3) * ld + c] * w[k + 3], f[k + 3], acc3);
        }
        for (; k < k1; k++) acc0 = __builtin_fma(J[k * ld + c] * w[k], f[k], acc0);
        part[s * n_cols + c] = (acc0 + acc1) + (acc2 + acc3);
        return;
    }
    for (; k + 3 < k1; k += 4) {
        acc0 = __builtin_fma(J[k * ld + c], f[k], acc0);
        acc1 = __builtin_fma(J[(k + 1) * ld + c], f[k + 1], acc1);
        acc2 = __builtin_fma(J[(k + 2) * ld + c], f[k + 2], acc2);
        acc3 = __builtin_fma(J[(k + 3) * ld + c], f[k + 3], acc3);
    }
    for (; k < k1; k++) acc0 = __builtin_fma(J[k * ld + c], f[k], acc0);
    part[s * n_cols + c] = (acc0 + acc1) + (acc2 + acc3);
}
__global__ void jtf_reduce_kernel(const double* __restrict__ part, int n_slabs, int n_cols, double* __restrict__ y)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    double sum = 0.0;
    int s = 0;
    for (; s + 8 <= n_slabs; s += 8) {              // (loads in flight, added in slab order)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = part[(int64_t)(s + u) * n_cols + c];
#pragma unroll
        for (int u = 0; u < 8; u++) sum += v[u];
    }
    for (; s < n_slabs; s++) sum += part[(int64_t)s * n_cols + c];
    y[c] = sum;
}

// Element-wise objective maps (row f1): probabilities -> least-squares vector and the row scale of its Jacobian.
//   kind 0  chi^2            RawChi2Function.lsvec/dlsvec, _weights/_dweights      objectivefns.py:1814-1885, 2040-2084
//   kind 1  Poisson dlogl    RawPoissonPicDeltaLogLFunction ('minp', harsh radius) objectivefns.py:2944-3160, 3185-3195
//   rowscale = (|lsvec| < 1e-100 ? 0 : 0.5 / lsvec) * dterms      TimeIndependentMDCObjectiveFunction.dlsvec :4633-4665
// Same operation order as the numpy expressions (no contraction), so chi^2 is bit-identical and dlogl differs only by
// the last bit of log().  Per-block partial sums of `terms` go to `part` (summed on the host in block order).
__global__ void objective_rows_kernel(int kind, double* __restrict__ probs, const double* __restrict__ counts,
                                      const double* __restrict__ totals, int64_t n, double min_p, double radius,
                                      double clip_lo, double clip_hi, double* __restrict__ lsvec,
                                      double* __restrict__ rowscale, double* __restrict__ terms_out,
                                      double* __restrict__ part)
{
    __shared__ double red[256];
    double local = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        double p = probs[k];
        if (clip_lo < clip_hi) {                                   // _clip_probs, objectivefns.py:4766-4774
            p = p < clip_lo ? clip_lo : (p > clip_hi ? clip_hi : p);
            probs[k] = p;
        }
        const double c = counts[k], N = totals[k];
        const double f = c / N;
        double terms, ls, dterms;
        if (kind == 0) {
            const double cp = p > min_p ? p : min_p;
            const double w = sqrt(N / cp);
            ls = (p - f) * w;
            const double dw = (p < min_p) ? 0.0 : -0.5 * w / cp;
            const double dls = w + (p - f) * dw;
            terms = ls * ls;
            dterms = 2 * ls * dls;
        } else {
            const double fnz = (c == 0) ? 1.0 : f;
            const double pos = (p < min_p) ? min_p : p;
            const double c0 = N - c / min_p;
            const double c1 = 0.5 * c / (min_p * min_p);
            double t = c * (log(fnz) - 1.0) - c * log(pos) + N * pos;
            t = t > 0.0 ? t : 0.0;
            const double dpm = p - min_p;
            if (p < min_p) t = t + c0 * dpm + c1 * (dpm * dpm);
            const double a = radius;
            const double zf = N * (p >= a ? p : (-1.0 / (3 * a * a)) * (p * p * p) + (p * p) / a + a / 3.0);
            terms = (c == 0) ? zf : t;
            ls = sqrt(terms);
            const double d = (p < min_p) ? c0 + 2 * c1 * dpm : N - c / pos;
            const double dzf = N * (p >= a ? 1.0 : (-1.0 / (a * a)) * (p * p) + 2 * p / a);
            dterms = (c == 0) ? dzf : d;
        }
        const double p5 = (fabs(ls) < 1e-100) ? 0.0 : 0.5 / ls;
        lsvec[k] = ls;
        rowscale[k] = p5 * dterms;
        if (terms_out) terms_out[k] = terms;
        local += terms;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

hipError_t launch_objective_rows(int kind, double* probs, const double* counts, const double* totals, int64_t n, double min_p,
                                 double radius, double clip_lo, double clip_hi, double* lsvec, double* rowscale,
                                 double* terms_out, double* part, int n_blocks, hipStream_t s)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(objective_rows_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, kind, probs, counts, totals, n, min_p,
                       radius, clip_lo, clip_hi, lsvec, rowscale, terms_out, part);
    return hipGetLastError();
}

// Second-derivative coefficients of the objective (the reference's raw_objfn.dterms / .hterms, objectivefns.py:631-669,
// 755-794 with RawChi2Function.hlsvec :1886-1921 and _hweights :2086-2108; RawPoissonPicDeltaLogLFunction.dterms
// :3098-3150 and .hterms :3152-3183) for the Hessian of the objective, element by element.
__global__ void objective_coeffs_kernel(int kind, const double* __restrict__ probs, const double* __restrict__ counts,
                                        const double* __restrict__ totals, int64_t n, double min_p, double radius,
                                        double* __restrict__ dterms_out, double* __restrict__ hterms_out)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const double p = probs[k], c = counts[k], N = totals[k];
        const double f = c / N;
        double dterms, hterms;
        if (kind == 0) {
            const double cp = p > min_p ? p : min_p;
            const double w = sqrt(N / cp);
            const bool clipped = p < min_p;
            const double dw = clipped ? 0.0 : -0.5 * w / cp;
            const double hw = clipped ? 0.0 : 0.75 * w / (cp * cp);
            const double ls = (p - f) * w;
            const double dls = w + (p - f) * dw;
            const double hls = 2 * dw + (p - f) * hw;
            dterms = 2 * ls * dls;
            hterms = 2 * (dls * dls + ls * hls);
        } else {
            const double pos = (p < min_p) ? min_p : p;
            const double c0 = N - c / min_p;
            const double c1 = 0.5 * c / (min_p * min_p);
            const double a = radius;
            const double d = (p < min_p) ? c0 + 2 * c1 * (p - min_p) : N - c / pos;
            const double dzf = N * (p >= a ? 1.0 : (-1.0 / (a * a)) * (p * p) + 2 * p / a);
            dterms = (c == 0) ? dzf : d;
            const double h = (p < min_p) ? 2 * c1 : c / (pos * pos);
            const double hzf = (p >= a) ? 0.0 : N * ((-2.0 / (a * a)) * p + 2.0 / a);
            hterms = (c == 0) ? hzf : h;
        }
        dterms_out[k] = dterms;
        hterms_out[k] = hterms;
    }
}

hipError_t launch_objective_coeffs(int kind, const double* probs, const double* counts, const double* totals, int64_t n, double min_p,
                                   double radius, double* dterms, double* hterms, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(objective_coeffs_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, s, kind, probs,
                       counts, totals, n, min_p, radius, dterms, hterms);
    return hipGetLastError();
}

// Hessian block of the objective from device-resident pieces (TimeIndependentMDCObjectiveFunction._hessian_from_block,
// objectivefns.py:4914-4968):  out[i][j] = sum_e  hterms[e] * d1[e][i] * d2[e][j] + dterms[e] * H[e][i][j].
// Thread = column j, 8 rows i per thread, one slab of elements per blockIdx.y; H is read exactly once, coalesced.
constexpr int HB_IC = 8;
__global__ __launch_bounds__(256) void hessian_block_kernel(const double* __restrict__ H, const double* __restrict__ d1,
                                                           const double* __restrict__ d2, const double* __restrict__ dco,
                                                           const double* __restrict__ hco, int64_t nE, int n1, int n2,
                                                           int64_t slab, double* __restrict__ part)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.z * HB_IC;
    const int64_t e0 = (int64_t)blockIdx.y * slab, e1 = (e0 + slab < nE) ? e0 + slab : nE;
    if (j >= n2) return;
    double acc[HB_IC];
#pragma unroll
    for (int t = 0; t < HB_IC; t++) acc[t] = 0.0;
    for (int64_t e = e0; e < e1; e++) {
        const double hd2 = hco[e] * d2[e * n2 + j];
        const double dc = dco[e];
        const double* Hrow = H + (e * n1 + i0) * (int64_t)n2 + j;
        const double* d1row = d1 + e * n1 + i0;
#pragma unroll
        for (int t = 0; t < HB_IC; t++) {
            if (i0 + t < n1) acc[t] = __builtin_fma(dc, Hrow[(int64_t)t * n2], __builtin_fma(hd2, d1row[t], acc[t]));
        }
    }
#pragma unroll
    for (int t = 0; t < HB_IC; t++)
        if (i0 + t < n1) part[((int64_t)blockIdx.y * n1 + i0 + t) * n2 + j] = acc[t];
}
__global__ void hessian_block_reduce_kernel(const double* __restrict__ part, int n_slabs, int64_t total, double* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < n_slabs; k++) s += part[(int64_t)k * total + i];
        out[i] = s;
    }
}
int hessian_block_slabs(int64_t nE, int n1, int n2)
{
    const int64_t tiles = (int64_t)((n2 + 255) / 256) * ((n1 + HB_IC - 1) / HB_IC);
    int64_t slabs = (4096 + tiles - 1) / tiles;
    if (slabs > (nE + 63) / 64) slabs = (nE + 63) / 64;
    if (slabs < 1) slabs = 1;
    if (slabs > 1024) slabs = 1024;
    return (int)slabs;
}
hipError_t launch_hessian_block(const double* H, const double* d1, const double* d2, const double* dco, const double* hco, int64_t nE,
                                int n1, int n2, double* part, int n_slabs, double* out, hipStream_t s)
{
    if (nE <= 0 || n1 <= 0 || n2 <= 0) return hipSuccess;
    const int64_t slab = (nE + n_slabs - 1) / n_slabs;
    (void)hipGetLastError();
    hipLaunchKernelGGL(hessian_block_kernel, dim3((unsigned)((n2 + 255) / 256), (unsigned)n_slabs, (unsigned)((n1 + HB_IC - 1) / HB_IC)),
                       dim3(256), 0, s, H, d1, d2, dco, hco, nE, n1, n2, slab, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t total = (int64_t)n1 * n2;
    hipLaunchKernelGGL(hessian_block_reduce_kernel, dim3((unsigned)std::min<int64_t>(1024, (total + 255) / 256)), dim3(256), 0, s, part,
                       n_slabs, total, out);
    return hipGetLastError();
}

// Chain rule of general parameterisations (gst_set_derivs): C[:, colmap[j]] (+)= A[:, a_col0 : a_col0 + K] . B[:, j]
// with A the element Jacobian (row-major, ldA), B = d(element)/d(parameter) of one object (row-major [K][n]).
// Workgroup = 64 rows x 64 columns (wavefront w: rows 16 w ... 16 w + 15, 4 column tiles of v_mfma_f64_16x16x4_f64); A and B
// go through LDS in chunks of 32 k, loaded with coalesced rows and read back in the MFMA operand layouts from padded,
// conflict-free arrays.  (Round 3.  The first form took every operand straight from memory: 5 loads of 8 bytes per lane and
// k-step, the A one touching 16 cache lines -- timing ablations: 13.3 of the kernel's 16 ms were the operand loads, the
// matrix pipes 28 % busy.  Through LDS a workgroup moves 32 KB per chunk instead of 80.)
// Rows / columns beyond the matrix are CLAMPED to valid ones (their products are never stored); k beyond K reads zeros.
constexpr int CR_BK = 32;              // k per chunk
constexpr int CR_AS = CR_BK + 2;       // row stride of the A chunk in LDS: 16 rows x {kk, kk + 1} hit 64 distinct banks
constexpr int CR_BS = 64 + 16;         // row stride of the B chunk: consecutive k rows start 32 banks apart
__global__ __launch_bounds__(256) void chain_rule_gemm_kernel(const double* __restrict__ A, int64_t ldA, int64_t a_col0, int K,
                                                              const double* __restrict__ B, int n,
                                                              const int32_t* __restrict__ colmap, double* __restrict__ C,
                                                              int64_t ldC, int64_t n_rows, const int overwrite)
{
    __shared__ double As[64 * CR_AS];
    __shared__ double Bs[CR_BK * CR_BS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_n = (n + 63) / 64;
    const int64_t R0 = ((int64_t)blockIdx.x / tiles_n) * 64;
    const int c0 = (int)((int64_t)blockIdx.x % tiles_n) * 64;
    const int i = lane & 15, kk = lane >> 4;
    d4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
    // loaders: A chunk 64 rows x 32 k -- thread (row tid >> 5 (+ 8 e), k tid & 31); B chunk 32 k x 64 columns -- thread
    // (k tid >> 6 (+ 4 e), column tid & 63): every load instruction reads whole 256- / 512-byte row pieces
    const int a_r = tid >> 5, a_k = tid & 31;
    const int b_k = tid >> 6, b_c = tid & 63;
    const int b_col = (c0 + b_c < n) ? c0 + b_c : n - 1;
    const double* a_src[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int64_t row = R0 + a_r + 8 * e;
        a_src[e] = A + (row < n_rows ? row : n_rows - 1) * ldA + a_col0 + a_k;
    }
    double av[8], bv[8];
    auto fetch = [&](const int k0) {             // the chunk's 16 global loads per thread, into registers
#pragma unroll
        for (int e = 0; e < 8; e++) av[e] = (k0 + a_k < K) ? a_src[e][k0] : 0.0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = k0 + b_k + 4 * e;
            bv[e] = (k < K) ? B[(int64_t)k * n + b_col] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += CR_BK) {
        __syncthreads();                         // (the previous chunk has been consumed)
#pragma unroll
        for (int e = 0; e < 8; e++) As[(a_r + 8 * e) * CR_AS + a_k] = av[e];
#pragma unroll
        for (int e = 0; e < 8; e++) Bs[(b_k + 4 * e) * CR_BS + b_c] = bv[e];
        __syncthreads();
        if (k0 + CR_BK < K) fetch(k0 + CR_BK);   // the next chunk is in flight while this one is multiplied
        const double* as = As + (16 * wv + i) * CR_AS + kk;
        const double* bs = Bs + kk * CR_BS + i;
#pragma unroll
        for (int s = 0; s < CR_BK / 4; s++) {
            const double a = as[4 * s];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bs[4 * s * CR_BS + 16 * t], acc[t], 0, 0, 0);
        }
    }
    const int64_t r0 = R0 + 16 * wv;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int col = c0 + 16 * t + i;
        const int32_t cc = col < n ? colmap[col] : -1;
        if (cc < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t row = r0 + kk + 4 * r;
            if (row < n_rows) {                     // (the first object that owns these columns stores, later ones add)
                double* c = C + row * ldC + cc;
                *c = overwrite ? acc[t][r] : *c + acc[t][r];
            }
        }
    }
}

// The same product for the big objects (a gate's K = D^2 element columns x its parameters; round 4).  Workgroup = 128 rows
// x 80 columns (the 240 parameters of a two-qubit CPTPLND gate are three such tiles exactly), wavefront w = rows 32 w ..
// 32 w + 31 as 2 x 5 accumulator tiles: 10 MFMAs per 7 LDS operand reads instead of 4 per 5.  k in chunks of 16 (28 KB of
// LDS: five workgroups per CU).  Every wavefront stages ITS OWN 32 rows of A, so it knows whether the chunk is all zero
// -- the element Jacobian's block of a gate none of its 8 circuits applies -- and skips the chunk's MFMAs (a fifth of
// them on a GST design); the B chunk is shared.  Column tiles are the fastest grid index: the workgroups that re-read an
// A tile run side by side and find it in L2.
constexpr int C2_BK = 16;
constexpr int C2_AS = C2_BK + 2;       // 36 dwords: 16 rows land on 16 distinct multiples of 4 banks, {kk, kk + 1} interleave
constexpr int C2_BN = 80;
__global__ __launch_bounds__(256, 3) void chain_rule_gemm2_kernel(const double* __restrict__ A, int64_t ldA, int64_t a_col0, int K,
                                                               const double* __restrict__ B, int n,
                                                               const int32_t* __restrict__ colmap, double* __restrict__ C,
                                                               int64_t ldC, int64_t n_rows, const int overwrite,
                                                               const uint64_t* __restrict__ rowmask, const int mask_bit)
{
    __shared__ double As[4 * 32 * C2_AS];
    __shared__ double Bs[C2_BK * C2_BN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (n + C2_BN - 1) / C2_BN;
    const int64_t R0 = ((int64_t)blockIdx.x / tiles_n) * 128;
    const int c0 = (int)((int64_t)blockIdx.x % tiles_n) * C2_BN;
    const int i = lane & 15, kk = lane >> 4;
    // rowmask (optional, from the plan's structure): bit g of word r says whether any of rows 32 r .. 32 r + 31 can be
    // non-zero in gate g's element columns (some circuit of those rows applies g).  A wavefront whose 32 rows cannot
    // does not even load them; a workgroup none of whose wavefronts can stores its zeros and leaves.
    bool live = true;
    if (rowmask) {
        const uint64_t* m = rowmask + (R0 >> 5);
        live = ((m[wv] >> mask_bit) & 1ull) != 0;
        if ((((m[0] | m[1] | m[2] | m[3]) >> mask_bit) & 1ull) == 0) {
            if (!overwrite) return;
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const int col = c0 + 16 * t + i;
                const int32_t cc = col < n ? colmap[col] : -1;
                if (cc < 0) continue;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int64_t row = R0 + 32 * wv + kk + 4 * r;
                    if (row < n_rows) C[row * ldC + cc] = 0.0;
                }
            }
            return;
        }
    }
    d4_t acc[2][5];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int t = 0; t < 5; t++) acc[m][t] = (d4_t){0.0, 0.0, 0.0, 0.0};
    // loaders: A -- this wavefront's rows 32 wv + (lane >> 4) + 4 e, k = lane & 15 (four 128-byte row pieces per load);
    // B -- element tid + 256 e of the 16 x 80 chunk
    const double* a_src[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int64_t row = R0 + 32 * wv + kk + 4 * e;
        a_src[e] = A + (row < n_rows ? row : n_rows - 1) * ldA + a_col0 + i;
    }
    int b_k[5], b_off[5];
#pragma unroll
    for (int e = 0; e < 5; e++) {
        const int idx = tid + 256 * e;
        b_k[e] = idx / C2_BN;
        const int bc = idx - b_k[e] * C2_BN;
        b_off[e] = (c0 + bc < n) ? c0 + bc : n - 1;
    }
    double av[8], bv[5];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int e = 0; e < 8; e++) av[e] = (live && k0 + i < K) ? a_src[e][k0] : 0.0;
#pragma unroll
        for (int e = 0; e < 5; e++) bv[e] = (k0 + b_k[e] < K) ? B[(int64_t)(k0 + b_k[e]) * n + b_off[e]] : 0.0;
    };
    double* const my_as = As + wv * 32 * C2_AS;
    bool any = false;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += C2_BK) {
        __syncthreads();                         // (the previous chunk has been consumed)
        bool nzl = false;
#pragma unroll
        for (int e = 0; e < 8; e++) { my_as[(kk + 4 * e) * C2_AS + i] = av[e]; nzl = nzl || av[e] != 0.0; }
#pragma unroll
        for (int e = 0; e < 5; e++) Bs[tid + 256 * e] = bv[e];
        const bool nz = __ballot(nzl) != 0;      // wave-uniform: do this wavefront's 32 rows hold anything in this chunk?
        __syncthreads();
        if (k0 + C2_BK < K) fetch(k0 + C2_BK);   // the next chunk is in flight while this one is multiplied
        if (nz) {
            any = true;
            const double* as = my_as + i * C2_AS + kk;
            const double* bs = Bs + kk * C2_BN + i;
#pragma unroll 1
            for (int s = 0; s < C2_BK / 4; s++) {
                const double a0 = as[4 * s], a1 = as[16 * C2_AS + 4 * s];
#pragma unroll
                for (int t = 0; t < 5; t++) {
                    const double b = bs[4 * s * C2_BN + 16 * t];
                    acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][t], 0, 0, 0);
                }
            }
        }
    }
    if (!any && !overwrite) return;              // (adding zeros)
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const int col = c0 + 16 * t + i;
        const int32_t cc = col < n ? colmap[col] : -1;
        if (cc < 0) continue;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t row = R0 + 32 * wv + 16 * m + kk + 4 * r;
                if (row < n_rows) {
                    double* c = C + row * ldC + cc;
                    *c = overwrite ? acc[m][t][r] : *c + acc[m][t][r];
                }
            }
    }
}

hipError_t launch_chain_rule_gemm(const double* A, int64_t ldA, int64_t a_col0, int K, const double* B, int n, const int32_t* colmap,
                                  double* C, int64_t ldC, int64_t n_rows, hipStream_t s, bool overwrite, const uint64_t* rowmask, int mask_bit)
{
    if (n <= 0 || n_rows <= 0 || K <= 0) return hipSuccess;
    if (K >= 64 && n >= 40 && n_rows >= 128) {
        const int64_t blocks2 = ((n_rows + 127) / 128) * ((n + C2_BN - 1) / C2_BN);
        if (blocks2 > 0x7fffffffLL) return hipErrorInvalidValue;
        (void)hipGetLastError();
        hipLaunchKernelGGL(chain_rule_gemm2_kernel, dim3((unsigned)blocks2), dim3(256), 0, s, A, ldA, a_col0, K, B, n, colmap, C, ldC, n_rows, overwrite ? 1 : 0,
                           rowmask, mask_bit);
        return hipGetLastError();
    }
    const int64_t blocks = ((n_rows + 63) / 64) * ((n + 63) / 64);       // workgroup = 64 rows x 64 columns
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(chain_rule_gemm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, A, ldA, a_col0, K, B, n, colmap, C, ldC, n_rows, overwrite ? 1 : 0);
    return hipGetLastError();
}

// development switch: GST_JTJ_FAST=0 keeps the one-panel-ahead form for every shape
static bool jtj_fast_path()
{
    static const int v = [] { const char* e = std::getenv("GST_JTJ_FAST"); return e ? std::atoi(e) : 1; }();
    return v != 0;
}

int jtj_num_slabs(int64_t n_rows, int n_cols, int n_cus)
{
    const int n_tiles = (n_cols + JTJ_TILE - 1) / JTJ_TILE;
    const int n_pairs = n_tiles * (n_tiles + 1) / 2;
    const int64_t max_by_rows = std::max<int64_t>(1, (n_rows + 63) / 64);          // at least 64 rows per slab
    // One group of slabs per XCD (a multiple of 8; empty slabs are harmless), ~4096 workgroups or more -- and among the
    // candidates the one whose workgroups fill the chip's 2 * n_cus slots in whole rounds: 48 slabs x 91 tile pairs of the
    // 2Q design are 8.53 rounds (the ninth half empty), 56 slabs are 9.95.
    const int slots = 2 * std::max(n_cus, 1);
    int best = 8;
    double best_eff = -1.0;
    for (int sl = 8; sl <= 64; sl += 8) {
        if (sl > 8 && sl - 7 > max_by_rows) break;
        const double rounds = (double)sl * n_pairs / slots;
        const double eff = rounds / std::ceil(rounds);
        const bool enough = (int64_t)sl * n_pairs >= 3584;
        const double score = (enough ? 1.0 : 0.0) + eff * ((int64_t)sl * n_pairs >= slots ? 1.0 : 0.5);
        if (score > best_eff + 1e-9) { best_eff = score; best = sl; }
        if (enough && (int64_t)sl * n_pairs >= 6144) break;                       // more slabs only add partial sums
    }
    return best;
}

hipError_t launch_jtj_panel_masks(double* J, int64_t n_rows, int n_cols, int64_t ld, const double* w, uint32_t* pmask, hipStream_t s,
                                  bool write_back)
{
    const int n_tiles = (n_cols + JTJ_TILE - 1) / JTJ_TILE;
    if (n_tiles > 32 || n_rows <= 0) return hipErrorInvalidValue;
    const int64_t n_panels = (n_rows + JTJ_PANEL - 1) / JTJ_PANEL;
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(pmask, 0, (size_t)(n_panels + JTJ_MAX_PAIRS) * 4, s);      // masks + pair counts
    if (e != hipSuccess) return e;
    const int64_t items = n_panels * n_tiles;
    hipLaunchKernelGGL(jtj_panel_mask_kernel, dim3((unsigned)std::min<int64_t>((items + 3) / 4, 65536)), dim3(256), 0, s, J, n_rows, n_cols, ld, w,
                       n_tiles, pmask, write_back);
    return hipGetLastError();
}
#if GST_JTJ_TIMING
extern "C" __attribute__((visibility("default"))) int gst_debug_jtj_phases(unsigned long long* out, int reset)
{
    if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(jtj_dbg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(jtj_dbg), 8 * sizeof(unsigned long long));
}
#endif
int jtj_mask_jtf_ranges(int64_t n_rows, int n_cols)
{
    const int n_tiles = (n_cols + JTJ_TILE - 1) / JTJ_TILE;
    const int64_t n_panels = (n_rows + JTJ_PANEL - 1) / JTJ_PANEL;
    int64_t ranges = (4096 + n_tiles - 1) / n_tiles;                      // ~4096 wavefronts
    if (ranges > n_panels / 8) ranges = n_panels / 8;                     // at least 8 panels each
    if (ranges > 512) ranges = 512;
    return (int)(ranges < 1 ? 1 : ranges);
}
hipError_t launch_jtj_mask_jtf(const double* J, int64_t n_rows, int n_cols, int64_t ld, const double* w, const double* f, uint32_t* pmask,
                               double* part, int n_ranges, double* y, hipStream_t s)
{
    const int n_tiles = (n_cols + JTJ_TILE - 1) / JTJ_TILE;
    if (n_tiles > 32 || n_rows <= 0 || n_ranges < 1) return hipErrorInvalidValue;
    const int64_t n_panels = (n_rows + JTJ_PANEL - 1) / JTJ_PANEL;
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(pmask, 0, (size_t)(n_panels + JTJ_MAX_PAIRS) * 4, s);      // masks + pair counts
    if (e != hipSuccess) return e;
    const int64_t ppr = (n_panels + n_ranges - 1) / n_ranges;
    const int64_t items = (int64_t)n_ranges * n_tiles;
    hipLaunchKernelGGL(jtj_mask_jtf_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, J, n_rows, n_cols, ld, w, f, n_tiles, pmask,
                       n_ranges, ppr, part);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(jtf_reduce_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, s, part, n_ranges, n_cols, y);
    return hipGetLastError();
}
int jtj_mask_tiles(int n_cols) { return (n_cols + JTJ_TILE - 1) / JTJ_TILE; }
int64_t jtj_mask_panels(int64_t n_rows) { return (n_rows + JTJ_PANEL - 1) / JTJ_PANEL; }
int64_t jtj_mask_words(int64_t n_rows) { return jtj_mask_panels(n_rows) + 2 * JTJ_MAX_PAIRS; }

hipError_t launch_jtj(const double* J, int64_t n_rows, int n_cols, int64_t ld, double* part, int n_slabs, double* C,
                      hipStream_t s, const uint32_t* pmask, const double* w)
{
    const int n_tiles = (n_cols + JTJ_TILE - 1) / JTJ_TILE;
    const int n_pairs = n_tiles * (n_tiles + 1) / 2;
    int64_t slab_rows = (n_rows + n_slabs - 1) / n_slabs;
    slab_rows = (slab_rows + JTJ_PANEL - 1) / JTJ_PANEL * JTJ_PANEL;        // the k loop advances one panel per iteration
    (void)hipGetLastError();
    const size_t lds_bytes = (size_t)2 * JTJ_PANEL * JTJ_LDS_STRIDE * sizeof(double);
    // the branch-free, two-panels-ahead form needs 16-byte loads to be legal everywhere
    const bool fast = jtj_fast_path() && (n_cols % 8 == 0) && (ld % 2 == 0) && (((uintptr_t)J & 15) == 0) && n_rows > 0 && ld < (1 << 24);
    typedef void (*jtj_kernel_t)(const double*, int64_t, int, int64_t, int64_t, int, double*, const uint32_t*, const double*, const uint32_t*);
    const jtj_kernel_t kern = fast ? (w ? jtj_mfma_lds_kernel<true, true> : jtj_mfma_lds_kernel<false, true>)
                                   : (w ? jtj_mfma_lds_kernel<true, false> : jtj_mfma_lds_kernel<false, false>);
    if (lds_bytes > 64 * 1024) {          // per device (a process may drive several GPUs): set on every launch, it is cheap
        hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (ea != hipSuccess) return ea;
    }
    const uint32_t* order = nullptr;
    if (pmask && n_pairs <= JTJ_MAX_PAIRS && n_pairs <= 1024) {
        // (the words behind the masks: pair counts, zeroed with the masks, then the ranked pairs -- jtj_mask_words())
        const int64_t n_panels = (n_rows + JTJ_PANEL - 1) / JTJ_PANEL;
        uint32_t* counts = const_cast<uint32_t*>(pmask) + n_panels;
        uint32_t* ord = counts + JTJ_MAX_PAIRS;
        hipLaunchKernelGGL(jtj_pair_count_kernel, dim3(64), dim3(256), 0, s, pmask, n_panels, n_tiles, counts);
        hipLaunchKernelGGL(jtj_pair_rank_kernel, dim3(1), dim3(1024), 0, s, counts, n_pairs, ord);
        hipError_t eo = hipGetLastError();
        if (eo != hipSuccess) return eo;
        order = ord;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_pairs * n_slabs)), dim3(256), lds_bytes, s, J, n_rows, n_cols, ld, slab_rows, n_tiles, part, pmask, w, order);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    {
        const int nb = (n_cols + 31) / 32;
        hipLaunchKernelGGL(jtj_reduce_kernel, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, s, part, n_slabs, n_cols, C);
    }
    return hipGetLastError();
}

// Slabs of J^T f: 256 rows each for the large problems (at most 256 slabs), 64 rows each where that still leaves few -- a 1Q
// design's 2,240 rows were 9 slabs of 60 threads walking 250 rows one after the other (23 us of a 160 us LM step).
int jtf_num_slabs(int64_t n_rows)
{
    const int64_t by256 = (n_rows + 255) / 256;
    if (by256 >= 64) return (int)std::min<int64_t>(256, by256);
    return (int)std::max<int64_t>(1, std::min<int64_t>(64, (n_rows + 63) / 64));
}

hipError_t launch_jtf(const double* J, const double* f, int64_t n_rows, int n_cols, int64_t ld, double* part, int n_slabs,
                      double* y, hipStream_t s, const double* w)
{
    const int64_t slab_rows = (n_rows + n_slabs - 1) / n_slabs;
    (void)hipGetLastError();
    hipLaunchKernelGGL(jtf_kernel, dim3((unsigned)((n_cols + 255) / 256), (unsigned)n_slabs), dim3(256), 0, s, J, f, n_rows,
                       n_cols, ld, slab_rows, part, w);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(jtf_reduce_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, s, part, n_slabs, n_cols, y);
    return hipGetLastError();
}

// ---- chain rule of a Hessian block (linear parameterisations) --------------------------------------------------
__global__ __launch_bounds__(256) void hessian_chain_rule_kernel(const double* __restrict__ H, int64_t nE, int m1, int m2,
                                                                 const int32_t* __restrict__ ptr1, const int32_t* __restrict__ row1,
                                                                 const double* __restrict__ w1, const int32_t* __restrict__ dest1, int n1,
                                                                 const int32_t* __restrict__ ptr2, const int32_t* __restrict__ row2,
                                                                 const double* __restrict__ w2, const int32_t* __restrict__ dest2, int n2,
                                                                 double* __restrict__ out, int64_t ld1, int64_t ld2)
{
    const int64_t total = nE * (int64_t)n1 * n2;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int j = (int)(t % n2);
        const int64_t r = t / n2;
        const int i = (int)(r % n1);
        const int64_t e = r / n1;
        const double* He = H + e * (int64_t)m1 * m2;
        double acc = 0.0;
        for (int32_t x = ptr1[i]; x < ptr1[i + 1]; x++) {
            const double* Ha = He + (int64_t)row1[x] * m2;
            double inner = 0.0;
            for (int32_t y = ptr2[j]; y < ptr2[j + 1]; y++) inner += w2[y] * Ha[row2[y]];
            acc += w1[x] * inner;
        }
        out[(e * ld1 + dest1[i]) * ld2 + dest2[j]] = acc;
    }
}

hipError_t launch_hessian_chain_rule(const double* H, int64_t nE, int m1, int m2, const int32_t* ptr1, const int32_t* row1,
                                     const double* w1, const int32_t* dest1, int n1, const int32_t* ptr2, const int32_t* row2,
                                     const double* w2, const int32_t* dest2, int n2, double* out, int64_t ld1, int64_t ld2,
                                     hipStream_t s)
{
    const int64_t total = nE * (int64_t)n1 * n2;
    if (total <= 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>((total + 255) / 256, 65536);
    (void)hipGetLastError();
    hipLaunchKernelGGL(hessian_chain_rule_kernel, dim3((unsigned)blocks), dim3(256), 0, s, H, nE, m1, m2, ptr1, row1, w1, dest1, n1,
                       ptr2, row2, w2, dest2, n2, out, ld1, ld2);
    return hipGetLastError();
}

// Sum of per-rank copies in rank order (the deterministic all-reduce of the IPC transport, gst_comm.cpp).
__global__ void sum_slots_kernel(const double* __restrict__ slots, int n_slots, int64_t stride, int64_t n, double* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double acc = slots[i];
        for (int r = 1; r < n_slots; r++) acc += slots[(int64_t)r * stride + i];
        out[i] = acc;
    }
}

__global__ __launch_bounds__(256) void check_finite_kernel(const double* __restrict__ w, int64_t n, uint32_t* __restrict__ flag)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = w[i];
        bad = bad || !(x - x == 0.0);                  // inf - inf and nan - nan are nan
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) *flag = 0u;
}

hipError_t launch_check_finite(const double* w, int64_t n, uint32_t* flag, hipStream_t s)
{
    if (n <= 0 || !flag) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, s, w, n, flag);
    return hipGetLastError();
}

hipError_t launch_sum_slots(const double* slots, int n_slots, int64_t stride, int64_t n, double* out, hipStream_t s)
{
    if (n <= 0 || n_slots <= 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>((n + 255) / 256, 4096);
    (void)hipGetLastError();
    hipLaunchKernelGGL(sum_slots_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slots, n_slots, stride, n, out);
    return hipGetLastError();
}

}  // namespace gst
