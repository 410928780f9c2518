// gst_kernels_lindblad.hip -- dense model members from Lindblad parameters, on the device.
//
// What the reference does on the host for every finite-difference step of a CPTPLND / GLND / H+S model
// (mapfill_dprobs_atom, mapforwardsim_calc_densitymx.pyx:349-381: `model.set_parameter_value(i, orig + eps)` ->
// LindbladErrorgen.from_vector -> coefficient blocks (lindbladcoefficients.py:164-470) -> error generator
// L = Re(sum_k c_k S_k) (lindbladerrorgen.py:658-742) -> ExpErrorgenOp._update_rep = scipy expm
// (experrorgenop.py:114-123) -> the composed member), this kernel does for ALL requested parameters at once: one
// workgroup per model set builds the one member its parameter belongs to and copies the rest of the base model,
// straight into the [gates_t | rhos | effects] layout the walk kernels' whole-model mode reads.  No host to_dense per
// column, no PCIe traffic per column.
//
// Numerics: not a bit-exact path (the reference's own expm is a Pade approximant; its einsum order is numpy's).  The
// exponential is a scaled Taylor series -- ||L / 2^s||_1 <= 1/4, 18 terms (remainder < 1e-24), then s squarings -- in
// fp64 with FMA; dense members agree with the reference's to ~1e-16 (tests/test_gpu_lindblad.py: <= 1e-14).
#include "gst_kernels.hpp"

#include "../../include/gstfwd.h"

namespace gst {

namespace {

constexpr int LB_TAYLOR = 18;

// c(theta) of member m into c_re / c_im (lindbladcoefficients.py:164-470); returns the number of coefficients.  `th`: the
// member's parameters (LDS).  All threads of the workgroup call it.
__device__ __forceinline__ int lb_coefficients(const LbArgs& a, const int m, const double* th, double* c_re, double* c_im, const int t, const int nt)
{
    int K = 0, poff = 0;
    for (int b = 0; b < a.n_blocks[m]; b++) {
        const int bt = a.blk_type[m * LB_MAX_BLOCKS + b], md = a.blk_mode[m * LB_MAX_BLOCKS + b], n = a.blk_n[m * LB_MAX_BLOCKS + b];
        if (bt != 2) {                        // 'ham' / 'other_diagonal': one coefficient per basis element
            for (int k = t; k < n; k += nt) {
                const double x = th[poff + k];
                c_re[K + k] = (md == 1) ? x * x : x;      // 'cholesky' (diagonal): v^2; 'elements': v
                c_im[K + k] = 0.0;
            }
            K += n; poff += n;
        } else {                              // 'other': n x n, parameters read as an n x n matrix p (row-major)
            const double* p = th + poff;
            for (int k = t; k < n * n; k += nt) {
                const int r = k / n, s = k % n;
                double re = 0.0, im = 0.0;
                if (md == 1) {                // 'cholesky': c = C C^dag, C_ii = p_ii, C_ij = p_ij + i p_ji (i > j), lower triangular
                    const int lim = r < s ? r : s;
                    for (int l = 0; l <= lim; l++) {
                        const double ar = p[r * n + l], ai = (l < r) ? p[l * n + r] : 0.0;   // C_rl
                        const double br = p[s * n + l], bi = (l < s) ? p[l * n + s] : 0.0;   // C_sl
                        re += ar * br + ai * bi;                                              // C_rl conj(C_sl)
                        im += ai * br - ar * bi;
                    }
                } else {                      // 'elements': Hermitian, c_rs = p_rs + i p_sr (r > s)
                    if (r == s) re = p[r * n + r];
                    else if (r > s) { re = p[r * n + s]; im = p[s * n + r]; }
                    else { re = p[s * n + r]; im = -p[r * n + s]; }
                }
                c_re[K + k] = re; c_im[K + k] = im;
            }
            K += n * n; poff += n * n;
        }
    }
    return K;
}

// dc/dtheta_q of member m into d_re / d_im (`block_data_jacobian`, lindbladcoefficients.py:178, 235, 348-380, 453-470).
__device__ __forceinline__ void lb_coefficient_derivs(const LbArgs& a, const int m, const int q, const double* th, double* d_re, double* d_im,
                                                      const int t, const int nt)
{
    int K = 0, poff = 0;
    for (int b = 0; b < a.n_blocks[m]; b++) {
        const int bt = a.blk_type[m * LB_MAX_BLOCKS + b], md = a.blk_mode[m * LB_MAX_BLOCKS + b], n = a.blk_n[m * LB_MAX_BLOCKS + b];
        if (bt != 2) {
            for (int k = t; k < n; k += nt) {
                d_re[K + k] = (q == poff + k) ? ((md == 1) ? 2.0 * th[poff + k] : 1.0) : 0.0;
                d_im[K + k] = 0.0;
            }
            K += n; poff += n;
        } else {
            const double* p = th + poff;
            const bool mine = q >= poff && q < poff + n * n;
            const int pa = mine ? (q - poff) / n : 0, pb = mine ? (q - poff) % n : 0;
            // dC/dp_ab has ONE entry: (r0, c0) = (max, min) of (a, b), value z = 1 (a >= b: a real part) or i (a < b: an imaginary part)
            const int r0 = pa >= pb ? pa : pb, c0 = pa >= pb ? pb : pa;
            const double zr = pa >= pb ? 1.0 : 0.0, zi = pa >= pb ? 0.0 : 1.0;
            for (int k = t; k < n * n; k += nt) {
                const int r = k / n, s = k % n;
                double re = 0.0, im = 0.0;
                if (mine && md == 1) {        // d(C C^dag) = dC C^dag + C dC^dag
                    if (r == r0 && c0 <= s) { // z conj(C_{s,c0})
                        const double cr = p[s * n + c0], ci = (c0 < s) ? p[c0 * n + s] : 0.0;
                        re += zr * cr + zi * ci; im += zi * cr - zr * ci;
                    }
                    if (s == r0 && c0 <= r) { // C_{r,c0} conj(z)
                        const double cr = p[r * n + c0], ci = (c0 < r) ? p[c0 * n + r] : 0.0;
                        re += cr * zr + ci * zi; im += ci * zr - cr * zi;
                    }
                } else if (mine) {            // Hermitian 'elements'
                    if (r0 == c0) { if (r == r0 && s == r0) re = 1.0; }
                    else if (r == r0 && s == c0) { re = zr; im = zi; }
                    else if (r == c0 && s == r0) { re = zr; im = -zi; }
                }
                d_re[K + k] = re; d_im[K + k] = im;
            }
            K += n * n; poff += n * n;
        }
    }
}

// D = 16: 256 threads, thread (i, j) owns element [i][j]; D = 4: 64 threads, the first 16 active.
template <int D>
__global__ __launch_bounds__(D * D < 64 ? 64 : D * D) void lindblad_build_kernel(const LbArgs a)
{
    constexpr int DD = D * D;
    constexpr int KMAX = lb_max_coeffs(D);        // parameters / coefficients of one member (the host checks)
    __shared__ double th[KMAX];
    __shared__ double c_re[KMAX], c_im[KMAX];
    __shared__ double A[DD], T0[DD], T1[DD], S[DD];
    __shared__ double colsum[D];
    __shared__ int s_shift;
    const int t = threadIdx.x;
    const bool act = t < DD;
    const int i = t / D, j = t % D;
    // ---- which member, which parameter ---------------------------------------------------------------------------------
    const int64_t set = blockIdx.x;
    int m = -1, q = -1;                           // member, local index of the stepped parameter (-1: none)
    if (a.set_param) {
        const int64_t gp = a.set_param[set];
        for (int mm = 0; mm < a.n_members; mm++)
            if (gp >= a.param0[mm] && gp < a.param0[mm] + a.n_params[mm]) { m = mm; q = (int)(gp - a.param0[mm]); }
    } else {
        m = (int)set;                             // base build: workgroup = member
    }
    double* const dst = a.sets + (a.set_param ? set * a.set_stride : 0);
    const bool member_only = a.set_param && a.member_only;
    const size_t ng = (size_t)a.n_gates * DD, nr = (size_t)a.n_rhos * D;
    if (m < 0) {                                  // a parameter no member owns: the set is the base model
        if (!member_only)
            for (int64_t k = t; k < a.set_stride; k += blockDim.x) dst[k] = a.base_set[k];
        return;
    }
    const int nP = a.n_params[m];
    for (int k = t; k < nP; k += blockDim.x) {
        const double x = a.theta[a.param0[m] + k];
        th[k] = (k == q) ? x + a.eps : x;         // theta_p + eps, as set_parameter_value does
    }
    __syncthreads();
    const int K = lb_coefficients(a, m, th, c_re, c_im, t, (int)blockDim.x);
    __syncthreads();
    // ---- L = sum_k Re(c_k) term_re[k] + Im(c_k) term_im[k] (lindbladerrorgen.py:699-703) --------------------------------------
    if (act) {
        const double* tr = a.term_re + (size_t)a.term_off[m] * DD + t;
        const double* ti = a.term_im + (size_t)a.term_off[m] * DD + t;
        double acc = 0.0;
        for (int k = 0; k < K; k++) acc += c_re[k] * tr[(size_t)k * DD] + c_im[k] * ti[(size_t)k * DD];
        A[t] = acc;
    }
    __syncthreads();
    // ---- E = exp(L): scale to ||.||_1 <= 1/4, Taylor, square back -------------------------------------------------------------------
    if (t < D) {
        double cs = 0.0;
        for (int r = 0; r < D; r++) cs += fabs(A[r * D + t]);
        colsum[t] = cs;
    }
    __syncthreads();
    if (t == 0) {
        double nrm = 0.0;
        for (int r = 0; r < D; r++) nrm = fmax(nrm, colsum[r]);
        int s = 0;
        while (nrm > 0.25 && s < 60) { nrm *= 0.5; s++; }
        s_shift = s;
    }
    __syncthreads();
    const int shift = s_shift;
    if (act) {
        const double x = ldexp(A[t], -shift);
        A[t] = x; T0[t] = x;
        S[t] = (i == j ? 1.0 : 0.0) + x;
    }
    __syncthreads();
    double* Tc = T0; double* Tn = T1;
    for (int k = 2; k <= LB_TAYLOR; k++) {
        if (act) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) acc += Tc[i * D + l] * A[l * D + j];
            acc /= (double)k;
            Tn[t] = acc;
            S[t] += acc;
        }
        __syncthreads();
        double* tmp = Tc; Tc = Tn; Tn = tmp;
    }
    for (int sq = 0; sq < shift; sq++) {
        double acc = 0.0;
        if (act) {
#pragma unroll
            for (int l = 0; l < D; l++) acc += S[i * D + l] * S[l * D + j];
        }
        __syncthreads();
        if (act) S[t] = acc;
        __syncthreads();
    }
    // ---- compose with the static factor and write the member; the rest of the set is the base model ------------------------------
    const int kind = a.kind[m], obj = a.obj[m];
    const double* stat = a.statics + a.static_off[m];
    size_t lo = 0, hi = 0;                        // [lo, hi): the doubles of the set this member owns
    if (kind == GST_KIND_GATE) {                  // G = E . U, stored transposed: gates_t[obj][j][i] = G[i][j]
        lo = member_only ? 0 : (size_t)obj * DD; hi = lo + DD;
        if (act) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) acc += S[i * D + l] * stat[l * D + j];
            dst[lo + (size_t)j * D + i] = acc;
            if (a.gates_rowmajor) a.gates_rowmajor[(size_t)obj * DD + t] = acc;
        }
    } else if (kind == GST_KIND_RHO) {            // rho = E . rho0
        lo = member_only ? 0 : ng + (size_t)obj * D; hi = lo + D;
        if (t < D) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) acc += S[t * D + l] * stat[l];
            dst[lo + t] = acc;
        }
    } else {                                      // POVM: effect_e = E^T e_e, i.e. row e of (static . E)
        const int ne = a.n_eff[m];
        lo = member_only ? 0 : ng + nr + (size_t)obj * D; hi = lo + (size_t)ne * D;
        for (int k = t; k < ne * D; k += blockDim.x) {
            const int e = k / D, c = k % D;
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) acc += stat[e * D + l] * S[l * D + c];
            dst[lo + k] = acc;
        }
    }
    if (a.set_param && !member_only)
        for (int64_t k = t; k < a.set_stride; k += blockDim.x)
            if ((size_t)k < lo || (size_t)k >= hi) dst[k] = a.base_set[k];
}

// d(dense member)/d(parameter) for EVERY parameter of every member -- what the reference gets from
// `ExpErrorgenOp.deriv_wrt_params()` (experrorgenop.py:213-260) composed with the static factor, the input of the
// analytic mode's chain rule (matrixforwardsim.py:126-190 `_doperation`).  Workgroup = (member, parameter q):
//   dL = sum_k Re(dc_k/dq) term_re[k] + Im(dc_k/dq) term_im[k],
//   d exp(L)[dL] by differentiating the scaled Taylor series term by term (M_k = A^k / k!, dM_k = (dM_{k-1} A + M_{k-1} dA) / k)
//   and the squarings (d(S S) = dS S + S dS),
// then the composition with the static factor; column q of the member's [n_elem][n_params] matrices (one per effect for a
// POVM) at deriv_out + deriv_off[m].
template <int D>
__global__ __launch_bounds__(D * D < 64 ? 64 : D * D) void lindblad_deriv_kernel(const LbArgs a)
{
    constexpr int DD = D * D;
    constexpr int KMAX = lb_max_coeffs(D);
    __shared__ double th[KMAX];
    __shared__ double c_re[KMAX], c_im[KMAX];
    __shared__ double A[DD], dA[DD], T0[DD], T1[DD], U0[DD], U1[DD], S[DD], dS[DD];
    __shared__ double colsum[D];
    __shared__ int s_shift;
    const int t = threadIdx.x;
    const bool act = t < DD;
    const int i = t / D, j = t % D;
    const int64_t gp = a.set_param[blockIdx.x];
    int m = -1, q = -1;
    for (int mm = 0; mm < a.n_members; mm++)
        if (gp >= a.param0[mm] && gp < a.param0[mm] + a.n_params[mm]) { m = mm; q = (int)(gp - a.param0[mm]); }
    if (m < 0) return;
    const int nP = a.n_params[m];
    for (int k = t; k < nP; k += blockDim.x) th[k] = a.theta[a.param0[m] + k];
    __syncthreads();
    const int K = lb_coefficients(a, m, th, c_re, c_im, t, (int)blockDim.x);
    __syncthreads();
    const double* tr = a.term_re + (size_t)a.term_off[m] * DD + t;
    const double* ti = a.term_im + (size_t)a.term_off[m] * DD + t;
    if (act) {
        double acc = 0.0;
        for (int k = 0; k < K; k++) acc += c_re[k] * tr[(size_t)k * DD] + c_im[k] * ti[(size_t)k * DD];
        A[t] = acc;
    }
    __syncthreads();
    lb_coefficient_derivs(a, m, q, th, c_re, c_im, t, (int)blockDim.x);      // (c_re / c_im now hold dc/dq)
    __syncthreads();
    if (act) {
        double acc = 0.0;
        for (int k = 0; k < K; k++) {
            const double dr = c_re[k], di = c_im[k];                         // (wave-uniform: most are zero)
            if (dr != 0.0 || di != 0.0) acc += dr * tr[(size_t)k * DD] + di * ti[(size_t)k * DD];
        }
        dA[t] = acc;
    }
    __syncthreads();
    if (t < D) {
        double cs = 0.0;
        for (int r = 0; r < D; r++) cs += fabs(A[r * D + t]);
        colsum[t] = cs;
    }
    __syncthreads();
    if (t == 0) {
        double nrm = 0.0;
        for (int r = 0; r < D; r++) nrm = fmax(nrm, colsum[r]);
        int s = 0;
        while (nrm > 0.25 && s < 60) { nrm *= 0.5; s++; }
        s_shift = s;
    }
    __syncthreads();
    const int shift = s_shift;
    if (act) {
        const double x = ldexp(A[t], -shift), dx = ldexp(dA[t], -shift);
        A[t] = x; dA[t] = dx; T0[t] = x; U0[t] = dx;
        S[t] = (i == j ? 1.0 : 0.0) + x; dS[t] = dx;
    }
    __syncthreads();
    double *Tc = T0, *Tn = T1, *Uc = U0, *Un = U1;
    for (int k = 2; k <= LB_TAYLOR; k++) {
        if (act) {
            double acc = 0.0, dacc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) {
                acc += Tc[i * D + l] * A[l * D + j];
                dacc += Uc[i * D + l] * A[l * D + j] + Tc[i * D + l] * dA[l * D + j];
            }
            acc /= (double)k; dacc /= (double)k;
            Tn[t] = acc; Un[t] = dacc;
            S[t] += acc; dS[t] += dacc;
        }
        __syncthreads();
        double* tmp = Tc; Tc = Tn; Tn = tmp;
        tmp = Uc; Uc = Un; Un = tmp;
    }
    for (int sq = 0; sq < shift; sq++) {
        double acc = 0.0, dacc = 0.0;
        if (act) {
#pragma unroll
            for (int l = 0; l < D; l++) {
                acc += S[i * D + l] * S[l * D + j];
                dacc += dS[i * D + l] * S[l * D + j] + S[i * D + l] * dS[l * D + j];
            }
        }
        __syncthreads();
        if (act) { S[t] = acc; dS[t] = dacc; }
        __syncthreads();
    }
    // ---- compose with the static factor; write column q ---------------------------------------------------------------------
    const int kind = a.kind[m];
    const double* stat = a.statics + a.static_off[m];
    double* const out = a.deriv_out + a.deriv_off[m];
    if (kind == GST_KIND_GATE) {                  // d(E U)[i][j] = sum_l dE[i][l] U[l][j]
        if (act) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) acc += dS[i * D + l] * stat[l * D + j];
            out[(size_t)t * nP + q] = acc;
        }
    } else if (kind == GST_KIND_RHO) {
        if (t < D) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) acc += dS[t * D + l] * stat[l];
            out[(size_t)t * nP + q] = acc;
        }
    } else {                                      // effect e, component c: sum_l e_e[l] dE[l][c]; one [D][n_params] matrix per effect
        const int ne = a.n_eff[m];
        for (int k = t; k < ne * D; k += blockDim.x) {
            const int e = k / D, c = k % D;
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < D; l++) acc += stat[e * D + l] * dS[l * D + c];
            out[((size_t)e * D + c) * nP + q] = acc;
        }
    }
}


// ---- D = 64 (three qubits): the same members with 64 x 64 products on the matrix cores -------------------------------------
// A three-qubit Lindblad member has 63 + 63^2 = 4,032 coefficients; its term table is 4,032 x 64 x 64 doubles (132 MB, twice:
// real and imaginary parts), shared by every member over the same basis.  Three kernels:
//   lindblad64_assemble_kernel  L = sum_k Re(c_k) term_re[k] + Im(c_k) term_im[k]: grid (member, 16 slices of 256 elements),
//                               every thread one element, the term table streamed once per member (coalesced over elements)
//   lindblad64_exp_kernel       one workgroup per member: scale, Taylor series, squarings -- every 64 x 64 product is 4
//                               wavefronts x 64 v_mfma_f64_16x16x4_f64 with both operands in LDS -- and the composition with
//                               the static factor; the scaled generator, EVERY Taylor term M_k = A^k / k! and EVERY square
//                               S_j are kept in a per-member workspace, because ...
//   lindblad64_deriv_kernel     ... one workgroup per (member, parameter) differentiates exactly that computation:
//                               dM_k = (dM_{k-1} A + M_{k-1} dA) / k, d(S_j S_j) = dS_j S_j + S_j dS_j, without recomputing
//                               what all 4,032 parameters of a member share.  dA = dL/dtheta_q needs only the coefficients
//                               that depend on theta_q (one for 'ham' / diagonal blocks, two for Hermitian 'other' blocks,
//                               <= 2 n for Cholesky ones): a sparse list instead of a 4,032-term sum.
// Numerics as at D <= 16: fp64, FMA, 18 Taylor terms at ||A||_1 <= 1/4; the MFMA re-associates the inner sums.
constexpr int L64_D = 64, L64_DD = 64 * 64;
constexpr int L64_XS = 68;                       // LDS row stride (doubles): A-operand reads of 16 rows are conflict-free
constexpr int L64_MAT = 64 * L64_XS;
constexpr int L64_MAX_SHIFT = 24;
constexpr int L64_LIST = 256;                    // nonzero dc/dtheta_q entries (<= 2 * 63)
// workspace of one member (doubles): A (scaled) | M_1 .. M_18 | S_0 .. S_MAX_SHIFT | shift
constexpr size_t L64_WS_A = 0, L64_WS_M = L64_DD, L64_WS_S = (size_t)(1 + LB_TAYLOR) * L64_DD,
                 L64_WS_SHIFT = (size_t)(1 + LB_TAYLOR + L64_MAX_SHIFT + 1) * L64_DD, L64_WS_STRIDE = L64_WS_SHIFT + 8;
typedef double l64_d4 __attribute__((ext_vector_type(4)));

// acc[c] (16 x 16 tile: rows 16 w .., columns 16 c ..) += X[16 w .. +15][:] . Y[:][16 c .. +15], X and Y in LDS (stride L64_XS)
__device__ __forceinline__ void l64_mm(const double* X, const double* Y, l64_d4 (&acc)[4], const int w, const int lr, const int lk)
{
    const double* xa = X + (16 * w + lr) * L64_XS + lk;
    const double* yb = Y + lk * L64_XS + lr;
#pragma unroll 4
    for (int s = 0; s < 16; s++) {
        const double av = xa[4 * s];
        const double* yr = yb + 4 * s * L64_XS;
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, yr[16 * c], acc[c], 0, 0, 0);
    }
}
// D-matrix layout of the tiles: lane (lr, lk), register r of tile c <-> element (16 w + lk + 4 r, 16 c + lr)
#define L64_FOR_TILE(c_, r_) _Pragma("unroll") for (int c_ = 0; c_ < 4; c_++) _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++)
#define L64_ROW(r_) (16 * w + lk + 4 * (r_))
#define L64_COL(c_) (16 * (c_) + lr)

__device__ __forceinline__ void l64_load_global(double* dst_lds, const double* src, const int t)
{
    for (int e = t; e < L64_DD; e += 256) dst_lds[(e >> 6) * L64_XS + (e & 63)] = src[e];
}

__global__ __launch_bounds__(256) void lindblad64_assemble_kernel(const LbArgs a, double* ws)
{
    extern __shared__ double l64_lds[];          // th | c_re | c_im, each n_params[m] (<= 4,032) doubles
    const int m = blockIdx.x, t = threadIdx.x;
    const int nP = a.n_params[m];
    double* th = l64_lds; double* c_re = th + nP; double* c_im = c_re + nP;
    for (int k = t; k < nP; k += 256) th[k] = a.theta[a.param0[m] + k];
    __syncthreads();
    const int K = lb_coefficients(a, m, th, c_re, c_im, t, 256);
    __syncthreads();
    const int e = blockIdx.y * 256 + t;
    const double* tr = a.term_re + (size_t)a.term_off[m] * L64_DD + e;
    const double* ti = a.term_im + (size_t)a.term_off[m] * L64_DD + e;
    double acc0 = 0.0, acc1 = 0.0;
    int k = 0;
    for (; k + 1 < K; k += 2) {
        acc0 += c_re[k] * tr[(size_t)k * L64_DD] + c_im[k] * ti[(size_t)k * L64_DD];
        acc1 += c_re[k + 1] * tr[(size_t)(k + 1) * L64_DD] + c_im[k + 1] * ti[(size_t)(k + 1) * L64_DD];
    }
    if (k < K) acc0 += c_re[k] * tr[(size_t)k * L64_DD] + c_im[k] * ti[(size_t)k * L64_DD];
    ws[(size_t)m * L64_WS_STRIDE + L64_WS_A + e] = acc0 + acc1;
}

__global__ __launch_bounds__(256) void lindblad64_exp_kernel(const LbArgs a, double* ws_all)
{
    extern __shared__ double l64_lds[];          // A | T0 | T1 (L64_MAT each) | colsum[64]
    double* A = l64_lds; double* Tc = A + L64_MAT; double* Tn = Tc + L64_MAT; double* colsum = Tn + L64_MAT;
    __shared__ int s_shift;
    const int m = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lr = lane & 15, lk = lane >> 4;
    double* ws = ws_all + (size_t)m * L64_WS_STRIDE;
    l64_load_global(A, ws + L64_WS_A, t);
    __syncthreads();
    if (t < 64) {
        double cs = 0.0;
        for (int r = 0; r < 64; r++) cs += fabs(A[r * L64_XS + t]);
        colsum[t] = cs;
    }
    __syncthreads();
    if (t == 0) {
        double nrm = 0.0;
        for (int r = 0; r < 64; r++) nrm = fmax(nrm, colsum[r]);
        int s = 0;
        while (nrm > 0.25 && s < L64_MAX_SHIFT) { nrm *= 0.5; s++; }
        s_shift = s;
        ws[L64_WS_SHIFT] = (double)s;
    }
    __syncthreads();
    const int shift = s_shift;
    for (int e = t; e < L64_DD; e += 256) {
        const double x = ldexp(A[(e >> 6) * L64_XS + (e & 63)], -shift);
        A[(e >> 6) * L64_XS + (e & 63)] = x; Tc[(e >> 6) * L64_XS + (e & 63)] = x;
        ws[L64_WS_A + e] = x; ws[L64_WS_M + e] = x;                                 // M_1 = A
    }
    __syncthreads();
    l64_d4 S[4];
    L64_FOR_TILE(c, r) S[c][r] = (L64_ROW(r) == L64_COL(c) ? 1.0 : 0.0) + A[L64_ROW(r) * L64_XS + L64_COL(c)];
    for (int k = 2; k <= LB_TAYLOR; k++) {
        l64_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        l64_mm(Tc, A, acc, w, lr, lk);
        const double inv = 1.0 / (double)k;
        double* Mk = ws + L64_WS_M + (size_t)(k - 1) * L64_DD;
        L64_FOR_TILE(c, r) {
            const double v = acc[c][r] * inv;
            Tn[L64_ROW(r) * L64_XS + L64_COL(c)] = v; Mk[L64_ROW(r) * 64 + L64_COL(c)] = v; S[c][r] += v;
        }
        __syncthreads();
        double* tmp = Tc; Tc = Tn; Tn = tmp;
    }
    // S_0 = the Taylor sum, S_{j+1} = S_j^2; the last one is exp(L)
    L64_FOR_TILE(c, r) { Tc[L64_ROW(r) * L64_XS + L64_COL(c)] = S[c][r]; ws[L64_WS_S + L64_ROW(r) * 64 + L64_COL(c)] = S[c][r]; }
    __syncthreads();
    for (int sq = 0; sq < shift; sq++) {
        l64_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        l64_mm(Tc, Tc, acc, w, lr, lk);
        double* Sj = ws + L64_WS_S + (size_t)(sq + 1) * L64_DD;
        L64_FOR_TILE(c, r) { Tn[L64_ROW(r) * L64_XS + L64_COL(c)] = acc[c][r]; Sj[L64_ROW(r) * 64 + L64_COL(c)] = acc[c][r]; }
        __syncthreads();
        double* tmp = Tc; Tc = Tn; Tn = tmp;
    }
    // ---- compose with the static factor (Tc = exp(L)) and write the member into the base set ------------------------------
    const int kind = a.kind[m], obj = a.obj[m];
    const double* stat = a.statics + a.static_off[m];
    const size_t ng = (size_t)a.n_gates * L64_DD, nr = (size_t)a.n_rhos * 64;
    double* const dst = a.sets;
    if (kind == GST_KIND_GATE) {                  // G = E . U, stored transposed: gates_t[obj][j][i] = G[i][j]
        l64_load_global(A, stat, t);
        __syncthreads();
        l64_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        l64_mm(Tc, A, acc, w, lr, lk);
        L64_FOR_TILE(c, r) {
            dst[(size_t)obj * L64_DD + (size_t)L64_COL(c) * 64 + L64_ROW(r)] = acc[c][r];
            if (a.gates_rowmajor) a.gates_rowmajor[(size_t)obj * L64_DD + (size_t)L64_ROW(r) * 64 + L64_COL(c)] = acc[c][r];
        }
    } else if (kind == GST_KIND_RHO) {            // rho = E . rho0
        if (t < 64) {
            double acc = 0.0;
            for (int l = 0; l < 64; l++) acc += Tc[t * L64_XS + l] * stat[l];
            dst[ng + (size_t)obj * 64 + t] = acc;
        }
    } else {                                      // effect_e = E^T e_e: row e of (static . E)
        const int ne = a.n_eff[m];
        for (int k = t; k < ne * 64; k += 256) {
            const int e = k >> 6, c = k & 63;
            double acc = 0.0;
            for (int l = 0; l < 64; l++) acc += stat[e * 64 + l] * Tc[l * L64_XS + c];
            dst[ng + nr + (size_t)obj * 64 + k] = acc;
        }
    }
}

// The coefficients that depend on parameter q of member m, as a list (index into the member's coefficients, d Re, d Im).
// `th`: the member's parameters (LDS).  Returns the count; every thread must call it (the list is written cooperatively).
__device__ __forceinline__ int l64_dcoef_list(const LbArgs& a, const int m, const int q, const double* th, int* lk_, double* lre, double* lim, const int t)
{
    int K = 0, poff = 0;
    for (int b = 0; b < a.n_blocks[m]; b++) {
        const int bt = a.blk_type[m * LB_MAX_BLOCKS + b], md = a.blk_mode[m * LB_MAX_BLOCKS + b], n = a.blk_n[m * LB_MAX_BLOCKS + b];
        const int np = (bt == 2) ? n * n : n;
        if (q >= poff && q < poff + np) {
            const int ql = q - poff;
            if (bt != 2) {
                if (t == 0) { lk_[0] = K + ql; lre[0] = (md == 1) ? 2.0 * th[poff + ql] : 1.0; lim[0] = 0.0; }
                return 1;
            }
            const double* p = th + poff;
            const int pa = ql / n, pb = ql % n;
            const int r0 = pa >= pb ? pa : pb, c0 = pa >= pb ? pb : pa;
            const double zr = pa >= pb ? 1.0 : 0.0, zi = pa >= pb ? 0.0 : 1.0;
            if (md == 0) {                        // Hermitian 'elements'
                if (r0 == c0) { if (t == 0) { lk_[0] = K + r0 * n + r0; lre[0] = 1.0; lim[0] = 0.0; } return 1; }
                if (t == 0) { lk_[0] = K + r0 * n + c0; lre[0] = zr; lim[0] = zi; lk_[1] = K + c0 * n + r0; lre[1] = zr; lim[1] = -zi; }
                return 2;
            }
            // 'cholesky': d(C C^dag) = dC C^dag + C dC^dag with dC = z at (r0, c0):
            //   row r0:    d c[r0][s] += z conj(C[s][c0])     (s >= c0)
            //   column r0: d c[r][r0] += C[r][c0] conj(z)     (r >= c0)
            const int cnt = n - c0;
            for (int u = t; u < 2 * cnt; u += 256) {
                const int s = c0 + (u < cnt ? u : u - cnt);
                const double cr = p[s * n + c0], ci = (c0 < s) ? p[c0 * n + s] : 0.0;          // C[s][c0]
                if (u < cnt) { lk_[u] = K + r0 * n + s; lre[u] = zr * cr + zi * ci; lim[u] = zi * cr - zr * ci; }
                else { lk_[u] = K + s * n + r0; lre[u] = cr * zr + ci * zi; lim[u] = ci * zr - cr * zi; }
            }
            return 2 * cnt;
        }
        K += np; poff += np;
    }
    return 0;
}

__global__ __launch_bounds__(256) void lindblad64_deriv_kernel(const LbArgs a, const double* ws_all)
{
    extern __shared__ double l64_lds[];          // A | dA | U | Mst (L64_MAT each) | list
    double* A = l64_lds; double* dA = A + L64_MAT; double* U = dA + L64_MAT; double* Mst = U + L64_MAT;
    double* lre = Mst + L64_MAT; double* lim = lre + L64_LIST; int* lki = (int*)(lim + L64_LIST);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lr = lane & 15, lk = lane >> 4;
    const int64_t gp = a.set_param[blockIdx.x];
    int m = -1, q = -1;
    for (int mm = 0; mm < a.n_members; mm++)
        if (gp >= a.param0[mm] && gp < a.param0[mm] + a.n_params[mm]) { m = mm; q = (int)(gp - a.param0[mm]); }
    if (m < 0) return;
    const int nP = a.n_params[m];
    const double* ws = ws_all + (size_t)m * L64_WS_STRIDE;
    double* th = Mst;                            // (the member's parameters: needed until the list exists; nP <= L64_MAT)
    for (int k = t; k < nP; k += 256) th[k] = a.theta[a.param0[m] + k];
    __syncthreads();
    const int n_list = l64_dcoef_list(a, m, q, th, lki, lre, lim, t);
    __syncthreads();
    const int shift = (int)ws[L64_WS_SHIFT];
    {   // dA = sum over the list, element e = t + 256 j (coalesced over the term matrices), scaled like A
        double acc[16];
#pragma unroll
        for (int j = 0; j < 16; j++) acc[j] = 0.0;
        const double* tr = a.term_re + (size_t)a.term_off[m] * L64_DD + t;
        const double* ti = a.term_im + (size_t)a.term_off[m] * L64_DD + t;
        for (int u = 0; u < n_list; u++) {
            const size_t ko = (size_t)lki[u] * L64_DD;
            const double dr = lre[u], di = lim[u];
#pragma unroll
            for (int j = 0; j < 16; j++) acc[j] += dr * tr[ko + 256 * j] + di * ti[ko + 256 * j];
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int e = t + 256 * j;
            const double v = ldexp(acc[j], -shift);
            dA[(e >> 6) * L64_XS + (e & 63)] = v; U[(e >> 6) * L64_XS + (e & 63)] = v;
        }
    }
    l64_load_global(A, ws + L64_WS_A, t);
    __syncthreads();
    l64_d4 dS[4];
    L64_FOR_TILE(c, r) dS[c][r] = dA[L64_ROW(r) * L64_XS + L64_COL(c)];
    for (int k = 2; k <= LB_TAYLOR; k++) {
        l64_load_global(Mst, ws + L64_WS_M + (size_t)(k - 2) * L64_DD, t);         // M_{k-1}
        __syncthreads();
        l64_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        l64_mm(U, A, acc, w, lr, lk);
        l64_mm(Mst, dA, acc, w, lr, lk);
        __syncthreads();                                                          // (every read of U is done)
        const double inv = 1.0 / (double)k;
        L64_FOR_TILE(c, r) { const double v = acc[c][r] * inv; U[L64_ROW(r) * L64_XS + L64_COL(c)] = v; dS[c][r] += v; }
        __syncthreads();
    }
    for (int sq = 0; sq < shift; sq++) {
        L64_FOR_TILE(c, r) U[L64_ROW(r) * L64_XS + L64_COL(c)] = dS[c][r];
        l64_load_global(Mst, ws + L64_WS_S + (size_t)sq * L64_DD, t);              // S_sq
        __syncthreads();
        l64_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        l64_mm(U, Mst, acc, w, lr, lk);
        l64_mm(Mst, U, acc, w, lr, lk);
        __syncthreads();
        L64_FOR_TILE(c, r) dS[c][r] = acc[c][r];
    }
    // ---- compose with the static factor; write column q ---------------------------------------------------------------------
    L64_FOR_TILE(c, r) U[L64_ROW(r) * L64_XS + L64_COL(c)] = dS[c][r];
    const int kind = a.kind[m];
    const double* stat = a.statics + a.static_off[m];
    double* const out = a.deriv_out + a.deriv_off[m];
    if (kind == GST_KIND_GATE) {                  // d(E U)[i][j] = sum_l dE[i][l] U[l][j]
        l64_load_global(A, stat, t);
        __syncthreads();
        l64_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        l64_mm(U, A, acc, w, lr, lk);
        L64_FOR_TILE(c, r) out[((size_t)L64_ROW(r) * 64 + L64_COL(c)) * nP + q] = acc[c][r];
    } else if (kind == GST_KIND_RHO) {
        __syncthreads();
        if (t < 64) {
            double acc = 0.0;
            for (int l = 0; l < 64; l++) acc += U[t * L64_XS + l] * stat[l];
            out[(size_t)t * nP + q] = acc;
        }
    } else {
        __syncthreads();
        const int ne = a.n_eff[m];
        for (int k = t; k < ne * 64; k += 256) {
            const int e = k >> 6, c = k & 63;
            double acc = 0.0;
            for (int l = 0; l < 64; l++) acc += stat[e * 64 + l] * U[l * L64_XS + c];
            out[((size_t)e * 64 + c) * nP + q] = acc;
        }
    }
}

}  // namespace

hipError_t launch_lindblad_build(int D, const LbArgs& a, int64_t n_sets, hipStream_t stream)
{
    if (n_sets <= 0) return hipSuccess;
    if (n_sets > 0x7fffffffLL) return hipErrorInvalidValue;
    (void)hipGetLastError();
    if (D == 4) hipLaunchKernelGGL((lindblad_build_kernel<4>), dim3((unsigned)n_sets), dim3(64), 0, stream, a);
    else if (D == 16) hipLaunchKernelGGL((lindblad_build_kernel<16>), dim3((unsigned)n_sets), dim3(256), 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

size_t lindblad64_workspace_doubles(int n_members) { return (size_t)n_members * L64_WS_STRIDE; }

// D = 64: the base model only (a.set_param == NULL): assemble every member's generator, exponentiate, compose.
hipError_t launch_lindblad64_build(const LbArgs& a, double* ws, int max_member_params, hipStream_t stream)
{
    if (a.set_param || !ws || max_member_params > L64_MAT) return hipErrorInvalidValue;
    (void)hipGetLastError();
    const size_t lds_a = (size_t)3 * max_member_params * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*)lindblad64_assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lindblad64_assemble_kernel, dim3((unsigned)a.n_members, 16), dim3(256), lds_a, stream, a, ws);
    const size_t lds_e = ((size_t)3 * L64_MAT + 64) * sizeof(double);
    e = hipFuncSetAttribute((const void*)lindblad64_exp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lindblad64_exp_kernel, dim3((unsigned)a.n_members), dim3(256), lds_e, stream, a, ws);
    return hipGetLastError();
}

}  // namespace gst

namespace gst {
hipError_t launch_lindblad_derivs(int D, const LbArgs& a, int64_t n_params_total, hipStream_t stream)
{
    if (n_params_total <= 0) return hipSuccess;
    if (n_params_total > 0x7fffffffLL || !a.set_param || !a.deriv_out || !a.deriv_off) return hipErrorInvalidValue;
    (void)hipGetLastError();
    if (D == 4) hipLaunchKernelGGL((lindblad_deriv_kernel<4>), dim3((unsigned)n_params_total), dim3(64), 0, stream, a);
    else if (D == 16) hipLaunchKernelGGL((lindblad_deriv_kernel<16>), dim3((unsigned)n_params_total), dim3(256), 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// D = 64: needs the workspace launch_lindblad64_build filled for the SAME parameter vector.
hipError_t launch_lindblad64_derivs(const LbArgs& a, const double* ws, int64_t n_params_total, hipStream_t stream)
{
    if (n_params_total <= 0) return hipSuccess;
    if (n_params_total > 0x7fffffffLL || !a.set_param || !a.deriv_out || !a.deriv_off || !ws) return hipErrorInvalidValue;
    (void)hipGetLastError();
    const size_t lds = ((size_t)4 * L64_MAT + 2 * L64_LIST) * sizeof(double) + (size_t)L64_LIST * sizeof(int);
    hipError_t e = hipFuncSetAttribute((const void*)lindblad64_deriv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lindblad64_deriv_kernel, dim3((unsigned)n_params_total), dim3(256), lds, stream, a, ws);
    return hipGetLastError();
}
}  // namespace gst
