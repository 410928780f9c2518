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
}  // namespace gst
