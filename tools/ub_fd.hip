// development microbenchmark: the column loop of the FD walk kernel (matvec_t<16,1>: coefficients through the scalar cache
// into SGPRs, one stage ahead) with W wavefronts per SIMD.  What is a lone wavefront waiting for?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pygsti_amd/csrc tools/ub_fd.hip -o /tmp/ub_fd && /tmp/ub_fd
#include "../pygsti_amd/csrc/gst_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace gst;

// V = 0: the product's matvec_t (SMEM feed, one stage ahead)      V = 1: no loads at all (coefficients of gate 0 held
// in SGPRs: the pure VALU ceiling)     V = 2: SMEM feed, but every application reads the SAME gate (hot lines)
// V = 3: pointer chase through the scalar cache (latency of one dependent s_load_dwordx2 on 12 KB of data)
// two models per lane share every SGPR coefficient: o2 = M v2 alongside o = M v (+ one special row each)
template <int D>
__device__ __forceinline__ void matvec_t2(cdouble_p __restrict__ Mt, const double (&v)[D], const double (&w)[D], double (&o)[D], double (&p)[D],
                                          const double (&sp)[2][D], double (&r)[2])
{
#pragma unroll
    for (int i = 0; i < D; i++) { o[i] = 0.0; p[i] = 0.0; }
    r[0] = 0.0; r[1] = 0.0;
    double cur[D], nxt[D];
    {
        cdouble_p q = Mt;
        asm volatile("" : "+s"(q));
#pragma unroll
        for (int i = 0; i < D; i++) cur[i] = q[i];
    }
#pragma unroll
    for (int j = 0; j < D; j++) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (j + 1 < D) {
            cdouble_p q = Mt + (j + 1) * D;
            asm volatile("" : "+s"(q));
#pragma unroll
            for (int i = 0; i < D; i++) nxt[i] = q[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double vj = v[j], wj = w[j];
#pragma unroll
        for (int i = 0; i < D; i++) o[i] = o[i] + cur[i] * vj;
        r[0] = r[0] + sp[0][j] * vj;
#pragma unroll
        for (int i = 0; i < D; i++) p[i] = p[i] + cur[i] * wj;
        r[1] = r[1] + sp[1][j] * wj;
        pin<D>(o); pin<D>(p);
#pragma unroll
        for (int i = 0; i < D; i++) cur[i] = nxt[i];
    }
}

// two models per lane + half-column stages (one s_load_dwordx16 per 2 x (8 + special) multiply-add pairs)
template <int D>
__device__ __forceinline__ void matvec_t2h(cdouble_p __restrict__ Mt, const double (&v)[D], const double (&w)[D], double (&o)[D], double (&p)[D],
                                           const double (&sp)[2][D], double (&r)[2])
{
    constexpr int H = D / 2;
#pragma unroll
    for (int i = 0; i < D; i++) { o[i] = 0.0; p[i] = 0.0; }
    r[0] = 0.0; r[1] = 0.0;
    double cur[H], nxt[H];
    {
        cdouble_p q = Mt;
        asm volatile("" : "+s"(q));
#pragma unroll
        for (int i = 0; i < H; i++) cur[i] = q[i];
    }
#pragma unroll
    for (int st = 0; st < 2 * D; st++) {
        const int j = st / 2, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (st + 1 < 2 * D) {
            cdouble_p q = Mt + (st + 1) * H;
            asm volatile("" : "+s"(q));
#pragma unroll
            for (int i = 0; i < H; i++) nxt[i] = q[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double vj = v[j], wj = w[j];
#pragma unroll
        for (int i = 0; i < H; i++) o[h * H + i] = o[h * H + i] + cur[i] * vj;
        if (h) r[0] = r[0] + sp[0][j] * vj;
#pragma unroll
        for (int i = 0; i < H; i++) p[h * H + i] = p[h * H + i] + cur[i] * wj;
        if (h) r[1] = r[1] + sp[1][j] * wj;
        pin<D>(o); pin<D>(p);
#pragma unroll
        for (int i = 0; i < H; i++) cur[i] = nxt[i];
    }
}

// hybrid feed: even columns through the scalar cache into SGPRs, odd columns broadcast out of LDS into VGPRs (every lane
// reads the same 16 bytes: ds_read_b128 of a uniform address); a stage is TWO columns, so one wavefront's serial scalar
// stream (2 loads per stage) has 272 cycles of arithmetic to hide under
template <int D>
__device__ __forceinline__ void matvec_hy(cdouble_p __restrict__ Mt, const double* __restrict__ Lt, const double (&v)[D], double (&o)[D],
                                          const double (&sp)[1][D], double (&r)[1])
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < D; i++) o[i] = 0.0;
    r[0] = 0.0;
    double cs[D], ns[D];          // scalar-fed column (SGPRs)
    double cl[D], nl[D];          // LDS-fed column (VGPRs)
    {
        cdouble_p q = Mt;
        asm volatile("" : "+s"(q));
#pragma unroll
        for (int i = 0; i < D; i++) cs[i] = q[i];
        const d2_t* l = (const d2_t*)__builtin_assume_aligned(Lt + D, 16);
#pragma unroll
        for (int i = 0; i < D; i += 2) { const d2_t t = l[i / 2]; cl[i] = t.x; cl[i + 1] = t.y; }
    }
#pragma unroll
    for (int j = 0; j < D; j += 2) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (j + 2 < D) {
            cdouble_p q = Mt + (j + 2) * D;
            asm volatile("" : "+s"(q));
#pragma unroll
            for (int i = 0; i < D; i++) ns[i] = q[i];
            const d2_t* l = (const d2_t*)__builtin_assume_aligned(Lt + (j + 3) * D, 16);
#pragma unroll
            for (int i = 0; i < D; i += 2) { const d2_t t = l[i / 2]; nl[i] = t.x; nl[i + 1] = t.y; }
        }
        __builtin_amdgcn_sched_barrier(0);
        const double v0 = v[j], v1 = v[j + 1];
#pragma unroll
        for (int i = 0; i < D; i++) o[i] = o[i] + cs[i] * v0;
        r[0] = r[0] + sp[0][j] * v0;
        pin<D>(o);
#pragma unroll
        for (int i = 0; i < D; i++) o[i] = o[i] + cl[i] * v1;
        r[0] = r[0] + sp[0][j + 1] * v1;
        pin<D>(o);
#pragma unroll
        for (int i = 0; i < D; i++) { cs[i] = ns[i]; cl[i] = nl[i]; }
    }
}

// half-column stages: one s_load_dwordx16 (8 coefficients) per stage of 16 + 1 VALU pairs
template <int D>
__device__ __forceinline__ void matvec_th(cdouble_p __restrict__ Mt, const double (&v)[D], double (&o)[D], const double (&sp)[1][D], double (&r)[1])
{
    constexpr int H = D / 2;
#pragma unroll
    for (int i = 0; i < D; i++) o[i] = 0.0;
    r[0] = 0.0;
    double cur[H], nxt[H];
    {
        cdouble_p q = Mt;
        asm volatile("" : "+s"(q));
#pragma unroll
        for (int i = 0; i < H; i++) cur[i] = q[i];
    }
#pragma unroll
    for (int st = 0; st < 2 * D; st++) {
        const int j = st / 2, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (st + 1 < 2 * D) {
            cdouble_p q = Mt + (st + 1) * H;
            asm volatile("" : "+s"(q));
#pragma unroll
            for (int i = 0; i < H; i++) nxt[i] = q[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double vj = v[j];
#pragma unroll
        for (int i = 0; i < H; i++) o[h * H + i] = o[h * H + i] + cur[i] * vj;
        if (h) r[0] = r[0] + sp[0][j] * vj;
        pin<D>(o);
#pragma unroll
        for (int i = 0; i < H; i++) cur[i] = nxt[i];
    }
}

template <int V>
__global__ __launch_bounds__(64, ((V == 4 || V == 6) ? 2 : (V == 7 ? 3 : 4))) void k(const double* gates_t, int n_apps, double* out, unsigned long long* cyc, const int* chase)
{
    constexpr int D = 16;
    const int lane = threadIdx.x;
    cdouble_p gt = as_const(gates_t);
    double v[D], sp[1][D];
    for (int j = 0; j < D; j++) { v[j] = 1.0 + 0.001 * lane + 0.01 * j; sp[0][j] = 0.5 + 0.001 * j; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (V == 4 || V == 6) {
        double w[D], sp2[2][D];
        for (int j = 0; j < D; j++) { w[j] = 0.9 + 0.002 * lane + 0.01 * j; sp2[0][j] = sp[0][j]; sp2[1][j] = 0.4 + 0.002 * j; }
        int g = 0;
        for (int s = 0; s < n_apps; s++) {
            double o[D], p[D], r[2];
            cdouble_p Mt = gt + (int64_t)g * D * D;
            asm volatile("" : "+s"(Mt));
            if constexpr (V == 4) matvec_t2<D>(Mt, v, w, o, p, sp2, r); else matvec_t2h<D>(Mt, v, w, o, p, sp2, r);
            for (int j = 0; j < D; j++) { v[j] = (lane == j) ? r[0] : o[j] * 0.05; w[j] = (lane == j) ? r[1] : p[j] * 0.05; }
            g = (g == 5) ? 0 : g + 1;
        }
        for (int j = 0; j < D; j++) v[j] += w[j];
    } else if constexpr (V == 7) {
        __shared__ __attribute__((aligned(16))) double ldsG[6 * D * D];
        for (int k = lane; k < 6 * D * D; k += 64) ldsG[k] = gates_t[k];
        __syncthreads();
        int g = 0;
        for (int s = 0; s < n_apps; s++) {
            double o[D], r[1];
            cdouble_p Mt = gt + (int64_t)g * D * D;
            asm volatile("" : "+s"(Mt));
            matvec_hy<D>(Mt, ldsG + g * D * D, v, o, sp, r);
            for (int j = 0; j < D; j++) v[j] = (lane == j) ? r[0] : o[j] * 0.05;
            g = (g == 5) ? 0 : g + 1;
        }
    } else if constexpr (V == 5) {
        int g = 0;
        for (int s = 0; s < n_apps; s++) {
            double o[D], r[1];
            cdouble_p Mt = gt + (int64_t)g * D * D;
            asm volatile("" : "+s"(Mt));
            matvec_th<D>(Mt, v, o, sp, r);
            for (int j = 0; j < D; j++) v[j] = (lane == j) ? r[0] : o[j] * 0.05;
            g = (g == 5) ? 0 : g + 1;
        }
    } else if constexpr (V == 3) {
        const __attribute__((address_space(4))) int* c = (const __attribute__((address_space(4))) int*)chase;
        int idx = 0;
        for (int s = 0; s < n_apps * 16; s++) { idx = c[idx]; asm volatile("" : "+s"(idx)); }
        v[0] += idx;
    } else {
        int g = 0;
        for (int s = 0; s < n_apps; s++) {
            double o[D], r[1];
            if constexpr (V == 1) {
                // coefficients already in registers: the same arithmetic without any memory operation
                double c0[D];
                for (int i = 0; i < D; i++) c0[i] = gt[i];
                for (int i = 0; i < D; i++) o[i] = 0.0;
                r[0] = 0.0;
#pragma unroll
                for (int j = 0; j < D; j++) {
                    const double vj = v[j];
#pragma unroll
                    for (int i = 0; i < D; i++) o[i] = o[i] + c0[i] * vj;
                    r[0] = r[0] + sp[0][j] * vj;
                    pin<D>(o);
                }
            } else {
                cdouble_p Mt = gt + (int64_t)(V == 2 ? 0 : g) * D * D;
                asm volatile("" : "+s"(Mt));
                matvec_t<D, 1, D>(Mt, 0, v, o, sp, r);
            }
            for (int j = 0; j < D; j++) v[j] = (lane == j) ? r[0] : o[j] * 0.05;      // keep the values bounded, state in place
            g = (g == 5) ? 0 : g + 1;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double acc = 0; for (int j = 0; j < D; j++) acc += v[j];
    out[(size_t)blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int D = 16, nG = 6, n_apps = 1200;
    std::vector<double> g((size_t)nG * D * D);
    for (size_t i = 0; i < g.size(); i++) g[i] = 0.1 * ((i * 7919) % 13) / 13.0;
    std::vector<int> chase(1536);          // 12 KB of ints... a 64-byte-stride cycle
    for (int i = 0; i < 1536; i++) chase[i] = (i + 16 * 37) % 1536;
    double *d_g, *d_out; unsigned long long* d_c; int* d_chase;
    hipMalloc(&d_g, g.size() * 8); hipMemcpy(d_g, g.data(), g.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&d_chase, chase.size() * 4); hipMemcpy(d_chase, chase.data(), chase.size() * 4, hipMemcpyHostToDevice);
    const int maxb = 1024 * 8;
    hipMalloc(&d_out, (size_t)maxb * 64 * 8); hipMalloc(&d_c, (size_t)maxb * 8);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1e6;
    printf("device clock %.2f GHz, %d CUs\n", ghz, prop.multiProcessorCount);
    for (int V : {0, 5, 7}) {
        for (int W : {1, 2, 3, 4}) {
            if ((V == 4 || V == 6) && W > 2) continue;
            if (V == 7 && W > 3) continue;
            const int blocks = 1024 * W;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (V == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                if (V == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                if (V == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                if (V == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                if (V == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                if (V == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                if (V == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                if (V == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, d_g, n_apps, d_out, d_c, d_chase);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c(blocks);
            hipMemcpy(c.data(), d_c, (size_t)blocks * 8, hipMemcpyDeviceToHost);
            std::sort(c.begin(), c.end());
            const double med = (double)c[blocks / 2], mx = (double)c[blocks - 1];
            const double per = med / (V == 3 ? n_apps * 16.0 : (double)n_apps);
            printf("V=%d W=%d: kernel %.3f ms; per wave median %.0f cycles/%s (max wave %.0f); aggregate %.2f applications/us per SIMD\n",
                   V, W, ms, per, V == 3 ? "dependent load" : "application", mx / (V == 3 ? n_apps * 16.0 : n_apps),
                   V == 3 ? 0.0 : (double)n_apps * W * ((V == 4 || V == 6) ? 2 : 1) / (ms * 1e3));
        }
    }
    return 0;
}
