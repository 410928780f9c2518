// development aid: does a wavefront's VALU work proceed while another wavefront of the same SIMD runs v_mfma_f64_16x16x4_f64?
// Workgroups of 8 wavefronts (2 per SIMD): wavefronts 0-3 issue MFMAs, 4-7 issue VALU instructions of one kind.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ub_mfma_mix tools/ub_mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int KIND>   // VALU kind: 0 none (idle partner), 1 v_fma_f64, 2 v_add_u32, 3 v_mul_lo_u32
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, int iters, int mfma_on)
{
    const int wv = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    if (wv < 4) {
        if (mfma_on) {
            d4_t acc[4];
            for (int x = 0; x < 4; x++) acc[x] = (d4_t){0.0, 0.0, 0.0, 0.0};
            double a = threadIdx.x * 0.001 + 1.0, b = threadIdx.x * 0.002 + 0.5;
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int x = 0; x < 4; x++) acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[x], 0, 0, 0);
            }
            for (int x = 0; x < 4; x++) s += acc[x][0] + acc[x][1] + acc[x][2] + acc[x][3];
        }
    } else if (KIND == 1) {
        double v[8];
        for (int x = 0; x < 8; x++) v[x] = threadIdx.x * 0.5 + x;
        for (int it = 0; it < iters * 4; it++) {
#pragma unroll
            for (int x = 0; x < 8; x++) v[x] = __builtin_fma(v[x], 1.0000001, 0.5);
        }
        for (int x = 0; x < 8; x++) s += v[x];
    } else if (KIND == 2 || KIND == 3) {
        unsigned v[8];
        for (int x = 0; x < 8; x++) v[x] = threadIdx.x + x;
        for (int it = 0; it < iters * 4; it++) {
#pragma unroll
            for (int x = 0; x < 8; x++) v[x] = (KIND == 2) ? v[x] + 0x9e3779b9u * (unsigned)x + 1u : v[x] * 2654435761u;
        }
        for (int x = 0; x < 8; x++) s += (double)v[x];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int KIND>
void run(const char* what, int mfma_on)
{
    const int blocks = 256, iters = 2000;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 512 * 8); hipMalloc(&cyc, (size_t)blocks * 8 * 8);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, mfma_on);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, mfma_on);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("%-46s MFMA wavefront: %7.1f cycles per MFMA | VALU wavefront: %6.2f cycles per VALU instruction\n", what,
                mfma_on ? (double)h[0] / (iters * 4.0) : 0.0, KIND ? (double)h[4] / (iters * 32.0) : 0.0);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("MFMA alone (partner wavefront idle)", 1);
    run<1>("v_fma_f64 alone", 0);
    run<1>("MFMA + v_fma_f64 on the same SIMD", 1);
    run<2>("v_add_u32 alone", 0);
    run<2>("MFMA + v_add_u32 on the same SIMD", 1);
    run<3>("v_mul_lo_u32 alone", 0);
    run<3>("MFMA + v_mul_lo_u32 on the same SIMD", 1);
    return 0;
}
