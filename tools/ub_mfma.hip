// development aid: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- cycles per instruction for one wavefront with NACC
// independent accumulators, and the aggregate rate with 1..4 wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ub_mfma tools/ub_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double* out, unsigned long long* cyc, int iters)
{
    d4_t acc[NACC];
    for (int x = 0; x < NACC; x++) acc[x] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 0.001 + 1.0, b = threadIdx.x * 0.002 + 0.5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int x = 0; x < NACC; x++) acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[x], 0, 0, 0);
    }
    double s = 0.0;
    for (int x = 0; x < NACC; x++) s += acc[x][0] + acc[x][1] + acc[x][2] + acc[x][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int waves_per_block, int blocks, const char* what)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * waves_per_block * 64 * 8);
    hipMalloc(&cyc, (size_t)blocks * 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC;
    const double total = n * blocks * waves_per_block;
    std::printf("%-44s NACC %d: %.1f memtime ticks per MFMA per wave; %.3f ms -> %.2f TFLOP/s f64, %.1f ns per MFMA per wave\n", what, NACC,
                (double)h / n, ms, total * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / n);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1>(1, 1, "1 wave, dependent chain");
    run<4>(1, 1, "1 wave, 4 independent accumulators");
    run<8>(1, 1, "1 wave, 8 independent accumulators");
    run<8>(4, 256, "256 blocks x 4 waves (1 wave / SIMD)");
    run<8>(4, 512, "512 blocks x 4 waves (2 waves / SIMD)");
    run<8>(4, 1024, "1024 blocks x 4 waves (4 waves / SIMD)");
    run<4>(4, 1024, "1024 blocks x 4 waves (4 waves / SIMD)");
    return 0;
}
