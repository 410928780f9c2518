// development aid: the step of chain64_resident_kernel in isolation -- 16 DEPENDENT v_mfma_f64_16x16x4_f64 with 16 different A
// and B register pairs, accumulator and B operands in the architectural ("v") or the accumulation ("a") file; ns per MFMA for a
// lone wavefront per SIMD on one CU and on every CU (clock / power effects).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ub_mfma2 tools/ub_mfma2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

#define CHAIN(ACON, BCON)                                                                                   \
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %17, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %18, %0\n\t"     \
                 "v_mfma_f64_16x16x4_f64 %0, %3, %19, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %20, %0\n\t"     \
                 "v_mfma_f64_16x16x4_f64 %0, %5, %21, %0\n\tv_mfma_f64_16x16x4_f64 %0, %6, %22, %0\n\t"     \
                 "v_mfma_f64_16x16x4_f64 %0, %7, %23, %0\n\tv_mfma_f64_16x16x4_f64 %0, %8, %24, %0\n\t"     \
                 "v_mfma_f64_16x16x4_f64 %0, %9, %25, %0\n\tv_mfma_f64_16x16x4_f64 %0, %10, %26, %0\n\t"    \
                 "v_mfma_f64_16x16x4_f64 %0, %11, %27, %0\n\tv_mfma_f64_16x16x4_f64 %0, %12, %28, %0\n\t"   \
                 "v_mfma_f64_16x16x4_f64 %0, %13, %29, %0\n\tv_mfma_f64_16x16x4_f64 %0, %14, %30, %0\n\t"   \
                 "v_mfma_f64_16x16x4_f64 %0, %15, %31, %0\n\tv_mfma_f64_16x16x4_f64 %0, %16, %32, %0\n\t"   \
                 : ACON(acc)                                                                                \
                 : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7]), "v"(A[8]), "v"(A[9]),     \
                   "v"(A[10]), "v"(A[11]), "v"(A[12]), "v"(A[13]), "v"(A[14]), "v"(A[15]),                                           \
                   BCON(B[0]), BCON(B[1]), BCON(B[2]), BCON(B[3]), BCON(B[4]), BCON(B[5]), BCON(B[6]), BCON(B[7]), BCON(B[8]),      \
                   BCON(B[9]), BCON(B[10]), BCON(B[11]), BCON(B[12]), BCON(B[13]), BCON(B[14]), BCON(B[15]))

template <int V>      // bit 0: accumulator in the accumulation file; bit 1: B operands in the accumulation file; bit 2: two independent chains
__global__ __launch_bounds__(256) void k(double* out, int iters)
{
    double A[16], B[16];
    for (int i = 0; i < 16; i++) { A[i] = threadIdx.x * 1e-3 + i; B[i] = threadIdx.x * 2e-3 - i; }
    d4_t acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; it++) {
        if constexpr ((V & 3) == 0) CHAIN("+v", "v");
        if constexpr ((V & 3) == 1) CHAIN("+a", "v");
        if constexpr ((V & 3) == 2) CHAIN("+v", "a");
        if constexpr ((V & 3) == 3) CHAIN("+a", "a");
        if constexpr (V & 4) {
            d4_t t = acc; acc = acc2; acc2 = t;         // (the second chain: same code, other accumulator)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc2[0];
}

template <int V>
void run(int blocks, const char* what)
{
    double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-52s blocks %4d: %.1f ns per MFMA per wave (%.2f TFLOP/s)\n", what, blocks, ms * 1e6 / (iters * 16.0),
                (double)blocks * 4 * iters * 16 * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    for (int blocks : {1, 256, 512, 1024}) {
        run<0>(blocks, "acc v, B v");
        run<1>(blocks, "acc a, B v");
        run<2>(blocks, "acc v, B a");
        run<3>(blocks, "acc a, B a");
    }
    return 0;
}
