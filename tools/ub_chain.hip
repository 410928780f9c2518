// development microbenchmark: latency of one chain step of the base pass (lone wavefront), piece by piece.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub_chain.hip -o /tmp/ub_chain && /tmp/ub_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
constexpr int D = 16;
template <int J> __device__ __forceinline__ double bc(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xf, 0xf, true); }
template <int J> __device__ __forceinline__ void bcast_all(double (&bv)[D], double v) { bv[J] = bc<J>(v); if constexpr (J + 1 < D) bcast_all<J + 1>(bv, v); }

// V bits: 1 stream coefficient loads from LDS each step, 2 fetch two program words via VGPR window, 4 ring push+flush,
//         8 use readlane instead of DPP, 16 interleaved (mul/add per term) instead of phase order
template <int V>
__global__ __launch_bounds__(64) void k(const double* gates_t, const uint32_t* prog, int n_steps, double* out, double* cache)
{
    __shared__ double ldsG[6 * D * D];
    __shared__ double sring[32 * D];
    __shared__ int sid[32];
    const int lane = threadIdx.x, li = lane % D, grp = lane / D;
    for (int i = lane; i < 6 * D * D; i += 64) ldsG[i] = gates_t[i];
    __syncthreads();
    double v = 1.0 + li * 0.01;
    double c[D];
    for (int j = 0; j < D; j++) c[j] = ldsG[j * D + li];
    int pc = 0, wbase = 0, n_sr = 0;
    uint32_t win_cur = prog[lane], win_nxt = prog[64 + lane];
    uint32_t g = 0, node = 0;
    for (int s = 0; s < n_steps; s++) {
        if constexpr (V & 2) {
            for (int r = 0; r < 2; r++) {
                if (pc - wbase == 64) { wbase += 64; win_cur = win_nxt; win_nxt = prog[(wbase + 64 + lane) & 0xffff]; }
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, pc - wbase);
                pc++;
                if (r == 0) node = w & 0xfffffff; else g = (w & 0xfffffff) % 6;
            }
        } else { g = (g + 1) % 6; node = s; }
        const double* Gn = ldsG + g * D * D + li;
        double acc = 0.0;
        if constexpr (V & 16) {
            double bv[D];
            if constexpr (V & 8) { for (int j = 0; j < D; j++) { long long b = __double_as_longlong(v); int lo = __builtin_amdgcn_readlane((int)b, j), hi = __builtin_amdgcn_readlane((int)(b >> 32), j); bv[j] = __longlong_as_double(((long long)hi << 32) | (unsigned)lo); } }
            else bcast_all<0>(bv, v);
#pragma unroll
            for (int j = 0; j < D; j++) { acc = acc + c[j] * bv[j]; if constexpr (V & 1) c[j] = Gn[j * D]; }
        } else {
            double bv[D];
            if constexpr (V & 8) { for (int j = 0; j < D; j++) { long long b = __double_as_longlong(v); int lo = __builtin_amdgcn_readlane((int)b, j), hi = __builtin_amdgcn_readlane((int)(b >> 32), j); bv[j] = __longlong_as_double(((long long)hi << 32) | (unsigned)lo); } }
            else bcast_all<0>(bv, v);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < D; j++) bv[j] = c[j] * bv[j];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (V & 1) {
#pragma unroll
                for (int j = 0; j < D; j++) c[j] = Gn[j * D];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < D; j++) acc = acc + bv[j];
            __builtin_amdgcn_sched_barrier(0);
        }
        v = acc * 1e-1;   // keep magnitudes bounded (one extra mul; same in all variants)
        if constexpr (V & 4) {
            if (grp == 0) sring[n_sr * D + lane] = v;
            if (lane == 0) sid[n_sr] = (int)node;
            if (++n_sr == 32) {
                for (int e0 = 0; e0 < 32; e0 += 4) { const int e = e0 + grp; cache[(int64_t)(sid[e] & 0xffff) * D + li] = sring[e * D + li]; }
                n_sr = 0;
            }
        }
    }
    out[blockIdx.x * 64 + lane] = v + c[3];
}
template <int V> void run(const char* name, const double* g, const uint32_t* p, double* out, double* cache, int blocks)
{
    const int n = 100000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, g, p, 1000, out, cache);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, g, p, n, out, cache);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s blocks %5d : %7.1f ns/step  (~%5.0f cycles @2.37GHz)\n", name, blocks, ms * 1e6 / n, ms * 1e6 / n * 2.37);
}
int main()
{
    std::vector<double> g(6 * D * D); for (size_t i = 0; i < g.size(); i++) g[i] = 0.05 + 0.001 * (i % 97);
    std::vector<uint32_t> p(70000); for (size_t i = 0; i < p.size(); i++) p[i] = (i & 1) ? (2u << 28 | (uint32_t)(i % 6)) : (6u << 28 | (uint32_t)(i / 2));
    double *dg, *out, *cache; uint32_t* dp;
    hipMalloc(&dg, g.size() * 8); hipMalloc(&dp, p.size() * 4); hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&cache, 8 * 16 * 70000);
    hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {1, 1024, 2048}) {
        run<0>("phase: matvec only", dg, dp, out, cache, blocks);
        run<16>("interleaved: matvec only", dg, dp, out, cache, blocks);
        run<8>("phase, readlane bcast", dg, dp, out, cache, blocks);
        run<1>("phase + LDS stream", dg, dp, out, cache, blocks);
        run<3>("phase + LDS stream + fetch", dg, dp, out, cache, blocks);
        run<5>("phase + LDS stream + ring", dg, dp, out, cache, blocks);
        run<7>("phase + LDS stream + fetch + ring", dg, dp, out, cache, blocks);
        run<23>("interleaved + LDS stream + fetch + ring", dg, dp, out, cache, blocks);
    }
    return 0;
}
