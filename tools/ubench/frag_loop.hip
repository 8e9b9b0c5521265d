// Micro-benchmark: cycles per weight fragment of the fused row-panel kernel's inner loop (LDS fragment read -> MFMA),
// one workgroup per CU, no global traffic.  Variants: MFMA shape (16x16x32 x TM | 32x32x16), waves per SIMD, wait style.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -o frag_loop frag_loop.hip && ./frag_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;

template <int OFF> __device__ __forceinline__ void frag_read_to(half8& r, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); }
template <int N> __device__ __forceinline__ void frag_wait(half8& r) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N)); }

// MODE 0: 16x16x32, TM tiles per fragment ; MODE 1: 32x32x16, one MFMA per fragment
template <int MODE, int TM, int NW, int PF, int NF>
__global__ __launch_bounds__(NW * 64, NW / 4) void k(long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < NF * 1024 / 4; i += NW * 64) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned la = (unsigned)(uintptr_t)(lptr_t)(smem + lane * 16);
    half8 x[TM];
    for (int t = 0; t < TM; ++t) for (int j = 0; j < 8; ++j) x[t][j] = (_Float16)(0.01f * (lane + j + t));
    float4v acc[6][TM];
    float16v acc32[3];
    for (int a = 0; a < 6; ++a) for (int t = 0; t < TM; ++t) acc[a][t] = (float4v){0, 0, 0, 0};
    for (int a = 0; a < 3; ++a) for (int j = 0; j < 16; ++j) acc32[a][j] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        half8 q[PF];
        [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read_to<I * 1024>(q[I], la), ...); }(std::make_integer_sequence<int, PF>{});
        [&]<int... Fi>(std::integer_sequence<int, Fi...>) {
            ([&] {
                constexpr int f = Fi;
                constexpr int pending = (NF - f < PF ? NF - f : PF) - 1;
                frag_wait<pending>(q[f % PF]);
                const half8 w = q[f % PF];
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) acc[f % 6][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x[t], acc[f % 6][t], 0, 0, 0);
                } else {
                    acc32[f % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x[0], acc32[f % 3], 0, 0, 0);
                }
                if constexpr (f + PF < NF) frag_read_to<(f + PF) * 1024>(q[f % PF], la);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, NF>{});
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < 6; ++a) for (int t = 0; t < TM; ++t) s += acc[a][t][0];
    for (int a = 0; a < 3; ++a) s += acc32[a][0];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int TM, int NW, int PF>
void run(const char* name) {
    constexpr int NF = 24;
    long long* d; float* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4);
    const int iters = 2000;
    auto kern = k<MODE, TM, NW, PF, NF>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), 128 * 1024, 0, d, iters, sink);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), 128 * 1024, 0, d, iters, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    const double frags = (double)iters * NF;
    const double flop_per_frag_wave = MODE == 0 ? 16384.0 * TM : 32768.0;
    const double tf = 256.0 * NW * frags * flop_per_frag_wave / (ms * 1e-3) / 1e12;
    printf("%-44s  %7.1f clk/fragment/wave (clock64 100MHz ticks -> x%.1f)  %8.1f TFLOP/s  wall %.3f ms\n", name, avg / frags, 1.0, tf, ms);
}

int main() {
    run<0, 2, 4, 6>("16x16x32 TM=2, 4 waves (1/SIMD), PF 6");
    run<0, 2, 4, 8>("16x16x32 TM=2, 4 waves (1/SIMD), PF 8");
    run<0, 1, 8, 6>("16x16x32 TM=1, 8 waves (2/SIMD), PF 6");
    run<0, 1, 4, 6>("16x16x32 TM=1, 4 waves (1/SIMD), PF 6");
    run<1, 1, 4, 6>("32x32x16,       4 waves (1/SIMD), PF 6");
    run<1, 1, 4, 8>("32x32x16,       4 waves (1/SIMD), PF 8");
    run<1, 1, 8, 6>("32x32x16,       8 waves (2/SIMD), PF 6");
    return 0;
}
