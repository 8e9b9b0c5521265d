// Issue-rate probe: v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands, K = 128) against two v_mfma_f32_16x16x32_f16 per
// 64 channels -- the trade the "fp16 + fp8 corrections" conv mode makes.  One wave per SIMD and two, random operands.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f8_rate mfma_f8_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: fp16 16x16x32 x2 ; 1: scaled fp8 16x16x128 ; 2: one fp16 pair + one fp8 (the new mix) ; 3: three fp16 pairs (fp16x3)
__global__ __launch_bounds__(256) void loop(const int* src, float* out, int iters) {
    v8i a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = src[(threadIdx.x * 8 + j + 64 * i) & 4095]; b[i][j] = src[(threadIdx.x * 8 + j + 64 * i + 1000) & 4095]; }
    v4f acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h8 a0 = __builtin_bit_cast(h8, (int __attribute__((ext_vector_type(4)))){a[i][0], a[i][1], a[i][2], a[i][3]});
                const h8 a1 = __builtin_bit_cast(h8, (int __attribute__((ext_vector_type(4)))){a[i][4], a[i][5], a[i][6], a[i][7]});
                const h8 b0 = __builtin_bit_cast(h8, (int __attribute__((ext_vector_type(4)))){b[j][0], b[j][1], b[j][2], b[j][3]});
                const h8 b1 = __builtin_bit_cast(h8, (int __attribute__((ext_vector_type(4)))){b[j][4], b[j][5], b[j][6], b[j][7]});
                if (MODE == 0 || MODE == 2 || MODE == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[i][j], 0, 0, 0);
                }
                if (MODE == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[i][j], 0, 0, 0);
                }
                if (MODE == 1 || MODE == 2)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x74747474);
            }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<int> h(4096);
    srand(1);
    for (auto& v : h) {   // fp16 pairs in [-2, 2) that are also harmless e4m3 bytes
        unsigned x = 0;
        for (int b = 0; b < 4; ++b) x |= (unsigned)((rand() % 96) + 8) << (8 * b);   // bytes 0x08..0x67: finite in both readings
        v = (int)x;
    }
    int* d; float* o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024 * 256 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int blocks : {256, 512}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto run = [&]() {
                if (mode == 0) hipLaunchKernelGGL(loop<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                else if (mode == 1) hipLaunchKernelGGL(loop<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                else if (mode == 2) hipLaunchKernelGGL(loop<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(loop<3>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            // "64-channel tile products" per wave per iteration: 16 (i, j) pairs; MODE 0/1: 1 unit each, MODE 2: 2 units, MODE 3: 3 units
            const double units = (double)blocks * 4 * iters * 16 * (mode == 2 ? 2 : (mode == 3 ? 3 : 1));
            const double flop = units * 2.0 * 16 * 16 * 64;
            printf("blocks %d mode %d: %.3f ms, %.1f TFLOP/s of 64-deep tile products (fp16-equivalent)\n", blocks, mode, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
