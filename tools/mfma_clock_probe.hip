// Diagnostic: sustained bf16 MFMA rate of the two shapes at one wave per SIMD
// (256 workgroups x 4 waves), operands random, 128 accumulator registers per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256, 1) void k32(const bf16x8* src, float* out, int iters) {
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[threadIdx.x + 256 * i];
    for (int i = 0; i < 2; ++i) b[i] = src[threadIdx.x + 256 * (4 + i)];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[1], acc[i][1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) s += acc[i][j][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 1) void k16(const bf16x8* src, float* out, int iters) {
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[threadIdx.x + 256 * i];
    for (int i = 0; i < 4; ++i) b[i] = src[threadIdx.x + 256 * (8 + i)];
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int v = 0; v < 4; ++v) acc[i][j][v] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int v = 0; v < 4; ++v) s += acc[i][j][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3;  // many repetitions give a power probe time to sample
    const int n = 256 * 12 * 8;
    std::vector<unsigned short> h(n);
    srand(1);
    for (auto& x : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.1f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    void *src, *out;
    hipMalloc(&src, n * 2); hipMalloc(&out, 256 * 256 * 4);
    hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape) {
        for (int rep = 0; rep < reps; ++rep) {
            const int iters = shape == 0 ? 60000 : 15000;  // same FLOPs: 8 x 32768 vs 32 x 16384 per iteration
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, (const bf16x8*)src, (float*)out, iters);
            else hipLaunchKernelGGL(k16, dim3(256), dim3(256), 0, 0, (const bf16x8*)src, (float*)out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (shape == 0 ? 8.0 * 32768 : 32.0 * 16384) * iters * 1024.0;
            if (rep < 3 || rep == reps - 1) printf("%s: %.2f ms, %.1f TFLOP/s\n", shape == 0 ? "32x32x16" : "16x16x32", ms, flop / ms * 1e-9);
        }
    }
    return 0;
}
