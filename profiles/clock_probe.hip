// Shader clock under load: s_memtime (clock64) against the constant 100 MHz wall clock, for a bf16-MFMA-dense loop, an fp32-MFMA loop and a VALU loop.
// hipcc --offload-arch=gfx950 -O3 scratch/probe/clock_probe.hip -o scratch/probe/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void probe(long long* out, int iters, float* sink) {
    const long long w0 = wall_clock64(), c0 = clock64();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, v, acc[u & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) v = fmaf(v, 1.0001f, 0.5f);
        }
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    float s = v;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 8 + (threadIdx.x >> 6);
        out[2 * w] = w1 - w0; out[2 * w + 1] = c1 - c0;
    }
}
int main() {
    const int blocks = 256, waves = blocks * 8;
    long long* d; float* sink;
    hipMalloc(&d, waves * 16); hipMalloc(&sink, 4);
    std::vector<long long> h(waves * 2);
    const char* names[3] = {"bf16 MFMA 32x32x16", "fp32 MFMA 32x32x2", "VALU fma"};
    for (int mode = 0; mode < 3; ++mode)
        for (int iters : {200, 2000, 20000}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; ++rep) {
                if (rep == 2) hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 0, 0, d, iters, sink);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 0, 0, d, iters, sink);
                else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(512), 0, 0, d, iters, sink);
            }
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float kms = 0.f; hipEventElapsedTime(&kms, e0, e1);
            hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
            double sw = 0, sc = 0;
            for (int w = 0; w < waves; ++w) { sw += h[2 * w]; sc += h[2 * w + 1]; }
            const double us = sw / waves / 100.0;            // wall clock: 100 MHz
            if (mode < 2) printf("   kernel %.1f us -> %.1f TFLOP/s whole chip\n", kms * 1e3, (double)waves * iters * 8 * (mode == 0 ? 32768.0 : 4096.0) / (kms * 1e-3) / 1e12);
            printf("%-20s iters %6d: %8.1f us per wave, s_memtime ticks / wall tick = %.3f -> %.3f GHz if s_memtime counts shader cycles; per MFMA/step %.1f ticks\n",
                   names[mode], iters, us, sc / sw, sc / sw * 0.1, sc / waves / ((double)iters * 8));
        }
    return 0;
}
