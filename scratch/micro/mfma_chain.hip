// cycles per v_mfma_f32_32x32x16_bf16 for (a) chains of 6 dependent MFMAs per accumulator block, 16 blocks in turn, (b) the same 96 MFMAs round-robin over the 16 blocks,
// (c) chains of 6 with the block in VGPRs forced by an LDS round trip -- one wavefront per SIMD (256-thread workgroup, 1 per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ t, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = in[lane + 64 * i]; b[i] = in[lane + 64 * (3 + i)]; }
    f32x16 acc[16];
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float v0 = (float)lane, v1 = 1.5f, v2 = 0.25f, v3 = 3.f, v4 = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2 || MODE == 3) {           // 2: chains of 6 with five independent VALU between dependent MFMAs; 3: the same VALU, two blocks alternating
#pragma unroll
            for (int j = 0; j < 16; j += 2)
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const int jj = MODE == 2 ? j + (q / 6) : j + (q & 1);
                    const int qq = MODE == 2 ? q % 6 : q / 2;
                    acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[qq % 3], b[(qq + 1) % 3], acc[jj], 0, 0, 0);
                    v0 = v0 * 1.0001f + 0.5f; v1 = v1 * 0.9999f + v0 * 0.f + 0.25f; v2 = v2 * 1.0002f + 0.125f; v3 = v3 * 0.9998f + 1.f; v4 = v4 * 1.0003f + 2.f;
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q % 3], b[(q + 1) % 3], acc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q % 3], b[(q + 1) % 3], acc[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3 + v4;
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    bf16x8* in; float* out; unsigned long long* t;
    hipMalloc(&in, 64 * 6 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&t, 256 * 4 * 8);
    hipMemset(in, 0x3c, 64 * 6 * 16);
    const int iters = 200;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, in, out, t, iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, in, out, t, iters);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, in, out, t, iters);
            else hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, in, out, t, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h[1024]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
        printf("mode %d (%s): %.1f cycles per MFMA\n", mode, mode == 0 ? "chains of 6 per block" : mode == 1 ? "round robin over 16 blocks" : mode == 2 ? "chains of 6, five VALU between dependent MFMAs" : "two blocks alternating, five VALU per gap", s / 1024 / (iters * 96.0));
    }
    return 0;
}
