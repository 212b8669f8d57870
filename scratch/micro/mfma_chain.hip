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
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
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
    float s = 0.f;
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    bf16x8* in; float* out; unsigned long long* t;
    hipMalloc(&in, 64 * 6 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&t, 256 * 4 * 8);
    hipMemset(in, 0x3c, 64 * 6 * 16);
    const int iters = 200;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, in, out, t, iters);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, in, out, t, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h[1024]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
        printf("mode %d (%s): %.1f cycles per MFMA\n", mode, mode == 0 ? "chains of 6 per block" : "round robin over 16 blocks", s / 1024 / (iters * 96.0));
    }
    return 0;
}
