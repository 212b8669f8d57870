// conv_wino43b_kernel's multiply phase in isolation: 9 "points" x 12 MFMAs on 14 + 1 accumulator blocks (AGPRs), A fragments from two alternating
// register sets written by the gap fillers (the split of eight fresh values per point), B fragments from an 18-fragment ring, optionally refilled
// by buffer loads in the gaps.  One wavefront per SIMD.  Prints s_memtime ticks per MFMA.
//   MODE 0: MFMAs only   1: + split fillers (44 per point)   2: + fragment refills (6 buffer_load_dwordx4 per point)   3: split + refills
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u32x4* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ t, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 ub[6][3], af[2][3];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) ub[i][j] = in[lane + 64 * (3 * i + j)];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) af[i][j] = in[lane + 64 * (18 + 3 * i + j)];
    f32x16 acc[18];
    for (int j = 0; j < 18; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float x[8]; unsigned h[8];
    for (int i = 0; i < 8; ++i) { x[i] = (float)(lane * 3 + i) * 1.37f; h[i] = 0; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(in), 0, 64 * 24 * 16, 0x00020000);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
            constexpr int S[13] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 35, 38, 41, 44};
            const int u = 2 * p, sl0 = u % 6, sl1 = (u + 1) % 6;
            u32x4* const a = af[p & 1];
            u32x4* const fn = af[(p + 1) & 1];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int pr = q >> 1;
                const int blk = 2 * p + (q & 1);
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[pr]]), __builtin_bit_cast(bf16x8, ub[(q & 1) ? sl1 : sl0][PB[pr]]), acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE & 1) {
#pragma unroll
                    for (int idx = S[q]; idx < S[q + 1]; ++idx) {
                        const int lvl = idx < 4 ? 0 : idx < 12 ? 1 : idx < 20 ? 2 : idx < 24 ? 3 : idx < 32 ? 4 : idx < 40 ? 5 : 6;
                        const int kk = idx - (lvl == 0 ? 0 : lvl == 1 ? 4 : lvl == 2 ? 12 : lvl == 3 ? 20 : lvl == 4 ? 24 : lvl == 5 ? 32 : 40);
                        if (lvl == 0 || lvl == 3 || lvl == 6) fn[lvl / 3][kk] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[2 * kk + 1]), __builtin_bit_cast(unsigned, x[2 * kk]), 0x07060302u);
                        else if (lvl == 1 || lvl == 4) h[kk] = __builtin_bit_cast(unsigned, x[kk]) & 0xffff0000u;
                        else asm("v_sub_f32 %0, %1, %2" : "=v"(x[kk]) : "v"(x[kk]), "v"(h[kk]));
                    }
                    if (q == 11) { for (int e = 0; e < 8; ++e) x[e] = x[e] * 1.0001f + (float)e; }      // fresh values for the next point
                }
                if (MODE & 2) {
                    if (q == 3) ub[sl0][2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 2048, (3 * sl0) * 1024, 0));
                    if (q == 4) ub[sl1][2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 2048, (3 * sl1) * 1024, 0));
                    if (q == 7) ub[sl0][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 1024, (3 * sl0) * 1024, 0));
                    if (q == 8) ub[sl1][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 1024, (3 * sl1) * 1024, 0));
                    if (q == 11) {
                        ub[sl0][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (3 * sl0) * 1024, 0));
                        ub[sl1][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (3 * sl1) * 1024, 0));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)h[i];
    for (int j = 0; j < 18; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
static u32x4* in; static float* out; static unsigned long long* t;
template <int MODE> static double run() {
    const int iters = 100;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, in, out, t, iters); (void)hipDeviceSynchronize(); }
    static unsigned long long h[1024]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    return s / 1024 / (iters * 108.0);
}
int main() {
    (void)hipMalloc(&in, 64 * 24 * 16); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&t, 1024 * 8);
    (void)hipMemset(in, 0x3c, 64 * 24 * 16);
    printf("MFMAs only            %.1f ticks per MFMA\n", run<0>());
    printf("+ split fillers       %.1f\n", run<1>());
    printf("+ fragment refills    %.1f\n", run<2>());
    printf("+ both                %.1f\n", run<3>());
    return 0;
}
