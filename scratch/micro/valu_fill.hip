// cycles per v_mfma_f32_32x32x16_bf16 with N independent VALU instructions of one kind written between consecutive MFMAs (one wavefront per SIMD,
// two accumulator blocks alternating as in conv_wino43b_kernel): how many vector instructions hide under one MFMA, and whether packed fp32 counts as one.
//   hipcc --offload-arch=gfx950 -O3 -o valu_fill valu_fill.hip && ./valu_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f32   1: v_pk_fma_f32   2: v_and_b32   3: v_perm_b32   4: v_pk_add_f32   5: v_sub_f32
template <int N, int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const bf16x8* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ t, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = in[lane + 64 * i]; b[i] = in[lane + 64 * (3 + i)]; }
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[8]; f32x2 p[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = (float)(lane + i); p[i] = f32x2{(float)lane, (float)i}; u[i] = lane * 77 + i; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q % 3], b[(q + 1) % 3], acc[q & 1], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < N; ++f) {
                const int c = (q * N + f) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(v[(c + 1) & 7]), "v"(v[(c + 2) & 7]));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(p[(c + 1) & 7]), "v"(p[(c + 2) & 7]));
                else if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
                else if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) & 7]), "v"(u[(c + 2) & 7]));
                else if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(p[(c + 1) & 7]));
                else if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) & 7]));
            }
            if (KIND == 6) {                     // conv_wino43b_kernel's gap: ops [S(q), S(q+1)) of the 44-instruction split of eight values
                constexpr int S[13] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 35, 38, 41, 44};
                const int g = q % 12;
#pragma unroll
                for (int idx = S[g]; idx < S[g + 1]; ++idx) {
                    const int lvl = idx < 4 ? 0 : idx < 12 ? 1 : idx < 20 ? 2 : idx < 24 ? 3 : idx < 32 ? 4 : idx < 40 ? 5 : 6;
                    const int kk = idx - (lvl == 0 ? 0 : lvl == 1 ? 4 : lvl == 2 ? 12 : lvl == 3 ? 20 : lvl == 4 ? 24 : lvl == 5 ? 32 : 40);
                    if (lvl == 0 || lvl == 3 || lvl == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[(lvl / 3 + kk) & 7]) : "v"(v[2 * kk + 1]), "v"(v[2 * kk]), "s"(0x07060302u));
                    else if (lvl == 1 || lvl == 4) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(p[kk].x) : "v"(v[kk]));
                    else asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[kk]) : "v"(v[kk]), "v"(p[kk].x));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + (float)u[i];
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (lane == 0) t[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

static bf16x8* in; static float* out; static unsigned long long* t;
template <int N, int KIND, int WAVES> static double run() {
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<N, KIND, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, t, iters); hipDeviceSynchronize(); }
    static unsigned long long h[256 * 8]; hipMemcpy(h, t, sizeof(unsigned long long) * 256 * WAVES, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256 * WAVES; ++i) s += h[i];
    return s / (256 * WAVES) / (iters * 24.0);
}
template <int KIND, int WAVES> static void row(const char* name) {
    printf("%-14s %d waves/SIMD:", name, WAVES / 4);
    printf(" N=0 %.1f", run<0, KIND, WAVES>()); printf(" | 2 %.1f", run<2, KIND, WAVES>()); printf(" | 4 %.1f", run<4, KIND, WAVES>());
    printf(" | 5 %.1f", run<5, KIND, WAVES>()); printf(" | 6 %.1f", run<6, KIND, WAVES>()); printf(" | 7 %.1f", run<7, KIND, WAVES>());
    printf(" | 8 %.1f", run<8, KIND, WAVES>()); printf(" | 10 %.1f", run<10, KIND, WAVES>()); printf(" | 12 %.1f\n", run<12, KIND, WAVES>());
}
int main() {
    hipMalloc(&in, 64 * 6 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8 * 8);
    hipMemset(in, 0x3c, 64 * 6 * 16);
    printf("s_memtime ticks per MFMA (ratio to the N=0 column = the cost of the fillers)\n");
    row<0, 4>("v_fma_f32"); row<1, 4>("v_pk_fma_f32"); row<2, 4>("v_and_b32"); row<3, 4>("v_perm_b32"); row<4, 4>("v_pk_add_f32"); row<5, 4>("v_sub_f32");
    row<0, 8>("v_fma_f32"); row<1, 8>("v_pk_fma_f32"); row<4, 8>("v_pk_add_f32");
    printf("the 44-instruction split of conv_wino43b_kernel spread over 12 gaps (4,4,4,4,4,4,4,4,3,3,3,3): %.1f ticks per MFMA\n", run<0, 6, 4>());
    return 0;
}
