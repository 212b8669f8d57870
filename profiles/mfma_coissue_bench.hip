// Micro-benchmark: what can a SIMD issue beside v_mfma_f32_32x32x2_f32 (fp32 MFMA) on MI355X / gfx950?
//
// Evidence behind DESIGN.md's statement that the Winograd kernel's transform / staging work does not overlap its fp32 MFMAs.
//   build:  hipcc -O3 --offload-arch=gfx950 profiles/mfma_coissue_bench.hip -o profiles/mfma_coissue_bench
//   run:    profiles/mfma_coissue_bench            (prints one table; numbers are shader cycles from s_memtime)
//
// Experiment A ("partner"): one workgroup of 512 threads per CU = two wavefronts per SIMD (w and w+4 share a SIMD).  Wavefronts 0-3 issue
//   NM back-to-back MFMAs; wavefronts 4-7 issue NX instructions of one kind (VALU fma, packed fma, ds_read_b128, ds_write_b128,
//   L2-hit global_load_dwordx4).  Each role is timed alone and together.  If the two streams overlapped perfectly, "together"
//   would equal max(alone); if the SIMD serialises them it equals the sum.
// Experiment B ("in-wave"): 256 threads = one wavefront per SIMD; the loop body is 1 MFMA + k filler instructions (v_fma_f32, or
//   ds_write_b128 -- all four SIMDs store, so that row is also bounded by the CU's LDS store rate); cycles per iteration show how
//   many fillers hide under one MFMA.
// Both are repeated with v_mfma_f32_32x32x16_bf16 as a control (the microarchitecture guide documents co-issue for that one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { X_NONE = 0, X_VALU, X_PK, X_DSR, X_DSW, X_GLD, X_KINDS };
static const char* xname[] = {"(nothing)", "v_fma_f32", "v_pk_fma_f32", "ds_read_b128", "ds_write_b128", "global_load_dwordx4 (L2 hit)"};

template <bool BF16>
__device__ __forceinline__ void mfma_stream(int n, f32x16& acc0, f32x16& acc1, float a, float b) {
    if (BF16) {
        bf16x8 av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = (__bf16)a; bv[i] = (__bf16)b; }
        for (int i = 0; i < n; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc1, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < n; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        }
    }
}

// out[block][role] = cycles ; roles: 0 = MFMA wavefronts (wave 0), 1 = X wavefronts (wave 4)
template <bool BF16, int XK>
__global__ __launch_bounds__(512) void partner_kernel(int nm, int nx, const f32x4* __restrict__ gsrc, float* __restrict__ sink, unsigned long long* __restrict__ out) {
    __shared__ f32x4 lds[1024];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    lds[t] = f32x4{1.f, 2.f, 3.f, 4.f}; lds[t + 512] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (nm > 0) mfma_stream<BF16>(nm, acc0, acc1, 1.0f + lane, 0.5f);
    } else {
        f32x4 v0 = {1.f, 1.f, 1.f, 1.f}, v1 = v0, v2 = v0, v3 = v0;
        unsigned s = 0;
        const f32x4* gp = gsrc + lane;
        for (int i = 0; i < nx; i += 4) {
            if (XK == X_VALU) {
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3"
                             : "+v"(v0[0]), "+v"(v1[0]), "+v"(v2[0]), "+v"(v3[0]));
            } else if (XK == X_PK) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 p0 = {v0[0], v0[1]}, p1 = {v1[0], v1[1]}, p2 = {v2[0], v2[1]}, p3 = {v3[0], v3[1]};
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                v0[0] = p0[0]; v1[0] = p1[0]; v2[0] = p2[0]; v3[0] = p3[0];
            } else if (XK == X_DSR) {
                v0 += lds[(t + i) & 1023]; v1 += lds[(t + i + 64) & 1023]; v2 += lds[(t + i + 128) & 1023]; v3 += lds[(t + i + 192) & 1023];
            } else if (XK == X_DSW) {
                lds[(t + i) & 1023] = v0; lds[(t + i + 64) & 1023] = v1; lds[(t + i + 128) & 1023] = v2; lds[(t + i + 192) & 1023] = v3;
            } else if (XK == X_GLD) {
                v0 += gp[(i * 16) & 4095]; v1 += gp[(i * 16 + 64) & 4095]; v2 += gp[(i * 16 + 128) & 4095]; v3 += gp[(i * 16 + 192) & 4095];
            }
        }
        acc0[0] = v0[0] + v1[1] + v2[2] + v3[3] + (float)s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (wave == 0 || wave == 4)) out[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    if (r == 12345.678f) sink[t] = r;            // keep everything live
}

// in-wave: 1 MFMA + K fillers per iteration
template <bool BF16, int XK, int K>
__global__ __launch_bounds__(256) void inwave_kernel(int iters, float* __restrict__ sink, unsigned long long* __restrict__ out) {
    __shared__ f32x4 lds[1024];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 1024; i += 256) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + i;
    f32x4 lv = {0.f, 0.f, 0.f, 0.f};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)1.0f; bv[i] = (__bf16)0.5f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (BF16) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc0, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f + lane, 0.5f, acc0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (XK == X_VALU) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k & 15]));
            else if (XK == X_DSR) lv += lds[(t + 64 * k + it) & 1023];
            else if (XK == X_DSW) lds[(t + 64 * k + it) & 1023] = lv;
        }
        if (BF16) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc1, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f + lane, 0.5f, acc1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (XK == X_VALU) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k & 15]));
            else if (XK == X_DSR) lv += lds[(t + 64 * k + it + 512) & 1023];
            else if (XK == X_DSW) lds[(t + 64 * k + it + 512) & 1023] = lv;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && t < 64) out[blockIdx.x] = t1 - t0;
    float r = lv[0] + lv[1];
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + v[i];
    if (r == 12345.678f) sink[t] = r;
}

static unsigned long long median(std::vector<unsigned long long> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool BF16, int XK>
static void partner_row(int nm, int nx, const f32x4* gsrc, float* sink, unsigned long long* dout) {
    const int blocks = 256;
    std::vector<unsigned long long> h(blocks * 2);
    unsigned long long res[3][2];
    const int cfg[3][2] = {{nm, 0}, {0, nx}, {nm, nx}};
    for (int c = 0; c < 3; ++c) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((partner_kernel<BF16, XK>), dim3(blocks), dim3(512), 0, 0, cfg[c][0], cfg[c][1], gsrc, sink, dout);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), dout, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<unsigned long long> a, b;
        for (int i = 0; i < blocks; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
        res[c][0] = median(a); res[c][1] = median(b);
    }
    printf("  %-30s | MFMA alone %7llu | X alone %7llu | together: MFMA waves %7llu, X waves %7llu | sum-of-alone %7llu | overlap %5.1f %%\n",
           xname[XK], res[0][0], res[1][1], res[2][0], res[2][1], res[0][0] + res[1][1],
           100.0 * (double)(res[0][0] + res[1][1] - std::max(res[2][0], res[2][1])) / (double)std::min(res[0][0], res[1][1]));
}

template <bool BF16, int XK, int K>
static void inwave_row(float* sink, unsigned long long* dout) {
    const int blocks = 256, iters = 512;
    std::vector<unsigned long long> h(blocks);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((inwave_kernel<BF16, XK, K>), dim3(blocks), dim3(256), 0, 0, iters, sink, dout);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), dout, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    printf(" %6.1f", (double)median(h) / (2.0 * iters));
}

template <bool BF16>
static void run_all(const f32x4* gsrc, float* sink, unsigned long long* dout) {
    const int nm = 256;
    printf("A. partner wavefronts on one SIMD, %d x %s in waves 0-3 (cycles):\n", nm, BF16 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32");
    partner_row<BF16, X_VALU>(nm, 2048, gsrc, sink, dout);
    partner_row<BF16, X_PK>(nm, 2048, gsrc, sink, dout);
    partner_row<BF16, X_DSR>(nm, 1024, gsrc, sink, dout);
    partner_row<BF16, X_DSW>(nm, 512, gsrc, sink, dout);
    partner_row<BF16, X_GLD>(nm, 256, gsrc, sink, dout);
    printf("B. one wavefront per SIMD, cycles per (1 MFMA + K fillers), K = 0 1 2 4 8 16:\n");
    printf("  v_fma_f32     :"); inwave_row<BF16, X_VALU, 0>(sink, dout); inwave_row<BF16, X_VALU, 1>(sink, dout); inwave_row<BF16, X_VALU, 2>(sink, dout);
    inwave_row<BF16, X_VALU, 4>(sink, dout); inwave_row<BF16, X_VALU, 8>(sink, dout); inwave_row<BF16, X_VALU, 16>(sink, dout); printf("\n");
    printf("  ds_write_b128 :"); inwave_row<BF16, X_DSW, 0>(sink, dout); inwave_row<BF16, X_DSW, 1>(sink, dout); inwave_row<BF16, X_DSW, 2>(sink, dout);
    inwave_row<BF16, X_DSW, 4>(sink, dout); inwave_row<BF16, X_DSW, 8>(sink, dout); inwave_row<BF16, X_DSW, 16>(sink, dout); printf("\n");
}

int main() {
    f32x4* gsrc; float* sink; unsigned long long* dout;
    CHECK(hipMalloc(&gsrc, 8192 * sizeof(f32x4)));
    CHECK(hipMemset(gsrc, 0, 8192 * sizeof(f32x4)));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&dout, 4096 * sizeof(unsigned long long)));
    run_all<false>(gsrc, sink, dout);
    run_all<true>(gsrc, sink, dout);
    return 0;
}
