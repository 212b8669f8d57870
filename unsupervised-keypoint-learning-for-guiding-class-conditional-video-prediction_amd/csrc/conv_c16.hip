// 3x3 stride-1 SAME convolution that PRODUCES exactly 16 channels (forward of the key-point detector's last decoder block, 64 -> 16 and
// 16 -> 16 at full resolution, and the data gradient of the 16 -> 16 layer; reference models/networks/__init__.py:50-54), fp32.
//
// Every other MFMA kernel of this library multiplies 32-wide cout blocks, so these layers ran with half of each tile padding (42-75
// TFLOP/s).  Here the block is v_mfma_f32_16x16x4_f32: M = 16 pixels of one image row, N = the 16 couts, K = 4 channels -- no padding.
// One workgroup (4 wavefronts) = 16 x 16 output pixels; wavefront w owns rows 4w .. 4w+3 (four accumulators of 16x16 = 16 VGPRs).
// Per 16-channel chunk the 18 x 18-pixel fp32 patch is staged in LDS (80 B per pixel: conflict-free ds_read_b128 at a 16-lane pixel
// stride); lane (pixel column l % 16, channel group l / 16) reads its four channels 4g .. 4g+3 with ONE ds_read_b128 per (block, tap),
// which feeds the four k-steps of that tap: k-step s of lane group g is channel 4g + s, and the filter fragment is indexed to match
// (fragment-ordered through LDS: one ds_read_b128 per tap).  Double-buffered patch and filter, one barrier per chunk.
// Epilogue: bias + activation, 64 contiguous bytes per pixel; optional batch-norm sums per 16x16-pixel tile in the slab format of
// kpx_conv3x3_wino_stats_f32 (bitwise reproducible: fixed shuffle / LDS order).
#include "kpx_common.h"
#include "kpx_env.h"
#include <stdlib.h>

struct C16Geom {
    const float* x; const float* w; const float* bias; float* y; float* stats;
    int N, H, W, K, ldx, ldy, act, dgrad, wci, wco;        // K gathered channels; the filter is [3][3][wci][wco] (HWIO)
    int tiles_y, tiles_x;
};

#define C16_PS 20                              // floats per staged pixel (16 channels + 4 pad)
#define C16_PATCH (18 * 18 * C16_PS)

template <bool STATS>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const C16Geom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*patch)[C16_PATCH] = reinterpret_cast<float (*)[C16_PATCH]>(smem);                    // [2][C16_PATCH]
    float (*filt)[9 * 256] = reinterpret_cast<float (*)[9 * 256]>(smem + 2 * C16_PATCH);          // [2][tap 9][lane 64][s 4]
    float (*red)[2][16] = reinterpret_cast<float (*)[2][16]>(smem + 2 * C16_PATCH + 2 * 9 * 256); // [4][2][16]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, grp = lane >> 4;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = L / g.tiles_y;
    const int oy0 = by * 16, ox0 = bx * 16;
    const int nchunks = g.K >> 4;

    // staging: 324 pixels x 4 channel quads = 1296 16-B units over 256 threads (6 rounds, the last one partial)
    const float* sp[6]; int sd[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int u = t + 256 * i, px = u >> 2, q = u & 3, py = px / 18, pxx = px - py * 18;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
        const bool ok = u < 1296 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        sp[i] = ok ? g.x + ((size_t)(n * g.H + iy) * g.W + ix) * g.ldx + 4 * q : nullptr;
        sd[i] = u < 1296 ? px * C16_PS + 4 * q : -1;
    }
    f32x4 sr[6];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 6; ++i) sr[i] = sp[i] ? *reinterpret_cast<const f32x4*>(sp[i] + 16 * kc) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 6; ++i) if (sd[i] >= 0) *reinterpret_cast<f32x4*>(&patch[buf][sd[i]]) = sr[i];
    };
    // filter fragment of (tap, k-step s): lane supplies B[k = grp][n = col] = w'[tap][16 kc + 4 grp + s][col]
    //   forward:  w'[tap][c][n] = w[tap][c][n]                      gradient: w'[tap][c][n] = w[8 - tap][n][c]
    // The 2304 values of a chunk go through LDS in fragment order ([tap][lane][s]: one ds_read_b128 per tap and lane), loaded by the
    // workgroup together (9 per thread) instead of 36 strided loads per lane.
    float fr[9];
    auto load_filter = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int e = t + 256 * i, tap = e >> 8, fl = (e >> 2) & 63, fs = e & 3;
            const int c = 16 * kc + 4 * (fl >> 4) + fs, nn = fl & 15;
            fr[i] = g.dgrad ? g.w[((size_t)(8 - tap) * g.wci + nn) * g.wco + c] : g.w[((size_t)tap * g.wci + c) * g.wco + nn];
        }
    };
    auto store_filter = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 9; ++i) filt[buf][t + 256 * i] = fr[i];
    };
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    load_filter(0);
    store_chunk(0);
    store_filter(0);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nchunks) { load_chunk(kc + 1); load_filter(kc + 1); }
        f32x4 bw[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) bw[tap] = *reinterpret_cast<const f32x4*>(&filt[cur][(tap * 64 + lane) * 4]);
        const float* pb = &patch[cur][(col) * C16_PS + 4 * grp];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = 4 * wave + b;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(pb + ((row + tap / 3) * 18 + tap % 3) * C16_PS);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[tap][s], acc[b], 0, 0, 0);
            }
        }
        if (kc + 1 < nchunks) { store_chunk(cur ^ 1); store_filter(cur ^ 1); }
        __syncthreads();
    }

    // D[m][n]: lane holds m = 4 grp + i (pixel column), n = col (cout)
    const float bv = g.bias ? g.bias[col] : 0.f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int oy = oy0 + 4 * wave + b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[b][i] + bv;
            if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
            else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
            if (STATS) { s0 += v; s1 += v * v; }
            g.y[((size_t)(n * g.H + oy) * g.W + ox0 + 4 * grp + i) * g.ldy + col] = v;
        }
    }
    if (STATS) {
        s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16);
        s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
        if (lane < 16) { red[wave][0][lane] = s0; red[wave][1][lane] = s1; }
        __syncthreads();
        if (t < 32) {
            const int st = t >> 4, c = t & 15;
            const float r = (red[0][st][c] + red[1][st][c]) + (red[2][st][c] + red[3][st][c]);
            const size_t tile = ((size_t)n * g.tiles_y + by) * g.tiles_x + bx;
            g.stats[(tile * 2 + st) * 16 + c] = r;
        }
    }
}

static std::atomic<unsigned long long> c16_attr_mask{0};

// in [N,H,W,K] (pixel stride ldin) -> out [N,H,W,16] (pixel stride ldout).  forward: w = [3][3][K][16]; dgrad: w = [3][3][16][K] (the layer's
// own HWIO filter, produced channels = its Cin = 16).  tile_stats (or NULL): [N * H/16 * W/16][2][16] sums of the output.
extern "C" int kpx_conv3x3_c16_eligible(int N, int H, int W, int K, int Nn, int ldin, int ldout, const void* in_ptr) {
    return N > 0 && Nn == 16 && K >= 16 && K % 16 == 0 && H % 16 == 0 && W % 16 == 0 && ldin % 4 == 0 && ldout >= 16 && (((uintptr_t)in_ptr) & 15) == 0;
}
extern "C" int kpx_conv3x3_c16_f32(const float* in, int N, int H, int W, int K, int ldin, const float* w_hwio, int dgrad, const float* bias,
                                   float* out, int ldout, int act, float* tile_stats, void* stream) {
    if (!in || !w_hwio || !out || ldin < K || act < 0 || act > 2 || !kpx_conv3x3_c16_eligible(N, H, W, K, 16, ldin, ldout, in)) return KPX_EINVAL;
    C16Geom g{};
    g.x = in; g.w = w_hwio; g.bias = bias; g.y = out; g.stats = tile_stats;
    g.N = N; g.H = H; g.W = W; g.K = K; g.ldx = ldin; g.ldy = ldout; g.act = act; g.dgrad = dgrad ? 1 : 0;
    g.wci = dgrad ? 16 : K; g.wco = dgrad ? K : 16;
    g.tiles_y = H / 16; g.tiles_x = W / 16;
    const unsigned blocks = (unsigned)((size_t)N * g.tiles_y * g.tiles_x);
    constexpr int lds = (2 * C16_PATCH + 2 * 9 * 256 + 4 * 2 * 16) * 4;        // 70.8 KB: dynamic
    if (kpx_first_use_on_device(&c16_attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return -(int)e;
    }
    if (tile_stats) hipLaunchKernelGGL(conv3x3_c16_kernel<true>, dim3(blocks), dim3(256), lds, kpx_stream(stream), g);
    else hipLaunchKernelGGL(conv3x3_c16_kernel<false>, dim3(blocks), dim3(256), lds, kpx_stream(stream), g);
    return kpx_launch_status();
}
