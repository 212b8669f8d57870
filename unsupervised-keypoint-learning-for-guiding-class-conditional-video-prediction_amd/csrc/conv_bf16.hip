// bf16-MFMA 3x3 stride-1 SAME convolution (fp32 tensors in HBM, bf16 operands, fp32 accumulate) for gfx950:
// BASELINE configs[2] ("bf16"): the 3x3 layers of the translator / VGG19 / encoders / pose decoder
// (reference models/networks/__init__.py:13,22,50..., models/networks/vgg.py:51), forward and data gradient.
//
// Direct convolution, no Winograd: v_mfma_f32_32x32x16_bf16 is 16x the fp32 MFMA rate, so the 2.25x MAC saving of F(2x2,3x3) is
// worth less than its fp32 transform work; at this rate the kernel is bounded by HBM (fp32 activations in and out) and LDS reads.
//   * one workgroup (4 wavefronts) = 16x16 output pixels x 32*NBW output channels; wavefront w owns pixel rows 4w..4w+3
//     (two 32-pixel blocks) x all NBW cout blocks: 2*NBW accumulators of 32x32;
//   * per 16-channel chunk the 18x18-pixel fp32 patch is converted to bf16 while it is staged (32 B per pixel; the 16-B channel
//     half is XOR-swizzled by the row parity: conflict-free ds_read_b128 for all nine taps), and the nine taps' filter fragments
//     (pre-converted, fragment-ordered by conv_bf16_prepare_kernel) are copied to LDS linearly; the A fragment of tap (r,s) is the
//     same LDS patch read at a shifted address, so every staged byte feeds 9 x NBW MFMAs;
//   * LDS double buffered, global loads of chunk k+1 issued before the MFMAs of chunk k, one barrier per chunk, two workgroups per CU.
// Data gradient = the same kernel on weights prepared flipped / transposed.  Master weights, bias, BN statistics stay fp32.
#include "kpx_common.h"
#include "kpx_env.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Bf16Geom {
    const float* x; float* y; const u32x4* Wf; const float* bias;
    int N, H, W, K, ldx, Nn, ldy, act;      // K / Nn: gathered / produced channels (real counts)
    int KC, NB;                             // 16-channel chunks, 32-cout blocks of the prepared weights
    int tiles_y, tiles_x;
};

static __device__ __attribute__((aligned(16))) float bf16_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// Wf[tap 9][kc][nb][h 2][col 32][j 8] (bf16) = w'[tap][c = 16 kc + 8 h + j][n = 32 nb + col]: the B fragment of one (tap, chunk, cout
// block) is 64 consecutive 16-B units, one per lane.  dgrad: w'[r][q][c'][n'] = w[2-r][2-q][n'][c'].
template <bool DGRAD>
__global__ __launch_bounds__(256) void conv_bf16_prepare_kernel(const float* __restrict__ w, int Cin, int Cout, int KC, int NB, unsigned short* __restrict__ Wf) {
    const int K = DGRAD ? Cout : Cin, Nn = DGRAD ? Cin : Cout;
    const size_t per_tap = (size_t)KC * NB * 512;
    const size_t total = 9 * per_tap;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int tap = (int)(idx / per_tap);
        const size_t rem = idx - (size_t)tap * per_tap;
        const int j = (int)(rem & 7), col = (int)((rem >> 3) & 31), h = (int)((rem >> 8) & 1);
        const size_t blk = rem >> 9;
        const int nb = (int)(blk % NB), kc = (int)(blk / NB);
        const int c = 16 * kc + 8 * h + j, n = 32 * nb + col;
        float v = 0.f;
        if (c < K && n < Nn) v = DGRAD ? w[((size_t)(8 - tap) * Cin + n) * Cout + c] : w[((size_t)tap * Cin + c) * Cout + n];
        const __bf16 b = (__bf16)v;
        Wf[idx] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

#define B16_RAWB (18 * 18 * 32)            // bytes of one staged patch: 18 x 18 pixels x 16 bf16

template <int NBW>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(const Bf16Geom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BTILE = 9 * NBW * 1024;                     // bytes of one chunk's filter fragments
    unsigned char* const rawb = smem;                         // [2][B16_RAWB]
    unsigned char* const Bb = smem + 2 * B16_RAWB;            // [2][BTILE]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int ntc = g.NB / NBW;
    const int nti = L % ntc; L /= ntc;
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = L / g.tiles_y;
    const int oy0 = by * 16, ox0 = bx * 16, n0 = nti * 32 * NBW;

    // raw staging: 648 half-pixels (pixel, 8-channel half) over 256 threads -> 3 units per thread (the last one partial)
    const float* rp[3]; int rdst[3]; bool rok[3]; int rvalid[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = t + 256 * i, px = u >> 1, half = u & 1, py = px / 18, pxx = px - py * 18;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
        rok[i] = u < 648 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        rp[i] = rok[i] ? g.x + ((size_t)(n * g.H + iy) * g.W + ix) * g.ldx + half * 8 : bf16_zero16;
        rdst[i] = u < 648 ? px * 32 + ((half ^ (py & 1)) << 4) : -1;
        rvalid[i] = g.K - half * 8;                           // valid channels of this 8-channel half in chunk 0 (decreases by 16 per chunk)
    }
    // filter staging: BTILE / 16 units of 16 B, linear copy
    constexpr int BU = BTILE / 16, BPT = (BU + 255) / 256;
    const u32x4* bsrc = g.Wf + (size_t)nti * NBW * 64;           // + tap * KC*NB*64 + kc * NB*64 + (unit within the NBW blocks)
    const size_t tap_stride = (size_t)g.KC * g.NB * 64, kc_stride = (size_t)g.NB * 64;

    // A-fragment byte offsets inside a patch for the 9 taps x 2 pixel blocks of this wavefront (block pb = rows 2pb, 2pb+1 of the tile)
    int a_off[2][9];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int row = 2 * (2 * wave + p) + (li >> 4) + tap / 3, col = (li & 15) + tap % 3;
            a_off[p][tap] = (row * 18 + col) * 32 + ((lh ^ (row & 1)) << 4);
        }

    f32x16 acc[2][NBW];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][b][r] = 0.f;

    f32x4 rr[3][2];
    u32x4 rb[BPT];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // a half whose channels lie (partly) beyond K: load only what exists (the row's pad bytes are not ours to read as data)
            const int valid = rvalid[i] - 16 * kc;
            const bool lo = rok[i] && valid >= 4, hi = rok[i] && valid >= 8;
            rr[i][0] = *reinterpret_cast<const f32x4*>(lo ? rp[i] + 16 * kc : bf16_zero16);
            rr[i][1] = *reinterpret_cast<const f32x4*>(hi ? rp[i] + 16 * kc + 4 : bf16_zero16);
            if (rok[i] && valid > 0 && valid < 8 && (valid & 3)) {          // K not a multiple of 4: element-wise tail
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < valid ? rp[i][16 * kc + e] : 0.f;
                    if (e < 4) rr[i][0][e] = v; else rr[i][1][e - 4] = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int u = t + 256 * i;                         // unit = tap * (NBW*64) + (block-local unit)
            if (BU % 256 == 0 || u < BU) {
                const int tap = u / (NBW * 64), loc = u - tap * (NBW * 64);
                rb[i] = bsrc[(size_t)tap * tap_stride + (size_t)kc * kc_stride + loc];
            }
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* const rw = rawb + buf * B16_RAWB;
        unsigned char* const bw = Bb + buf * BTILE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (__bf16)rr[i][0][e]; v[4 + e] = (__bf16)rr[i][1][e]; }
            if (rdst[i] >= 0) *reinterpret_cast<bf16x8*>(rw + rdst[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int u = t + 256 * i;
            if (BU % 256 == 0 || u < BU) *reinterpret_cast<u32x4*>(bw + u * 16) = rb[i];
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < g.KC; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < g.KC) load_chunk(kc + 1);
        const unsigned char* const rr_ = rawb + cur * B16_RAWB;
        const unsigned char* const br_ = Bb + cur * BTILE + lane * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            bf16x8 av[2], bv[NBW];
#pragma unroll
            for (int p = 0; p < 2; ++p) av[p] = *reinterpret_cast<const bf16x8*>(rr_ + a_off[p][tap]);
#pragma unroll
            for (int b = 0; b < NBW; ++b) bv[b] = *reinterpret_cast<const bf16x8*>(br_ + (tap * NBW + b) * 1024);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int b = 0; b < NBW; ++b)
                    acc[p][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[p], bv[b], acc[p][b], 0, 0, 0);
        }
        if (kc + 1 < g.KC) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // epilogue straight from the accumulators: lane = cout (128 contiguous bytes per pixel and cout block), register = pixel
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int oc = n0 + 32 * b + li;
        const bool ocv = oc < g.Nn;
        const float bvv = (g.bias && ocv) ? g.bias[oc] : 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = (r & 3) + 8 * (r >> 2) + 4 * lh;          // 0..31 inside the block: row pix>>4, column pix&15
                const int oy = oy0 + 2 * (2 * wave + p) + (pix >> 4), ox = ox0 + (pix & 15);
                float v = acc[p][b][r] + bvv;
                if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
                else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (ocv) g.y[((size_t)(n * g.H + oy) * g.W + ox) * g.ldy + oc] = v;
            }
    }
}


// ---- wide variant: one workgroup (8 wavefronts) = 16 x 32 output pixels x 32*NBW output channels (NBW = 2 or 4).  The narrow kernel moves
// 38.7 KB (patch + filter fragments) per chunk for 256 pixels x 64 couts and, with one chunk of MFMAs (2 304 cycles for both resident
// workgroups) to cover an HBM round trip, is latency-bound at 9.3 k cycles per chunk; here a chunk is 76 KB for four times the work and
// 4 608 cycles of MFMAs.  Wavefront w owns pixel rows 2w, 2w+1 (two 32-pixel blocks of 2 rows x 16 columns) x all NBW cout blocks.
#define B16W_RAWB (18 * 34 * 32)           // bytes of one staged patch: 18 x 34 pixels x 16 bf16

template <int NBW>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16_wide_kernel(const Bf16Geom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BTILE = 9 * NBW * 1024;
    unsigned char* const rawb = smem;                         // [2][B16W_RAWB]
    unsigned char* const Bb = smem + 2 * B16W_RAWB;           // [2][BTILE]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int ntc = g.NB / NBW;
    const int nti = L % ntc; L /= ntc;
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = L / g.tiles_y;
    const int oy0 = by * 16, ox0 = bx * 32, n0 = nti * 32 * NBW;

    // raw staging: 1224 half-pixels (pixel, 8-channel half) over 512 threads -> 3 units per thread (the last one partial)
    const float* rp[3]; int rdst[3]; bool rok[3]; int rvalid[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = t + 512 * i, px = u >> 1, half = u & 1, py = px / 34, pxx = px - py * 34;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
        rok[i] = u < 1224 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        rp[i] = rok[i] ? g.x + ((size_t)(n * g.H + iy) * g.W + ix) * g.ldx + half * 8 : bf16_zero16;
        rdst[i] = u < 1224 ? px * 32 + ((half ^ (py & 1)) << 4) : -1;
        rvalid[i] = g.K - half * 8;
    }
    constexpr int BU = BTILE / 16, BPT = (BU + 511) / 512;
    const u32x4* bsrc = g.Wf + (size_t)nti * NBW * 64;
    const size_t tap_stride = (size_t)g.KC * g.NB * 64, kc_stride = (size_t)g.NB * 64;

    int a_off[2][9];                                          // block p = columns 16p .. 16p+15 of rows 2 wave, 2 wave + 1
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int row = 2 * wave + (li >> 4) + tap / 3, col = 16 * p + (li & 15) + tap % 3;
            a_off[p][tap] = (row * 34 + col) * 32 + ((lh ^ (row & 1)) << 4);
        }

    f32x16 acc[2][NBW];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][b][r] = 0.f;

    f32x4 rr[3][2];
    u32x4 rb[BPT];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int valid = rvalid[i] - 16 * kc;
            const bool lo = rok[i] && valid >= 4, hi = rok[i] && valid >= 8;
            rr[i][0] = *reinterpret_cast<const f32x4*>(lo ? rp[i] + 16 * kc : bf16_zero16);
            rr[i][1] = *reinterpret_cast<const f32x4*>(hi ? rp[i] + 16 * kc + 4 : bf16_zero16);
            if (rok[i] && valid > 0 && valid < 8 && (valid & 3)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < valid ? rp[i][16 * kc + e] : 0.f;
                    if (e < 4) rr[i][0][e] = v; else rr[i][1][e - 4] = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int u = t + 512 * i;
            if (BU % 512 == 0 || u < BU) {
                const int tap = u / (NBW * 64), loc = u - tap * (NBW * 64);
                rb[i] = bsrc[(size_t)tap * tap_stride + (size_t)kc * kc_stride + loc];
            }
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* const rw = rawb + buf * B16W_RAWB;
        unsigned char* const bw = Bb + buf * BTILE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (__bf16)rr[i][0][e]; v[4 + e] = (__bf16)rr[i][1][e]; }
            if (rdst[i] >= 0) *reinterpret_cast<bf16x8*>(rw + rdst[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int u = t + 512 * i;
            if (BU % 512 == 0 || u < BU) *reinterpret_cast<u32x4*>(bw + u * 16) = rb[i];
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < g.KC; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < g.KC) load_chunk(kc + 1);
        const unsigned char* const rr_ = rawb + cur * B16W_RAWB;
        const unsigned char* const br_ = Bb + cur * BTILE + lane * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            bf16x8 av[2], bv[NBW];
#pragma unroll
            for (int p = 0; p < 2; ++p) av[p] = *reinterpret_cast<const bf16x8*>(rr_ + a_off[p][tap]);
#pragma unroll
            for (int b = 0; b < NBW; ++b) bv[b] = *reinterpret_cast<const bf16x8*>(br_ + (tap * NBW + b) * 1024);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int b = 0; b < NBW; ++b)
                    acc[p][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[p], bv[b], acc[p][b], 0, 0, 0);
        }
        if (kc + 1 < g.KC) store_chunk(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int oc = n0 + 32 * b + li;
        const bool ocv = oc < g.Nn;
        const float bvv = (g.bias && ocv) ? g.bias[oc] : 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = (r & 3) + 8 * (r >> 2) + 4 * lh;          // 0..31 inside the block: row pix>>4, column pix&15
                const int oy = oy0 + 2 * wave + (pix >> 4), ox = ox0 + 16 * p + (pix & 15);
                float v = acc[p][b][r] + bvv;
                if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
                else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (ocv) g.y[((size_t)(n * g.H + oy) * g.W + ox) * g.ldy + oc] = v;
            }
    }
}

static std::atomic<unsigned long long> bf16_attr_mask{0};
static inline int bf16_wide_lds_bytes(int nbw) { return 2 * B16W_RAWB + 2 * 9 * nbw * 1024; }
static inline int bf16_lds_bytes(int nbw) { return 2 * B16_RAWB + 2 * 9 * nbw * 1024; }

extern "C" size_t kpx_conv3x3_bf16_weights_bytes(int Cin, int Cout) {
    // large enough for either direction: chunks of 16 over max(Cin, Cout), blocks of 32 over max(Cin, Cout)
    const int m = Cin > Cout ? Cin : Cout;
    return (size_t)9 * ((m + 15) / 16) * ((m + 31) / 32) * 512 * 2;
}

extern "C" int kpx_conv3x3_bf16_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr) {
    if (N <= 0 || K <= 0 || Nn <= 0 || H % 16 || W % 16) return 0;
    if (ldin % 4 || (((uintptr_t)in_ptr) & 15)) return 0;
    return 1;
}

extern "C" int kpx_conv3x3_bf16_prepare_f32(const float* w_hwio, int Cin, int Cout, int dgrad, void* Wf, void* stream) {
    if (!w_hwio || !Wf || Cin <= 0 || Cout <= 0) return KPX_EINVAL;
    const int K = dgrad ? Cout : Cin, Nn = dgrad ? Cin : Cout;
    const int KC = (K + 15) / 16, NB = (Nn + 31) / 32;
    const size_t total = (size_t)9 * KC * NB * 512;
    size_t nb = (total + 255) / 256; if (nb > 2048) nb = 2048;
    if (dgrad) hipLaunchKernelGGL(conv_bf16_prepare_kernel<true>, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), w_hwio, Cin, Cout, KC, NB, (unsigned short*)Wf);
    else hipLaunchKernelGGL(conv_bf16_prepare_kernel<false>, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), w_hwio, Cin, Cout, KC, NB, (unsigned short*)Wf);
    return kpx_launch_status();
}

// in: [N,H,W,K] (pixel stride ldin), Wf: prepared for (K gathered, Nn produced) channels, out: [N,H,W,Nn] (pixel stride ldout)
extern "C" int kpx_conv3x3_bf16_f32(const float* in, int N, int H, int W, int K, int ldin, const void* Wf, const float* bias,
                                    float* out, int Nn, int ldout, int act, void* stream) {
    if (!in || !Wf || !out || !kpx_conv3x3_bf16_eligible(N, H, W, K, Nn, ldin, in) || ldin < K || ldout < Nn) return KPX_EINVAL;
    if (kpx_first_use_on_device(&bf16_attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bf16_lds_bytes(1));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bf16_lds_bytes(2));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16_wide_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bf16_wide_lds_bytes(2));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16_wide_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, bf16_wide_lds_bytes(4));
        if (e != hipSuccess) return -(int)e;
    }
    Bf16Geom g{};
    g.x = in; g.y = out; g.Wf = (const u32x4*)Wf; g.bias = bias;
    g.N = N; g.H = H; g.W = W; g.K = K; g.ldx = ldin; g.Nn = Nn; g.ldy = ldout; g.act = act;
    g.KC = (K + 15) / 16; g.NB = (Nn + 31) / 32;
    g.tiles_y = H / 16; g.tiles_x = W / 16;
    const size_t tiles = (size_t)N * g.tiles_y * g.tiles_x;
    hipStream_t s = kpx_stream(stream);
    // wide tiles (16 x 32 pixels, 8 wavefronts) when the launch still fills the chip with them
    const int wide_mode = kpx_env()->bf16_wide;
    if (wide_mode && W % 32 == 0 && g.NB % 2 == 0) {
        const int nbw = (g.NB % 4 == 0 && wide_mode != 2) ? 4 : 2;
        const size_t wgs = (size_t)N * g.tiles_y * (W / 32) * (g.NB / nbw);
        if (wgs >= 256 || (nbw == 4 && (size_t)N * g.tiles_y * (W / 32) * (g.NB / 2) >= 256)) {
            const int use = wgs >= 256 ? nbw : 2;
            g.tiles_x = W / 32;
            const unsigned blocks = (unsigned)((size_t)N * g.tiles_y * g.tiles_x * (g.NB / use));
            if (use == 4) hipLaunchKernelGGL(conv3x3_bf16_wide_kernel<4>, dim3(blocks), dim3(512), bf16_wide_lds_bytes(4), s, g);
            else hipLaunchKernelGGL(conv3x3_bf16_wide_kernel<2>, dim3(blocks), dim3(512), bf16_wide_lds_bytes(2), s, g);
            return kpx_launch_status();
        }
    }
    if (g.NB % 2 == 0) hipLaunchKernelGGL(conv3x3_bf16_kernel<2>, dim3((unsigned)(tiles * (g.NB / 2))), dim3(256), bf16_lds_bytes(2), s, g);
    else hipLaunchKernelGGL(conv3x3_bf16_kernel<1>, dim3((unsigned)(tiles * g.NB)), dim3(256), bf16_lds_bytes(1), s, g);
    return kpx_launch_status();
}
