// First-layer convolutions (image input, Cin <= 4, stride 1): the encoders' 7x7x3 -> 32 (reference models/networks/__init__.py:13)
// and VGG19 conv1_1 3x3x3 -> 64 (models/networks/vgg.py:51).  The implicit-GEMM kernel has to gather these with scalar, unaligned
// loads (a pixel is 12 bytes); here one workgroup stages the whole input patch of a 16x16 output tile (22x22x3 floats for 7x7) and
// the complete filter in LDS once and runs the K = KH*KW*Cin contraction out of LDS with v_mfma_f32_32x32x2_f32:
// the KW*Cin floats of a filter row are contiguous in NHWC, so k runs over (row, kk < KW*Cin) and the A operand of pixel (y, x) is
// patch[(y + r) * PWC + x * Cin + kk].  4 wavefronts x 2 M-tiles of 32 pixels (two output rows) x Cout/32 N-tiles.
#include "kpx_common.h"
#include "kpx_env.h"

struct RgbGeom {
    const float* x; const float* w; const float* bias; float* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, ldy;
    int KH, KW, pad_t, pad_l, act, stride;
    int y16;                         // bf16 configuration: y is a bf16 tensor (ldy in elements)
    int tiles_y, tiles_x, total_tiles, tpb;
    int PH, PWC, KWC, KWCp;          // patch rows, floats per patch row (+ slack), floats per filter row, padded to even
};

// Persistent workgroups: the filter is staged ONCE per workgroup, which then walks a contiguous range of 16x16 tiles.  On gfx950 nothing
// overlaps an fp32 MFMA on its SIMD, and a tile is only 60-154 MFMAs per wavefront, so the instructions around them decide the
// speed: the first version spent 1 700-2 000 VALU per wavefront and tile (per-element index divisions while staging the filter and the patch,
// 64 stores with their own bounds / address arithmetic) against 3 840-9 900 MFMA cycles -- PMC: matrix pipe busy 18-40 %.  Here the patch is
// staged by rows (one division-free row per wavefront and trip), and the epilogue works from one base pointer with a tile-uniform fast path.
// VEC (Cout a multiple of 8, 16-B aligned pixel rows): the MFMA roles are swapped -- A = filter value (rows = output channels), B = pixel value
// (columns = pixels) -- so a lane holds 16 output channels of ONE pixel in groups of four consecutive ones and stores 16 bytes at a time (four
// stores per 32 x 32 block instead of sixteen 4-byte ones; bf16 output: two, after a half-wave exchange as in conv_bf16s.hip).
template <int NT, bool VEC>
__global__ __launch_bounds__(256) void conv_rgb_kernel(const RgbGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                   // [PH][PWC] (+4 zero floats)
    float* Ws = smem + ((g.PH * g.PWC + 4 + 3) & ~3);      // [KH][KWCp][32*NT]
    constexpr int NC = 32 * NT;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;

    // ---- the filter, once: thread t owns column nn = t % NC and walks the (row, kk) pairs with a stride of 256 / NC (no divisions per element)
    {
        const int nn = t & (NC - 1), j0 = t / NC, jstep = 256 / NC;
        int r = 0, kk = j0;
        while (kk >= g.KWCp) { kk -= g.KWCp; ++r; }
        for (; r < g.KH; ) {
            Ws[(r * g.KWCp + kk) * NC + nn] = (kk < g.KWC && nn < g.Cout) ? g.w[((size_t)r * g.KWC + kk) * g.Cout + nn] : 0.f;
            kk += jstep;
            while (kk >= g.KWCp) { kk -= g.KWCp; ++r; }
        }
        if (t < 4) patch[g.PH * g.PWC + t] = 0.f;           // the slack the last pixel's padded k reads
    }
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (g.bias && nt * 32 + li < g.Cout) ? g.bias[nt * 32 + li] : 0.f;
    f32x4 bvv[VEC ? NT : 1][4];                           // VEC: bias of channels nt * 32 + 8 gq + 4 lh .. + 3
    if (VEC) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ch = nt * 32 + 8 * gq + 4 * lh;
                bvv[nt][gq] = (g.bias && ch < g.Cout) ? *reinterpret_cast<const f32x4*>(g.bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    const int rowlen = g.Wi * g.Cin;
    // lane's pixel inside M-tile mt: row 2*mt + li/16, column li%16
    const int a0 = g.stride * (2 * (2 * wave) + (li >> 4)) * g.PWC + g.stride * (li & 15) * g.Cin + lh;
    const int a1 = a0 + 2 * g.stride * g.PWC;
    const int b0 = lh * NC + li;
    const int ks = g.KWCp >> 1;
    const float lo = g.act == KPX_ACT_RELU ? 0.f : -__builtin_inff();
    const float slope = g.act == KPX_ACT_LRELU ? 0.01f : 1.f;

    for (int tile = blockIdx.x * g.tpb, tend = min(tile + g.tpb, g.total_tiles); tile < tend; ++tile) {
        int L = tile;
        const int bx = L % g.tiles_x; L /= g.tiles_x;
        const int by = L % g.tiles_y;
        const int n = L / g.tiles_y;
        const int oy0 = by * 16, ox0 = bx * 16;
        // ---- stage the patch (zero outside the image): wavefront w takes patch rows w, w + 4, ..
        const int iy0 = oy0 * g.stride - g.pad_t, ixc0 = (ox0 * g.stride - g.pad_l) * g.Cin;
        __syncthreads();                                   // the previous tile's reads (and, first, the filter) are done
        for (int pr = wave; pr < g.PH; pr += 4) {
            const int iy = iy0 + pr;
            const bool rowok = (unsigned)iy < (unsigned)g.Hi;
            const float* src = g.x + ((size_t)n * g.Hi + (rowok ? iy : 0)) * rowlen;
            for (int pc = lane; pc < g.PWC; pc += 64) {
                const int ic = ixc0 + pc;
                patch[pr * g.PWC + pc] = (rowok && (unsigned)ic < (unsigned)rowlen) ? src[ic] : 0.f;
            }
        }
        __syncthreads();

        f32x16 acc[2][NT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int r = 0; r < g.KH; ++r) {
            const float* pa = patch + r * g.PWC;
            const float* pb = Ws + r * g.KWCp * NC + b0;
            for (int s = 0; s < ks; ++s) {
                const float va0 = pa[a0 + 2 * s], va1 = pa[a1 + 2 * s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float vb = pb[2 * s * NC + nt * 32];
                    if (VEC) {
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, va0, acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, va1, acc[1][nt], 0, 0, 0);
                    } else {
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, vb, acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, vb, acc[1][nt], 0, 0, 0);
                    }
                }
            }
        }
        // ---- epilogue: accumulator register r of M-tile mt is output row 2 * (2 * wave + mt) + (r >> 3), column ((r >> 2) & 1) * 8 + 4 * lh + (r & 3)
        const bool full = oy0 + 16 <= g.Ho && ox0 + 16 <= g.Wo;            // tile-uniform
        if (VEC) {
            // accumulator register e of (mt, nt): pixel = lane li of M-tile mt (row 4 wave + 2 mt + li / 16, column li % 16), channel
            // nt * 32 + 8 (e >> 2) + 4 lh + (e & 3)
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int oy = oy0 + 4 * wave + 2 * mt + (li >> 4), ox = ox0 + (li & 15);
                const bool ok = full || (oy < g.Ho && ox < g.Wo);
                const size_t base = (((size_t)n * g.Ho + oy) * g.Wo + ox) * g.ldy;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float a = acc[mt][nt][e] + bvv[nt][e >> 2][e & 3];
                        if (g.act == KPX_ACT_TANH) a = tanhf(a);
                        else { a = fmaxf(a, lo); a = a > 0.f ? a : a * slope; }
                        v[e] = a;
                    }
                    if (g.y16) {
                        unsigned short* const yo = reinterpret_cast<unsigned short*>(g.y) + base;
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                            auto pk = [](float p0, float p1) { const bf16x2_t h = {(__bf16)p0, (__bf16)p1}; return __builtin_bit_cast(unsigned, h); };
                            const unsigned a0 = pk(v[4 * k], v[4 * k + 1]), a1 = pk(v[4 * k + 2], v[4 * k + 3]);
                            const unsigned b0 = pk(v[4 * k + 4], v[4 * k + 5]), b1 = pk(v[4 * k + 6], v[4 * k + 7]);
                            auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                            auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                            const int ch = nt * 32 + 8 * (k + lh);
                            if (ok && ch < g.Cout) *reinterpret_cast<u32x4_t*>(yo + ch) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
                        }
                    } else {
                        float* const yo = g.y + base;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int ch = nt * 32 + 8 * gq + 4 * lh;
                            if (ok && ch < g.Cout) *reinterpret_cast<f32x4*>(yo + ch) = f32x4{v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
                        }
                    }
                }
            }
            continue;
        }
        const size_t ybase_off = (((size_t)n * g.Ho + oy0 + 4 * wave) * g.Wo + ox0 + 4 * lh) * g.ldy + li;
        float* const ybase = g.y + ybase_off;
        unsigned short* const ybase16 = reinterpret_cast<unsigned short*>(g.y) + ybase_off;
        const size_t rstep = (size_t)g.Wo * g.ldy;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt * 32 + li >= g.Cout) continue;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 2 * mt + (r >> 3), col = ((r >> 2) & 1) * 8 + (r & 3);      // relative to (oy0 + 4 wave, ox0 + 4 lh)
                    float v = acc[mt][nt][r] + bv[nt];
                    if (g.act == KPX_ACT_TANH) v = tanhf(v);
                    else { v = fmaxf(v, lo); v = v > 0.f ? v : v * slope; }
                    if (full || (oy0 + 4 * wave + row < g.Ho && ox0 + 4 * lh + col < g.Wo)) {
                        if (g.y16) { const __bf16 hb = (__bf16)v; ybase16[row * rstep + (size_t)col * g.ldy + nt * 32] = __builtin_bit_cast(unsigned short, hb); }
                        else ybase[row * rstep + (size_t)col * g.ldy + nt * 32] = v;
                    }
                }
        }
    }
}

static std::atomic<unsigned long long> rgb_attr_mask{0};

// stride-1 / stride-2 convolutions of a dense (ldx == Cin) image with Cin <= 4 and Cout <= 64; returns -2 when the shape is not handled
extern "C" __attribute__((visibility("hidden"))) int kpx_conv_rgb_fwd(const float* x, int N, int Hi, int Wi, int Cin, const float* w, int KH, int KW,
                                                                  const float* bias, float* y, int Ho, int Wo, int Cout, int ldy,
                                                                  int stride, int pad_t, int pad_l, int act, hipStream_t s, int y16) {
    if (Cin > 4 || Cout > 64 || KH > 7 || KW > 7 || stride < 1 || stride > 2) return -2;
    RgbGeom g{};
    g.x = x; g.w = w; g.bias = bias; g.y = y; g.y16 = y16;
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.ldy = ldy;
    g.KH = KH; g.KW = KW; g.pad_t = pad_t; g.pad_l = pad_l; g.act = act; g.stride = stride;
    g.tiles_y = (Ho + 15) / 16; g.tiles_x = (Wo + 15) / 16;
    g.KWC = KW * Cin; g.KWCp = (g.KWC + 1) & ~1;
    g.PH = 15 * stride + KH; g.PWC = (15 * stride + KW) * Cin + 1;          // +1: the padded k of the last pixel stays inside the row
    const int nt = Cout <= 32 ? 1 : 2;
    const size_t lds = ((size_t)((g.PH * g.PWC + 4 + 3) & ~3) + (size_t)KH * g.KWCp * 32 * nt) * 4;
    if (kpx_first_use_on_device(&rgb_attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rgb_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rgb_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rgb_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rgb_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        if (e != hipSuccess) return -(int)e;
    }
    if (lds > 98304) return -2;
    g.total_tiles = N * g.tiles_y * g.tiles_x;
    // persistent workgroups: as many as are resident at once (LDS-limited, at most 8 of 4 wavefronts per CU), a contiguous range of tiles each
    int per_cu = (int)((160 * 1024) / (lds + 256)); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    int want = 256 * per_cu; if (want > g.total_tiles) want = g.total_tiles;
    g.tpb = (g.total_tiles + want - 1) / want;
    const unsigned blocks = (unsigned)((g.total_tiles + g.tpb - 1) / g.tpb);
    // 16-byte stores: whole 8-channel groups, aligned pixel rows (and a 16-B aligned bias: 4-channel groups are read as one)
    const bool vec = Cout % 8 == 0 && ldy % 8 == 0 && ((((uintptr_t)y) | ((uintptr_t)bias)) & 15) == 0;
    if (vec) {
        if (nt == 1) hipLaunchKernelGGL((conv_rgb_kernel<1, true>), dim3(blocks), dim3(256), lds, s, g);
        else hipLaunchKernelGGL((conv_rgb_kernel<2, true>), dim3(blocks), dim3(256), lds, s, g);
    } else if (nt == 1) hipLaunchKernelGGL((conv_rgb_kernel<1, false>), dim3(blocks), dim3(256), lds, s, g);
    else hipLaunchKernelGGL((conv_rgb_kernel<2, false>), dim3(blocks), dim3(256), lds, s, g);
    return kpx_launch_status();
}


// ------------------------------------------------------------------------------------------ data gradient towards an image (Cin <= 4)
// dx[n, iy, ix, c] = sum_{ky, kx, co} dy[n, (iy + pad - ky) / S, (ix + pad - kx) / S, co] * w[ky][kx][c][co] for VGG19 conv1_1 (3x3 s1,
// 64 -> 3: the perceptual loss's gradient towards the generated frame, reference models/networks/vgg.py:51) and img_discr conv_0 (4x4 s2,
// 64 -> 3: the adversarial loss's, models/networks/__init__.py:143).  The implicit-GEMM kernel pads the 3 produced channels to an MFMA
// tile of 32-64 and runs at ~8 TFLOP/s; this one is plain VALU work over LDS: one workgroup stages the dy patch of a (16 S)^2 input
// tile 16 channels at a time (18 x 18 pixels, 20 floats per pixel: conflict-free ds_read_b128 at a 16-lane pixel stride), each thread
// owns S x S input pixels, and the filter taps are wave-uniform scalar operands.
//
// FWD = true runs the same loop as a FORWARD convolution with few output channels (the translator's fused crude + mask head, 3x3 s1
// 64 -> 4, reference models/networks/__init__.py:97-99): y[oy, ox, n] = sum x[oy + ky - pad, ox + kx - pad, ci] w[ky][kx][ci][n] is the
// gradient form above with dy := x, co := ci, c := n, the taps mirrored (ky' = KS - 1 - ky, pad' = KS - 1 - pad) and the filter read as
// [tap][ci][n]; bias and activation are applied at the store.
struct RgbDgradGeom {
    const float* dy; const float* w; float* dx; const float* bias;
    int N, Ho, Wo, Cout, lddy, Hi, Wi, Cin, lddx, pad_t, pad_l, tiles_y, tiles_x, act;
    int dy16;                        // bf16 configuration: the gathered tensor (dy, or x of the few-channel forward) is bf16 (lddy in elements)
};

template <int KS, int S, int CIN, bool FWD>
__global__ __launch_bounds__(256) void conv_rgb_dgrad_kernel(const RgbDgradGeom g) {
    constexpr int TI = 16 * S, PR = 18, PS = 20;
    static_assert((TI + KS - 2) / S + 1 <= PR, "dy patch does not fit");      // floor((m + TI - 1) / S) - floor((m - KS + 1) / S) + 1
    __shared__ __attribute__((aligned(16))) float patch[PR * PR * PS];
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    int L = blockIdx.x;
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = L / g.tiles_y;
    const int iy0 = by * TI, ix0 = bx * TI;
    // first dy row / column any pixel of the tile can touch: floor((i0 + pad - (KS - 1)) / S)
    const int ny = iy0 + g.pad_t - (KS - 1), nx = ix0 + g.pad_l - (KS - 1);
    const int oy_min = ny >= 0 ? ny / S : -((-ny + S - 1) / S), ox_min = nx >= 0 ? nx / S : -((-nx + S - 1) / S);
    float acc[S][S][CIN];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b)
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc[a][b][c] = 0.f;

    for (int c0 = 0; c0 < g.Cout; c0 += 16) {
        __syncthreads();
        for (int i = t; i < PR * PR * 4; i += 256) {
            const int px = i >> 2, q = i & 3, pr = px / PR, pc = px - pr * PR;
            const int oy = oy_min + pr, ox = ox_min + pc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)oy < (unsigned)g.Ho && (unsigned)ox < (unsigned)g.Wo) {
                const size_t off = (((size_t)n * g.Ho + oy) * g.Wo + ox) * g.lddy + c0 + 4 * q;
                if (g.dy16) {
                    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                    const u32x2_t h = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned short*>(g.dy) + off);
                    v = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u), __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
                } else v = *reinterpret_cast<const f32x4*>(g.dy + off);
            }
            *reinterpret_cast<f32x4*>(&patch[px * PS + 4 * q]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < S; ++a) {
            const int iyp = iy0 + S * ty + a + g.pad_t;
#pragma unroll
            for (int b = 0; b < S; ++b) {
                const int ixp = ix0 + S * tx + b + g.pad_l;
                for (int ky = (a + g.pad_t) % S; ky < KS; ky += S) {              // (iy0 is a multiple of S: the parity is the thread's a)
                    const int pr = (iyp - ky) / S - oy_min;
                    for (int kx = (b + g.pad_l) % S; kx < KS; kx += S) {
                        const int pc = (ixp - kx) / S - ox_min;
                        const float* pp = &patch[(pr * PR + pc) * PS];
                        // wave-uniform filter addresses: scalar loads.  gradient: w[tap][c][co]; forward: w[mirrored tap][ci = co][n = c]
                        const float* wp = FWD ? g.w + ((size_t)((KS - 1 - ky) * KS + (KS - 1 - kx)) * g.Cout + c0) * CIN
                                              : g.w + (size_t)((ky * KS + kx) * CIN) * g.Cout + c0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 d = *reinterpret_cast<const f32x4*>(pp + 4 * q);
#pragma unroll
                            for (int c = 0; c < CIN; ++c)
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    acc[a][b][c] = fmaf(d[e], FWD ? wp[(4 * q + e) * CIN + c] : wp[(size_t)c * g.Cout + 4 * q + e], acc[a][b][c]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) {
            const int iy = iy0 + S * ty + a, ix = ix0 + S * tx + b;
            if (iy < g.Hi && ix < g.Wi) {
                float* o = g.dx + (((size_t)n * g.Hi + iy) * g.Wi + ix) * g.lddx;
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    float v = acc[a][b][c];
                    if (FWD) {
                        if (g.bias) v += g.bias[c];
                        if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                        else if (g.act == KPX_ACT_TANH) v = tanhf(v);
                    }
                    o[c] = v;
                }
            }
        }
}

// 3x3 stride-1 and 4x4 stride-2 data gradients towards Cin <= 4 channels, Cout a multiple of 16; returns -2 when the shape is not handled
extern "C" __attribute__((visibility("hidden"))) int kpx_conv_rgb_dgrad(const float* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                                                                    float* dx, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l, hipStream_t s, int dy16) {
    const bool k3 = KH == 3 && KW == 3 && stride == 1, k4 = KH == 4 && KW == 4 && stride == 2;
    if (!(k3 || k4) || (Cin != 3 && Cin != 4) || Cout % 16 || lddy % (dy16 ? 8 : 4) || (((uintptr_t)dy) & 15) || (((uintptr_t)w) & 15) || pad_t < 0 || pad_l < 0 ||
        pad_t >= KH || pad_l >= KW)
        return -2;
    // every input pixel's taps must land inside the 18-pixel patch: true for Ho = ceil-type SAME / explicit pads of this path (checked per launch)
    RgbDgradGeom g{};
    g.dy = dy; g.w = w; g.dx = dx; g.dy16 = dy16;
    g.N = N; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.lddy = lddy; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.lddx = lddx; g.pad_t = pad_t; g.pad_l = pad_l;
    const int TI = 16 * stride;
    g.tiles_y = (Hi + TI - 1) / TI; g.tiles_x = (Wi + TI - 1) / TI;
    const unsigned blocks = (unsigned)((size_t)N * g.tiles_y * g.tiles_x);
    if (k3 && Cin == 3) hipLaunchKernelGGL((conv_rgb_dgrad_kernel<3, 1, 3, false>), dim3(blocks), dim3(256), 0, s, g);
    else if (k3) hipLaunchKernelGGL((conv_rgb_dgrad_kernel<3, 1, 4, false>), dim3(blocks), dim3(256), 0, s, g);
    else if (Cin == 3) hipLaunchKernelGGL((conv_rgb_dgrad_kernel<4, 2, 3, false>), dim3(blocks), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((conv_rgb_dgrad_kernel<4, 2, 4, false>), dim3(blocks), dim3(256), 0, s, g);
    return kpx_launch_status();
}

// 3x3 stride-1 forward convolutions with Cout <= 4 and Cin a multiple of 16 (the translator's 64 -> 4 head); -2 when the shape is not handled
extern "C" __attribute__((visibility("hidden"))) int kpx_conv_few_fwd(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const float* w, int KH, int KW,
                                                                  const float* bias, float* y, int Ho, int Wo, int Cout, int ldy,
                                                                  int stride, int pad_t, int pad_l, int act, hipStream_t s) {
    if (KH != 3 || KW != 3 || stride != 1 || (Cout != 3 && Cout != 4) || Cin % 16 || ldx % 4 || (((uintptr_t)x) & 15) || (((uintptr_t)w) & 15) ||
        pad_t < 0 || pad_l < 0 || pad_t > 2 || pad_l > 2 || Ho != Hi + 2 * pad_t - 2 || Wo != Wi + 2 * pad_l - 2)
        return -2;
    RgbDgradGeom g{};
    g.dy = x; g.w = w; g.dx = y; g.bias = bias; g.act = act;
    g.N = N; g.Ho = Hi; g.Wo = Wi; g.Cout = Cin; g.lddy = ldx;          // the gathered tensor
    g.Hi = Ho; g.Wi = Wo; g.Cin = Cout; g.lddx = ldy;                   // the produced one
    g.pad_t = 2 - pad_t; g.pad_l = 2 - pad_l;
    g.tiles_y = (Ho + 15) / 16; g.tiles_x = (Wo + 15) / 16;
    const unsigned blocks = (unsigned)((size_t)N * g.tiles_y * g.tiles_x);
    if (Cout == 3) hipLaunchKernelGGL((conv_rgb_dgrad_kernel<3, 1, 3, true>), dim3(blocks), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((conv_rgb_dgrad_kernel<3, 1, 4, true>), dim3(blocks), dim3(256), 0, s, g);
    return kpx_launch_status();
}
