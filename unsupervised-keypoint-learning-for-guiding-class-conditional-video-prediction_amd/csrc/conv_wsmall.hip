// Weight gradients of the layers whose filter is TINY and whose pixel count is huge: the image-input layers (7x7 3->32 of the encoders, the
// discriminator's 4x4 stride-2 3->64), the key-point detector's 16->16 layer at full resolution and the translator's 64->4 head
// (reference models/networks/__init__.py:9, 52-54, 86-88, 126).  dw[tap][c][n] = sum_pixels x[pixel + tap][c] * dy[pixel][n] is a GEMM with
// M = taps*Cin (48 .. 576), N = Cout (4 .. 64) and K = N*Ho*Wo (0.3 .. 1 M): the 32x32 / 64x64 tiles of the general kernels are mostly
// padding there (3.3 - 22 TF measured), while both operands together are only 15 - 140 MB -- these layers are bound by how fast the pixels
// stream through, not by the matrix pipe.
//
// Here the whole [taps*Cin] x [Cout] result lives in the accumulators of ONE workgroup as 16x16 blocks (v_mfma_f32_16x16x4_f32: K = four
// consecutive output pixels of a row): wavefront w owns RBW row blocks (16 consecutive (tap, channel) rows each) times all NB column
// blocks.  A workgroup walks output tiles of TH x TW pixels of one image: the input patch ((TH-1)*stride+KH rows) and the dy tile are
// staged in LDS once, zero-filled outside the image, and every wavefront sweeps all pixel quads: an A operand is ONE ds_read_b32 at
// patch[row*stride + ky][ (x + k)*stride + kx ][c] (consecutive lanes = consecutive (kx, c) = consecutive floats), a B operand one
// ds_read_b32 of dy.  Persistent workgroups (a contiguous range of tiles each) write one partial slab in HWIO order; the fixed-order
// wgrad_reduce kernel sums the slabs (bitwise reproducible, no atomics).
#include "kpx_common.h"
#include "kpx_env.h"
#include <type_traits>

struct WsmallGeom {
    const float* x; const float* dy; float* out;
    int N, Hi, Wi, ldx, Ho, Wo, Cout, lddy;
    int KH, KW, stride, pad_t, pad_l;
    int dy16;                   // bf16 configuration: dy is a bf16 tensor (lddy in elements), x stays fp32 (the image-input layers)
    int TH, TW;                 // output tile (TW a multiple of 4)
    int tiles_y, tiles_x, total_tiles, tpb;
    int in_rows, in_cols;       // staged patch
    size_t slab;                // floats per slab = KH*KW*Cin*Cout
};

typedef float wf32x4 __attribute__((ext_vector_type(4)));

// XU / DU: (upper bounds of the) staging units per thread for the input patch / the dy tile -- 16-B units when the tensor allows it (VX / VD),
// single floats otherwise.  A unit's place inside the tile never changes, so its (row, column, channel) and LDS offset are worked out once;
// per tile only the bounds check and the base address differ.  The NEXT tile's units are fetched into registers before the current tile is
// multiplied and stored to LDS after it: global latency hides behind the MFMAs.
// On gfx950 nothing overlaps an fp32 MFMA on its SIMD, so the sweep is written to issue as little else as possible: the
// tile width TW, the filter size KS and the stride ST are compile-time, which turns every operand address into one per-row VGPR plus an
// immediate (quad q of a row is q * 16 * ST * CIN bytes further) -- no address arithmetic between the MFMAs.
template <int CIN, int KS, int ST, int TW, int RBW, int NB, int WAVES, int XU, int DU, bool VX, bool VD>
__global__ __launch_bounds__(WAVES * 64) void conv_wgrad_small_kernel(const WsmallGeom g) {
    constexpr int T = WAVES * 64;
    constexpr int DP = NB * 16 + (NB > 1 ? 16 : 0);      // floats per dy pixel in LDS: k * DP mod 64 = 0, 16, 32, 48 -> conflict-free B reads
    constexpr int XW = VX ? 4 : 1, DW = VD ? 4 : 1;
    constexpr int IN_COLS = (TW - 1) * ST + KS, PITCH = IN_COLS * CIN, TWQ = TW / 4, QSTEP = 4 * ST * CIN;
    constexpr int ROWS = KS * KS * CIN;
    constexpr bool PADROWS = ROWS % 16 != 0;             // the last 16-row block overhangs the filter: those A values are forced to zero
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xs = smem;
    float* const dys = smem + ((g.in_rows * PITCH + 3) & ~3);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, k = lane >> 4;

    int offA[RBW];
    bool okA[RBW];
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
        const int idx = (wave * RBW + i) * 16 + r;
        const int tp = idx / CIN, c = idx - tp * CIN;
        const int ky = tp / KS, kx = tp - ky * KS;
        okA[i] = idx < ROWS;
        offA[i] = (okA[i] ? ky * PITCH + kx * CIN + c : 0) + k * ST * CIN;
    }
    wf32x4 acc[RBW][NB];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[i][n] = wf32x4{0.f, 0.f, 0.f, 0.f};

    // staging units of this thread: packed (row << 16 | column), global channel offset, LDS offset (-1: no such unit)
    int xrc[XU], xco[XU], xl[XU], drc[DU], dco[DU], dl[DU];
    {
        constexpr int per_pix = CIN / XW, per_row = IN_COLS * per_pix;
        const int total = g.in_rows * per_row;
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = t + i * T;
            const int rr = u / per_row, e = u - rr * per_row;
            const int col = e / per_pix, c = (e - col * per_pix) * XW;
            xrc[i] = (rr << 16) | col; xco[i] = c;
            xl[i] = u < total ? rr * PITCH + col * CIN + c : -1;
        }
        const int dper_pix = (g.Cout + DW - 1) / DW, dtotal = g.TH * TW * dper_pix;
#pragma unroll
        for (int i = 0; i < DU; ++i) {
            const int u = t + i * T;
            const int p = u / dper_pix, c = (u - p * dper_pix) * DW;
            const int rr = p / TW, col = p - rr * TW;
            drc[i] = (rr << 16) | col; dco[i] = c;
            dl[i] = u < dtotal ? p * DP + c : -1;
        }
    }
    typedef typename std::conditional<VX, wf32x4, float>::type xunit;
    typedef typename std::conditional<VD, wf32x4, float>::type dunit;
    xunit xv[XU];
    dunit dv[DU];
    // Operands through buffer descriptors (round 4; as 64-bit pointers with a bounds branch per unit the fetch was ~20 instructions per unit
    // beside an fp32 MFMA that hides none of them): voffset = the unit's place relative to the tile origin (fixed), soffset = the tile origin
    // (scalar), an out-of-image unit carries the out-of-range offset and reads as zero.  The x descriptor starts (pad_t rows + pad_l pixels)
    // before the tensor so that the padded patch origin is never negative; those positions are only addressed with the out-of-range offset.
    const int WS_OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x) - ((size_t)g.pad_t * g.Wi + g.pad_l) * g.ldx, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dy), 0, 0x7fffffff, 0x00020000);
    int xvo[XU], dvo[DU];
#pragma unroll
    for (int i = 0; i < XU; ++i) xvo[i] = xl[i] >= 0 ? (((xrc[i] >> 16) * g.Wi + (xrc[i] & 0xffff)) * g.ldx + xco[i]) * 4 : WS_OOB;
#pragma unroll
    for (int i = 0; i < DU; ++i) dvo[i] = dl[i] >= 0 ? (((drc[i] >> 16) * g.Wo + (drc[i] & 0xffff)) * g.lddy + dco[i]) * 4 : WS_OOB;
    // tile cursor (scalar; readfirstlane: integer division runs on the vector ALU): tiles of one workgroup are consecutive
    const int t0 = blockIdx.x * g.tpb, t1 = min(t0 + g.tpb, g.total_tiles);
    int ctx = __builtin_amdgcn_readfirstlane(t0 % g.tiles_x), cty = __builtin_amdgcn_readfirstlane((t0 / g.tiles_x) % g.tiles_y),
        cn = __builtin_amdgcn_readfirstlane(t0 / (g.tiles_x * g.tiles_y));
    auto fetch = [&]() {                                      // the tile under the cursor, then advance it
        const int oy0 = cty * g.TH, ox0 = ctx * TW;
        const int iy0 = oy0 * ST - g.pad_t, ix0 = ox0 * ST - g.pad_l;
        const int sx = ((cn * g.Hi + oy0 * ST) * g.Wi + ox0 * ST) * g.ldx * 4;       // byte offsets < 2^31 (checked by the planner)
        const int sd = ((cn * g.Ho + oy0) * g.Wo + ox0) * g.lddy * 4;
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int iy = iy0 + (xrc[i] >> 16), ix = ix0 + (xrc[i] & 0xffff);
            const bool ok = (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            const int vo = ok ? xvo[i] : WS_OOB;
            if constexpr (VX) xv[i] = __builtin_bit_cast(xunit, __builtin_amdgcn_raw_buffer_load_b128(rsx, vo, sx, 0));
            else xv[i] = __builtin_bit_cast(xunit, __builtin_amdgcn_raw_buffer_load_b32(rsx, vo, sx, 0));
        }
#pragma unroll
        for (int i = 0; i < DU; ++i) {
            const int oy = oy0 + (drc[i] >> 16), ox = ox0 + (drc[i] & 0xffff);
            const int vo = (oy < g.Ho && ox < g.Wo) ? dvo[i] : WS_OOB;
            if constexpr (VD) {
                if (g.dy16) {                                  // four bf16 channels = 8 bytes: two dword loads (raw_buffer_load_b64 is mis-lowered), widened
                    const int v2 = vo == WS_OOB ? WS_OOB : vo >> 1;
                    const unsigned lo = __builtin_amdgcn_raw_buffer_load_b32(rsd, v2, sd >> 1, 0), hi = __builtin_amdgcn_raw_buffer_load_b32(rsd, v2 == WS_OOB ? WS_OOB : v2 + 4, sd >> 1, 0);
                    dv[i] = wf32x4{__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u), __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
                } else dv[i] = __builtin_bit_cast(dunit, __builtin_amdgcn_raw_buffer_load_b128(rsd, vo, sd, 0));
            } else dv[i] = __builtin_bit_cast(dunit, __builtin_amdgcn_raw_buffer_load_b32(rsd, vo, sd, 0));
        }
        if (++ctx == g.tiles_x) { ctx = 0; if (++cty == g.tiles_y) { cty = 0; ++cn; } }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < XU; ++i) if (xl[i] >= 0) *reinterpret_cast<xunit*>(&xs[xl[i]]) = xv[i];
#pragma unroll
        for (int i = 0; i < DU; ++i) if (dl[i] >= 0) *reinterpret_cast<dunit*>(&dys[dl[i]]) = dv[i];
    };

    if (t0 < t1) fetch();
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();                                      // the previous tile's reads are done
        stage();
        __syncthreads();
        if (tile + 1 < t1) fetch();                           // in flight while this tile is multiplied
        // ---- every wavefront sweeps all pixel quads of the tile; the operands of quad q + 1 are read before quad q is multiplied ----
        for (int rr = 0; rr < g.TH; ++rr) {
            const float* xa[RBW];
#pragma unroll
            for (int i = 0; i < RBW; ++i) xa[i] = xs + rr * ST * PITCH + offA[i];
            const float* const db = dys + (rr * TW + k) * DP + r;
            float a[2][RBW], b[2][NB];
#pragma unroll
            for (int i = 0; i < RBW; ++i) a[0][i] = xa[i][0];
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) b[0][nn] = db[nn * 16];
#pragma unroll
            for (int q = 0; q < TWQ; ++q) {
                const int cur = q & 1;
                if (q + 1 < TWQ) {
#pragma unroll
                    for (int i = 0; i < RBW; ++i) a[cur ^ 1][i] = xa[i][(q + 1) * QSTEP];
#pragma unroll
                    for (int nn = 0; nn < NB; ++nn) b[cur ^ 1][nn] = db[(q + 1) * 4 * DP + nn * 16];
                }
#pragma unroll
                for (int i = 0; i < RBW; ++i) {
                    const float av = (PADROWS && !okA[i]) ? 0.f : a[cur][i];
#pragma unroll
                    for (int nn = 0; nn < NB; ++nn) acc[i][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[cur][nn], acc[i][nn], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial slab of this workgroup, HWIO order: row (tap, c) = block * 16 + 4 * (lane / 16) + v, column = nb * 16 + lane % 16 ----
    float* const out = g.out + (size_t)blockIdx.x * g.slab;
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int nn = 0; nn < NB; ++nn)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int idx = (wave * RBW + i) * 16 + 4 * k + v, col = nn * 16 + r;
                if (idx < ROWS && col < g.Cout) out[(size_t)idx * g.Cout + col] = acc[i][nn][v];
            }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
// variant: 0 = none; 1 = Cin 16, 3x3, Cout <= 16; 2 = Cin 3, 7x7, Cout <= 32; 3 = Cin 3, 4x4 stride 2, Cout <= 64; 4 = Cin 64, 3x3, Cout <= 16;
// 5 = Cin 32, 3x3, Cout <= 32
static int wsmall_variant(int Cin, int Cout, int KH, int KW, int stride) {
    if (Cin == 16 && KH == 3 && KW == 3 && stride == 1 && Cout <= 16) return 1;
    if (Cin == 3 && KH == 7 && KW == 7 && stride == 1 && Cout <= 32) return 2;
    if (Cin == 3 && KH == 4 && KW == 4 && stride == 2 && Cout <= 64) return 3;
    if (Cin == 64 && KH == 3 && KW == 3 && stride == 1 && Cout <= 4) return 4;
    if (Cin == 32 && KH == 3 && KW == 3 && stride == 1 && Cout <= 32) return 5;
    return 0;
}

static void wsmall_geom(int variant, int N, int Hi, int Wi, int ldx, int Ho, int Wo, int Cout, int lddy, int KH, int KW, int stride, int pad_t, int pad_l,
                        int Cin, WsmallGeom* g, int* lds_bytes, int* threads) {
    g->N = N; g->Hi = Hi; g->Wi = Wi; g->ldx = ldx; g->Ho = Ho; g->Wo = Wo; g->Cout = Cout; g->lddy = lddy;
    g->KH = KH; g->KW = KW; g->stride = stride; g->pad_t = pad_t; g->pad_l = pad_l;
    int th, tw, nb, waves;
    switch (variant) {
        case 1: th = 4; tw = 64; nb = 1; waves = 3; break;
        case 2: th = 2; tw = 64; nb = 2; waves = 5; break;      // (4 rows per tile measured slower: 0.195 vs 0.174 ms at N = 64)
        case 3: th = 2; tw = 68; nb = 4; waves = 3; break;
        case 5: th = 2; tw = 64; nb = 2; waves = 6; break;
        default: th = 4; tw = 32; nb = 1; waves = 4; break;
    }
    if (th > Ho) th = Ho;                                 // (the tile width is compile-time: columns beyond Wo are staged as zeros)
    g->TH = th; g->TW = tw;
    g->tiles_y = (Ho + th - 1) / th; g->tiles_x = (Wo + tw - 1) / tw;
    g->total_tiles = N * g->tiles_y * g->tiles_x;
    g->in_rows = (th - 1) * stride + KH; g->in_cols = (tw - 1) * stride + KW;
    g->slab = (size_t)KH * KW * Cin * Cout;
    const int dp = nb * 16 + (nb > 1 ? 16 : 0);
    *lds_bytes = (((g->in_rows * g->in_cols * Cin + 3) & ~3) + th * tw * dp) * 4;
    *threads = waves * 64;
}

// number of partial slabs (= workgroups) the layer is computed in; 0: the layer is not one of this family's
extern "C" __attribute__((visibility("hidden"))) int kpx_wsmall_splits(int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride) {
    const int variant = wsmall_variant(Cin, Cout, KH, KW, stride);
    if (!variant || (long)N * Ho * Wo < 32768) return 0;
    if ((size_t)N * Hi * Wi * (size_t)((Cin + 3) & ~3) * 4 >= 0x60000000ull || (size_t)N * Ho * Wo * (size_t)((Cout + 3) & ~3) * 4 >= 0x60000000ull) return 0;   // 32-bit byte offsets in the kernel
    WsmallGeom g{}; int lds, threads;
    wsmall_geom(variant, N, Hi, Wi, Cin, Ho, Wo, Cout, Cout, KH, KW, stride, 0, 0, Cin, &g, &lds, &threads);
    // persistent workgroups: as many as are resident at once (LDS and, at ~200 VGPRs, two wavefronts per SIMD), at least 2 tiles each
    int per_cu = (160 * 1024) / (lds + 512); if (per_cu < 1) per_cu = 1;
    const int by_regs = variant == 2 ? 3 : 2;
    if (per_cu > by_regs) per_cu = by_regs;
    long S = 256L * per_cu;
    if (S > g.total_tiles / 2) S = g.total_tiles / 2;
    if (S < 1) S = 1;
    const long tpb = (g.total_tiles + S - 1) / S;
    return (int)((g.total_tiles + tpb - 1) / tpb);        // exactly the workgroups kpx_wsmall_launch starts for this S
}

extern "C" __attribute__((visibility("hidden"))) int kpx_wsmall_launch(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const float* dy, int Ho, int Wo, int Cout, int lddy,
                                                                   int KH, int KW, int stride, int pad_t, int pad_l, float* slabs, int S, hipStream_t s, int dy16) {
    static std::atomic<unsigned long long> attr_mask{0};
    const int variant = wsmall_variant(Cin, Cout, KH, KW, stride);
    if (!variant) return KPX_EINVAL;
    WsmallGeom g{}; int lds, threads;
    wsmall_geom(variant, N, Hi, Wi, ldx, Ho, Wo, Cout, lddy, KH, KW, stride, pad_t, pad_l, Cin, &g, &lds, &threads);
    g.x = x; g.dy = dy; g.out = slabs; g.dy16 = dy16;
    g.tpb = (g.total_tiles + S - 1) / S;
#define WS_K1 conv_wgrad_small_kernel<16, 3, 1, 64, 3, 1, 3, 9, 6, true, true>
#define WS_K2 conv_wgrad_small_kernel<3, 7, 1, 64, 2, 2, 5, 6, 4, false, true>
#define WS_K3 conv_wgrad_small_kernel<3, 4, 2, 68, 1, 4, 3, 13, 12, false, true>
#define WS_K4 conv_wgrad_small_kernel<64, 3, 1, 32, 9, 1, 4, 13, 2, true, true>
#define WS_K5 conv_wgrad_small_kernel<32, 3, 1, 64, 3, 2, 6, 6, 3, true, true>
    if (kpx_first_use_on_device(&attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(WS_K1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(WS_K2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(WS_K3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(WS_K4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(WS_K5), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e;
    }
    const dim3 grid((unsigned)((g.total_tiles + g.tpb - 1) / g.tpb));
    if ((int)grid.x != S) return KPX_EINVAL;              // S must come from kpx_wsmall_splits (the reduce sums exactly S slabs)
    switch (variant) {
        case 1: hipLaunchKernelGGL(WS_K1, grid, dim3(threads), lds, s, g); break;
        case 2: hipLaunchKernelGGL(WS_K2, grid, dim3(threads), lds, s, g); break;
        case 3: hipLaunchKernelGGL(WS_K3, grid, dim3(threads), lds, s, g); break;
        case 5: hipLaunchKernelGGL(WS_K5, grid, dim3(threads), lds, s, g); break;
        default: hipLaunchKernelGGL(WS_K4, grid, dim3(threads), lds, s, g); break;
    }
    return kpx_launch_status();
}
