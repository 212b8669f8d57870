// Fused Winograd F(4x4, 3x3) convolution whose 36 transform-domain GEMMs run FP32-EQUIVALENT ON THE BF16 MATRIX PIPE of gfx950
// (v_mfma_f32_32x32x16_bf16, six products of exact three-term operands), for the 3x3 stride-1 SAME layers of the fp32 configuration
// (reference models/networks/layers.py:4-10 on models/networks/__init__.py:13-24,50-62,80-97 and models/networks/vgg.py:20-40):
// forward, and the data gradient on dgrad-transformed filters.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      as in conv_wino43.hip (points 0, +-1, +-2, inf), but
//   U = G g G^T (fp64, rounded once to fp32) is stored PRE-SPLIT as u = u0 + u1 + u2 (three bf16 planes, exact: trunc16 twice, the rest fits)
//   V = B^T d B (fp32 VALU) is split the same way in registers right after the input transform, and each point-wise product is
//   v.u ~ v1u1 + v0u2 + v2u0 + v0u1 + v1u0 + v0u0  (smallest first, fp32 accumulate; the dropped terms are < 2^-23 |vu|: conv_gemm3.hip).
// Cost per 32 x 32 x 16 block: 6 x 32 cycles instead of 8 x 64 of v_mfma_f32_32x32x2_f32, and -- the point of the exercise, DESIGN.md 4.1 --
// a bf16 MFMA lets the other instructions of its SIMD issue beside it.
//
// Why the structure differs from conv_wino43_kernel: with the multiplies 2.7x cheaper the kernel is bound by everything else -- V through LDS
// (110 KB per 16 channels as three bf16 planes: it cannot even be double buffered), the transform on two of four SIMDs, 20 k cycles of
// prologue / epilogue per tile.  Here V NEVER TOUCHES LDS:
//   * one workgroup = 4 wavefronts (one per SIMD, up to 512 registers each) = 16 x 32 output pixels (4 x 8 tiles = the 32 rows of the MFMA)
//     x 64 output channels x all 36 points; wavefront (rh, ch) owns the 3 x 3 block of points {rows 3rh..3rh+2} x {cols 3ch..3ch+2} for BOTH
//     32-channel halves: 18 accumulator blocks = 288 registers;
//   * lane (tile = lane & 31, channel octet = lane >> 5) is exactly the lane of the MFMA's A operand that needs V[point][tile][8 channels], so
//     every wavefront transforms ITS nine points for its own lanes: B^T d B restricted to a 3 x 3 block needs 5 x 5 of the 6 x 6 patch and
//     48 instead of 36 FMAs per channel (the separable passes share less) -- 1/3 more transform arithmetic buys no V traffic, no V
//     barriers and no transform / multiply role split;
//   * the raw patch (18 x 34 pixels x 16 channels per K step, fp32) reaches LDS by LDS-DMA (no staging registers), pixel-major (the four
//     channel quads of a pixel are consecutive lanes of a request: 64 contiguous bytes) with one skew slot per four pixels (conflict-free
//     transform reads), double buffered: ONE barrier per 16-channel K step;
//   * the filter fragments come straight from L2 in fragment order through a buffer descriptor with scalar offsets, a ring of six
//     (point, cout half) units = 72 registers, each fragment refilled in place as soon as its last MFMA has issued (~3 points ahead);
//   * per K step: ONE transform pass (the 5 x 5 patch of the lane's tile read once: 50 ds_read_b128 -- the four wavefronts transform at the
//     same time and that phase runs at the LDS read bandwidth), then the nine points' 108 MFMAs as one HAND-PLACED instruction stream: one
//     wavefront per SIMD hides at most six single-issue vector instructions under a 32-cycle MFMA and no packed-fp32 one
//     (scratch/micro/valu_fill.hip), so every MFMA is followed by its share of the NEXT point's split (44 single-issue VALU over twelve
//     gaps), a filter refill, an LDS-resident accumulator's store / fetch -- pinned by sched_barrier;
//   * the epilogue runs in two passes of ALL 36 points x 32 output channels through LDS (147 KB): every wavefront deposits nine blocks in
//     both passes, every thread owns (tile, 4 couts), reads its 36 values as ds_read_b128 and finishes the 4 x 4 output pixels in one go.
// STATS / mask / pool epilogue options are those of conv_wino43_kernel (same statistics strips: kpx_conv3x3_wino43_stats_tiles); PACK: two 16 x 16
// images side by side in one 16 x 32 region (VGG19 conv4_*, the 16 x 16 data gradients of the encoders), each with its own zero halo in the patch.
#include "kpx_common.h"
#include "kpx_env.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define W4B_OOB 0x7ffffff0                 // voffset beyond any buffer: the load returns / lands as zeros
#define W4B_ROWSLOTS 146                   // 16-B slots per patch row: pixel column c at slot 4 c + (c >> 2) (four channel quads per pixel, one skew slot per
                                           // four pixels), 34 columns = 144 slots, padded to 146 so that four rows are 8 (mod 16) slots
#define W4B_ROWSLOTS_PACK 154              // PACK (two 16 x 16 images side by side): image A's 18 patch columns at slots 0..75 of a row, image B's at 76..151 (tile
                                           // origins 17 tx and 17 tx + 8: the sixteen 16-B columns of a ds_read_b128 group stay distinct), 4 x 154 = 8 (mod 16) too
#define W4B_NPIECES 44                     // LDS-DMA pieces of 1 KB per buffer (18 x 146 = 2628 slots, rounded up to 11 pieces per wavefront)
#define W4B_RAW_BYTES (W4B_NPIECES * 1024) // one raw buffer
#define W4B_ACC_OFF (2 * W4B_RAW_BYTES)    // [4 wavefronts][4 blocks][4][64 lanes] x 16 B: four of a wavefront's 18 accumulator blocks (units 3, 9, 13, 17 of a K
                                           // step) live here between K steps; each is fetched one point before its own, multiplied into, and put back
                                           // during the point after (never two in registers at once).  18 blocks are 288 registers, the accumulator
                                           // file has 256: with all of them in registers hipcc spills two blocks to scratch every K step (vector-memory
                                           // operations in the middle of a counted-vmcnt pipeline)
#define W4B_NLDS 4
#ifndef W4B_DMA_IN_T
#define W4B_DMA_IN_T 1                    // the next patch's eleven LDS-DMA pieces between the transform's column groups (texture addresser beside LDS reads /
                                           // VALU: -2 ... -5 % per launch against all of them before the transform = 0; the offsets then wait in spare AGPRs)
#endif
#ifndef W4B_U_AUX
#define W4B_U_AUX 0                       // cache policy of the filter-fragment loads (measured: 1 = sc0 no change, 2 = nt 10-20 % slower -- every CU re-reads U from L2)
#endif
#ifndef W4B_RING
#define W4B_RING 6                         // filter-fragment ring: units (must divide 18)
#endif
#if W4B_RING == 6
#define W4B_RING3 18                        // fragment loads a full ring holds: the counted waits' argument
#else
#define W4B_RING3 27
#endif
#define W4B_STR2(x) #x
#define W4B_STR(x) W4B_STR2(x)
#define W4B_EPI_BYTES (36 * 32 * 32 * 4)   // P[point][tile][32 couts] fp32
#define W4B_MAIN_BYTES (W4B_ACC_OFF + 4 * W4B_NLDS * 4096)

struct Wino43bGeom {
    const float* x; float* y; const void* U; const float* bias;
    int N, H, W, Cin, ldx, Cout, ldy, act;
    int Kp, Np;                              // U is [36][Kp/16][Np/32][3 terms][64 lanes][8] bf16
    int tiles_y, tiles_x;                    // 16 x 32-pixel regions per image
    const float* mask_y; int ld_mask;        // optional: zero the output where mask_y <= 0
    float* pool_y; int ld_pool;              // optional: also write the 2x2 max-pool of the (activated) output
    float* stats;                            // STATS 1 / 2: as conv_wino43_kernel
    const float* bn_beta;
};

__device__ __forceinline__ unsigned w4b_bits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float w4b_float(unsigned v) { return __builtin_bit_cast(float, v); }
// {hi16(b), hi16(a)}: two truncated bf16 values in one dword, a in the low half
__device__ __forceinline__ unsigned w4b_pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// exact three-way split of eight floats into the three A-operand fragments of one point (element e of a fragment = channel e of the lane's octet)
__device__ __forceinline__ void w4b_split8(const float* __restrict__ v, u32x4* __restrict__ f) {
#if defined(W4B_EXP) && (W4B_EXP & 1)      // timing experiment (numerically meaningless): no split arithmetic
#pragma unroll
    for (int m = 0; m < 4; ++m) { f[0][m] = w4b_bits(v[2 * m]); f[1][m] = w4b_bits(v[2 * m + 1]); f[2][m] = w4b_bits(v[m]); }
    return;
#endif
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float a = v[2 * m], b = v[2 * m + 1];
        const unsigned a0 = w4b_bits(a), b0 = w4b_bits(b);
        f[0][m] = w4b_pack_hi(a0, b0);
        const float ra = a - w4b_float(a0 & 0xffff0000u), rb = b - w4b_float(b0 & 0xffff0000u);
        const unsigned a1 = w4b_bits(ra), b1 = w4b_bits(rb);
        f[1][m] = w4b_pack_hi(a1, b1);
        const float sa = ra - w4b_float(a1 & 0xffff0000u), sb = rb - w4b_float(b1 & 0xffff0000u);
        f[2][m] = w4b_pack_hi(w4b_bits(sa), w4b_bits(sb));
    }
}

// ---------------------------------------------------------------------------------------------- filter transform (once per optimiser update)
// U[p = 6 i + j] = (G g G^T)[i][j] in double precision, rounded once (conv_wino43.hip says why), then split into three bf16 terms and stored
// as the B operand of v_mfma_f32_32x32x16_bf16: Ub[p][ks][nb][term][lane = r + 32 h][e] = term of U[p][c = 16 ks + 8 h + e][n = 32 nb + r].
// One thread = (point row i, channel octet c8 = 2 ks + h, n): it reads the 3 x 3 taps of its eight channels (consecutive threads =
// consecutive n: coalesced forward; eight contiguous channels per thread for the transposed dgrad read), forms the six points of row i and
// writes ONE 16-byte fragment element group per (point, term) -- a wavefront stores 1 KB contiguous.  (Measured alternatives: a thread with
// all 36 points of a channel pair stores 108 dwords at a 16-byte lane stride, 0.39 ms per step for the batch of trainable filters; a thread
// per point re-reads the taps 36 times, 0.93 ms.  The fp32 form's transform is 0.03 ms.)
struct KpxWino43bDesc { const float* w; unsigned* u; int cin, cout, dgrad, reserved; };

__constant__ double w4b_G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6.0, -1.0 / 6.0, -1.0 / 6.0}, {-1.0 / 6.0, 1.0 / 6.0, -1.0 / 6.0},
                                   {1.0 / 24.0, 1.0 / 12.0, 1.0 / 6.0}, {1.0 / 24.0, -1.0 / 12.0, 1.0 / 6.0}, {0.0, 0.0, 1.0}};

__device__ __forceinline__ void w4b_transform_filter(const float* __restrict__ w, int Cin, int Cout, bool dg, size_t idx, int Kp, int Np, unsigned* __restrict__ Ub) {
    const int K = dg ? Cout : Cin, Nn = dg ? Cin : Cout;
    const int KS = Kp >> 4, NB = Np >> 5, C8 = Kp >> 3;
    const int n = (int)(idx % (size_t)Np);
    const int c8 = (int)((idx / (size_t)Np) % (size_t)C8), i = (int)(idx / ((size_t)Np * C8));
    const double gi0 = w4b_G[i][0], gi1 = w4b_G[i][1], gi2 = w4b_G[i][2];
    float u[6][8];
    // taps of the eight channels: g[tap (r, q) of the flipped / transposed filter][e]
    float g[9][8];
    const bool vec = dg && 8 * c8 + 8 <= K && n < Nn && (Cout & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    if (vec) {                                           // dgrad: the eight channels are contiguous in memory -- two 16-byte loads per tap
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const f32x4* src = reinterpret_cast<const f32x4*>(w + ((size_t)(8 - tp) * Cin + n) * Cout + 8 * c8);
            const f32x4 a = src[0], b = src[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[tp][e] = a[e]; g[tp][4 + e] = b[e]; }
        }
    } else {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 8 * c8 + e;
                g[tp][e] = !(c < K && n < Nn) ? 0.f : dg ? w[((size_t)(8 - tp) * Cin + n) * Cout + c] : w[((size_t)tp * Cin + c) * Cout + n];
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        double t[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) t[q] = gi0 * (double)g[q][e] + gi1 * (double)g[3 + q][e] + gi2 * (double)g[6 + q][e];
#pragma unroll
        for (int j = 0; j < 6; ++j) u[j][e] = (float)(t[0] * w4b_G[j][0] + t[1] * w4b_G[j][1] + t[2] * w4b_G[j][2]);
    }
    const int ks = c8 >> 1, h = c8 & 1, nb = n >> 5, r = n & 31;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        u32x4 f[3];
        w4b_split8(u[j], f);
        u32x4* const o = reinterpret_cast<u32x4*>(Ub + (((size_t)(6 * i + j) * KS + ks) * NB + nb) * 768) + (r + 32 * h);
        o[0] = f[0]; o[64] = f[1]; o[128] = f[2];
    }
}
__global__ __launch_bounds__(256) void wino43b_filter_transform_batch_kernel(const KpxWino43bDesc* __restrict__ descs) {
    const KpxWino43bDesc d = descs[blockIdx.y];
    const bool dg = d.dgrad != 0;
    const int K = dg ? d.cout : d.cin, Nn = dg ? d.cin : d.cout;
    const int Kp = (K + 15) & ~15, Np = (Nn + 63) & ~63;
    const size_t total = (size_t)6 * (Kp / 8) * Np;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
        w4b_transform_filter(d.w, d.cin, d.cout, dg, idx, Kp, Np, d.u);
}
__global__ __launch_bounds__(256) void wino43b_filter_transform_kernel(const float* __restrict__ w, int Cin, int Cout, int dgrad, int Kp, int Np, unsigned* __restrict__ Ub) {
    const size_t total = (size_t)6 * (Kp / 8) * Np;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
        w4b_transform_filter(w, Cin, Cout, dgrad != 0, idx, Kp, Np, Ub);
}

// ---------------------------------------------------------------------------------------------- the convolution
__constant__ float w4b_AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};

// unit u = 2 lp + nb of a K step: LDS-resident accumulator slot (0..3) or -1; register block index (0..13) of the others
// units whose accumulator block lives in VGPRs (VGPR-form MFMA, written as asm: the builtin selects the AGPR form): hipcc allocates 240 AGPRs to
// 15 blocks and time-shares three of them with these units' blocks through 96 v_accvgpr moves per K step whatever the live ranges are, while
// 48 VGPRs are free over the whole loop.  (The MFMA that next touches such a block follows two MFMAs later: no hazard the compiler would pad.)
#define W4B_VFORM(u) ((u) == 16)
#ifndef W4B_VFORM1
#define W4B_VFORM1(u) (w4b_lds_slot(u) >= 0)
#endif
__device__ __forceinline__ constexpr int w4b_lds_slot(int u) { return u == 3 ? 0 : u == 9 ? 1 : u == 13 ? 2 : u == 17 ? 3 : -1; }
__device__ __forceinline__ constexpr int w4b_reg_block(int u) { return u - (u > 3) - (u > 9) - (u > 13); }

// global row / column of the 6 x 6 point grid of a wavefront's local kind (0: the `single` one, 1 / 2: the pair) for block half hf (0: rows 0-2)
__device__ __forceinline__ constexpr int w4b_grid(int hf, int kind) { return kind == 0 ? (hf ? 5 : 0) : (hf ? 2 + kind : kind); }

#ifdef KPX_W4B_STAMP      // diagnostic build only (scratch/w43b_stamps.py): s_memtime stamps of every wavefront of the first 64 workgroups
static __device__ unsigned long long* w4b_dbg = nullptr;
extern "C" int kpx_debug_w4b_stamps(unsigned long long* buf) { return -(int)hipMemcpyToSymbol(HIP_SYMBOL(w4b_dbg), &buf, sizeof(buf)); }
#define W4B_STAMP(slot) do { if (dbgp) dbgp[(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W4B_KSTAMP(k, j) do { __builtin_amdgcn_sched_barrier(0); if (dbgp && (k) >= 1 && (k) < 5) dbgp[16 + ((k) - 1) * 8 + (j)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4B_USTAMP(k, u) do { if (dbgp && (k) == 2) dbgp[256 + (u)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4B_STAMP(slot) do { } while (0)
#define W4B_KSTAMP(k, j) do { } while (0)
#define W4B_USTAMP(k, u) do { } while (0)
#endif

// Packed-fp32 arithmetic of the transform phase, written out (the phase has no MFMA to disturb): c * a + b with a scalar-pair constant, a +- b.
// hipcc emits a - b on f32x4 as four v_sub_f32 and the negated products through v_xor sign flips: 316 vector instructions per K step for what is
// 192 packed ones.  (FMA contraction of 4 e0 - 5 e2 + e4 etc.: the transform rounds once less per term.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long w4b_c2(float c) { const unsigned u = __builtin_bit_cast(unsigned, c); return ((unsigned long long)u << 32) | u; }
__device__ __forceinline__ f32x4 w4b_fma4(unsigned long long c, const f32x4& a, const f32x4& b) {       // c * a + b
    f32x2 lo, hi;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "s"(c), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "s"(c), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 w4b_add4(const f32x4& a, const f32x4& b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 w4b_sub4(const f32x4& a, const f32x4& b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// The K loop of one wavefront role.  RH / CH: which half of the point rows / columns (compile time: the second transform pass selects
// REGISTERS by column, the first one patch rows).  acc[lp = 3 rk + ck][nb]: local point (row kind rk, column kind ck), 32-cout block nb.
template <bool PACK>
__device__ __forceinline__ void w4b_kloop(const int rh, const int ch, const Wino43bGeom& g, f32x16 (&accr)[14], unsigned char* const smem, const int lane, const int wave,
                                         const int n, const int oy0, const int ox0, const int nti, unsigned long long* const dbgp) {
    const int KS = g.Kp >> 4, NB = g.Np >> 5;
    // ---- LDS-DMA of the raw patch: piece wave * 12 + i covers slots 64 (wave * 12 + i) .. + 63 of the buffer
    constexpr int RS = PACK ? W4B_ROWSLOTS_PACK : W4B_ROWSLOTS;
    const unsigned img_bytes = (unsigned)g.H * g.W * g.ldx * 4u * (PACK ? 2u : 1u);      // PACK: images n and n + 1
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x) + (size_t)n * g.H * g.W * g.ldx, 0, img_bytes, 0x00020000);
    // slot S of row r = S / 146: within the row, slot 17 g + 4 k + q is channel quad q of pixel column 4 g + k (slot 17 g + 16: the skew slot; slots
    // 144, 145: row padding).  Consecutive lanes = the four quads of a pixel (64 contiguous bytes of the tensor) and four consecutive pixels.
    unsigned dtail = 0;                                  // bit i: piece i's channel quad exists in the LAST K step (channel tail; K % 4 == 0)
    int dv[11];                                          // buffer offsets of this lane's slots (tile constants; the compiler parks them in spare accumulator registers)
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int S = (wave * 11 + i) * 64 + lane;
        const int row = S / RS, rs0 = S - row * RS;
        const int half = PACK && rs0 >= 76 ? 1 : 0, rs = rs0 - 76 * half;               // PACK: which image's half row
        const int grp = rs / 17, r17 = rs - grp * 17;
        const int col = 4 * grp + (r17 >> 2), q = r17 & 3;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + col;
        const bool ok = r17 < 16 && rs < (PACK ? 76 : 144) && row < 18 && col < (PACK ? 18 : 34) && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        dv[i] = ok ? (((half * g.H + iy) * g.W + ix) * g.ldx + q * 4) * 4 : W4B_OOB;
        if ((KS - 1) * 16 + q * 4 < g.Cin) dtail |= 1u << i;
    }
    auto dma_piece = [&](int s, int buf, int i) {        // piece i of K step s -> raw buffer buf (past the last K step: zeros into the idle buffer)
        unsigned char* const dst = smem + buf * W4B_RAW_BYTES + wave * 11 * 1024;
        const int vo = (s >= KS || (s == KS - 1 && !((dtail >> i) & 1u))) ? W4B_OOB : dv[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(dst + i * 1024), 16, vo, s * 64, 0, 0);
    };
    auto dma = [&](int s, int buf) {                     // K step s -> raw buffer buf
#pragma unroll
        for (int i = 0; i < 11; ++i) dma_piece(s, buf, i);
    };
    // ---- filter fragments: unit u = 2 lp + nb of a K step; ring slot u % 6
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.U), 0, 36u * (unsigned)(KS * NB) * 3072u, 0x00020000);
    const int ulane = lane * 16;
    const int ustep = NB * 3072;                         // bytes per K step
    int ubase[9];
#pragma unroll
    for (int lp = 0; lp < 9; ++lp)
        ubase[lp] = __builtin_amdgcn_readfirstlane(((6 * w4b_grid(rh, lp / 3) + w4b_grid(ch, lp % 3)) * KS * NB + 2 * nti) * 3072);
    u32x4 ub[W4B_RING][3];
    auto uload = [&](int slot, int unit, int koff) {     // unit of the K step at byte offset koff -> ring slot
#if defined(W4B_EXP) && (W4B_EXP & 2)      // timing experiment: the ring is loaded once (prologue) and never refilled
        if (unit >= 6) return;
#endif
        const int so = ubase[unit >> 1] + (unit & 1) * 3072 + koff;
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
            ub[slot][tm] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane, so + tm * 1024, W4B_U_AUX));
    };

    // ---- transform read base of this lane: pixel (4 ty + RH, 4 tx + CH) of the patch, channel quad 2 * octet.  Tile origins are 17 slots apart
    // in x and 4 x 146 = 8 (mod 16) slots in y, so the sixteen lanes of every ds_read_b128 group (tiles {0-3, 12-15, 20-27} /
    // {4-11, 16-19, 28-31} of one octet) hit sixteen different 16-B columns of the 256-B bank row: {0..3}, {12..15}, {4..7}, {8..11}.
    const int tile = lane & 31, oct = lane >> 5, ty = tile >> 3, tx = tile & 7;
    const int rbase = ((4 * ty + rh) * RS + 17 * tx + (PACK && tx >= 4 ? 8 : 0) + 2 * oct) * 16;
    // patch column CH + m of the tile: slot 4 (CH + m) + ((CH + m) >> 2)
#define W4B_RD(buf, j, a, m) (*reinterpret_cast<const f32x4*>(smem + (buf) * W4B_RAW_BYTES + rbase + ((a) * RS + 4 * (CH + (m)) + ((CH + (m)) >> 2) + (j)) * 16))

    // the LDS-resident accumulator blocks (this lane's 16 registers of slot k as 4 x 16 B)
    unsigned char* const accsp = smem + W4B_ACC_OFF + wave * (W4B_NLDS * 4096) + lane * 16;
#pragma unroll
    for (int i = 0; i < 4 * W4B_NLDS; ++i) *reinterpret_cast<f32x4*>(accsp + i * 1024) = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 ctA, ctB, ctC, ctD;                           // the LDS-resident blocks (units 3, 9, 13, 17) while they are in registers
    // ---- the multiply phase is HAND PLACED (sched_barrier after every MFMA): one wavefront per SIMD hides at most six single-issue vector
    // instructions under a v_mfma_f32_32x32x16_bf16 and NO packed-fp32 one (v_pk_*_f32 beside an MFMA costs ~10 cycles each:
    // scratch/micro/valu_fill.hip), and hipcc left to itself clumps the split's arithmetic between runs of back-to-back MFMAs and SLP-packs
    // the subtractions.  A point = the two units (cout blocks 0 / 1) on the same V fragments: six products each, smallest first, the twelve
    // MFMAs alternating between the two accumulator blocks.  Gap q (after MFMA q) carries
    //   * ops [S(q), S(q+1)) of the 44-instruction split of the NEXT point's eight V values (level-major over the four channel pairs, so
    //     consecutive instructions are independent),
    //   * the refill of a filter fragment of THIS point as soon as its last MFMA has issued (term 2 after q = 2 / 3, term 1 after 6 / 7,
    //     term 0 after 10 / 11),
    //   * one LDS instruction of the PREVIOUS point's accumulator swap (its MFMAs have long finished: no dependency stall).
#if defined(W4B_EXP) && (W4B_EXP & 32)
    unsigned hmask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(hmask));
#endif
    auto split_op = [&](const int idx, float (&x)[8], unsigned (&h)[8], u32x4* f) {
        const int lvl = idx < 4 ? 0 : idx < 12 ? 1 : idx < 20 ? 2 : idx < 24 ? 3 : idx < 32 ? 4 : idx < 40 ? 5 : 6;
        const int k = idx - (lvl == 0 ? 0 : lvl == 1 ? 4 : lvl == 2 ? 12 : lvl == 3 ? 20 : lvl == 4 ? 24 : lvl == 5 ? 32 : 40);
        if (lvl == 0 || lvl == 3 || lvl == 6) f[lvl / 3][k] = w4b_pack_hi(w4b_bits(x[2 * k]), w4b_bits(x[2 * k + 1]));
#if defined(W4B_EXP) && (W4B_EXP & 32)
        else if (lvl == 1 || lvl == 4) h[k] = w4b_bits(x[k]) & hmask;
#else
        else if (lvl == 1 || lvl == 4) h[k] = w4b_bits(x[k]) & 0xffff0000u;
#endif
        else asm("v_sub_f32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(h[k]));      // (asm: never SLP-packed; exact -- the difference fits)
    };
    auto split_all = [&](const float* v, u32x4* f) {     // a split with no MFMAs to hide under (the first point of a K step)
        float x[8]; unsigned h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = v[e];
#if defined(W4B_EXP) && (W4B_EXP & 1)
        w4b_split8(v, f); return;
#endif
#pragma unroll
        for (int i = 0; i < 44; ++i) split_op(i, x, h, f);
    };
    // u: the point's even unit (accumulators c0 / c1); a: its V fragments; xn / fn: the next point's V values and fragment buffer (or null);
    // st / st_slot: an LDS-resident block to put back (the previous point's), ld / ld_slot: the one to fetch for the NEXT point -- each such block
    // lives in registers for three points only, and never two of them at once (the stores ride in gaps 1-4, the loads in gaps 5-8)
    auto point = [&](const int u, f32x16& c0, f32x16& c1, const u32x4* a, const float* xn, u32x4* fn,
                     const f32x16* st, const int st_slot, f32x16* ld, const int ld_slot, const int koff) {
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
        constexpr int S[13] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 35, 38, 41, 44};
        const int sl0 = u % W4B_RING, sl1 = (u + 1) % W4B_RING;
        // refill sources: unit + 6 of this K step, or unit - 12 of the next one (scalar base; the term's 1 KB steps ride in the immediate offset)
        const int un0 = u + W4B_RING < 18 ? u + W4B_RING : u + W4B_RING - 18, un1 = u + 1 + W4B_RING < 18 ? u + 1 + W4B_RING : u + 1 + W4B_RING - 18;
        const int so0 = ubase[un0 >> 1] + (un0 & 1) * 3072 + (u + W4B_RING < 18 ? koff : koff + ustep);
        const int so1 = ubase[un1 >> 1] + (un1 & 1) * 3072 + (u + 1 + W4B_RING < 18 ? koff : koff + ustep);
        float x[8]; unsigned h[8];
        if (xn) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = xn[e];
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int pr = q >> 1;
            if (q & 1) {
                if (W4B_VFORM1(u + 1)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a[PA[pr]]), "v"(ub[sl1][PB[pr]]));
                else c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[pr]]), __builtin_bit_cast(bf16x8, ub[sl1][PB[pr]]), c1, 0, 0, 0);
            }
            else if (W4B_VFORM(u)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a[PA[pr]]), "v"(ub[sl0][PB[pr]]));
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[pr]]), __builtin_bit_cast(bf16x8, ub[sl0][PB[pr]]), c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#if defined(W4B_EXP) && (W4B_EXP & 1)
#else
            if (xn) {
#pragma unroll
                for (int i = S[q]; i < S[q + 1]; ++i) split_op(i, x, h, fn);
            }
#endif
#if defined(W4B_EXP) && (W4B_EXP & 2)
#else
            if (q == 3) ub[sl0][2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane + 2048, so0, W4B_U_AUX));
            if (q == 4) ub[sl1][2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane + 2048, so1, W4B_U_AUX));
            if (q == 7) ub[sl0][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane + 1024, so0, W4B_U_AUX));
            if (q == 8) ub[sl1][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane + 1024, so1, W4B_U_AUX));
            if (q == 11) {
                ub[sl0][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane, so0, W4B_U_AUX));
                ub[sl1][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane, so1, W4B_U_AUX));
            }
#endif
            if (st && q >= 1 && q <= 4) {
                const int i = q - 1;
                *reinterpret_cast<f32x4*>(accsp + st_slot * 4096 + i * 1024) = f32x4{(*st)[4 * i], (*st)[4 * i + 1], (*st)[4 * i + 2], (*st)[4 * i + 3]};
            }
            if (ld && q >= 5 && q <= 8) {
                const int i = q - 5;
                const f32x4 v = *reinterpret_cast<const f32x4*>(accsp + ld_slot * 4096 + i * 1024);
#pragma unroll
                for (int e = 0; e < 4; ++e) (*ld)[4 * i + e] = v[e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // second transform pass over the five column sums t[m] (column CH + m) of one point row: the three points of this wavefront's column half
    //   single (B^T row 0 / 5): 4 t0 - 5 t2 + t4 ;  pair, CH = 0 (rows 1, 2 of B^T on columns 1..4): (t4 - 4 t2) +- (t3 - 4 t1)
    //                                               pair, CH = 1 (rows 3, 4 on columns 1..4 = t0..t3): (t3 - t1) +- 2 (t2 - t0)
    const unsigned long long k4 = w4b_c2(4.f), km5 = w4b_c2(-5.f), km4 = w4b_c2(-4.f), k2 = w4b_c2(2.f), km2 = w4b_c2(-2.f);
    auto second = [&](auto chc, const f32x4* t, f32x4* o) {
        constexpr int CH = decltype(chc)::value;
        o[0] = w4b_fma4(km5, t[2], w4b_fma4(k4, t[0], t[4]));
        if (CH == 0) {
            const f32x4 u = w4b_fma4(km4, t[2], t[4]), v = w4b_fma4(km4, t[1], t[3]);
            o[1] = w4b_add4(u, v); o[2] = w4b_sub4(u, v);
        } else {
            const f32x4 u = w4b_sub4(t[3], t[1]), v = w4b_sub4(t[2], t[0]);
            o[1] = w4b_fma4(k2, v, u); o[2] = w4b_fma4(km2, v, u);
        }
    };


    // (Walking the K steps in an order rotated from tile to tile, so that the workgroups of a round do not all read the same few KB of U at
    //  the same time, was measured: no change -- the L2 channels are not the limit.  Every tile sums its channels in the same order.)
    //
    // Schedule of a K step: request the next step's patch (11 LDS-DMA pieces), transform all nine points (T), split point 0, the nine points'
    // 108 MFMAs as one hand-placed stream (M), counted wait + barrier.  Measured alternatives, not kept: a rotated loop with T of the next
    // point group beside M of the previous one spills 11-36 registers per iteration to scratch; the DMA pieces between the reads of T (eleven
    // more registers live at T's peak: scratch reloads with vmcnt(0)), or two steps ahead in the gaps of points 0..2 with the one barrier after
    // T (+3 %: a filter fragment loaded behind a piece waits for that piece) -- DESIGN.md 4.2b.
    // both row groups in ONE pass over the patch (rows RH .. RH + 4 read once: 50 instead of 70 ds_read_b128 per K step -- the four wavefronts'
    // transform phases coincide and run at the LDS read bandwidth): v[0..2] the `single` row's points, v[3..8] the pair rows'
    auto t_all = [&](auto rhc, auto chc, int buf, float (&v)[9][8], const int dstep) {
        constexpr int RH = decltype(rhc)::value, CH = decltype(chc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 ts[5], t1[5], t2[5];
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const f32x4 e0 = W4B_RD(buf, j, 0, m), e1 = W4B_RD(buf, j, 1, m), e2 = W4B_RD(buf, j, 2, m), e3 = W4B_RD(buf, j, 3, m), e4 = W4B_RD(buf, j, 4, m);
#if W4B_DMA_IN_T
                dma_piece(dstep, buf ^ 1, 5 * j + m);     // the next K step's patch request rides between the column groups of the transform
                if (j == 1 && m == 4) dma_piece(dstep, buf ^ 1, 10);
#endif
                ts[m] = w4b_fma4(km5, e2, w4b_fma4(k4, e0, e4));
                if (RH == 0) {                           // pair rows on patch rows 1..4
                    const f32x4 u = w4b_fma4(km4, e2, e4), w = w4b_fma4(km4, e1, e3);
                    t1[m] = w4b_add4(u, w); t2[m] = w4b_sub4(u, w);
                } else {                                 // patch rows 1..4 = relative rows 0..3
                    const f32x4 u = w4b_sub4(e3, e1), w = w4b_sub4(e2, e0);
                    t1[m] = w4b_fma4(k2, w, u); t2[m] = w4b_fma4(km2, w, u);
                }
            }
            f32x4 o0[3], o1[3], o2[3];
            second(chc, ts, o0);
            second(chc, t1, o1);
            second(chc, t2, o2);
#pragma unroll
            for (int ck = 0; ck < 3; ++ck)
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[ck][4 * j + q] = o0[ck][q]; v[3 + ck][4 * j + q] = o1[ck][q]; v[6 + ck][4 * j + q] = o2[ck][q]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: the first K step's patch, the first six filter units
    dma(0, 0);
#pragma unroll
    for (int u = 0; u < W4B_RING; ++u) uload(u, u, 0);
    asm volatile("s_waitcnt vmcnt(" W4B_STR(W4B_RING3) ")" ::: "memory");     // the 11 DMA pieces are older than the 18 fragment loads
    __builtin_amdgcn_s_barrier();
    W4B_STAMP(1);

    for (int it = 0; it < KS; ++it) {
        const int buf = it & 1;
        const int koff = it * ustep;
        W4B_KSTAMP(it, 0);
#if defined(W4B_EXP) && (W4B_EXP & 8)
#else
#if !W4B_DMA_IN_T
        if (it + 1 < KS) dma(it + 1, buf ^ 1);
#endif
#endif
        W4B_KSTAMP(it, 1);
        __builtin_amdgcn_sched_barrier(0);
        float v[9][8];                                   // V of this wavefront's nine points, this lane's 8 channels
        {   // the transform is the only role-dependent code of the K loop: the multiply phase below is ONE instruction stream for the four
            // wavefronts (four copies of it cost 9 % in instruction fetch: with every wavefront on the same role the launch takes 0.0955
            // instead of 0.1049 ms)
            const std::integral_constant<int, 0> i0; const std::integral_constant<int, 1> i1;
            if (rh == 0) { if (ch == 0) t_all(i0, i0, buf, v, it + 1); else t_all(i0, i1, buf, v, it + 1); }
            else { if (ch == 0) t_all(i1, i0, buf, v, it + 1); else t_all(i1, i1, buf, v, it + 1); }
        }
        W4B_KSTAMP(it, 2);
        u32x4 af[2][3];
        split_all(v[0], af[0]);
        __builtin_amdgcn_sched_barrier(0);
        W4B_KSTAMP(it, 3);
        W4B_KSTAMP(it, 4);
        // ---- the nine points, units 0..17: every point's gaps carry the next point's split
        point(0, accr[w4b_reg_block(0)], accr[w4b_reg_block(1)], af[0], v[1], af[1], nullptr, 0, &ctA, 0, koff);
        point(2, accr[w4b_reg_block(2)], ctA, af[1], v[2], af[0], nullptr, 0, nullptr, 0, koff);
        point(4, accr[w4b_reg_block(4)], accr[w4b_reg_block(5)], af[0], v[3], af[1], &ctA, 0, nullptr, 0, koff);
        point(6, accr[w4b_reg_block(6)], accr[w4b_reg_block(7)], af[1], v[4], af[0], nullptr, 0, &ctB, 1, koff);
        point(8, accr[w4b_reg_block(8)], ctB, af[0], v[5], af[1], nullptr, 0, nullptr, 0, koff);
        point(10, accr[w4b_reg_block(10)], accr[w4b_reg_block(11)], af[1], v[6], af[0], &ctB, 1, &ctC, 2, koff);
        point(12, accr[w4b_reg_block(12)], ctC, af[0], v[7], af[1], nullptr, 0, nullptr, 0, koff);
        point(14, accr[w4b_reg_block(14)], accr[w4b_reg_block(15)], af[1], v[8], af[0], &ctC, 2, &ctD, 3, koff);
        point(16, accr[w4b_reg_block(16)], ctD, af[0], nullptr, nullptr, nullptr, 0, nullptr, 0, koff);
        // (the last MFMA on ctD is asm: the compiler does not pad the MFMA-result -> LDS-data hazard behind it.  20 wait states cover a 16-pass MFMA.)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(accsp + 3 * 4096 + i * 1024) = f32x4{ctD[4 * i], ctD[4 * i + 1], ctD[4 * i + 2], ctD[4 * i + 3]};
        W4B_KSTAMP(it, 5);
        if (it + 1 < KS) {
            asm volatile("s_waitcnt vmcnt(" W4B_STR(W4B_RING3) ")" ::: "memory");         // the next patch has landed: only the 18 fragment loads issued after its DMA may be in flight
            W4B_KSTAMP(it, 6);
            __builtin_amdgcn_s_barrier();
        }
    }
    W4B_STAMP(2);
#undef W4B_RD
}

template <int STATS, bool PACK = false>
__global__ __launch_bounds__(256, 1) void conv_wino43b_kernel(const Wino43bGeom g) {
    static_assert(!PACK || STATS == 0, "packed 16 x 16 images: no statistics epilogue (kpx_conv3x3_wino43_stats_tiles is 0 for them)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int ntc = g.Np / 64;
    const int nti = L % ntc; L /= ntc;
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = PACK ? 2 * (L / g.tiles_y) : L / g.tiles_y;      // PACK: tiles_y = tiles_x = 1, the region holds images n and n + 1
    const int oy0 = by * 16, ox0 = bx * 32, n0 = nti * 64;

    f32x16 accr[14];                                     // the register-resident accumulator blocks (w4b_reg_block)
#pragma unroll
    for (int b = 0; b < 14; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accr[b][r] = 0.f;

#ifdef KPX_W4B_STAMP
    unsigned long long* const dbgp = (w4b_dbg && lane == 0 && blockIdx.x < 64) ? w4b_dbg + ((size_t)blockIdx.x * 4 + wave) * 512 : nullptr;
    if (dbgp) { dbgp[0] = __builtin_amdgcn_s_memtime(); dbgp[8] = __builtin_amdgcn_s_memrealtime(); }
#else
    unsigned long long* const dbgp = nullptr;
#endif
    const int rh = wave >> 1, ch = wave & 1;
    w4b_kloop<PACK>(rh, ch, g, accr, smem, lane, wave, n, oy0, ox0, nti, dbgp);

    // ---- epilogue: two passes (one per 32-cout block) of all 36 points through LDS; thread = (tile, 4 couts)
    f32x16 accl[W4B_NLDS];                               // the LDS-resident blocks, before the passes overwrite them
    {
        const unsigned char* const accsp = smem + W4B_ACC_OFF + wave * (W4B_NLDS * 4096) + lane * 16;
#pragma unroll
        for (int b = 0; b < W4B_NLDS; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(accsp + (b * 4 + i) * 1024);
#pragma unroll
                for (int q = 0; q < 4; ++q) accl[b][4 * i + q] = v[q];
            }
    }
    float* const P = reinterpret_cast<float*>(smem);
    const int li = lane & 31, lh = lane >> 5;
    const int otile = t >> 3, ocq = t & 7;
    const float* const Pr = P + otile * 32 + ocq * 4;
    const float lo = g.act == KPX_ACT_RELU ? 0.f : -__builtin_inff();
    const float slope = g.act == KPX_ACT_LRELU ? 0.01f : 1.f;
    const int oy = oy0 + 4 * (otile >> 3), ox = PACK ? 4 * (otile & 3) : ox0 + 4 * (otile & 7);
    const int on = PACK ? n + ((otile >> 2) & 1) : n;    // the image of this thread's tile
    const size_t cstr = (size_t)g.ldy, rstr = (size_t)g.W * g.ldy;
#pragma unroll 1
    for (int nb = 0; nb < 2; ++nb) {
        __syncthreads();                                 // main-loop LDS reads (nb = 0) / the first pass's P reads (nb = 1) are done
        if (nb == 0) W4B_STAMP(3);
#pragma unroll
        for (int lp = 0; lp < 9; ++lp) {
            const int p = 6 * w4b_grid(rh, lp / 3) + w4b_grid(ch, lp % 3);
            float* const Pw = P + (p * 32 + 4 * lh) * 32 + li;
            // register r of a block is tile row (r & 3) + 8 (r >> 2) + 4 lh, 128 B apart: registers r and r + 2 go out as ONE
            // ds_write2st64_b32 (two dwords 256 B apart) -- hipcc pairs the stores of a block held in VGPRs by itself, not those of one in AGPRs
            const int u = 2 * lp + nb;                    // (nb is a run-time loop variable: both forms are written out)
#define W4B_DEP(blk, cons) do { \
                const unsigned pa = (unsigned)(size_t)(lds_ptr_t)Pw, pb = pa + 128u; \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:1" :: "v"(pa), cons((blk)[0]), cons((blk)[2]) : "memory"); \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:1" :: "v"(pb), cons((blk)[1]), cons((blk)[3]) : "memory"); \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:4 offset1:5" :: "v"(pa), cons((blk)[4]), cons((blk)[6]) : "memory"); \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:4 offset1:5" :: "v"(pb), cons((blk)[5]), cons((blk)[7]) : "memory"); \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:8 offset1:9" :: "v"(pa), cons((blk)[8]), cons((blk)[10]) : "memory"); \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:8 offset1:9" :: "v"(pb), cons((blk)[9]), cons((blk)[11]) : "memory"); \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:12 offset1:13" :: "v"(pa), cons((blk)[12]), cons((blk)[14]) : "memory"); \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:12 offset1:13" :: "v"(pb), cons((blk)[13]), cons((blk)[15]) : "memory"); \
            } while (0)
            if (nb == 0) {
                if (w4b_lds_slot(2 * lp) >= 0) W4B_DEP(accl[w4b_lds_slot(2 * lp) & 3], "v");
                else if (W4B_VFORM(2 * lp)) W4B_DEP(accr[w4b_reg_block(2 * lp)], "v");
                else W4B_DEP(accr[w4b_reg_block(2 * lp)], "a");
            } else {
                if (w4b_lds_slot(2 * lp + 1) >= 0) W4B_DEP(accl[w4b_lds_slot(2 * lp + 1) & 3], "v"); else W4B_DEP(accr[w4b_reg_block(2 * lp + 1)], "a");
            }
            (void)u;
#undef W4B_DEP
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the deposits are asm: the compiler's own wait before the barrier does not count them)
        if (nb == 0) W4B_STAMP(4);
        __syncthreads();
        if (nb == 0) W4B_STAMP(5);
        // Q[a][jj] = sum_b M[a][b] A[b][jj], then Y[ii][jj] = sum_a A^T[ii][a] Q[a][jj]
        f32x4 Q[6][4];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            f32x4 m[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) m[b] = *reinterpret_cast<const f32x4*>(&Pr[(a * 6 + b) * 1024]);
            const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            Q[a][0] = m[0] + s12 + s34;
            Q[a][1] = d12 + 2.f * d34;
            Q[a][2] = s12 + 4.f * s34;
            Q[a][3] = d12 + 8.f * d34 + m[5];
        }
        f32x4 Y[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x4 s12 = Q[1][jj] + Q[2][jj], d12 = Q[1][jj] - Q[2][jj], s34 = Q[3][jj] + Q[4][jj], d34 = Q[3][jj] - Q[4][jj];
            Y[0][jj] = Q[0][jj] + s12 + s34;
            Y[1][jj] = d12 + 2.f * d34;
            Y[2][jj] = s12 + 4.f * s34;
            Y[3][jj] = d12 + 8.f * d34 + Q[5][jj];
        }
        if (nb == 0) W4B_STAMP(6);
        const int c0o = n0 + 32 * nb + ocq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (c0o + q < g.Cout) bv[q] = g.bias[c0o + q];
        }
        float* const obase = g.y + ((size_t)(on * g.H + oy) * g.W + ox) * g.ldy + c0o;
        const bool fast = (g.ldy & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.y) & 15) == 0) && n0 + 32 * nb + 32 <= g.Cout;     // block-uniform
        f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};
        if (STATS == 2) {
            // data gradient dz of a ReLU'd batch norm's output z (= mask_y): store dz * [z > 0] and reduce sum(dz), sum(dz * (z - beta))
            // (launch preconditions checked by the entry: fast stores, no bias / act)
            const float* const mbase = g.mask_y + ((size_t)(on * g.H + oy) * g.W + ox) * g.ld_mask + c0o;
            const size_t mc = (size_t)g.ld_mask, mr = (size_t)g.W * g.ld_mask;
            const f32x4 be = *reinterpret_cast<const f32x4*>(g.bn_beta + c0o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 zm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) zm[j] = *reinterpret_cast<const f32x4*>(mbase + i * mr + j * mc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dz = zm[j][q] > 0.f ? Y[i][j][q] : 0.f;
                        v[q] = dz; st_s[q] += dz; st_q[q] += dz * (zm[j][q] - be[q]);
                    }
                    *reinterpret_cast<f32x4*>(obase + i * rstr + j * cstr) = v;
                }
            }
        } else if (fast && !STATS && (g.mask_y || g.pool_y)) {       // block-uniform: VGG19's fused ReLU backward / max-pool forward
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = Y[i][j] + bv;
                    if (g.act == KPX_ACT_RELU) {          // (block-uniform; VGG19's forward: one max per element)
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                    } else if (g.act == KPX_ACT_LRELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { const float z = v[q]; v[q] = z > 0.f ? z : z * slope; }
                    }
                    Y[i][j] = v;
                }
            if (g.mask_y) {
                const float* const mbase = g.mask_y + ((size_t)(on * g.H + oy) * g.W + ox) * g.ld_mask + c0o;
                const size_t mc = (size_t)g.ld_mask, mr = (size_t)g.W * g.ld_mask;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 ym = *reinterpret_cast<const f32x4*>(mbase + i * mr + j * mc);
#pragma unroll
                        for (int q = 0; q < 4; ++q) Y[i][j][q] = ym[q] > 0.f ? Y[i][j][q] : 0.f;
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(obase + i * rstr + j * cstr) = Y[i][j];
            if (g.pool_y) {
                const int Hp = g.H >> 1, Wp = g.W >> 1;
                float* const pbase = g.pool_y + ((size_t)(on * Hp + (oy >> 1)) * Wp + (ox >> 1)) * g.ld_pool + c0o;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 pv;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            pv[q] = fmaxf(fmaxf(Y[2 * i][2 * j][q], Y[2 * i][2 * j + 1][q]), fmaxf(Y[2 * i + 1][2 * j][q], Y[2 * i + 1][2 * j + 1][q]));
                        *reinterpret_cast<f32x4*>(pbase + ((size_t)i * Wp + j) * g.ld_pool) = pv;
                    }
            }
        } else if (fast && g.act != KPX_ACT_LRELU) {
            // no activation / ReLU (block-uniform): bias + one max per element -- the general form below spends 19 vector instructions per 16-byte
            // store on the leaky slope (compare, select, multiply per element), 600 per tile
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = Y[i][j] + bv;
                    if (g.act == KPX_ACT_RELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                    }
                    if (STATS) { st_s += v; st_q += v * v; }
                    *reinterpret_cast<f32x4*>(obase + i * rstr + j * cstr) = v;
                }
        } else if (fast) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = Y[i][j] + bv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float z = fmaxf(v[q], lo); v[q] = z > 0.f ? z : z * slope; }
                    if (STATS) { st_s += v; st_q += v * v; }
                    *reinterpret_cast<f32x4*>(obase + i * rstr + j * cstr) = v;
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = Y[i][j] + bv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float z = fmaxf(v[q], lo); v[q] = z > 0.f ? z : z * slope; }
                    if (STATS) { st_s += v; st_q += v * v; }
                    float* o = obase + i * rstr + j * cstr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (c0o + q < g.Cout) o[q] = v[q];
                }
        }
        if (STATS != 0) {
            // batch-norm statistics per 4 x 16-pixel strip (4 tiles: lane bits 3-4), fixed butterfly order: bitwise reproducible.
            // A wavefront's 64 threads = 8 tiles x 8 cout quads = two strips (2 wave + (lane >> 5)) x 32 couts.
#pragma unroll
            for (int sh = 8; sh <= 16; sh <<= 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) { st_s[q] += __shfl_xor(st_s[q], sh); st_q[q] += __shfl_xor(st_q[q], sh); }
            if ((lane & 24) == 0) {
                const size_t strip = (((size_t)n * g.tiles_y + by) * g.tiles_x + bx) * 8 + 2 * wave + lh;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c0o + q < g.Cout) { g.stats[(strip * 2) * g.Cout + c0o + q] = st_s[q]; g.stats[(strip * 2 + 1) * g.Cout + c0o + q] = st_q[q]; }
            }
        }
        if (nb == 0) W4B_STAMP(10);
    }
    W4B_STAMP(7);
#ifdef KPX_W4B_STAMP
    if (dbgp) dbgp[9] = __builtin_amdgcn_s_memrealtime();
#endif
}

static std::atomic<unsigned long long> w4b_attr_mask{0};
static inline int w4b_lds_bytes() { return W4B_EPI_BYTES > W4B_MAIN_BYTES ? W4B_EPI_BYTES : W4B_MAIN_BYTES; }

extern "C" int kpx_conv3x3_wino43b_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr) {
    if (kpx_env()->no_wino || N <= 0) return 0;
    const bool shape = (H % 16 == 0 && W % 32 == 0) || (H == 16 && W == 16 && N % 2 == 0);      // 16 x 16 images are packed two to a workgroup
    return shape && K >= 16 && K % 4 == 0 && Nn >= 33 && ldin >= K && ldin % 4 == 0 && (((uintptr_t)in_ptr) & 15) == 0 &&
           (size_t)H * W * ldin * 8 < 0x7fffff00u && (size_t)36 * ((K + 15) & ~15) * ((Nn + 63) & ~63) * 6 < 0x7fffff00u;
}
extern "C" size_t kpx_wino43b_u_bytes(int Cin, int Cout) {
    const size_t a = (size_t)((Cin + 15) & ~15) * ((Cout + 63) & ~63), b = (size_t)((Cout + 15) & ~15) * ((Cin + 63) & ~63);
    return 36 * 6 * (a > b ? a : b);
}
extern "C" int kpx_wino43b_filter_transform_f32(const float* w_hwio, int Cin, int Cout, int dgrad, void* U, void* stream) {
    if (!w_hwio || !U || Cin <= 0 || Cout <= 0) return KPX_EINVAL;
    const int K = dgrad ? Cout : Cin, Nn = dgrad ? Cin : Cout;
    const int Kp = (K + 15) & ~15, Np = (Nn + 63) & ~63;
    size_t nb = ((size_t)6 * (Kp / 8) * Np + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(wino43b_filter_transform_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), w_hwio, Cin, Cout, dgrad, Kp, Np, (unsigned*)U);
    return kpx_launch_status();
}
extern "C" int kpx_wino43b_filter_transform_batch_f32(const void* descs_dev, int n, void* stream) {
    if (!descs_dev || n <= 0 || n > 65535) return KPX_EINVAL;
    hipLaunchKernelGGL(wino43b_filter_transform_batch_kernel, dim3(96, (unsigned)n), dim3(256), 0, kpx_stream(stream), (const KpxWino43bDesc*)descs_dev);
    return kpx_launch_status();
}

// One entry for every form of the launch (the options are those of kpx_conv3x3_wino43_f32 / _stats_f32 / _ex_f32 / _bnbwd_stats_f32):
//   tile_stats alone: batch-norm sums of the output per 4 x 16-pixel strip;  tile_stats + bn_y + bn_beta: data gradient towards a ReLU'd batch
//   norm's output, masked, with that batch norm's backward sums;  mask_y / pool_y: VGG19's fused ReLU backward / 2x2 max-pool.
extern "C" int kpx_conv3x3_wino43b_f32(const float* in, int N, int H, int W, int K, int ldin, const void* U, const float* bias,
                                       float* out, int Nn, int ldout, int act, const float* mask_y, int ld_mask, float* pool_y, int ld_pool,
                                       float* tile_stats, const float* bn_y, int ld_bn_y, const float* bn_beta, void* stream) {
    if (!in || !U || !out || ldin < K || ldout < Nn || act < 0 || act > 2 || !kpx_conv3x3_wino43b_eligible(N, H, W, K, Nn, ldin, in)) return KPX_EINVAL;
    const bool bnbwd = bn_y != nullptr || bn_beta != nullptr;
    if (bnbwd && (!tile_stats || !bn_y || !bn_beta || mask_y || pool_y || bias || act != KPX_ACT_NONE || Nn % 64 || ldout % 4 || (((uintptr_t)out) & 15) ||
                  ld_bn_y % 4 || ld_bn_y < Nn || (((uintptr_t)bn_y) & 15) || (((uintptr_t)bn_beta) & 15)))
        return KPX_EINVAL;
    if ((mask_y || pool_y) && (tile_stats || Nn % 64 || ldout % 4 || (((uintptr_t)out) & 15) ||
                               (mask_y && (ld_mask % 4 || ld_mask < Nn || (((uintptr_t)mask_y) & 15))) ||
                               (pool_y && (ld_pool % 4 || ld_pool < Nn || (((uintptr_t)pool_y) & 15) || (H & 1) || (W & 1)))))
        return KPX_EINVAL;
    if (kpx_first_use_on_device(&w4b_attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43b_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, w4b_lds_bytes());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43b_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, w4b_lds_bytes());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43b_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, w4b_lds_bytes());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43b_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, w4b_lds_bytes());
        if (e != hipSuccess) return -(int)e;
    }
    Wino43bGeom g{};
    g.x = in; g.y = out; g.U = U; g.bias = bias;
    g.N = N; g.H = H; g.W = W; g.Cin = K; g.ldx = ldin; g.Cout = Nn; g.ldy = ldout; g.act = act;
    g.Kp = (K + 15) & ~15; g.Np = (Nn + 63) & ~63;
    const bool pack = W == 16;
    if (pack && tile_stats) return KPX_EINVAL;           // (no statistics epilogue for packed images: kpx_conv3x3_wino43_stats_tiles is 0 for them)
    g.tiles_y = H / 16; g.tiles_x = pack ? 1 : W / 32;
    g.stats = tile_stats;
    g.mask_y = bnbwd ? bn_y : mask_y; g.ld_mask = bnbwd ? ld_bn_y : ld_mask; g.pool_y = pool_y; g.ld_pool = ld_pool; g.bn_beta = bn_beta;
    const unsigned blocks = (unsigned)((size_t)(pack ? N / 2 : N) * g.tiles_y * g.tiles_x * (g.Np / 64));
    if (pack) hipLaunchKernelGGL((conv_wino43b_kernel<0, true>), dim3(blocks), dim3(256), w4b_lds_bytes(), kpx_stream(stream), g);
    else if (bnbwd) hipLaunchKernelGGL((conv_wino43b_kernel<2>), dim3(blocks), dim3(256), w4b_lds_bytes(), kpx_stream(stream), g);
    else if (tile_stats) hipLaunchKernelGGL((conv_wino43b_kernel<1>), dim3(blocks), dim3(256), w4b_lds_bytes(), kpx_stream(stream), g);
    else hipLaunchKernelGGL((conv_wino43b_kernel<0>), dim3(blocks), dim3(256), w4b_lds_bytes(), kpx_stream(stream), g);
    return kpx_launch_status();
}
