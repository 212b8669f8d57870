// fp32-equivalent implicit-GEMM convolution on the bf16 matrix pipe of gfx950 (MI355X): forward and data gradient of the layers the
// Winograd kernels cannot take (strided, 4x4, 1x1: the discriminator, the encoders' stride-2 layers, the key-point head), through the same
// geometry (conv_geom.h) and the same C entry points as the fp32 kernel of conv_igemm.hip (kpx_conv2d_fwd_f32 / kpx_conv2d_dgrad_f32).
//
// Arithmetic.  Every fp32 operand is split EXACTLY into three bf16 terms while its tile is staged in LDS:
//     a = a0 + a1 + a2,   a0 = trunc16(a),  a1 = trunc16(a - a0),  a2 = a - a0 - a1
// (a - a0 has at most 16 significant bits and a - a0 - a1 at most 8, so both subtractions are exact and a2 is a bf16 number: no bit of
// the fp32 value is lost).  The product a*b = sum_ij ai*bj is accumulated with the six terms of weight >= 2^-16,
//     a0*b0 + (a0*b1 + a1*b0) + (a1*b1 + a0*b2 + a2*b0),
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; each bf16 x bf16 product is exact in fp32.  The dropped terms (a1*b2, a2*b1, a2*b2)
// are below 2^-23 |a*b| -- smaller than the rounding of one fp32 accumulation.  Measured against a float64 convolution the result is as
// close as the v_mfma_f32_32x32x2_f32 kernel's (tests/test_ops_gpu.py::test_gemm3_*): this is the fp32 configuration's arithmetic, not the
// bf16 mode.  Six bf16 MFMAs cost 6 x 32 = 192 cycles per 32x32x16 block against 8 x 64 = 512 for the fp32 MFMA, and -- unlike the fp32
// MFMA, which holds its SIMD's issue port -- the bf16 MFMA lets the splitting VALU work and the LDS traffic issue beside it.
// TERMS = 1 keeps only a0*b0: the bf16 configuration's arithmetic (BASELINE configs[2]) from the same kernel.
//
// One workgroup = 8 wavefronts (WM x WN) = BM x BN outputs as 32x32 accumulators; K chunks of 32 channels.
// LDS per stage: TERMS planes of A [BM][32] bf16 and of B [BN][32] bf16 (k contiguous: one ds_read_b128 = one MFMA operand of a lane) at a
// row pitch of 80 bytes (conflict-free reads and writes, see G3_PITCH).
#include "conv_geom.h"
#include "kpx_env.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned g3_bits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float g3_float(unsigned v) { return __builtin_bit_cast(float, v); }
// {hi16(b), hi16(a)}: two truncated bf16 values in one dword, a in the low half
__device__ __forceinline__ unsigned g3_pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// exact three-way split of two floats; p[t] = packed (term t of a, term t of b)
template <int TERMS>
__device__ __forceinline__ void g3_split2(float a, float b, unsigned* p) {
    const unsigned ua = g3_bits(a), ub = g3_bits(b);
    p[0] = g3_pack_hi(ua, ub);
    if (TERMS > 1) {
        const float ra = a - g3_float(ua & 0xffff0000u), rb = b - g3_float(ub & 0xffff0000u);
        const unsigned va = g3_bits(ra), vb = g3_bits(rb);
        p[1] = g3_pack_hi(va, vb);
        if (TERMS > 2) {
            const float sa = ra - g3_float(va & 0xffff0000u), sb = rb - g3_float(vb & 0xffff0000u);
            p[2] = g3_pack_hi(g3_bits(sa), g3_bits(sb));
        }
    }
}

template <int V> struct g3_ic { static constexpr int value = V; };

#define G3_PITCH 80            // bytes of one LDS row (32 bf16 + 16 pad): 80 = 5 x 16 and 5 is odd, so sixteen consecutive rows start in sixteen
                               // different 16-B slots of the 256-B bank window -- ds_read_b128 fragment reads, the 8-lane ds_write_b128 groups and
                               // the 16-lane ds_write_b64 groups of the transposing B store are all conflict-free without an XOR swizzle
#define G3_OOB 0x7ffffff0      // voffset beyond any buffer: the load returns zeros (out-of-image taps, channel tails, chunks past the end)

// IO16 (bf16 configuration, TERMS == 1): the gathered tensor is bf16 in HBM -- a staging unit is ONE 16-byte load of eight channels and
// goes to LDS as it is; the produced tensor is bf16 when g.io16 says so.
template <int BM, int BN, int WM, int WN, bool BT, int TERMS, bool IO16 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm3_kernel(const ConvGeom g) {
    static_assert(!IO16 || TERMS == 1, "bf16 tensors carry one term");
    constexpr int NT = WM * WN * 64, BK = 32, ESZ = IO16 ? 2 : 4;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int APL = BM * G3_PITCH, BPL = BN * G3_PITCH;        // bytes of one bf16 plane
    constexpr int ASZ = TERMS * APL;
    // staging units, RA / RB per thread (no guards, no branches in the chunk body):
    //   A: (row, 8-channel octet);  B read k-contiguous (dgrad): KU consecutive k of one n row;  B read n-contiguous (forward): 2 k x CW n,
    //   transposed in registers -- each n becomes one bf16 pair.
    constexpr int RA = BM * 4 / NT;
    constexpr int KU = BN * 4 >= NT ? 8 : 4, CW = BN * 4 >= NT ? 4 : 2;
    constexpr int RB = BT ? BN * 32 / KU / NT : 16 * BN / CW / NT;
    static_assert(TM >= 1 && TN >= 1 && BM % (WM * 32) == 0 && BN % (WN * 32) == 0 && RA >= 1 && RB >= 1, "tile shape");
    static_assert(RA * NT == BM * 4 && RB * NT == (BT ? BN * 32 / KU : 16 * BN / CW), "whole units per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [A planes][B planes] (one stage)
    unsigned char* const Asm = smem;
    unsigned char* const Bsm = smem + ASZ;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const ConvClass k = g.cls[blockIdx.y];
    const int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    if (L >= k.mt * g.nt) return;
    const int m0 = (L / g.nt) * BM, n0 = (L % g.nt) * BN;
    const int HW = k.Ha * k.Wa;
    const int wrow = wm * TM * 32, wcol = wn * TN * 32;

    // Operands are fetched through buffer descriptors: an invalid unit (tap outside the image, channel tail, tile row / column past the
    // tensor, chunk past the end of the K loop) simply carries an out-of-range offset and reads as zeros -- the chunk body has no branch.
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0,
                                            (int)((size_t)g.N * g.Hi * g.Wi * g.ldx * ESZ), 0x00020000);
    const size_t w_bytes = (size_t)g.KW * ((size_t)k.wr0 + (size_t)(k.Tr > 0 ? k.Tr - 1 : 0) * g.wrs + 1) * g.wts * 4;     // up to the last filter row this class reads
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w), 0, (int)(w_bytes < 0x7fffffffu ? w_bytes : 0x7fffffffu), 0x00020000);

    // ---- loop-invariant per-thread state
    int a_voff[RA], a_st[RA], a_k[RA];
    unsigned a_vm[RA];                                  // bit tr * Tq + tq: the tap lies inside the image for this unit's pixel (<= 32 taps)
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int u = t + NT * i, oct = u & 3, row = u >> 2;
        const int m = m0 + row;
        const bool ok = m < k.M;
        const int mm = ok ? m : 0;
        const int n = mm / HW, rem = mm - n * HW, a = rem / k.Wa, b = rem - a * k.Wa;
        const int ih0 = a * g.isy + k.iy0, iw0 = b * g.isx + k.ix0;
        a_voff[i] = ((n * g.Hi * g.Wi + ih0 * g.Wi + iw0) * g.ldx + oct * 8) * ESZ;
        unsigned vm = 0;
        for (int r = 0; r < k.Tr; ++r)
            for (int q = 0; q < k.Tq; ++q)
                if (ok && (unsigned)(ih0 + r * g.ity) < (unsigned)g.Hi && (unsigned)(iw0 + q * g.itx) < (unsigned)g.Wi) vm |= 1u << (r * k.Tq + q);
        a_vm[i] = vm;
        a_st[i] = row * G3_PITCH + oct * 16;
        a_k[i] = oct * 8;
    }
    int b_voff[RB], b_st[RB], b_k[RB];                 // b_k: first gathered channel of the unit within a chunk
    bool b_ok[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int u = t + NT * i;
        if (BT) {
            const int ku = u % (32 / KU), nn = u / (32 / KU), n = n0 + nn;
            b_ok[i] = n < g.Cout;
            b_voff[i] = (n * g.ldw + ku * KU) * 4;
            b_st[i] = nn * G3_PITCH + ku * KU * 2;
            b_k[i] = ku * KU;
        } else {
            const int kp = u & 15, nq = u >> 4, n = n0 + nq * CW;                 // sixteen k pairs x BN / CW column groups
            b_ok[i] = n < g.Cout;
            b_voff[i] = ((kp * 2) * g.ldw + n) * 4;
            b_st[i] = (nq * CW) * G3_PITCH + kp * 4;
            b_k[i] = kp * 2;
        }
    }
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = (wrow + i * 32 + li) * G3_PITCH + lh * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = (wcol + j * 32 + li) * G3_PITCH + lh * 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nck = (g.Cin + BK - 1) / BK;
    int nchunks = k.Tr * k.Tq * nck;
    int tr = 0, tq = 0, c0 = 0;
    if (g.ksplit > 1) {
        const int cps = (nchunks + g.ksplit - 1) / g.ksplit;
        const int cb = blockIdx.z * cps, ce = min(nchunks, cb + cps);
        const int tap0 = cb / nck;
        c0 = (cb - tap0 * nck) * BK; tr = tap0 / k.Tq; tq = tap0 - tr * k.Tq;
        nchunks = ce > cb ? ce - cb : 0;
    }
    int left = nchunks;                                 // chunks not yet loaded

    f32x4 ra[RA][2], rb[RB][2];                         // (the narrow B units fill only part of rb)
    auto load_chunk = [&]() {                           // registers <- the next chunk in K order (zeros once the K loop is exhausted)
        const int a_off = (((tr * g.ity) * g.Wi + tq * g.itx) * g.ldx + c0) * ESZ;        // wave-uniform byte offsets
        const int tapbit = tr * k.Tq + tq;
        const int tap = (k.wr0 + tr * g.wrs) * g.KW + (k.wq0 + tq * g.wqs);
        const int b_off = (tap * g.wts + (BT ? c0 : c0 * g.ldw)) * 4;
        const bool live = left > 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const bool va = live & (((a_vm[i] >> tapbit) & 1u) != 0) & (c0 + a_k[i] < g.Cin);       // (bitwise: no short-circuit branches)
            const int av = va ? a_voff[i] + a_off : G3_OOB;
            ra[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, av, 0, 0));
            if (!IO16) ra[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, av + 16, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (BT) {
                const bool vb = live & b_ok[i] & (c0 + b_k[i] < g.Cin);
                const int bv = vb ? b_voff[i] + b_off : G3_OOB;
                rb[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, bv, 0, 0));
                if (KU == 8) rb[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, bv + 16, 0, 0));
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool vb = live & b_ok[i] & (c0 + b_k[i] + j < g.Cin);
                    const int bv = vb ? b_voff[i] + b_off + j * g.ldw * 4 : G3_OOB;
                    if (CW == 4) rb[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, bv, 0, 0));
                    else {
                        // (two dword loads: hipcc of ROCm 7.2 lowers __builtin_amdgcn_raw_buffer_load_b64 to ONE buffer_load_dword and
                        //  hands back its value twice -- found with the 64-cout forward parity cases)
                        rb[i][j][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, bv, 0, 0));
                        rb[i][j][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, bv + 4, 0, 0));
                    }
                }
            }
        }
        --left;
        c0 += BK;
        const bool wrap = c0 >= g.Cin;
        c0 = wrap ? 0 : c0;
        tq += wrap ? 1 : 0;
        const bool wrapq = tq == k.Tq;
        tq = wrapq ? 0 : tq;
        tr += wrapq ? 1 : 0;
    };
    // split the staged fp32 units into their bf16 terms and write the planes
    auto store_ab = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            unsigned char* const Ab = Asm + a_st[i];
            if (IO16) { *reinterpret_cast<u32x4*>(Ab) = __builtin_bit_cast(u32x4, ra[i][0]); continue; }
            unsigned p[4][3];
            g3_split2<TERMS>(ra[i][0][0], ra[i][0][1], p[0]); g3_split2<TERMS>(ra[i][0][2], ra[i][0][3], p[1]);
            g3_split2<TERMS>(ra[i][1][0], ra[i][1][1], p[2]); g3_split2<TERMS>(ra[i][1][2], ra[i][1][3], p[3]);
#pragma unroll
            for (int tm = 0; tm < TERMS; ++tm) *reinterpret_cast<u32x4*>(Ab + tm * APL) = u32x4{p[0][tm], p[1][tm], p[2][tm], p[3][tm]};
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            unsigned char* const Bb = Bsm + b_st[i];
            if (BT) {
                unsigned p[4][3];
                g3_split2<TERMS>(rb[i][0][0], rb[i][0][1], p[0]); g3_split2<TERMS>(rb[i][0][2], rb[i][0][3], p[1]);
                if (KU == 8) {
                    g3_split2<TERMS>(rb[i][1][0], rb[i][1][1], p[2]); g3_split2<TERMS>(rb[i][1][2], rb[i][1][3], p[3]);
#pragma unroll
                    for (int tm = 0; tm < TERMS; ++tm) *reinterpret_cast<u32x4*>(Bb + tm * BPL) = u32x4{p[0][tm], p[1][tm], p[2][tm], p[3][tm]};
                } else {
#pragma unroll
                    for (int tm = 0; tm < TERMS; ++tm) *reinterpret_cast<u32x2*>(Bb + tm * BPL) = u32x2{p[0][tm], p[1][tm]};
                }
            } else {
                // rb[i][j][c] = w[k = 2 kp + j][n = nq*CW + c]: column c becomes one bf16 pair (4 bytes) of LDS row nq*CW + c
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    unsigned p[3];
                    g3_split2<TERMS>(rb[i][0][c], rb[i][1][c], p);
#pragma unroll
                    for (int tm = 0; tm < TERMS; ++tm) *reinterpret_cast<unsigned*>(Bb + tm * BPL + c * G3_PITCH) = p[tm];
                }
            }
        }
    };

    struct Frag { bf16x8 a[TM][TERMS], b[TN][TERMS]; };
    auto read_frag = [&](int s, Frag& f) {               // k16 step s of the chunk: bytes 32 s + 16 lh of the row
        const unsigned char* const Ab = Asm + s * 32;
        const unsigned char* const Bb = Bsm + s * 32;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int tm = 0; tm < TERMS; ++tm) f.a[i][tm] = *reinterpret_cast<const bf16x8*>(Ab + tm * APL + a_rd[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int tm = 0; tm < TERMS; ++tm) f.b[j][tm] = *reinterpret_cast<const bf16x8*>(Bb + tm * BPL + b_rd[j]);
    };
    auto mfma_frag = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                if (TERMS > 2) {                          // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][1], c, 0, 0, 0);
                }
                if (TERMS > 1) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], c, 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], c, 0, 0, 0);
            }
    };
    // ONE LDS stage, TWO workgroups per CU.  Measured on MI355X (img_discr/conv_2, N = 64, 90 us of pure MFMA time at the clock the chip
    // held): fragment reads + MFMAs alone 130 us, + operand split and LDS stores 154 us, + global loads 175 us -- the phases of a chunk ADD
    // on a SIMD instead of overlapping, whatever the software pipelining inside the workgroup (double-buffered LDS, barrier in mid-chunk,
    // two register sets, pinned instruction order: all within 3 %; removing 8 of the 11 split VALU per pair: -5 %).  Two independent
    // workgroups per CU (one 60 KB LDS stage each, <= 128 VGPRs) fill each other's barrier / store phases: -5..10 %.
    auto chunk = [&]() {
        Frag f;
        read_frag(0, f);
        load_chunk();                                    // registers <- next chunk: lands under this chunk's MFMAs
        mfma_frag(f);
        read_frag(1, f);
        mfma_frag(f);
        __syncthreads();                                 // every wavefront has read the stage
        store_ab();
        __syncthreads();
    };

    load_chunk();
    store_ab();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) chunk();

    // ---- epilogue (as conv_igemm_kernel): the output pixel of every tile row goes through LDS
    int* rowpix = reinterpret_cast<int*>(smem);
    if (t < BM) {
        const int m = m0 + t;
        int pix = -1;
        if (m < k.M) {
            const int n = m / HW, rem = m - n * HW, a = rem / k.Wa, b = rem - a * k.Wa;
            pix = (n * g.Ho + a * g.osy + k.oy0) * g.Wo + b * g.osx + k.ox0;
        }
        rowpix[t] = pix;
    }
    __syncthreads();
    if (g.ksplit > 1) {                                  // raw partial sums; bias / activation happen in splitk_reduce_kernel
        float* wsl = g.ws + (size_t)blockIdx.z * g.ws_slab;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + wcol + n * 32 + li;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pix = rowpix[wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
                    if (col < g.Cout && pix >= 0) wsl[(size_t)pix * g.Cout + col] = acc[i][n][r];
                }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int col = n0 + wcol + n * 32 + li;
        const bool cok = col < g.Cout;
        const float bv = (cok && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = rowpix[wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
                if (cok && pix >= 0) {
                    float v = acc[i][n][r] + bv;
                    if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                    else if (g.act == KPX_ACT_TANH) v = tanhf(v);
                    if (g.mul_y) {
                        const float my = (g.io16 & 4) ? g3_float((unsigned)reinterpret_cast<const unsigned short*>(g.mul_y)[(size_t)pix * g.ld_mul + col] << 16)
                                                      : g.mul_y[(size_t)pix * g.ld_mul + col];
                        v *= kpx_act_grad_from_y(my, g.mul_act);
                    }
                    if (g.io16 & 2) { const __bf16 b = (__bf16)v; reinterpret_cast<unsigned short*>(g.y)[(size_t)pix * g.ldy + col] = __builtin_bit_cast(unsigned short, b); }
                    else g.y[(size_t)pix * g.ldy + col] = v;
                }
            }
        }
    }
}

static std::atomic<unsigned long long> g3_attr_mask{0};

// wavefront grids (8 wavefronts per workgroup, two workgroups per CU): 128 x 128 tiles as 2 x 4 wavefronts of 64 x 32, 128 x 64 tiles as 4 x 2
// of 32 x 32.  (Measured alternatives, img_discr layers at N = 64, forward / data gradient per step: 4 wavefronts of 64 x 64: 1.98 / 1.49 ms;
// this: 1.82 / 1.34 ms; one double-buffered workgroup per CU: 1.89 / 1.50 ms; the fp32-MFMA kernel: 2.75 / 1.99 ms.)
template <int BN> struct g3_waves { static constexpr int wm = BN == 128 ? 2 : 4, wn = BN == 128 ? 4 : 2; };
template <int BM, int BN, bool BT, int TERMS, bool IO16 = false>
static int g3_launch_one(const ConvGeom& g, dim3 grid, hipStream_t s) {
    constexpr int lds = TERMS * (BM + BN) * G3_PITCH;
    hipLaunchKernelGGL((conv_gemm3_kernel<BM, BN, g3_waves<BN>::wm, g3_waves<BN>::wn, BT, TERMS, IO16>), grid, dim3(g3_waves<BN>::wm * g3_waves<BN>::wn * 64), lds, s, g);
    return kpx_launch_status();
}
template <int BM, int BN, bool BT, int TERMS, bool IO16 = false>
static hipError_t g3_set_attr() {
    constexpr int lds = TERMS * (BM + BN) * G3_PITCH;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm3_kernel<BM, BN, g3_waves<BN>::wm, g3_waves<BN>::wn, BT, TERMS, IO16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

// Shapes the bf16x3 kernel takes (everything else stays on conv_igemm_kernel): 16-B aligned operands, the gathered channel count a
// multiple of 8 (one LDS unit = 8 channels of one pixel), produced channels a multiple of 4, no row-merged taps.
extern "C" __attribute__((visibility("hidden"))) int kpx_gemm3_eligible(const ConvGeom* g) {
    const KpxEnv* e = kpx_env();
    if (e->no_gemm3 || g->merge || !g->vecA || !g->vecB) return 0;
    if (g->Cin % 8 != 0 || g->Cin < 16 || g->Cout % 4 != 0 || g->Cout < 16 || g->ldx % 4 != 0 || g->ldw % 4 != 0) return 0;
    if ((g->io16 & 1) && (g->ldx % 8 != 0 || g->terms != 1)) return 0;
    return 1;
}

// Launch the bf16x3 (TERMS = 3; 1 = plain bf16) gather convolution for a geometry prepared by kpx_conv2d_fwd_f32 / kpx_conv2d_dgrad_f32
// (classes, split-K slice count and workspace already decided).  bt: weights read transposed (dgrad).
extern "C" __attribute__((visibility("hidden"))) int kpx_gemm3_launch(ConvGeom g, int bt, int terms, hipStream_t s) {
    if (kpx_first_use_on_device(&g3_attr_mask)) {
        hipError_t e = g3_set_attr<128, 128, false, 3>();
        if (e == hipSuccess) e = g3_set_attr<128, 128, true, 3>();
        if (e == hipSuccess) e = g3_set_attr<128, 64, false, 3>();
        if (e == hipSuccess) e = g3_set_attr<128, 64, true, 3>();
        if (e == hipSuccess) e = g3_set_attr<128, 128, false, 1>();
        if (e == hipSuccess) e = g3_set_attr<128, 128, true, 1>();
        if (e == hipSuccess) e = g3_set_attr<128, 64, false, 1>();
        if (e == hipSuccess) e = g3_set_attr<128, 64, true, 1>();
        if (e == hipSuccess) e = g3_set_attr<128, 128, false, 1, true>();
        if (e == hipSuccess) e = g3_set_attr<128, 128, true, 1, true>();
        if (e == hipSuccess) e = g3_set_attr<128, 64, false, 1, true>();
        if (e == hipSuccess) e = g3_set_attr<128, 64, true, 1, true>();
        if (e != hipSuccess) return -(int)e;
    }
    if (g.ncls <= 0) { g.ncls = 1; g.cls[0] = ConvClass{g.Ha, g.Wa, g.oy0, g.ox0, g.Tr, g.Tq, g.iy0, g.ix0, g.wr0, g.wq0, 0, 0}; }
    g.M = 0;
    for (int i = 0; i < g.ncls; ++i) { g.cls[i].M = g.N * g.cls[i].Ha * g.cls[i].Wa; if (g.cls[i].M > g.M) g.M = g.cls[i].M; }
    if (g.M <= 0 || g.Cout <= 0) return 0;
    const int BM = 128, BN = g.Cout > 64 ? 128 : 64;
    g.nt = (g.Cout + BN - 1) / BN;
    int mtmax = 0;
    for (int i = 0; i < g.ncls; ++i) { g.cls[i].mt = (g.cls[i].M + BM - 1) / BM; if (g.cls[i].mt > mtmax) mtmax = g.cls[i].mt; }
    g.mt = mtmax;
    const dim3 grid((unsigned)(g.mt * g.nt), (unsigned)g.ncls, (unsigned)(g.ksplit > 1 ? g.ksplit : 1));
#define G3_GO(bn, btv, tv) return g3_launch_one<128, bn, btv, tv>(g, grid, s)
    if (g.io16 & 1) {                                    // bf16 tensors (terms == 1 by eligibility)
        if (BN == 128) { if (bt) return g3_launch_one<128, 128, true, 1, true>(g, grid, s); return g3_launch_one<128, 128, false, 1, true>(g, grid, s); }
        if (bt) return g3_launch_one<128, 64, true, 1, true>(g, grid, s);
        return g3_launch_one<128, 64, false, 1, true>(g, grid, s);
    }
    if (terms == 1) {
        if (BN == 128) { if (bt) G3_GO(128, true, 1); else G3_GO(128, false, 1); }
        else { if (bt) G3_GO(64, true, 1); else G3_GO(64, false, 1); }
    }
    if (BN == 128) { if (bt) G3_GO(128, true, 3); else G3_GO(128, false, 3); }
    if (bt) G3_GO(64, true, 3);
    G3_GO(64, false, 3);
#undef G3_GO
}

// ------------------------------------------------------------------------------------------------------------------------ weight gradient
// dw[tap][c][k] = sum_p x[p shifted by tap][c] * dy[p][k] on the same bf16x3 arithmetic: M = Cin, N = Cout, K = output pixels.  Both
// operands are pixel-major in HBM (channels contiguous) while an MFMA fragment wants 8 consecutive K = pixels per lane, so BOTH are staged
// like the forward kernel's n-contiguous weights: a unit = 2 pixels x 4 channels (two 16-B loads), transposed in registers -- each channel
// becomes one bf16 pair per term, written to LDS row `channel` of the operand's plane (pitch G3_PITCH: conflict-free).  Workgroup =
// (split, tap, Cin tile, Cout tile) as in conv_wgrad_kernel (conv_igemm.hip), partial slabs reduced in fixed order by wgrad_reduce_kernel.
template <int BM, int WM, int WN, int TERMS, bool IO16 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_wgrad3_kernel(const WgradGeom g) {
    static_assert(!IO16 || TERMS == 1, "bf16 tensors carry one term");
    constexpr int NT = WM * WN * 64, BKP = 32, BN = BM, ESZ = IO16 ? 2 : 4;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int APL = BM * G3_PITCH, BPL = BN * G3_PITCH;
    constexpr int ASZ = TERMS * APL;
    static_assert(NT == 16 * (BM / 4), "one A unit and one B unit (2 pixels x 4 channels) per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Asm = smem;
    unsigned char* const Bsm = smem + ASZ;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int taps = g.KH * g.KW;
    const int kti = L % g.kt; L /= g.kt;
    const int cti = L % g.ct; L /= g.ct;
    const int tap = L % taps; L /= taps;
    const int split = L;
    const int r = tap / g.KW, q = tap - r * g.KW;
    const int cbase = cti * BM, kbase = kti * BN;
    const int pbeg = split * g.pps;
    const int pend = min(g.P, pbeg + g.pps);
    const int wrow = wm * TM * 32, wcol = wn * TN * 32;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, (int)((size_t)g.N * g.Hi * g.Wi * g.ldx * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dy), 0, (int)((size_t)g.P * g.lddy * ESZ), 0x00020000);
    // this thread's unit: pixel pair kp of the 32-pixel chunk, channel quad cq (kp fastest: conflict-free ds_write_b32 at pitch 80)
    // bf16 tensors (IO16): a unit is two pixels x EIGHT channels (one 16-byte load per pixel) of ONE operand -- the first half of the
    // workgroup's threads stage x, the second half dy
    const bool isB16 = IO16 && t >= NT / 2;
    const int tt = IO16 ? (isB16 ? t - NT / 2 : t) : t;
    const int kp = tt & 15, cq = tt >> 4;
    const int ac = cbase + cq * (IO16 ? 8 : 4), bc = kbase + cq * (IO16 ? 8 : 4);
    const bool a_cok = ac < g.Cin, b_cok = bc < g.Cout;            // (Cin, Cout multiples of 4 / 8: a unit is all in or all out)
    const int a_st = (cq * (IO16 ? 8 : 4)) * G3_PITCH + kp * 4, b_st = a_st;
    // pixel coordinates of the unit's two pixels, advanced by 32 pixels per chunk
    int pn[2], pho[2], pwo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = pbeg + 2 * kp + j;
        pn[j] = p / (g.Ho * g.Wo);
        const int rem = p - pn[j] * g.Ho * g.Wo;
        pho[j] = rem / g.Wo;
        pwo[j] = rem - pho[j] * g.Wo;
    }
    int p0 = pbeg + 2 * kp;

    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = (wrow + i * 32 + li) * G3_PITCH + lh * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = (wcol + j * 32 + li) * G3_PITCH + lh * 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[2], rb[2];
    auto load_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ih = pho[j] * g.stride + r - g.pad_t, iw = pwo[j] * g.stride + q - g.pad_l;
            const bool va = a_cok & (p0 + j < pend) & ((unsigned)ih < (unsigned)g.Hi) & ((unsigned)iw < (unsigned)g.Wi);
            const int av = va ? (((pn[j] * g.Hi + ih) * g.Wi + iw) * g.ldx + ac) * ESZ : G3_OOB;
            const bool vb = b_cok & (p0 + j < pend);
            const int bv = vb ? ((p0 + j) * g.lddy + bc) * ESZ : G3_OOB;
            if (IO16) {                                    // one operand per thread: ra[j] = eight bf16 channels of pixel j
                ra[j] = __builtin_bit_cast(f32x4, isB16 ? __builtin_amdgcn_raw_buffer_load_b128(rs_b, bv, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs_a, av, 0, 0));
            } else {
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, av, 0, 0));
                rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, bv, 0, 0));
            }
            // advance this pixel by one chunk (32 pixels): at most a few row wraps for narrow images
            pwo[j] += BKP;
            while (pwo[j] >= g.Wo) {
                pwo[j] -= g.Wo;
                if (++pho[j] == g.Ho) { pho[j] = 0; ++pn[j]; }
            }
        }
        p0 += BKP;
    };
    auto store_ab = [&]() {
        unsigned char* const Ab = Asm + a_st;
        unsigned char* const Bb = Bsm + b_st;
        if (IO16) {
            // channel c of the unit: (pixel 0, pixel 1) as one bf16 pair in LDS row c of the thread's operand
            unsigned char* const Ob = isB16 ? Bb : Ab;
            const u32x4 v0 = __builtin_bit_cast(u32x4, ra[0]), v1 = __builtin_bit_cast(u32x4, ra[1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned pr = (c & 1) ? __builtin_amdgcn_perm(v1[c >> 1], v0[c >> 1], 0x07060302u) : __builtin_amdgcn_perm(v1[c >> 1], v0[c >> 1], 0x05040100u);
                *reinterpret_cast<unsigned*>(Ob + c * G3_PITCH) = pr;
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned pa[3], pb[3];
            g3_split2<TERMS>(ra[0][c], ra[1][c], pa);
            g3_split2<TERMS>(rb[0][c], rb[1][c], pb);
#pragma unroll
            for (int tm = 0; tm < TERMS; ++tm) {
                *reinterpret_cast<unsigned*>(Ab + tm * APL + c * G3_PITCH) = pa[tm];
                *reinterpret_cast<unsigned*>(Bb + tm * BPL + c * G3_PITCH) = pb[tm];
            }
        }
    };
    struct Frag { bf16x8 a[TM][TERMS], b[TN][TERMS]; };
    auto read_frag = [&](int s, Frag& f) {
        const unsigned char* const Ab = Asm + s * 32;
        const unsigned char* const Bb = Bsm + s * 32;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int tm = 0; tm < TERMS; ++tm) f.a[i][tm] = *reinterpret_cast<const bf16x8*>(Ab + tm * APL + a_rd[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int tm = 0; tm < TERMS; ++tm) f.b[j][tm] = *reinterpret_cast<const bf16x8*>(Bb + tm * BPL + b_rd[j]);
    };
    auto mfma_frag = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                if (TERMS > 2) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][1], c, 0, 0, 0);
                }
                if (TERMS > 1) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], c, 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], c, 0, 0, 0);
            }
    };

    const int nchunks = pend > pbeg ? (pend - pbeg + BKP - 1) / BKP : 0;
    load_chunk();
    store_ab();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        Frag f;
        read_frag(0, f);
        load_chunk();                                    // next chunk (zeros past the end of the split: p >= pend)
        mfma_frag(f);
        read_frag(1, f);
        mfma_frag(f);
        __syncthreads();
        store_ab();
        __syncthreads();
    }

    float* out = g.out + (size_t)split * g.slab + (size_t)tap * g.Cin * g.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int k = kbase + wcol + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = cbase + wrow + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (c < g.Cin && k < g.Cout) out[(size_t)c * g.Cout + k] = acc[i][j][e];
            }
    }
}

extern "C" __attribute__((visibility("hidden"))) int kpx_wgrad3_eligible(const WgradGeom* g) {
    const KpxEnv* e = kpx_env();
    if (e->no_gemm3 || g->merge || !g->vecA || !g->vecB) return 0;
    if (g->Cin % 4 != 0 || g->Cout % 4 != 0 || g->Cin < 16 || g->Cout < 16) return 0;
    if ((size_t)g->N * g->Hi * g->Wi * g->ldx * 4 >= 0x7fffffffu || (size_t)g->P * g->lddy * 4 >= 0x7fffffffu) return 0;     // 32-bit buffer offsets
    if (g->io16 && g->terms != 1) return 0;
    return 1;
}
// bm: the square channel tile conv_igemm.hip planned the grid for (128 or 64); the grid is g.S * taps * g.ct * g.kt workgroups.
extern "C" __attribute__((visibility("hidden"))) int kpx_wgrad3_launch(WgradGeom g, int bm, int terms, hipStream_t s) {
    const dim3 grid((unsigned)(g.S * g.KH * g.KW * g.ct * g.kt));
    if (g.io16) {
        if (bm == 128) hipLaunchKernelGGL((conv_wgrad3_kernel<128, 2, 4, 1, true>), grid, dim3(512), 256 * G3_PITCH, s, g);
        else hipLaunchKernelGGL((conv_wgrad3_kernel<64, 2, 2, 1, true>), grid, dim3(256), 128 * G3_PITCH, s, g);
        return kpx_launch_status();
    }
    if (bm == 128) {
        constexpr int lds = 3 * 256 * G3_PITCH;
        if (terms == 1) hipLaunchKernelGGL((conv_wgrad3_kernel<128, 2, 4, 1>), grid, dim3(512), lds / 3, s, g);
        else hipLaunchKernelGGL((conv_wgrad3_kernel<128, 2, 4, 3>), grid, dim3(512), lds, s, g);
    } else {
        constexpr int lds = 3 * 128 * G3_PITCH;
        if (terms == 1) hipLaunchKernelGGL((conv_wgrad3_kernel<64, 2, 2, 1>), grid, dim3(256), lds / 3, s, g);
        else hipLaunchKernelGGL((conv_wgrad3_kernel<64, 2, 2, 3>), grid, dim3(256), lds, s, g);
    }
    return kpx_launch_status();
}
