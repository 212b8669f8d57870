// 3x3 stride-1 SAME convolution with bf16 TENSORS IN HBM (bf16 in, bf16 or fp32 out, fp32 accumulate) on v_mfma_f32_32x32x16_bf16 for
// gfx950: the bf16 configuration (BASELINE configs[2]) of layers.conv (reference models/networks/layers.py:4-10) for the translator /
// VGG19 / encoders / key-point detector 3x3 layers (models/networks/__init__.py:13-24,50-62,80-97, models/networks/vgg.py:20-40), forward
// and data gradient (the same kernel on filters prepared flipped / transposed).
//
// Structure (one workgroup = 8 wavefronts = MB x 32 output pixels x NBT x 32 output channels; one workgroup per CU):
//   * operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write.  Input patch: pixel q of the
//     (TH+2) x PW patch owns five 16-B slots (32 channels of a chunk + one pad slot: pitch 80 B), so sixteen consecutive pixels start in
//     sixteen different 16-B columns of the 256-B bank row -- every fragment read of every tap is conflict-free (scratch/lds_conflicts_bf16s.py)
//     and a tap is an IMMEDIATE offset from one base register per pixel block.  Out-of-image pixels, pad slots and channel tails carry an
//     out-of-range buffer offset and land as zeros.  Filters: fragment-ordered by kpx_conv3x3_bf16s_prepare (once per optimiser update),
//     1 KB = one wave-instruction per (tap, k16 step, cout block).
//   * K loop: chunks of 32 channels x three phases (one filter ROW each: 3 taps x 2 k16 steps).  The filter row of phase p+1 and a third
//     of the next chunk's patch are in flight during phase p; counted s_waitcnt vmcnt + one raw s_barrier per phase (the patch is double
//     buffered per chunk, the filter rows per phase).
//   * MFMA operand roles are swapped against the textbook im2col form: A = filter fragment (rows = output channels), B = pixel fragment
//     (columns = pixels).  The accumulator then holds, per lane = pixel, sixteen output channels in groups of four consecutive ones: after
//     one v_permlane32_swap per dword (guide T21) a lane stores 16 contiguous bytes of its pixel's NHWC row.
//   * wavefront (wm, wn) owns P pixel blocks x Q cout blocks: P + Q fragment reads per P*Q MFMAs.
// Epilogue options: bias, activation, a ReLU mask (data gradient towards an activated tensor: VGG19), per-workgroup batch-norm sums of the
// fp32 accumulators (consumed by kpx_bn_train_fwd: no statistics pass over the activation).
#include "kpx_common.h"
#include "kpx_env.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define S16_OOB 0x7ffffff0
#define S16_PIX 80                       // bytes of one patch pixel in LDS: 32 channels + one 16-B pad slot

struct S16Geom {
    const void* x; void* y; const void* Wf; const float* bias; const void* mask; float* stats;
    int N, H, W, K, ldx, Nn, ldy, ldm, act;
    int KC, NB;                          // 32-channel chunks of K, 32-cout blocks of the prepared filters (Nn rounded up to the tile)
    int tiles_y, tiles_x, ngrp, ntc;     // ngrp: image groups (G images per tile), ntc: cout tiles
    int total_tiles;                     // ngrp * tiles_y * tiles_x * ntc: walked by persistent workgroups
    int out_f32;
};

// Wf[kc][tap 9][ks 2][nb][lane 64][8] bf16: element j of lane (li, lh) = w'[tap][c = 32 kc + 16 ks + 8 lh + j][n = 32 nb + li]
// (the A operand of v_mfma_f32_32x32x16_bf16: row li = output channel, k = 8 lh + j).  dgrad: w'[r][q][c'][n'] = w[2-r][2-q][n'][c'].
template <bool DGRAD>
__global__ __launch_bounds__(256) void conv_bf16s_prepare_kernel(const float* __restrict__ w, int Cin, int Cout, int KC, int NB, unsigned short* __restrict__ Wf) {
    const int K = DGRAD ? Cout : Cin, Nn = DGRAD ? Cin : Cout;
    const size_t total = (size_t)KC * 9 * 2 * NB * 512;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx & 7), li = (int)((idx >> 3) & 31), lh = (int)((idx >> 8) & 1);
        size_t blk = idx >> 9;
        const int nb = (int)(blk % NB); blk /= NB;
        const int ks = (int)(blk & 1); blk >>= 1;
        const int tap = (int)(blk % 9);
        const int kc = (int)(blk / 9);
        const int c = 32 * kc + 16 * ks + 8 * lh + j, n = 32 * nb + li;
        float v = 0.f;
        if (c < K && n < Nn) v = DGRAD ? w[((size_t)(8 - tap) * Cin + n) * Cout + c] : w[((size_t)tap * Cin + c) * Cout + n];
        const __bf16 b = (__bf16)v;
        Wf[idx] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

// batched form: one launch per optimiser update for every trainable 3x3 filter (table entry: w, Wf, Cin, Cout, dgrad, NB)
struct S16PrepDesc { const float* w; unsigned short* Wf; int Cin, Cout, dgrad, NB; };
__global__ __launch_bounds__(256) void conv_bf16s_prepare_batch_kernel(const S16PrepDesc* __restrict__ table, int ndesc) {
    const S16PrepDesc d = table[blockIdx.y];
    const int K = d.dgrad ? d.Cout : d.Cin, Nn = d.dgrad ? d.Cin : d.Cout;
    const int KC = (K + 31) / 32;
    const size_t total = (size_t)KC * 9 * 2 * d.NB * 512;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx & 7), li = (int)((idx >> 3) & 31), lh = (int)((idx >> 8) & 1);
        size_t blk = idx >> 9;
        const int nb = (int)(blk % d.NB); blk /= d.NB;
        const int ks = (int)(blk & 1); blk >>= 1;
        const int tap = (int)(blk % 9);
        const int kc = (int)(blk / 9);
        const int c = 32 * kc + 16 * ks + 8 * lh + j, n = 32 * nb + li;
        float v = 0.f;
        if (c < K && n < Nn) v = d.dgrad ? d.w[((size_t)(8 - tap) * d.Cin + n) * d.Cout + c] : d.w[((size_t)tap * d.Cin + c) * d.Cout + n];
        const __bf16 b = (__bf16)v;
        d.Wf[idx] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

__device__ __forceinline__ unsigned s16_pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float s16_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float s16_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

// outstanding vector-memory operations allowed to remain (s_waitcnt takes an immediate)
__device__ __forceinline__ void s16_wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: if (n > 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;       // (n < 0: no wait; more than 20: waiting for more is safe)
    }
}
// depth of the filter-row ring by tile variant (LDS budget: s16_plan adds it up)
// (Measured: rings of 3 / 4 rows on the small tiles change nothing -- 0.112 vs 0.116 ms on VGG19 conv1_2 -- those launches are bound by the
//  per-tile instruction overhead, not by the filter rows' latency; every variant keeps two.)
template <int NBT, int P> struct S16Ring { static constexpr int nbb = 2; };

// geometry of a tile at compile time
template <int WN, int Q, int P, int BW>
struct S16Tile {
    static constexpr int WM = 8 / WN, MB = WM * P, NBT = WN * Q;
    static constexpr int R = 32 / BW;                       // rows of one 32-pixel block
    static constexpr int ROWS = MB * R;                     // output rows of a tile (stacked over its images)
    static constexpr int PW = BW + 2;
};

// STATS: 0 none; 1 per-workgroup channel sums and sums of squares of the fp32 outputs before the activation -> stats[tile][2][Nn] (the batch
// norm BEHIND a forward convolution); 2 (data gradient towards y = relu(batch norm)): `mask` is y, `bias` is the batch norm's beta -- the
// gradient is gated by y > 0 and the workgroup writes sum(dz), sum(dz * (y - beta)) per channel: the batch norm's backward sums
// (x_hat = (y - beta) / gamma wherever dz != 0), as kpx_conv3x3_wino43_bnbwd_stats_f32 does in the fp32 configuration
template <int WN, int Q, int P, int BW, int STATS>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16s_kernel(const S16Geom g, const int TH, const int G) {
    using T = S16Tile<WN, Q, P, BW>;
    constexpr int WM = T::WM, NBT = T::NBT, R = T::R, PW = T::PW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PH = TH + 2;
    const int NPX = G * PH * PW;                            // patch pixels
    const int APIECES = (NPX * 5 + 63) >> 6;                // 1-KB LDS-DMA pieces of one patch
    const int ABYTES = APIECES << 10;
    constexpr int BPIECES = 6 * NBT, BBYTES = BPIECES << 10;
    // filter-row ring: a row is requested D = NBB - 1 phases ahead (S16Ring: two rows everywhere, deeper rings measured no gain)
    constexpr int NBB = S16Ring<NBT, P>::nbb, D = NBB - 1;
    constexpr int BPW = (BPIECES + 7) / 8;                  // filter pieces per wavefront and phase
    constexpr int APW = 9;                                  // patch pieces per wavefront and chunk (host: APIECES <= 72), up to three per phase
    unsigned char* const Asm = smem;                        // [2][ABYTES]
    unsigned char* const Bsm = smem + 2 * ABYTES;           // [NBB][BBYTES]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    // PERSISTENT workgroups: tile L, L + gridDim.x, ..  The next tile's first patch and filter row are requested BEFORE the current tile's
    // epilogue (its stores then drain beside the next tile's loads); tiles of one XCD's workgroups are neighbours (shared halo / the cout tiles
    // of one input tile hit the same L2)
    struct Tile { int nti, bx, by, grp; };
    auto decode = [&](int L) {
        Tile tl;
        tl.nti = L % g.ntc; L /= g.ntc;
        tl.bx = L % g.tiles_x; L /= g.tiles_x;
        tl.by = L % g.tiles_y;
        tl.grp = L / g.tiles_y;
        return tl;
    };
    unsigned char* const Rsm = smem + 2 * ABYTES + NBB * BBYTES;     // [bias NBT * 32 floats][statistics 2 x WM x NBT * 32 floats]

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, (int)((size_t)g.N * g.H * g.W * g.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wf), 0, (int)((size_t)g.KC * 18 * g.NB * 1024), 0x00020000);

    // ---- patch pieces of this wavefront: piece wave + 8 i covers slots 64 (wave + 8 i) .. + 63; slot = 5 q + u
    int a_voff[APW];
    unsigned a_tail = 0;                                    // bit i: the unit exists in the LAST chunk too (channel tail)
    const int ktail = g.K - 32 * (g.KC - 1);
    int a_pos[APW];                                         // patch position of this lane's slot of piece i (tile independent), packed
                                                            // row | column << 8 | image << 16 | unit << 24; -1: no slot
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int piece = wave + 8 * i;
        const int S = piece * 64 + lane, q = S / 5, u = S - 5 * q;
        const int gi = q / (PH * PW), rem = q - gi * (PH * PW), pr = rem / PW, pc = rem - pr * PW;
        a_pos[i] = (piece < APIECES && u < 4 && q < NPX) ? (pr | (pc << 8) | (gi << 16) | (u << 24)) : -1;
    }
    auto set_tile = [&](const Tile& tl) {                   // buffer offsets of this lane's patch slots for tile tl
        const int oy0 = tl.by * TH, ox0 = tl.bx * BW, n0 = tl.grp * G;
        a_tail = 0;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int pos = a_pos[i], u = (pos >> 24) & 3;
            const int n = n0 + ((pos >> 16) & 255), iy = oy0 - 1 + (pos & 255), ix = ox0 - 1 + ((pos >> 8) & 255);
            const bool ok = pos >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W && n < g.N;
            a_voff[i] = ok ? (int)((((size_t)n * g.H + iy) * g.W + ix) * g.ldx * 2 + u * 16) : S16_OOB;
            if (ok && u * 8 < ktail) a_tail |= 1u << i;
        }
    };
    // ---- fragment read bases
    int pb[P];                                              // pixel block p of this wavefront: LDS byte offset of (lane's pixel, tap (0,0), unit lh)
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int mb = wm * P + p;
        const int tr0 = mb * R + li / BW;                   // row inside the tile's stacked images
        const int gi = tr0 / TH, row = tr0 - gi * TH;
        const int j = li % BW;
        const int rot = (BW == 16 && (li / BW) == 1) ? 2 : (BW == 8 && (li / BW) == 3) ? 2 : 0;
        const int col = (j - rot) & (BW - 1);
        pb[p] = ((gi * PH + row) * PW + col) * S16_PIX + lh * 16;
    }
    const int wb = (wn * Q) * 1024 + lane * 16;

    f32x16 acc[P][Q];

    auto issue_b = [&](int nti, int kc, int r, int buf) {  // filter row r of chunk kc, cout tile nti -> Bsm[buf]
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int piece = wave * BPW + i;               // = (s * 2 + ks) * NBT + nb
            if (BPIECES % 8 == 0 || piece < BPIECES) {
                const int sk = piece / NBT, nb = piece - sk * NBT;
                const int soff = ((((kc * 3 + r) * 3) * 2 + sk) * g.NB + nti * NBT + nb) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bsm + buf * BBYTES + piece * 1024), 16, lane * 16, soff, 0, 0);
            }
        }
    };
    auto issue_a = [&](int kc, int part, int buf) {        // pieces part, part + 2, .. of this wavefront (part 0 / 1), chunk kc -> Asm[buf]
        const bool last = kc == g.KC - 1;
#pragma unroll
        for (int ii = 0; ii < 5; ++ii) {
            const int i = part + 2 * ii;
            if (i >= APW) continue;
            const int piece = wave + 8 * i;
            if (piece < APIECES) {
                const int vo = (last && !((a_tail >> i) & 1u)) ? S16_OOB : a_voff[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(Asm + buf * ABYTES + piece * 1024), 16, vo, kc * 64, 0, 0);
            }
        }
    };

    // bias (forward) / beta (STATS 2) of a tile's output channels travel through LDS: [2 slots by tile parity][NBT * 32] floats at Rsm.  The
    // value is loaded BEFORE the tile's LDS-DMA requests (vector-memory results return in order) and parked in LDS after the previous tile's epilogue.
    float bias_next = 0.f;
    auto issue_first = [&](const Tile& tl) {               // a tile's first operands: chunk 0's patch and filter row 0, its bias values
        const int ch = tl.nti * NBT * 32 + t;
        bias_next = (g.bias && t < NBT * 32 && ch < g.Nn) ? g.bias[ch] : 0.f;
        set_tile(tl);
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < 3 * g.KC) issue_b(tl.nti, j / 3, j % 3, j);
        issue_a(0, 0, 0); issue_a(0, 1, 0);
        if (g.KC > 1) { issue_a(1, 0, 1); issue_a(1, 1, 1); }     // both patch buffers are free at a tile boundary: two chunks ahead
    };
    float* const Bias = reinterpret_cast<float*>(Rsm);      // [2][NBT * 32]
    int Lcur = kpx_xcd_remap(blockIdx.x, gridDim.x);
    Tile cur = decode(Lcur);
    int it = 0;
    issue_first(cur);
    if (t < NBT * 32) Bias[t] = bias_next;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

  for (;;) {
    const int nti = cur.nti, bx = cur.bx, by = cur.by, grp = cur.grp;
    const int oy0 = by * TH, ox0 = bx * BW, n0 = grp * G, c0 = nti * NBT * 32;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;
    int phase = 0, hist_a1 = 0, hist_a2 = 0, hist_b1 = 0;
    const int nphases = 3 * g.KC;
    for (int kc = 0; kc < g.KC; ++kc) {
        const int abuf = (kc & 1) * ABYTES;
        const bool more = kc + 1 < g.KC;
#pragma unroll
        for (int r = 0; r < 3; ++r, ++phase) {
            const int bbuf = (phase % NBB) * BBYTES;
            // prefetch: the filter row of phase + D; the next chunk's patch in the first two phases of a chunk (never in the third: the wait
            // before a new chunk then leaves this phase's filter request in flight)
            int nbi = 0, na = 0;                             // LDS-DMA instructions this wavefront issues in this phase (wave-uniform)
            if (phase + D < nphases) {
                const int pp = phase + D;
                issue_b(nti, pp / 3, pp % 3, pp % NBB);
                nbi = BPIECES % 8 == 0 ? BPW : max(0, min(BPW, BPIECES - wave * BPW));
            }
            if (more && kc >= 1 && r < 2) {
                issue_a(kc + 1, r, (kc + 1) & 1);
#pragma unroll
                for (int ii = 0; ii < 5; ++ii) na += (r + 2 * ii < APW && wave + 8 * (r + 2 * ii) < APIECES) ? 1 : 0;
            }
            const unsigned char* const Ab = Asm + abuf;
            const unsigned char* const Bb = Bsm + bbuf + wb;
            // six (tap, k16) steps; the fragments of step i+1 are read before the MFMAs of step i (two register sets)
            auto rd = [&](int st, bf16x8* xf, bf16x8* wf) {
                const int s = st >> 1, ks = st & 1;
#pragma unroll
                for (int q = 0; q < Q; ++q) wf[q] = *reinterpret_cast<const bf16x8*>(Bb + (st * NBT + q) * 1024);
#pragma unroll
                for (int p = 0; p < P; ++p) xf[p] = *reinterpret_cast<const bf16x8*>(Ab + pb[p] + (r * PW + s) * S16_PIX + ks * 32);
            };
            auto mm = [&](const bf16x8* xf, const bf16x8* wf) {
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int q = 0; q < Q; ++q)
                        acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q], xf[p], acc[p][q], 0, 0, 0);
            };
            bf16x8 xa[P], wa[Q], xb[P], wc[Q];
            rd(0, xa, wa);
#pragma unroll
            for (int st = 0; st < 6; st += 2) {
                // (sched_barrier: hipcc otherwise sinks every fragment read to its first use and waits lgkmcnt(0) there)
                rd(st + 1, xb, wc);
                __builtin_amdgcn_sched_barrier(0);
                mm(xa, wa);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 2 < 6) rd(st + 2, xa, wa);
                __builtin_amdgcn_sched_barrier(0);
                mm(xb, wc);
                __builtin_amdgcn_sched_barrier(0);
            }
            // Everything the NEXT phase reads must have landed: its filter row (requested D - 1 phases ago) and, before a new chunk, the
            // chunk's patch (requested in the first two phases of this chunk or at the tile boundary).  Vector-memory operations retire in
            // order: wait until at most `young` = the requests issued AFTER the youngest required one are outstanding.
            int young;
            if (r == 2) young = D >= 2 ? nbi : 0;
            else young = D == 1 ? na : D == 2 ? hist_a1 + nbi + na : hist_a2 + hist_b1 + hist_a1 + nbi + na;
            if (phase + 1 >= nphases) young = -1;            // last phase of the tile: nothing to wait for here
            hist_a2 = hist_a1; hist_b1 = nbi; hist_a1 = na;
            s16_wait_vmcnt(young);
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- the next tile's first operands are requested before this tile's epilogue -- unless the epilogue has loads of its own (mask / batch
    // norm output): vector-memory operations complete in order, so those loads would wait for the whole prologue
    const int Lnext = Lcur + (int)gridDim.x;
    const bool has_next = Lnext < g.total_tiles;
    Tile nxt = cur;
    const bool early = has_next && !g.mask;
    if (has_next) nxt = decode(Lnext);
    if (early) issue_first(nxt);

    // ---- epilogue.  acc[p][q][e]: pixel = lane li of block p, output channel = c0 + 32 (wn Q + q) + 8 (e >> 2) + 4 lh + (e & 3)
    const float* const bias_cur = Bias + (it & 1) * (NBT * 32);
    if (STATS != 2 && g.bias) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias_cur + 32 * (wn * Q + q) + 8 * gq + 4 * lh);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p][q][4 * gq + j] += b[j];
            }
    }
    size_t pix_off[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int mb = wm * P + p;
        const int tr0 = mb * R + li / BW;
        const int gi = tr0 / TH, row = tr0 - gi * TH;
        const int j = li % BW;
        const int rot = (BW == 16 && (li / BW) == 1) ? 2 : (BW == 8 && (li / BW) == 3) ? 2 : 0;
        const int col = (j - rot) & (BW - 1);
        pix_off[p] = (((size_t)(n0 + gi) * g.H + oy0 + row) * g.W + ox0 + col);
    }
    if (STATS) {
        // per-channel sums over the workgroup's pixels, from the fp32 (biased, not yet activated) outputs: per lane over its P blocks, then a halving butterfly over
        // the 32 lanes of a half-wave (16 values -> 1 per lane), then over the WM wavefronts through LDS
        float* const red = reinterpret_cast<float*>(Rsm) + 2 * NBT * 32;   // [2 (sum, sumsq)][WM][NBT * 32] behind the two bias slots
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float s1[16], s2[16];
            if (STATS == 2) {
                const int cbq = c0 + 32 * (wn * Q + q);
                float be[16];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_cur + 32 * (wn * Q + q) + 8 * gq + 4 * lh);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { be[4 * gq + j] = b[j]; s1[4 * gq + j] = 0.f; s2[4 * gq + j] = 0.f; }
                }
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const unsigned short* const yo = reinterpret_cast<const unsigned short*>(g.mask) + pix_off[p] * g.ldm;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int ch = cbq + 8 * gq + 4 * lh;
                        const u32x2 yy = ch < g.Nn ? *reinterpret_cast<const u32x2*>(yo + ch) : u32x2{0u, 0u};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float yv = (j & 1) ? s16_hi(yy[j >> 1]) : s16_lo(yy[j >> 1]);
                            const int e = 4 * gq + j;
                            const float v = yv > 0.f ? acc[p][q][e] : 0.f;
                            acc[p][q][e] = v;
                            s1[e] += v; s2[e] = fmaf(v, yv - be[e], s2[e]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int p = 0; p < P; ++p) { const float v = acc[p][q][e]; a += v; b = fmaf(v, v, b); }
                    s1[e] = a; s2[e] = b;
                }
            }
            // butterfly: after the step with distance d a lane keeps half of its values
#pragma unroll
            for (int d = 16, cnt = 16; d >= 1; d >>= 1, cnt >>= 1) {
                const bool up = (li & d) != 0;
#pragma unroll
                for (int e = 0; e < cnt / 2; ++e) {
                    const float keep1 = up ? s1[e + cnt / 2] : s1[e], send1 = up ? s1[e] : s1[e + cnt / 2];
                    const float keep2 = up ? s2[e + cnt / 2] : s2[e], send2 = up ? s2[e] : s2[e + cnt / 2];
                    s1[e] = keep1 + __shfl_xor(send1, d, 64);
                    s2[e] = keep2 + __shfl_xor(send2, d, 64);
                }
                if (cnt == 2) break;
            }
            // now s1[0] / s2[0] hold the sum over 16 of the 32 lanes for value index ((li>>4)&1)*8 + ((li>>3)&1)*4 + ((li>>2)&1)*2 + ((li>>1)&1);
            // lanes li and li^1 hold the two halves of the same value
            const float t1 = s1[0] + __shfl_xor(s1[0], 1, 64), t2 = s2[0] + __shfl_xor(s2[0], 1, 64);
            const int e = ((li >> 4) & 1) * 8 + ((li >> 3) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 1) & 1);
            const int ch = 32 * (wn * Q + q) + 8 * (e >> 2) + 4 * lh + (e & 3);
            if ((li & 1) == 0) {
                red[wm * (NBT * 32) + ch] = t1;
                red[(WM + wm) * (NBT * 32) + ch] = t2;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (a raw barrier: __syncthreads() would also wait for the LDS-DMA in flight)
        __builtin_amdgcn_s_barrier();
        if (t < 2 * NBT * 32) {
            const int which = t / (NBT * 32), ch = t - which * (NBT * 32);
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) a += red[(which * WM + w) * (NBT * 32) + ch];
            const int tile = ((grp * g.tiles_y + by) * g.tiles_x + bx);
            if (c0 + ch < g.Nn) g.stats[((size_t)tile * 2 + which) * g.Nn + c0 + ch] = a;
        }
    }

    const bool img_ok = true;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int cb = c0 + 32 * (wn * Q + q);              // first output channel of this 32-block
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float a = acc[p][q][e];
                if (g.act == KPX_ACT_RELU) a = fmaxf(a, 0.f);
                else if (g.act == KPX_ACT_LRELU) a = a > 0.f ? a : 0.01f * a;
                v[e] = a;
            }
            if (g.out_f32) {
                float* const yo = reinterpret_cast<float*>(g.y) + pix_off[p] * g.ldy;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int ch = cb + 8 * gq + 4 * lh;
                    if (img_ok && ch < g.Nn) *reinterpret_cast<f32x4*>(yo + ch) = f32x4{v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
                }
            } else {
                unsigned short* const yo = reinterpret_cast<unsigned short*>(g.y) + pix_off[p] * g.ldy;
                const unsigned short* const mo = (STATS != 2 && g.mask) ? reinterpret_cast<const unsigned short*>(g.mask) + pix_off[p] * g.ldm : nullptr;
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    // groups k, k+1 (channels 8k + 4lh .. and 8(k+1) + 4lh ..): after the half exchange lanes 0-31 hold channels 8k .. 8k+7,
                    // lanes 32-63 channels 8(k+1) .. 8(k+1)+7 of their pixel
                    unsigned a0 = s16_pack2(v[4 * k], v[4 * k + 1]), a1 = s16_pack2(v[4 * k + 2], v[4 * k + 3]);
                    unsigned b0 = s16_pack2(v[4 * k + 4], v[4 * k + 5]), b1 = s16_pack2(v[4 * k + 6], v[4 * k + 7]);
                    auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                    const int ch = cb + 8 * (k + lh);
                    if (img_ok && ch < g.Nn) {
                        if (mo) {
                            const u32x4 m = *reinterpret_cast<const u32x4*>(mo + ch);
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                // zero where the mask tensor is <= 0 (bf16 sign / zero test on the raw halves)
                                const unsigned ml = m[d] & 0xffffu, mh = m[d] >> 16;
                                const bool pl = ml != 0 && ml < 0x8000u, ph = mh != 0 && mh < 0x8000u;
                                o[d] = (pl ? (o[d] & 0xffffu) : 0u) | (ph ? (o[d] & 0xffff0000u) : 0u);
                            }
                        }
                        *reinterpret_cast<u32x4*>(yo + ch) = o;
                    }
                }
            }
        }
    }
    if (!has_next) break;
    if (!early) issue_first(nxt);
    ++it;
    if (t < NBT * 32) Bias[(it & 1) * (NBT * 32) + t] = bias_next;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur = nxt; Lcur = Lnext;
  }
}

static std::atomic<unsigned long long> s16_attr_mask{0};

struct S16Plan { int variant, BW, TH, G, MB, NBT, lds; };
// variant 0: 512 pixels x 128 couts (WN 2, Q 2, P 4); 1: 256 x 128 (P 2); 2: 512 x 64 (WN 1, Q 2, P 2); 3: 512 x 32 (WN 1, Q 1, P 2);
// 4: 128 x 128 (WN 2, Q 2, P 1); 5: 128 x 64 (WN 2, Q 1, P 1) -- the small tiles for layers of few pixels
static bool s16_plan(int N, int H, int W, int K, int Nn, S16Plan* pl) {
    int BW = 0;
    if (W % 32 == 0) BW = 32; else if (W == 16) BW = 16; else if (W == 8) BW = 8; else return false;
    if (!((K % 32 == 0) || K == 16 || K == 8)) return false;
    const int R = 32 / BW;
    int variant, MB, NBT;
    const long px = (long)N * H * W;
    const int force = kpx_env()->bf16s_variant;        // KPX_BF16S_VARIANT (experiments): 0 = planner's choice, v + 1 forces variant v
    if (Nn > 64) {
        NBT = 4;
        const long ct = (Nn + 127) / 128;
        if ((px / 512) * ct >= 256) { variant = 0; MB = 16; }
        else if ((px / 256) * ct >= 192) { variant = 1; MB = 8; }
        else if ((px / 128) * ct >= 192) { variant = 4; MB = 4; }
        else { variant = 5; MB = 4; NBT = 2; }
    } else if (Nn > 32) { variant = 2; MB = 16; NBT = 2; }
    else { variant = 3; MB = 16; NBT = 1; }
    if (force > 0) {
        variant = force - 1;
        static const int mbs[6] = {16, 8, 16, 16, 4, 4}, nbts[6] = {4, 4, 2, 1, 4, 2};
        if (variant > 5) return false;
        MB = mbs[variant]; NBT = nbts[variant];
    }
    int rows = MB * R, TH, G;
    if (H >= rows) { if (H % rows) return false; TH = rows; G = 1; }
    else { if (rows % H) return false; TH = H; G = rows / H; if (N % G) return false; }
    if (BW != 32 && W != BW) return false;
    const int NPX = G * (TH + 2) * (BW + 2);
    const int apieces = (NPX * 5 + 63) / 64;
    if (apieces > 72) return false;
    pl->variant = variant; pl->BW = BW; pl->TH = TH; pl->G = G; pl->MB = MB; pl->NBT = NBT;
    const int nbb = 2;                                       // filter-row ring depth (S16Ring)
    pl->lds = 2 * apieces * 1024 + nbb * 6 * NBT * 1024 + 8192;        // + bias / statistics staging
    if (pl->lds > 160 * 1024) return false;
    return true;
}

extern "C" int kpx_conv3x3_bf16s_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr) {
    S16Plan pl;
    if (N <= 0 || K <= 0 || Nn <= 0 || ldin % 8 || (((uintptr_t)in_ptr) & 15) || ldin < K) return 0;
    if ((size_t)N * H * W * ldin * 2 >= 0x7fffffffu) return 0;
    return s16_plan(N, H, W, K, Nn, &pl) ? 1 : 0;
}

// bytes of the prepared filters of one direction (chunks of 32 over the gathered, blocks of 32 over the produced channels rounded up to 128)
extern "C" size_t kpx_conv3x3_bf16s_weights_bytes(int K, int Nn) {
    const int KC = (K + 31) / 32, NB = ((Nn + 127) / 128) * 4;
    return (size_t)KC * 18 * NB * 1024;
}

extern "C" int kpx_conv3x3_bf16s_prepare_f32(const float* w_hwio, int Cin, int Cout, int dgrad, void* Wf, void* stream) {
    if (!w_hwio || !Wf || Cin <= 0 || Cout <= 0) return KPX_EINVAL;
    const int K = dgrad ? Cout : Cin, Nn = dgrad ? Cin : Cout;
    const int KC = (K + 31) / 32, NB = ((Nn + 127) / 128) * 4;
    const size_t total = (size_t)KC * 18 * NB * 512;
    size_t nb = (total + 255) / 256; if (nb > 2048) nb = 2048;
    if (dgrad) hipLaunchKernelGGL(conv_bf16s_prepare_kernel<true>, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), w_hwio, Cin, Cout, KC, NB, (unsigned short*)Wf);
    else hipLaunchKernelGGL(conv_bf16s_prepare_kernel<false>, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), w_hwio, Cin, Cout, KC, NB, (unsigned short*)Wf);
    return kpx_launch_status();
}

// table: ndesc entries of {const float* w; void* Wf; int Cin, Cout, dgrad, NB} (32 bytes each) in device memory
extern "C" int kpx_conv3x3_bf16s_prepare_batch_f32(const void* table, int ndesc, void* stream) {
    if (!table || ndesc <= 0) return KPX_EINVAL;
    hipLaunchKernelGGL(conv_bf16s_prepare_batch_kernel, dim3(64, (unsigned)ndesc), dim3(256), 0, kpx_stream(stream), (const S16PrepDesc*)table, ndesc);
    return kpx_launch_status();
}

extern "C" int kpx_conv3x3_bf16s_stats_tiles(int N, int H, int W, int K, int Nn) {
    S16Plan pl;
    if (!s16_plan(N, H, W, K, Nn, &pl)) return 0;
    return (int)(((long)N * H * W) / (pl.MB * 32));
}

template <int WN, int Q, int P, int BW, int STATS>
static hipError_t s16_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16s_kernel<WN, Q, P, BW, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int WN, int Q, int P, int STATS>
static void s16_go_bw(const S16Geom& g, const S16Plan& pl, unsigned blocks, hipStream_t s) {
    if (pl.BW == 32) hipLaunchKernelGGL((conv3x3_bf16s_kernel<WN, Q, P, 32, STATS>), dim3(blocks), dim3(512), pl.lds, s, g, pl.TH, pl.G);
    else if (pl.BW == 16) hipLaunchKernelGGL((conv3x3_bf16s_kernel<WN, Q, P, 16, STATS>), dim3(blocks), dim3(512), pl.lds, s, g, pl.TH, pl.G);
    else hipLaunchKernelGGL((conv3x3_bf16s_kernel<WN, Q, P, 8, STATS>), dim3(blocks), dim3(512), pl.lds, s, g, pl.TH, pl.G);
}
template <int STATS>
static void s16_go(const S16Geom& g, const S16Plan& pl, unsigned blocks, hipStream_t s) {
    switch (pl.variant) {
        case 0: s16_go_bw<2, 2, 4, STATS>(g, pl, blocks, s); break;
        case 1: s16_go_bw<2, 2, 2, STATS>(g, pl, blocks, s); break;
        case 2: s16_go_bw<1, 2, 2, STATS>(g, pl, blocks, s); break;
        case 3: s16_go_bw<1, 1, 2, STATS>(g, pl, blocks, s); break;
        case 4: s16_go_bw<2, 2, 1, STATS>(g, pl, blocks, s); break;
        default: s16_go_bw<2, 1, 1, STATS>(g, pl, blocks, s); break;
    }
}

// persistent grid: one workgroup per CU (147 KB of LDS each), every workgroup the same number of tiles where that divides
static unsigned s16_grid(int total) {
    if (total <= 256) return (unsigned)total;
    const int per = (total + 255) / 256;
    return (unsigned)((total + per - 1) / per);
}

static int kpx_conv3x3_bf16s_attrs() {
    if (kpx_first_use_on_device(&s16_attr_mask)) {
        hipError_t e = hipSuccess;
#define S16_ATTR(wn, q, p) \
        if (e == hipSuccess) e = s16_attr<wn, q, p, 32, 0>(); if (e == hipSuccess) e = s16_attr<wn, q, p, 16, 0>(); if (e == hipSuccess) e = s16_attr<wn, q, p, 8, 0>(); \
        if (e == hipSuccess) e = s16_attr<wn, q, p, 32, 1>(); if (e == hipSuccess) e = s16_attr<wn, q, p, 16, 1>(); if (e == hipSuccess) e = s16_attr<wn, q, p, 8, 1>(); \
        if (e == hipSuccess) e = s16_attr<wn, q, p, 32, 2>(); if (e == hipSuccess) e = s16_attr<wn, q, p, 16, 2>(); if (e == hipSuccess) e = s16_attr<wn, q, p, 8, 2>();
        S16_ATTR(2, 2, 4) S16_ATTR(2, 2, 2) S16_ATTR(1, 2, 2) S16_ATTR(1, 1, 2) S16_ATTR(2, 2, 1) S16_ATTR(2, 1, 1)
#undef S16_ATTR
        if (e != hipSuccess) return -(int)e;
    }
    return 0;
}

// in [N,H,W,K] bf16 (pixel stride ldin elements), Wf prepared for (K gathered, Nn produced), out [N,H,W,Nn] bf16 (out_f32 = 0) or fp32
// (pixel stride ldout elements).  mask (optional, bf16 [N,H,W,Nn], pixel stride ldmask): the output is zeroed where mask <= 0.
// stats (optional): [tiles][2][Nn] fp32 per-workgroup sums / sums of squares of the fp32 outputs before the activation (kpx_conv3x3_bf16s_stats_tiles tiles).
extern "C" int kpx_conv3x3_bf16s(const void* in, int N, int H, int W, int K, int ldin, const void* Wf, const float* bias,
                                 void* out, int Nn, int ldout, int out_f32, int act, const void* mask, int ldmask, float* stats, void* stream) {
    S16Plan pl;
    if (!in || !Wf || !out || !kpx_conv3x3_bf16s_eligible(N, H, W, K, Nn, ldin, in) || ldout < Nn || (mask && (out_f32 || ldmask % 8 || ldmask < Nn))) return KPX_EINVAL;
    if (out_f32 ? (Nn % 4 || ldout % 4 || (((uintptr_t)out) & 15)) : (Nn % 8 || ldout % 8 || (((uintptr_t)out) & 15))) return KPX_EINVAL;
    if (!s16_plan(N, H, W, K, Nn, &pl)) return KPX_EINVAL;
    { const int rc = kpx_conv3x3_bf16s_attrs(); if (rc) return rc; }
    S16Geom g{};
    g.x = in; g.y = out; g.Wf = Wf; g.bias = bias; g.mask = mask; g.stats = stats;
    g.N = N; g.H = H; g.W = W; g.K = K; g.ldx = ldin; g.Nn = Nn; g.ldy = ldout; g.ldm = ldmask; g.act = act; g.out_f32 = out_f32;
    g.KC = (K + 31) / 32; g.NB = ((Nn + 127) / 128) * 4;
    g.tiles_y = H / pl.TH; g.tiles_x = W / pl.BW; g.ngrp = N / pl.G; g.ntc = (Nn + pl.NBT * 32 - 1) / (pl.NBT * 32);
    g.total_tiles = g.ngrp * g.tiles_y * g.tiles_x * g.ntc;
    const unsigned blocks = s16_grid(g.total_tiles);
    hipStream_t s = kpx_stream(stream);
    if (stats) s16_go<1>(g, pl, blocks, s); else s16_go<0>(g, pl, blocks, s);
    return kpx_launch_status();
}

// Data gradient towards y = relu(batch norm(.)) with that batch norm's backward sums from the epilogue (STATS 2 above): dz = conv(dy, wf) gated
// by bn_y > 0, stored bf16; stats[tile][2][Nn] = sum(dz), sum(dz * (bn_y - beta)) per workgroup (kpx_conv3x3_bf16s_stats_tiles tiles).
extern "C" int kpx_conv3x3_bf16s_bnbwd(const void* in, int N, int H, int W, int K, int ldin, const void* Wf, void* out, int Nn, int ldout,
                                       const void* bn_y, int ld_bn_y, const float* beta, float* stats, void* stream) {
    S16Plan pl;
    if (!in || !Wf || !out || !bn_y || !beta || !stats || !kpx_conv3x3_bf16s_eligible(N, H, W, K, Nn, ldin, in) || ldout < Nn || Nn % 8 || ldout % 8 || ld_bn_y % 8 ||
        ld_bn_y < Nn || ((((uintptr_t)out) | ((uintptr_t)bn_y)) & 15))
        return KPX_EINVAL;
    if (!s16_plan(N, H, W, K, Nn, &pl)) return KPX_EINVAL;
    int rc = kpx_conv3x3_bf16s_attrs();
    if (rc) return rc;
    S16Geom g{};
    g.x = in; g.y = out; g.Wf = Wf; g.bias = beta; g.mask = bn_y; g.stats = stats;
    g.N = N; g.H = H; g.W = W; g.K = K; g.ldx = ldin; g.Nn = Nn; g.ldy = ldout; g.ldm = ld_bn_y; g.act = KPX_ACT_NONE; g.out_f32 = 0;
    g.KC = (K + 31) / 32; g.NB = ((Nn + 127) / 128) * 4;
    g.tiles_y = H / pl.TH; g.tiles_x = W / pl.BW; g.ngrp = N / pl.G; g.ntc = (Nn + pl.NBT * 32 - 1) / (pl.NBT * 32);
    g.total_tiles = g.ngrp * g.tiles_y * g.tiles_x * g.ntc;
    const unsigned blocks = s16_grid(g.total_tiles);
    s16_go<2>(g, pl, blocks, kpx_stream(stream));
    return kpx_launch_status();
}
