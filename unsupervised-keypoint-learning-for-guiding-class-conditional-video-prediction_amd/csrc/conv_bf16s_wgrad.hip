// Weight gradient of the 3x3 stride-1 SAME layers in the bf16 configuration (gradient of layers.conv, reference models/networks/layers.py:6-9,
// for the translator / encoder / key-point-detector layers of models/networks/__init__.py:13-24,50-62,80-97):
//     dw[tap (r,s)][ci][co] = sum over pixels p of x[p + (r-1, s-1)][ci] * dy[p][co],      x, dy bf16 in HBM, fp32 accumulate, dw fp32.
// A GEMM per tap with K = pixels, on v_mfma_f32_32x32x16_bf16: A = x^T (rows = input channels), B = dy (columns = output channels).  Both
// operands are pixel-major in memory (NHWC) while an MFMA fragment wants eight consecutive K = pixels of ONE channel per lane: the tiles are
// staged exactly as they lie in HBM ([pixel][channel], LDS-DMA, no staging registers) and read back TRANSPOSED by ds_read_b64_tr_b16
// (4 pixels x 16 channels per 16-lane group, one 16-bit element per pixel in a lane's result).
//   * workgroup = 8 wavefronts = AB x BB blocks of 32 x 32 (ci, co) x WS pixel sub-ranges (AB * BB * WS = 8); every wavefront keeps the NINE
//     taps of its block: 9 accumulators of 32 x 32;
//   * a stage = 128 M output pixels (TR rows x TW columns, 8 M k-steps of 16 consecutive pixels of a row): the dy tile and the (TR+2) x (TW+4)
//     x patch, double buffered, one barrier per stage;
//   * per k-step a wavefront needs 2 quads of dy and, per filter ROW, three consecutive quads of x (12 pixels): the three column taps are
//     the same 12 pixels shifted by 0 / 1 / 2 -- a register funnel shift (v_alignbit) instead of two more LDS reads -- and a wavefront walks
//     consecutive rows, so two of the three patch rows are the previous k-step's: 5 transposed reads (2.5 KB) per 9 MFMAs;
//   * bank conflicts: the four pixel rows of a transposed read must fall in four different 64-B quarters of the 256-B bank row: the 64-B
//     channel blocks of a pixel are XOR-swizzled with the pixel index (on the SOURCE side of the LDS-DMA, the LDS image stays lane-linear);
//   * the K range (all pixels) is split over workgroups into partial slabs [splits][9][Cin][Cout] fp32 (the WS wavefronts of a block are summed
//     through LDS first), summed in a fixed tree by wgrad16_reduce_kernel (bitwise reproducible, no float atomics).
#include "kpx_common.h"
#include "kpx_env.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

#define W16_OOB 0x7ffffff0

struct W16Geom {
    const void* x; const void* dy; float* out;
    int N, H, W, Cin, ldx, Cout, lddy;
    int TR, TW;                     // stage tile: TR rows x TW columns = 128 pixels
    int tiles_x, tiles_y, tiles;    // per image / total
    int tps;                        // tiles per split
    int ct, kt;                     // channel tiles (32 AB input, 32 BB output channels each)
    size_t slab;                    // floats per slab = 9 * Cin * Cout
};

// One LDS-DMA instruction (16 B per lane to LDS address `lds` + 16 lane) issued from inline assembly: hipcc treats the transposed LDS reads
// (an intrinsic without memory operands) as readers of everything an LDS-DMA it knows about may write and puts `s_waitcnt vmcnt(0)` in front
// of the first read after a request -- the next stage's prefetch then completed before the current stage's first MFMA (found in the ISA of
// round 5's first version).  The waits of this kernel are explicit (vmcnt(0) + barrier at the end of a stage), so the compiler need not know.
typedef int w16_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w16_i32x4 w16_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    w16_i32x4 r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}
__device__ __forceinline__ void w16_dma16(const w16_i32x4 rsrc, const void* lds, int voffset) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr_t)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(l), "v"(voffset), "s"(rsrc) : "memory");       // (m0 is a reserved register to hipcc: it cannot be named as a clobber, and nothing else in this kernel uses it)
}

template <int V> struct w16_ic { static constexpr int value = V; };
// XOR swizzle of the 64-B channel block `cb` of pixel q, for NBLK blocks per pixel (see the header)
template <int NBLK>
__device__ __forceinline__ int w16_swz(int cb, int q) { return NBLK == 4 ? (cb ^ (q & 3)) : NBLK == 2 ? (cb ^ ((q >> 1) & 1)) : cb; }

template <int AB, int BB, int WS, int M>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_bf16_kernel(const W16Geom g) {
    static_assert(AB * BB * WS == 8, "eight wavefronts");
    constexpr int SP = 128 * M;                              // output pixels of a stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int TR = g.TR, TW = g.TW, PWX = TW + 4;
    const int XPIX = (TR + 2) * PWX;                        // patch pixels (the last row may be read up to 3 pixels past its end: + slack below)
    const int XUNITS = XPIX * 4 * AB, YUNITS = SP * 4 * BB;
    const int XPIECES = (XUNITS + 63) >> 6, YPIECES = (YUNITS + 63) >> 6;
    const int XBYTES = (XPIECES << 10) + 1024, YBYTES = YPIECES << 10;
    unsigned char* const Xs = smem;                         // [2][XBYTES]
    unsigned char* const Ys = smem + 2 * XBYTES;            // [2][YBYTES]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5, g16 = li >> 4, i16 = li & 15;
    const int ai = wave % AB, bi = (wave / AB) % BB, wsi = wave / (AB * BB);
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int kti = L % g.kt; L /= g.kt;
    const int cti = L % g.ct; L /= g.ct;
    const int split = L;
    const int ci0 = cti * 32 * AB, co0 = kti * 32 * BB;
    const int cin8 = (g.Cin + 7) & ~7, cout8 = (g.Cout + 7) & ~7;
    const int t_beg = split * g.tps, t_end = min(g.tiles, t_beg + g.tps);

    const w16_i32x4 rs_x = w16_rsrc(g.x, (unsigned)((size_t)g.N * g.H * g.W * g.ldx * 2));
    const w16_i32x4 rs_y = w16_rsrc(g.dy, (unsigned)((size_t)g.N * g.H * g.W * g.lddy * 2));

    // ---- LDS-DMA pieces of this wavefront (piece = wave + 8 i): per lane the patch / tile position and the channel unit it fetches
    constexpr int XPW = 6, YPW = 9;                          // pieces per wavefront: host keeps XPIECES <= 48, YPIECES <= 72
    int x_rel[XPW], x_pr[XPW], x_pc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int S = (wave + 8 * i) * 64 + lane, q = S / (4 * AB), u = S - q * (4 * AB);
        const int pr = q / PWX, pc = q - pr * PWX;
        const int cu = (w16_swz<AB>(u >> 2, q) << 2) | (u & 3);
        const bool ok = q < XPIX && ci0 + 8 * cu < cin8;
        x_pr[i] = ok ? pr : -100000; x_pc[i] = pc;
        x_rel[i] = ((pr - 1) * g.W + (pc - 1)) * g.ldx * 2 + (ci0 + 8 * cu) * 2;
    }
    int y_rel[YPW];
#pragma unroll
    for (int i = 0; i < YPW; ++i) {
        const int S = (wave + 8 * i) * 64 + lane, q = S / (4 * BB), u = S - q * (4 * BB);
        const int pr = q / TW, pc = q - pr * TW;
        const int cu = (w16_swz<BB>(u >> 2, q) << 2) | (u & 3);
        const bool ok = wave + 8 * i < YPIECES && q < SP && co0 + 8 * cu < cout8;
        y_rel[i] = ok ? (pr * g.W + pc) * g.lddy * 2 + (co0 + 8 * cu) * 2 : -1;
    }
    auto issue = [&](int tile, int buf) {
        const int n = tile / (g.tiles_x * g.tiles_y), rem = tile - n * (g.tiles_x * g.tiles_y);
        const int y0 = (rem / g.tiles_x) * TR, x0 = (rem % g.tiles_x) * TW;
        const int base_x = ((n * g.H + y0) * g.W + x0) * g.ldx * 2, base_y = ((n * g.H + y0) * g.W + x0) * g.lddy * 2;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            if (wave + 8 * i < XPIECES) {
                const int iy = y0 - 1 + x_pr[i], ix = x0 - 1 + x_pc[i];
                const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                w16_dma16(rs_x, Xs + buf * XBYTES + (wave + 8 * i) * 1024, ok ? base_x + x_rel[i] : W16_OOB);
            }
        }
#pragma unroll
        for (int i = 0; i < YPW; ++i) {
            if (wave + 8 * i < YPIECES)
                w16_dma16(rs_y, Ys + buf * YBYTES + (wave + 8 * i) * 1024, y_rel[i] >= 0 ? base_y + y_rel[i] : W16_OOB);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tp][e] = 0.f;

    // transposed-read lane roles: this lane supplies the address of pixel (quad base + i16 / 4), channels 16 g16 + 4 (i16 % 4) .. + 3 of the block
    const int rq = i16 >> 2, rp = i16 & 3;
    const int xch = (16 * g16 + 4 * rp) * 2, ych = xch;       // byte offset inside the 64-B channel block
    constexpr int XPB = 64 * AB, YPB = 64 * BB;                // bytes per pixel
    // Every transposed read addresses pixel q = (multiple of 4) + rq: the patch / tile widths, the k-step columns and the lane's 8 lh + 4 n
    // are multiples of 4, so the XOR swizzle of q is a constant of the lane, and an address is (lane constant) + (wave-uniform offset of the
    // k-step) + (immediate of the read) -- one v_add per filter row and k-step instead of the index arithmetic of eleven reads.
    const int swx = AB == 4 ? (ai ^ rq) : AB == 2 ? (ai ^ ((rq >> 1) & 1)) : ai;
    const int swy = BB == 4 ? (bi ^ rq) : BB == 2 ? (bi ^ ((rq >> 1) & 1)) : bi;
    const int x_lane = (8 * lh + rq) * XPB + swx * 64 + xch;
    const int y_lane = (8 * lh + rq) * YPB + swy * 64 + ych;
    const int rowstep = __builtin_amdgcn_readfirstlane(PWX * XPB);

    if (t_beg < t_end) {
        // zero the slack behind the patches once (reads past the last patch row's end must stay finite: they meet no accumulator, but NaN bits would)
        for (int i = t; i < 256; i += 512) {
            reinterpret_cast<u32x4*>(Xs + (XPIECES << 10))[i & 63] = u32x4{0, 0, 0, 0};
            reinterpret_cast<u32x4*>(Xs + XBYTES + (XPIECES << 10))[i & 63] = u32x4{0, 0, 0, 0};
        }
        issue(t_beg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int tile = t_beg; tile < t_end; ++tile) {
        const int buf = (tile - t_beg) & 1;
        if (tile + 1 < t_end) issue(tile + 1, buf ^ 1);
        const unsigned char* const Xb = Xs + buf * XBYTES;
        const unsigned char* const Yb = Ys + buf * YBYTES;
        // A k-step = 16 consecutive pixels of one tile row: patch rows row .. row + 2 against the dy row.  A wavefront walks a RUN of consecutive
        // rows at one 16-column position, so that two of a k-step's three patch-row fragments are the previous k-step's: a ring of four
        // patch-row fragments (three in use, one being read) and two dy fragments -- ONE new patch row (3 transposed reads) + the dy fragment
        // (2) per 9 MFMAs instead of 11 reads (the kernel was bound by the LDS reads: 54 % LDS against 39 % matrix-pipe activity).  The reads of
        // k-step j + 1 are issued before the MFMAs of k-step j (sched_barrier keeps hipcc from sinking them to their use).
        constexpr int KSTEPS = 8 * M / WS;
        unsigned xr[4][6], bq[2][4];
        auto rd_x = [&](const unsigned char* Xrow, int slot) {   // patch columns col + 8 lh + 0..11 of one patch row, input channel 32 ai + li
#pragma unroll
            for (int n3 = 0; n3 < 3; ++n3) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Xrow + 4 * n3 * XPB));
                const u32x2 vv = __builtin_bit_cast(u32x2, v);
                xr[slot][2 * n3] = vv[0]; xr[slot][2 * n3 + 1] = vv[1];
            }
        };
        auto rd_y = [&](const unsigned char* Yk, int slot) {     // pixels (row, col + 8 lh + 0..7), output channel 32 bi + li
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Yk + 4 * n2 * YPB));
                const u32x2 vv = __builtin_bit_cast(u32x2, v);
                bq[slot][2 * n2] = vv[0]; bq[slot][2 * n2 + 1] = vv[1];
            }
        };
        auto mm = [&](int s0, int ys) {                         // patch rows in ring slots s0, s0 + 1, s0 + 2 (mod 4) = filter rows 0, 1, 2
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, u32x4{bq[ys][0], bq[ys][1], bq[ys][2], bq[ys][3]});
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const unsigned* const d = xr[(s0 + r) & 3];
                // the three column taps are the same 12 pixels shifted by 0 / 1 / 2
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, u32x4{d[0], d[1], d[2], d[3]});
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbit(d[1], d[0], 16), __builtin_amdgcn_alignbit(d[2], d[1], 16),
                                                                  __builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(d[4], d[3], 16)});
                const bf16x8 a2 = __builtin_bit_cast(bf16x8, u32x4{d[1], d[2], d[3], d[4]});
                acc[3 * r + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfrag, acc[3 * r + 0], 0, 0, 0);
                acc[3 * r + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfrag, acc[3 * r + 1], 0, 0, 0);
                acc[3 * r + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bfrag, acc[3 * r + 2], 0, 0, 0);
            }
        };
        const int ystep = __builtin_amdgcn_readfirstlane(TW * YPB);
        auto run = [&](auto rl, int row0, int col) {            // RL k-steps: tile rows row0 .. row0 + RL - 1 at columns col .. col + 15
            constexpr int RL = decltype(rl)::value;
            const unsigned char* const Xk = Xb + (row0 * PWX + col) * XPB + x_lane;
            const unsigned char* const Yk = Yb + (row0 * TW + col) * YPB + y_lane;
            rd_x(Xk, 0); rd_x(Xk + rowstep, 1); rd_x(Xk + 2 * rowstep, 2); rd_y(Yk, 0);
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                if (j + 1 < RL) { rd_x(Xk + (j + 3) * rowstep, (j + 3) & 3); rd_y(Yk + (j + 1) * ystep, (j + 1) & 1); }
                __builtin_amdgcn_sched_barrier(0);
                mm(j & 3, j & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // this wavefront's KSTEPS consecutive (column half, row) units of the stage, column half major, in runs of RL rows (a lone wavefront
        // owns the whole stage: both column halves of a 32-wide tile / both row halves of a 16-wide one -- two runs, one loop body)
        constexpr int NRUN = WS == 1 ? 2 : 1, RL = KSTEPS / NRUN;
#pragma unroll 1
        for (int h = 0; h < NRUN; ++h) {
            const int u0 = __builtin_amdgcn_readfirstlane(wsi * KSTEPS + h * RL);
            const int colh = TW == 32 ? u0 / TR : 0, row0 = TW == 32 ? u0 - colh * TR : u0;
            run(w16_ic<RL>{}, row0, colh * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- partial slab of this workgroup: [split][tap][ci][co]; acc[tap][e]: ci = ci0 + 32 ai + 8 (e >> 2) + 4 lh + (e & 3), co = co0 + 32 bi + li
    float* const out = g.out + (size_t)split * g.slab;
    if (WS == 1) {
        const int co = co0 + 32 * bi + li;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + 32 * ai + 8 * (e >> 2) + 4 * lh + (e & 3);
                if (ci < g.Cin && co < g.Cout) out[((size_t)tp * g.Cin + ci) * g.Cout + co] = acc[tp][e];
            }
        return;
    }
    // the WS wavefronts of a block hold partial sums over different pixels: summed through LDS tap by tap, in wavefront order
    float* const red = reinterpret_cast<float*>(smem);      // [8 wavefronts][16 e][64 lanes]
    constexpr int NBLK = AB * BB, PER = 16 / WS;             // blocks of the workgroup; outputs per thread and tap
#pragma unroll 1
    for (int tp = 0; tp < 9; ++tp) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) v = tp == k ? acc[k][e] : v;
            red[(wave * 16 + e) * 64 + lane] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = t + 512 * j, ab = idx >> 10, rem = idx & 1023, e = rem >> 6, ln = rem & 63;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < WS; ++w) a += red[((w * NBLK + ab) * 16 + e) * 64 + ln];
            const int ci = ci0 + 32 * (ab % AB) + 8 * (e >> 2) + 4 * (ln >> 5) + (e & 3), co = co0 + 32 * (ab / AB) + (ln & 31);
            if (ci < g.Cin && co < g.Cout) out[((size_t)tp * g.Cin + ci) * g.Cout + co] = a;
        }
    }
}

// dw[i] = sum_s slabs[s][i]: a workgroup owns 64 float4 outputs; its four 64-thread segments sum the slabs s = seg, seg + 4, .. (eight loads in
// flight), the four segment sums are added in order -- a fixed tree: bitwise reproducible
__global__ __launch_bounds__(256) void wgrad16_reduce_kernel(const float* __restrict__ slabs, size_t slab, int S, float* __restrict__ dw, int vec) {
    __shared__ f32x4 part[4][64];
    const int seg = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (vec) {
        const size_t n4 = slab >> 2, i = (size_t)blockIdx.x * 64 + ln;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (i < n4) {
            int s = seg;
            for (; s + 28 < S; s += 32) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 4 * u) * slab + 4 * i);
#pragma unroll
                for (int u = 0; u < 8; ++u) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
            }
            for (; s < S; s += 4) { const f32x4 b = *reinterpret_cast<const f32x4*>(slabs + (size_t)s * slab + 4 * i); a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
        }
        part[seg][ln] = a;
        __syncthreads();
        if (seg == 0 && i < n4) {
            f32x4 r = part[0][ln];
#pragma unroll
            for (int k = 1; k < 4; ++k) { r[0] += part[k][ln][0]; r[1] += part[k][ln][1]; r[2] += part[k][ln][2]; r[3] += part[k][ln][3]; }
            *reinterpret_cast<f32x4*>(dw + 4 * i) = r;
        }
        return;
    }
    const size_t i = (size_t)blockIdx.x * 64 + ln;
    float a = 0.f;
    if (i < slab) for (int s = seg; s < S; s += 4) a += slabs[(size_t)s * slab + i];
    part[seg][ln][0] = a;
    __syncthreads();
    if (seg == 0 && i < slab) dw[i] = ((part[0][ln][0] + part[1][ln][0]) + part[2][ln][0]) + part[3][ln][0];
}

struct W16Plan { int variant, AB, BB, WS, M, TR, TW, ct, kt, tiles, S, tps, lds; };
// variants (AB, BB, WS, M): 0 (2,4,1,1) 64 x 128 channels; 1 (2,2,2,2) 64 x 64; 2 (1,1,8,4) 32 x 32; 3 (2,1,4,2) 64 x 32; 4 (1,2,4,2) 32 x 64; 5 (1,4,2,1) 32 x 128
static bool w16_plan(int N, int H, int W, int Cin, int Cout, W16Plan* pl) {
    if (W % 16 || N <= 0) return false;
    int v;
    if (Cin > 32) { v = Cout > 64 ? 0 : Cout > 32 ? 1 : 3; }
    else { v = Cout > 64 ? 5 : Cout > 32 ? 4 : 2; }
    static const int abs_[6] = {2, 2, 1, 2, 1, 1}, bbs[6] = {4, 2, 1, 1, 2, 4}, wss[6] = {1, 2, 8, 4, 4, 2}, ms[6] = {1, 2, 4, 2, 2, 1};
    const int TW = W >= 32 ? 32 : 16;
    int M = ms[v];
    while (M > 1 && (H % (128 * M / TW))) M >>= 1;           // (stages of whole rows inside one image)
    const int TR = 128 * M / TW;
    if (W % TW || H % TR) return false;
    if (M != ms[v]) return false;                            // (every layer of the path takes its variant's stage; other shapes: the fp32 kernels)
    pl->variant = v; pl->AB = abs_[v]; pl->BB = bbs[v]; pl->WS = wss[v]; pl->M = M; pl->TR = TR; pl->TW = TW;
    pl->ct = (Cin + 32 * pl->AB - 1) / (32 * pl->AB); pl->kt = (Cout + 32 * pl->BB - 1) / (32 * pl->BB);
    pl->tiles = N * (H / TR) * (W / TW);
    // splits: ONE round of the chip's 256 CUs (every workgroup writes a slab of its whole accumulator set: more workgroups = more slab bytes),
    // at least two stages per workgroup
    long S = 256 / ((long)pl->ct * pl->kt);
    if (S < 1) S = 1;
    if (S > pl->tiles / 2) S = pl->tiles / 2;
    if (S < 1) S = 1;
    pl->tps = (int)((pl->tiles + S - 1) / S);
    pl->S = (pl->tiles + pl->tps - 1) / pl->tps;
    const int xpix = (TR + 2) * (TW + 4);
    const int xp = (xpix * 4 * pl->AB + 63) / 64, yp = (128 * M * 4 * pl->BB + 63) / 64;
    if (xp > 48 || yp > 72) return false;
    pl->lds = 2 * (xp * 1024 + 1024) + 2 * yp * 1024;
    return pl->lds <= 160 * 1024;
}

extern "C" int kpx_conv3x3_wgrad_bf16_eligible(int N, int H, int W, int Cin, int ldx, int Cout, int lddy, const void* x, const void* dy) {
    W16Plan pl;
    if (!x || !dy || Cin <= 0 || Cout <= 0 || ldx % 8 || lddy % 8 || ldx < ((Cin + 7) & ~7) || lddy < ((Cout + 7) & ~7) || ((((uintptr_t)x) | ((uintptr_t)dy)) & 15)) return 0;
    if ((size_t)N * H * W * ldx * 2 >= 0x7fffffffu || (size_t)N * H * W * lddy * 2 >= 0x7fffffffu) return 0;
    return w16_plan(N, H, W, Cin, Cout, &pl) ? 1 : 0;
}
extern "C" size_t kpx_conv3x3_wgrad_bf16_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    W16Plan pl;
    if (!w16_plan(N, H, W, Cin, Cout, &pl)) return 0;
    return (size_t)pl.S * 9 * Cin * Cout * sizeof(float);
}

template <int AB, int BB, int WS, int M>
static hipError_t w16_attr() { return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_kernel<AB, BB, WS, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }

// x bf16 [N,H,W,>=Cin] (pixel stride ldx; channels Cin .. roundup8(Cin) must be finite: they are gathered and discarded), dy bf16 [N,H,W,>=Cout],
// dw fp32 [3,3,Cin,Cout] (written, not accumulated); workspace: kpx_conv3x3_wgrad_bf16_workspace_bytes
extern "C" int kpx_conv3x3_wgrad_bf16(const void* x, int N, int H, int W, int Cin, int ldx, const void* dy, int Cout, int lddy,
                                      float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    static std::atomic<unsigned long long> attr_mask{0};
    W16Plan pl;
    if (!dw || !kpx_conv3x3_wgrad_bf16_eligible(N, H, W, Cin, ldx, Cout, lddy, x, dy) || !w16_plan(N, H, W, Cin, Cout, &pl)) return KPX_EINVAL;
    const size_t need = (size_t)pl.S * 9 * Cin * Cout * sizeof(float);
    if (!workspace || workspace_bytes < need) return KPX_EINVAL;
    if (kpx_first_use_on_device(&attr_mask)) {
        hipError_t e = w16_attr<2, 4, 1, 1>();
        if (e == hipSuccess) e = w16_attr<2, 2, 2, 2>();
        if (e == hipSuccess) e = w16_attr<1, 1, 8, 4>();
        if (e == hipSuccess) e = w16_attr<2, 1, 4, 2>();
        if (e == hipSuccess) e = w16_attr<1, 2, 4, 2>();
        if (e == hipSuccess) e = w16_attr<1, 4, 2, 1>();
        if (e != hipSuccess) return -(int)e;
    }
    W16Geom g{};
    g.x = x; g.dy = dy; g.out = (float*)workspace;
    g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.ldx = ldx; g.Cout = Cout; g.lddy = lddy;
    g.TR = pl.TR; g.TW = pl.TW; g.tiles_x = W / pl.TW; g.tiles_y = H / pl.TR; g.tiles = pl.tiles; g.tps = pl.tps; g.ct = pl.ct; g.kt = pl.kt;
    g.slab = (size_t)9 * Cin * Cout;
    hipStream_t s = kpx_stream(stream);
    const dim3 grid((unsigned)(pl.S * pl.ct * pl.kt));
    switch (pl.variant) {
        case 0: hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<2, 4, 1, 1>), grid, dim3(512), pl.lds, s, g); break;
        case 1: hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<2, 2, 2, 2>), grid, dim3(512), pl.lds, s, g); break;
        case 2: hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<1, 1, 8, 4>), grid, dim3(512), pl.lds, s, g); break;
        case 3: hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<2, 1, 4, 2>), grid, dim3(512), pl.lds, s, g); break;
        case 4: hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<1, 2, 4, 2>), grid, dim3(512), pl.lds, s, g); break;
        default: hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<1, 4, 2, 1>), grid, dim3(512), pl.lds, s, g); break;
    }
    int rc = kpx_launch_status();
    if (rc) return rc;
    const int vec = (g.slab % 4 == 0) && ((((uintptr_t)workspace) | ((uintptr_t)dw)) & 15) == 0;
    const size_t nb = ((vec ? g.slab / 4 : g.slab) + 63) / 64;
    hipLaunchKernelGGL(wgrad16_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float*)workspace, g.slab, pl.S, dw, vec);
    return kpx_launch_status();
}
