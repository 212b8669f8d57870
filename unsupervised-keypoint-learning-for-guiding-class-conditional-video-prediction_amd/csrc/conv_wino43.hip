// Fused Winograd F(4x4, 3x3) convolution for gfx950 (fp32, v_mfma_f32_32x32x2_f32): 36 multiplies per 4x4 output tile instead of 144,
// i.e. 4x fewer MFMAs than the direct kernel and 1.78x fewer than F(2x2,3x3) (conv_wino.hip), for the large 3x3 stride-1 SAME layers
// of the translator, VGG19 and the image encoder (reference models/networks/__init__.py:13,22,80-97, models/networks/vgg.py:51).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 6x6 input patch, g: 3x3 filter, Y: 4x4 outputs   (Lavin & Gray points 0, +-1, +-2, inf)
//
// fp32 accuracy: rel-L2 1.5e-6 .. 3.3e-6 against float64 (F(2x2,3x3): 2.8e-7 .. 4.9e-7, direct fp32 chain: 4.4e-7 .. 8.9e-7), i.e.
// inside the 1e-5 parity bar of a layer but ~6x looser than F(2x2,3x3): which layers run it is the caller's per-layer policy
// (ops.WINO43_EXCLUDE_*); the library never picks it on its own.
//
// One workgroup (8 wavefronts) = 16 x 32 output pixels (4 x 8 tiles = one 32-row MFMA block) x 64 output channels x all 36 points.
// Same skeleton as conv_wino_v2_kernel -- B fragments straight from the fragment-ordered, pre-transformed filters, raw patch and
// V double buffered with ONE barrier per 8-channel chunk -- with these differences:
//   * every wavefront multiplies 9 (point, cout half) accumulator blocks of 32x32 (144 VGPRs); its B fragments are a 3-deep ring
//     refilled in place, pinned by a sched_barrier per point (under register pressure hipcc otherwise sinks each refill to its use);
//   * roles, each with its own copy of the chunk loop so that their live registers never add up: wavefronts 0-5 produce one row of
//     B^T d B each (wave-uniform coefficients; tile = lane & 31, channel half = lane >> 5; 18-24 ds_read_b128 of the raw patch,
//     6 ds_write_b128 of V; 4-5 multiply first and transform afterwards), wavefronts 6-7 fetch the whole patch of a chunk through a
//     buffer descriptor (out-of-image units read as zero by the range check);
//   * the raw patch is stored as two channel-half planes of 16-B pixels with one slot of skew per 4 columns (column c at slot
//     c + c/4, row stride 42 slots): tile origins are 5 slots apart in x and 8 (mod 16) in y, so the 16 lanes of a ds_read_b128
//     group -- tiles {0-3, 12-15, 20-27} -- hit 16 different 16-B bank groups for every patch element (checked exhaustively);
//   * the epilogue runs in two passes of 18 points (rows 0-2 / 3-5 of the point grid) through LDS (147 KB); the points of each half
//     are spread 5,5,4,4 / 4,4,5,5 over the wavefronts of a cout half, so all eight deposit in both passes; each thread owns
//     (tile, 4 couts), applies A^T . A to the three point rows of the pass and keeps its 16 output pixels in registers between them;
//   * STATS: per-strip batch-norm sums from the epilogue; PACK: two 16 x 16 images per workgroup.
#include "kpx_common.h"
#include "kpx_env.h"
#include <stdlib.h>

struct Wino43Geom {
    const float* x; float* y; const float* U; const float* bias;
    int N, H, W, Cin, ldx, Cout, ldy, act;
    int Kp, Np;                              // U is [36][Kp/8][Np/32][2][32][4]
    int tiles_y, tiles_x;                    // 16 x 32-pixel regions per image
    const float* mask_y; int ld_mask;        // optional: zero the output where mask_y <= 0 (the ReLU backward of the tensor this gradient belongs to)
    float* pool_y; int ld_pool;              // optional: also write the 2x2 max-pool of the (activated) output
    float* stats;                            // STATS 1: [N * tiles_y * tiles_x * 8 strips of 4 x 16 pixels][2][Cout] sum / sum of squares of the output
    const float* bn_beta;                    // STATS 2 (data gradient feeding a ReLU'd batch norm whose output is mask_y): per strip sum(dz), sum(dz * (y - beta)),
};                                           //          dz = out where mask_y > 0 else 0; the masked gradient is what gets stored


// U[p = 6 i + j][c][n] = (G g G^T)[i][j] in the fragment order of conv_wino.hip (Uf[p][kc][nb][lh][li][4]).
struct KpxWino43Desc { const float* w; float* u; int cin, cout, dgrad, reserved; };
__device__ __forceinline__ void w43_transform_filter(const float* __restrict__ w, int Cin, int Cout, bool dg, size_t idx, int Kp, int Np, float* __restrict__ Uf) {
    const int K = dg ? Cout : Cin, Nn = dg ? Cin : Cout;
    const int KC = Kp >> 3, NB = Np >> 5;
    const int j = (int)(idx & 3), li = (int)((idx >> 2) & 31), lh = (int)((idx >> 7) & 1);
    const size_t blk = idx >> 8;
    const int nb = (int)(blk % NB), kc = (int)(blk / NB);
    const int c = 8 * kc + 4 * lh + j, n = 32 * nb + li;
    const bool real = c < K && n < Nn;
    // In DOUBLE precision, rounded once at the end: G's entries (1/6, 1/12, 1/24) are not representable, and an error in U is the SAME for
    // every tile of the layer -- a systematic, input-correlated output error that sums over pixels do not average out (it showed as 6e-4 in
    // the bias / batch-norm gradient sums of a zero-mean test tensor).  The transform runs once per optimiser update: its cost is irrelevant.
    double g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            g[r][q] = !real ? 0.0 : (double)(dg ? w[((size_t)((2 - r) * 3 + (2 - q)) * Cin + n) * Cout + c] : w[((size_t)(r * 3 + q) * Cin + c) * Cout + n]);
    // rows of G: [1/4,0,0], [-1/6,-1/6,-1/6], [-1/6,1/6,-1/6], [1/24,1/12,1/6], [1/24,-1/12,1/6], [0,0,1]
    double t[6][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double a = g[0][q], b = g[1][q], c2 = g[2][q];
        t[0][q] = 0.25 * a;
        t[1][q] = (-1.0 / 6.0) * (a + b + c2);
        t[2][q] = (-1.0 / 6.0) * (a - b + c2);
        t[3][q] = (1.0 / 24.0) * a + (1.0 / 12.0) * b + (1.0 / 6.0) * c2;
        t[4][q] = (1.0 / 24.0) * a - (1.0 / 12.0) * b + (1.0 / 6.0) * c2;
        t[5][q] = c2;
    }
    const size_t pstride = (size_t)KC * NB * 256;
    float* o = Uf + blk * 256 + (idx & 255);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double a = t[i][0], b = t[i][1], c2 = t[i][2];
        o[(size_t)(i * 6 + 0) * pstride] = (float)(0.25 * a);
        o[(size_t)(i * 6 + 1) * pstride] = (float)((-1.0 / 6.0) * (a + b + c2));
        o[(size_t)(i * 6 + 2) * pstride] = (float)((-1.0 / 6.0) * (a - b + c2));
        o[(size_t)(i * 6 + 3) * pstride] = (float)((1.0 / 24.0) * a + (1.0 / 12.0) * b + (1.0 / 6.0) * c2);
        o[(size_t)(i * 6 + 4) * pstride] = (float)((1.0 / 24.0) * a - (1.0 / 12.0) * b + (1.0 / 6.0) * c2);
        o[(size_t)(i * 6 + 5) * pstride] = (float)c2;
    }
}
__global__ __launch_bounds__(256) void wino43_filter_transform_batch_kernel(const KpxWino43Desc* __restrict__ descs) {
    const KpxWino43Desc d = descs[blockIdx.y];
    const bool dg = d.dgrad != 0;
    const int K = dg ? d.cout : d.cin, Nn = dg ? d.cin : d.cout;
    const int Kp = (K + 7) & ~7, Np = (Nn + 63) & ~63;
    const size_t total = (size_t)Kp * Np;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
        w43_transform_filter(d.w, d.cin, d.cout, dg, idx, Kp, Np, d.u);
}
__global__ __launch_bounds__(256) void wino43_filter_transform_kernel(const float* __restrict__ w, int Cin, int Cout, int dgrad, int Kp, int Np, float* __restrict__ Uf) {
    const size_t total = (size_t)Kp * Np;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
        w43_transform_filter(w, Cin, Cout, dgrad != 0, idx, Kp, Np, Uf);
}

// PACK: two 16 x 16 images side by side in one 16 x 32 region (VGG19 conv4_*, the 16 x 16 layers of the encoders): each image has its own
// 18-column patch; the second one starts 36 slots after the first (row stride 58), which keeps the reads conflict-free (the tile
// origins of the second image are 16 slots = one full bank sweep further than they would be in a 34-column patch).
#define W4_RS_OF(PACK) ((PACK) ? 58 : 42)
#define W4_V (36 * 32 * 8)                     // floats per V buffer (9216)
#define W4_RAW_OF(PACK) (2 * 18 * W4_RS_OF(PACK) * 4)      // floats per raw buffer (6048 / 8352)
#define W4_MAIN (2 * W4_RAW_OF(true) + 2 * W4_V)
#define W4_EPI (18 * 32 * 64)

// rows of B^T (= rows of the input transform): value = sum_m A[m] * d[R[m]]
__constant__ int w43_R[6][4] = {{0, 2, 4, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 3, 5, 5}};
__constant__ float w43_A[6][4] = {{4.f, -5.f, 1.f, 0.f}, {-4.f, -4.f, 1.f, 1.f}, {4.f, -4.f, -1.f, 1.f}, {-2.f, -1.f, 2.f, 1.f}, {2.f, -1.f, -2.f, 1.f}, {4.f, -5.f, 1.f, 0.f}};

__constant__ float w43_AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};

#ifdef KPX_WINO_STAMP      // diagnostic build only (profiles/wino43_stamps.sh): s_memtime stamps of every wavefront of the first 64 workgroups
static __device__ unsigned long long* w43_dbg = nullptr;
extern "C" int kpx_debug_w43_stamps(unsigned long long* buf) { return -(int)hipMemcpyToSymbol(HIP_SYMBOL(w43_dbg), &buf, sizeof(buf)); }
#define W4_STAMP(slot) do { if (dbgp) dbgp[(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W4_KSTAMP(k, j) do { if (dbgp && (k) >= 4 && (k) < 12) dbgp[16 + ((k) - 4) * 4 + (j)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_KSTAMP(k, j) do { } while (0)
#define W4_STAMP(slot) do { } while (0)
#endif


template <int STATS, bool PACK>
__global__ __launch_bounds__(512, 2) void conv_wino43_kernel(const Wino43Geom g) {
    constexpr int W4_RS = W4_RS_OF(PACK), W4_PLANE = 18 * W4_RS, W4_RAW = W4_RAW_OF(PACK);
    constexpr int W4_LOADER_UNITS = PACK ? 8 : 10;       // (pixel, half) units of a raw patch per thread of wavefronts 6-7: 1224 / 1024 in all
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const rawb = smem;
    float* const Vb = smem + 2 * W4_RAW;

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int ct = wave & 1, pg = wave >> 1;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int ntc = g.Np / 64;
    const int nti = L % ntc; L /= ntc;
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = PACK ? 2 * (L / g.tiles_y) : L / g.tiles_y;      // PACK: tiles_y = tiles_x = 1, the region holds images n and n + 1
    const int oy0 = by * 16, ox0 = bx * 32, n0 = nti * 64;

    // MFMA operands (all wavefronts).  Points 0-17 (rows 0-2 of the 6x6 grid) and 18-35 are each spread over the four wavefronts of a
    // cout half as 5,5,4,4 and 4,4,5,5, so that BOTH epilogue passes (one per half of the grid) have work for every wavefront:
    // block b of wavefront pg is point b + offA (b < 4), b + offB (b > 4); block 4 belongs to the first half for pg < 2.
    const int offA = pg == 0 ? 0 : pg == 1 ? 5 : pg == 2 ? 10 : 14;
    const int offB = pg == 0 ? 13 : pg == 1 ? 17 : pg == 2 ? 22 : 27;
    const int nfirst = pg < 2 ? 5 : 4;                   // blocks [0, nfirst) are first-half points
    const int off4 = pg < 2 ? offA : offB;
    const int a_rd0 = li * 8 + ((lh ^ ((li >> 3) & 1)) << 2);
    const int a_rdA = a_rd0 + offA * 256, a_rd4 = a_rd0 + off4 * 256, a_rdB = a_rd0 + offB * 256;
    const int KC = g.Kp >> 3, NB = g.Np >> 5;
    const int nb = nti * 2 + ct;
    // B fragments through a buffer descriptor over U: voffset = lane * 16 (fixed), soffset = (point, chunk, cout block) in bytes -- scalar.
    // (As 64-bit pointers this was 12 v_lshl_add_u64 + 18 s_add / s_addc per chunk and wavefront, beside an fp32 MFMA that hides none of it.)
    const int ub_pstride = KC * NB * 1024, ub_step = NB * 1024;                        // bytes per point / per chunk
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.U), 0, 36u * (unsigned)ub_pstride, 0x00020000);
    const int ulane = lane * 16;
    int ubsA = __builtin_amdgcn_readfirstlane((offA * KC * NB + nb) * 1024), ubs4 = __builtin_amdgcn_readfirstlane((off4 * KC * NB + nb) * 1024),
        ubsB = __builtin_amdgcn_readfirstlane((offB * KC * NB + nb) * 1024);
#define W4_AOFF(b) (((b) < 4 ? a_rdA : (b) == 4 ? a_rd4 : a_rdB) + (b) * 256)
#define W4_ULOAD(b, extra) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane, ((b) < 4 ? ubsA : (b) == 4 ? ubs4 : ubsB) + (b) * ub_pstride + (extra), 0))
    const int nchunks = g.Kp / 8;

#ifdef KPX_WINO_STAMP
    unsigned long long* dbgp = (w43_dbg && lane == 0 && blockIdx.x < 64) ? w43_dbg + ((size_t)blockIdx.x * 8 + wave) * 64 : nullptr;
    if (dbgp) { dbgp[0] = __builtin_amdgcn_s_memtime(); dbgp[8] = __builtin_amdgcn_s_memrealtime(); }
#endif
    f32x16 acc[9];
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    f32x4 ub[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) ub[b] = W4_ULOAD(b, 0);

    if (PACK) {                                          // the halo slots of both raw buffers are never written again
        for (int i = t; i < 2 * W4_RAW / 4; i += 512) *reinterpret_cast<f32x4*>(&rawb[i * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
    // B fragments: a ring of 3 points, each refilled in place (for the point 3 ahead, possibly of the next chunk) right after its MFMAs
    auto mfma = [&](const float* Vr, bool refill_next) {
        f32x4 av[2];
        av[0] = *reinterpret_cast<const f32x4*>(&Vr[W4_AOFF(0)]);
        const int nx = refill_next ? ub_step : 0;                        // (last chunk: harmless re-read instead of a branch)
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            if (b < 8) av[(b + 1) & 1] = *reinterpret_cast<const f32x4*>(&Vr[W4_AOFF(b + 1)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][j], ub[b % 3][j], acc[b], 0, 0, 0);
            if (b < 6) ub[b % 3] = W4_ULOAD(b + 3, 0);
            else ub[b % 3] = W4_ULOAD(b - 6, nx);
            __builtin_amdgcn_sched_barrier(0);           // or the scheduler sinks each refill to its use, 3 points later, and waits for it there
        }
        ubsA += ub_step; ubs4 += ub_step; ubsB += ub_step;
    };

    // The two roles run separate copies of the chunk loop (their register needs differ: 44 transient transform registers vs 40 of
    // patch data in flight); both pass exactly one barrier per chunk, after one in the prologue.
    if (wave < 6) {
        // ---- input transform: one row of B^T d B per wavefront for (tile = lane & 31, channel half = lane >> 5); wavefronts 4 and 5 take
        // rows 0 and 5, whose three-term sums need one LDS read less per column ----
        const int trow = wave < 4 ? wave + 1 : wave == 4 ? 0 : 5;
        const int ttile = lane & 31, thf = lane >> 5, tty = ttile >> 3, ttx = ttile & 7;
        const int trd = (thf * W4_PLANE + 4 * tty * W4_RS + 5 * ttx + (PACK && ttx >= 4 ? 16 : 0)) * 4;
        const int r0 = w43_R[trow][0] * W4_RS * 4, r1 = w43_R[trow][1] * W4_RS * 4, r2 = w43_R[trow][2] * W4_RS * 4, r3 = w43_R[trow][3] * W4_RS * 4;
        const float a0 = w43_A[trow][0], a1 = w43_A[trow][1], a2 = w43_A[trow][2], a3 = w43_A[trow][3];
        const int vwr = (6 * trow) * 256 + ttile * 8 + ((thf ^ ((ttile >> 3) & 1)) << 2);
        const int ktail = g.Cin - (nchunks - 1) * 8 - thf * 4;       // valid channels of this thread's half in the LAST chunk
        // MASK (compile time): the channel-tail select of a padded LAST chunk; as a run-time flag it was 18 v_cndmask per chunk for every chunk
        auto transform_t = [&](const float* rawR, float* Vw, auto mask_tag, auto three_tag) {
            constexpr bool THREE = decltype(three_tag)::value;
            constexpr bool mask_tail = decltype(mask_tag)::value;
            f32x4 tc[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int co = trd + (c + (c >> 2)) * 4;
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(&rawR[co + r0]);
                const f32x4 d1 = *reinterpret_cast<const f32x4*>(&rawR[co + r1]);
                const f32x4 d2 = *reinterpret_cast<const f32x4*>(&rawR[co + r2]);
                if (THREE) tc[c] = a0 * d0 + a1 * d1 + a2 * d2;
                else tc[c] = a0 * d0 + a1 * d1 + a2 * d2 + a3 * *reinterpret_cast<const f32x4*>(&rawR[co + r3]);
            }
            if (mask_tail) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (j >= ktail) tc[c][j] = 0.f;
            }
            *reinterpret_cast<f32x4*>(&Vw[vwr]) = 4.f * tc[0] - 5.f * tc[2] + tc[4];
            *reinterpret_cast<f32x4*>(&Vw[vwr + 256]) = -4.f * (tc[1] + tc[2]) + tc[3] + tc[4];
            *reinterpret_cast<f32x4*>(&Vw[vwr + 512]) = 4.f * (tc[1] - tc[2]) - tc[3] + tc[4];
            *reinterpret_cast<f32x4*>(&Vw[vwr + 768]) = 2.f * (tc[3] - tc[1]) - tc[2] + tc[4];
            *reinterpret_cast<f32x4*>(&Vw[vwr + 1024]) = 2.f * (tc[1] - tc[3]) - tc[2] + tc[4];
            *reinterpret_cast<f32x4*>(&Vw[vwr + 1280]) = 4.f * tc[1] - 5.f * tc[3] + tc[5];
        };
        auto transform = [&](const float* rawR, float* Vw, bool mask_tail, auto three_tag) {
            if (mask_tail) transform_t(rawR, Vw, std::true_type{}, three_tag);          // (wave-uniform branch)
            else transform_t(rawR, Vw, std::false_type{}, three_tag);
        };
        W4_STAMP(10);
        __syncthreads();                                 // raw[0] (chunk 0) staged by the loaders
        W4_STAMP(12);
        transform(rawb, Vb, nchunks == 1 && ktail < 4, std::false_type{});
        W4_STAMP(1);
        // wavefronts 4-5 (the second transform wavefront of SIMD 0 / 1) multiply first and transform afterwards, so that one of the two
        // always has MFMAs to issue while the other waits for LDS; separate loop copies keep the register allocation of each tight
        auto chunk_loop = [&](auto late_tag) {
            constexpr bool LATE = decltype(late_tag)::value;
            for (int k = 0; k < nchunks; ++k) {
                const int cur = k & 1;
                W4_KSTAMP(k, 0);
                __syncthreads();                         // V[cur] (chunk k) and raw[cur^1] (chunk k+1) are complete
                W4_KSTAMP(k, 1);
                if (LATE) { mfma(Vb + cur * W4_V, k + 1 < nchunks); W4_KSTAMP(k, 2); }
                if (k + 1 < nchunks) transform(rawb + (cur ^ 1) * W4_RAW, Vb + (cur ^ 1) * W4_V, k + 1 == nchunks - 1 && ktail < 4, late_tag);
                if (!LATE) { W4_KSTAMP(k, 2); mfma(Vb + cur * W4_V, k + 1 < nchunks); }
                W4_KSTAMP(k, 3);
            }
        };
        if (wave >= 4) chunk_loop(std::true_type{}); else chunk_loop(std::false_type{});
    } else {
        // ---- patch loader: the whole 18 x 34-pixel x 8-channel patch of a chunk, 10 (pixel, half) units per thread, through a buffer
        // descriptor of image n: out-of-image units carry an out-of-range offset and read as zero ----
        const int tl = t - 384;
        const unsigned img_bytes = (unsigned)g.H * g.W * g.ldx * 4u * (PACK ? 2u : 1u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x) + (size_t)n * g.H * g.W * g.ldx, 0, img_bytes, 0x00020000);
        int voff[W4_LOADER_UNITS], rdst[W4_LOADER_UNITS];
#pragma unroll
        for (int i = 0; i < W4_LOADER_UNITS; ++i) {
            const int u = tl + 128 * i;
            if (PACK) {                                  // only the 16 x 16 in-image pixels of the two images: the halo slots stay zero
                const int hf = u & 1, px = u >> 1, img = px >> 8, iy = (px >> 4) & 15, ix = px & 15;
                voff[i] = (((img * 16 + iy) * 16 + ix) * g.ldx + hf * 4) * 4;
                rdst[i] = (hf * W4_PLANE + (iy + 1) * W4_RS + (ix + 1) + ((ix + 1) >> 2) + img * 36) * 4;
            } else {
                const int uu = u < 1224 ? u : tl;        // the 56 missing units of the last round re-store the thread's first unit
                const int px = uu >> 1, hf = uu & 1, row = px / 34, col = px - row * 34;
                const int iy = oy0 - 1 + row, ix = ox0 - 1 + col;
                const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                voff[i] = ok ? ((iy * g.W + ix) * g.ldx + hf * 4) * 4 : (int)0x80000000;
                rdst[i] = (hf * W4_PLANE + row * W4_RS + col + (col >> 2)) * 4;
            }
        }
        f32x4 rr[W4_LOADER_UNITS];
        int soff = 0;
        auto load_raw = [&]() {
#pragma unroll
            for (int i = 0; i < W4_LOADER_UNITS; ++i) rr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0));
            soff += 32;
        };
        auto stage = [&](float* rawW) {
#pragma unroll
            for (int i = 0; i < W4_LOADER_UNITS; ++i) *reinterpret_cast<f32x4*>(&rawW[rdst[i]]) = rr[i];
        };
        W4_STAMP(10);
        load_raw();
        stage(rawb);
        W4_STAMP(11);
        if (nchunks > 1) load_raw();
        __syncthreads();
        W4_STAMP(12);
        if (nchunks > 1) stage(rawb + W4_RAW);
        if (nchunks > 2) load_raw();
        W4_STAMP(1);
        for (int k = 0; k < nchunks; ++k) {
            const int cur = k & 1;
            W4_KSTAMP(k, 0);
            __syncthreads();
            W4_KSTAMP(k, 1);
            if (k + 2 < nchunks) stage(rawb + cur * W4_RAW);
            if (k + 3 < nchunks) load_raw();
            W4_KSTAMP(k, 2);
            mfma(Vb + cur * W4_V, k + 1 < nchunks);
            W4_KSTAMP(k, 3);
        }
    }

    // epilogue: two passes through LDS, one per half of the 6x6 point grid (rows 0-2, rows 3-5): every wavefront deposits its 4-5 blocks
    // of that half (P[point][tile][64 couts], 147 KB), then each of the 512 threads applies the three point rows to its (tile, 4 couts)
    // and keeps the 4x4 output pixels in registers between the passes.  Rows written by the upper lane half (tile rows 4-7 of each 8)
    // have their two 32-cout halves swapped, so that one ds_write_b32 covers all 64 banks.
    float* const P = smem;
    const int otile = t >> 4, ocq = t & 15;
    const float* const Pr = P + otile * 64 + ((ocq * 4) ^ (((otile >> 2) & 1) << 5));
    f32x4 Y[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    W4_STAMP(2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                 // main-loop LDS reads (h = 0) / the first pass's P reads (h = 1) are done
        if (h == 0) W4_STAMP(3);
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            if (b == 4 ? (nfirst == 5) == (h == 0) : (b < 4) == (h == 0)) {
                const int hp = b + (b < 4 ? offA : b == 4 ? off4 : offB) - 18 * h;      // point inside the half
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    P[(hp * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + ((ct * 32 + li) ^ (lh << 5))] = acc[b][r];
            }
        }
        if (h == 0) W4_STAMP(4);
        __syncthreads();
        if (h == 0) W4_STAMP(5);
#pragma unroll 1
        for (int al = 0; al < 3; ++al) {                 // point row a = 3h + al; Y[ii][jj] += A^T[ii][a] * Q[jj]   (rolled: bounds the registers)
            const int a = 3 * h + al;
            const float c0 = w43_AT[0][a], c1 = w43_AT[1][a], c2 = w43_AT[2][a], c3 = w43_AT[3][a];
            f32x4 m[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) m[b] = *reinterpret_cast<const f32x4*>(&Pr[(al * 6 + b) * 2048]);
            // Q[jj] = sum_b M[a][b] A[b][jj]   (A^T rows: [1,1,1,1,1,0], [0,1,-1,2,-2,0], [0,1,1,4,4,0], [0,1,-1,8,-8,1])
            const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            f32x4 Q[4];
            Q[0] = m[0] + s12 + s34;
            Q[1] = d12 + 2.f * d34;
            Q[2] = s12 + 4.f * s34;
            Q[3] = d12 + 8.f * d34 + m[5];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                Y[0][jj] += c0 * Q[jj];
                Y[1][jj] += c1 * Q[jj];
                Y[2][jj] += c2 * Q[jj];
                Y[3][jj] += c3 * Q[jj];
            }
        }
    }
    W4_STAMP(6);
    const int c0o = n0 + ocq * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (c0o + q < g.Cout) bv[q] = g.bias[c0o + q];
    }
    // activation without branches: max(v, lo) then v > 0 ? v : slope * v   (none: lo = -inf, slope 1; relu: lo = 0; leaky: slope 0.01)
    const float lo = g.act == KPX_ACT_RELU ? 0.f : -__builtin_inff();
    const float slope = g.act == KPX_ACT_LRELU ? 0.01f : 1.f;
    const int oy = oy0 + 4 * (otile >> 3), ox = PACK ? 4 * (otile & 3) : ox0 + 4 * (otile & 7);
    const int on = PACK ? n + ((otile >> 2) & 1) : n;
    float* const obase = g.y + ((size_t)(on * g.H + oy) * g.W + ox) * g.ldy + c0o;
    const size_t cstr = (size_t)g.ldy, rstr = (size_t)g.W * g.ldy;
    const bool fast = (g.ldy & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.y) & 15) == 0) && n0 + 64 <= g.Cout;    // block-uniform
    f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};
    if (STATS == 2) {
        // data gradient dz of a ReLU'd batch norm's output z (= mask_y): store dz * [z > 0] and reduce this batch norm's backward sums
        // sum(dz), sum(dz * (z - beta)) (x_hat = (z - beta) / gamma wherever z > 0; the finalize divides by gamma) -- the separate reduction
        // pass over (dz, y) of kpx_bn_train_bwd_f32 is then not needed.  Launch preconditions (checked by the entry): fast stores, no bias / act.
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
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float z = fmaxf(v[q], lo); v[q] = z > 0.f ? z : z * slope; }
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
                    f32x4 p;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        p[q] = fmaxf(fmaxf(Y[2 * i][2 * j][q], Y[2 * i][2 * j + 1][q]), fmaxf(Y[2 * i + 1][2 * j][q], Y[2 * i + 1][2 * j + 1][q]));
                    *reinterpret_cast<f32x4*>(pbase + ((size_t)i * Wp + j) * g.ld_pool) = p;
                }
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
        // batch-norm statistics: this wavefront's 64 threads cover one 4 x 16-pixel strip (4 tiles) x 64 couts; the 4 tiles are the lane
        // bits 4-5, added in a fixed butterfly order, so the slab -- and everything derived from it -- is bitwise reproducible.
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) { st_s[q] += __shfl_xor(st_s[q], sh); st_q[q] += __shfl_xor(st_q[q], sh); }
        if (lane < 16) {
            const size_t strip = (((size_t)n * g.tiles_y + by) * g.tiles_x + bx) * 8 + wave;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c0o + q < g.Cout) { g.stats[(strip * 2) * g.Cout + c0o + q] = st_s[q]; g.stats[(strip * 2 + 1) * g.Cout + c0o + q] = st_q[q]; }
        }
    }
    W4_STAMP(7);
#ifdef KPX_WINO_STAMP
    if (dbgp) dbgp[9] = __builtin_amdgcn_s_memrealtime();
#endif
}

static std::atomic<unsigned long long> w43_attr_mask{0};
static inline int w43_lds_bytes() { return (W4_MAIN > W4_EPI ? W4_MAIN : W4_EPI) * 4; }

extern "C" int kpx_conv3x3_wino43_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr) {
    if (kpx_env()->no_wino || N <= 0) return 0;
    const bool shape = (H % 16 == 0 && W % 32 == 0) || (H == 16 && W == 16 && N % 2 == 0);      // 16 x 16 images are packed two to a workgroup
    return shape && K >= 16 && Nn >= 33 && ldin >= ((K + 7) & ~7) && ldin % 4 == 0 && (((uintptr_t)in_ptr) & 15) == 0 &&
           (size_t)H * W * ldin * 8 < 0x7fffffffu;
}
extern "C" size_t kpx_wino43_u_bytes(int Cin, int Cout) {
    const size_t a = (size_t)((Cin + 7) & ~7) * ((Cout + 63) & ~63), b = (size_t)((Cout + 7) & ~7) * ((Cin + 63) & ~63);
    return 36 * 4 * (a > b ? a : b);
}
extern "C" int kpx_wino43_filter_transform_f32(const float* w_hwio, int Cin, int Cout, int dgrad, float* U, void* stream) {
    if (!w_hwio || !U || Cin <= 0 || Cout <= 0) return KPX_EINVAL;
    const int K = dgrad ? Cout : Cin, Nn = dgrad ? Cin : Cout;
    const int Kp = (K + 7) & ~7, Np = (Nn + 63) & ~63;
    size_t nb = ((size_t)Kp * Np + 255) / 256; if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(wino43_filter_transform_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), w_hwio, Cin, Cout, dgrad, Kp, Np, U);
    return kpx_launch_status();
}
extern "C" int kpx_wino43_filter_transform_batch_f32(const void* descs_dev, int n, void* stream) {
    if (!descs_dev || n <= 0 || n > 65535) return KPX_EINVAL;
    hipLaunchKernelGGL(wino43_filter_transform_batch_kernel, dim3(64, (unsigned)n), dim3(256), 0, kpx_stream(stream), (const KpxWino43Desc*)descs_dev);
    return kpx_launch_status();
}
static int w43_launch(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                      float* out, int Nn, int ldout, int act, float* tile_stats, void* stream,
                      const float* mask_y = nullptr, int ld_mask = 0, float* pool_y = nullptr, int ld_pool = 0, const float* bn_beta = nullptr) {
    if (!in || !U || !out || ldin < K || ldout < Nn || act < 0 || act > 2 || !kpx_conv3x3_wino43_eligible(N, H, W, K, Nn, ldin, in)) return KPX_EINVAL;
    if (kpx_first_use_on_device(&w43_attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, w43_lds_bytes());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, w43_lds_bytes());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, w43_lds_bytes());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino43_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, w43_lds_bytes());
        if (e != hipSuccess) return -(int)e;
    }
    Wino43Geom g{};
    g.x = in; g.y = out; g.U = U; g.bias = bias;
    g.N = N; g.H = H; g.W = W; g.Cin = K; g.ldx = ldin; g.Cout = Nn; g.ldy = ldout; g.act = act;
    g.Kp = (K + 7) & ~7; g.Np = (Nn + 63) & ~63;
    const bool pack = W == 16;
    if (pack && tile_stats) return KPX_EINVAL;           // (no batch-norm layer of the path is 16 x 16 with a forward on this kernel)
    g.tiles_y = H / 16; g.tiles_x = pack ? 1 : W / 32;
    g.stats = tile_stats;
    g.mask_y = mask_y; g.ld_mask = ld_mask; g.pool_y = pool_y; g.ld_pool = ld_pool; g.bn_beta = bn_beta;
    const unsigned blocks = (unsigned)((size_t)(pack ? N / 2 : N) * g.tiles_y * g.tiles_x * (g.Np / 64));
    if (pack) hipLaunchKernelGGL((conv_wino43_kernel<0, true>), dim3(blocks), dim3(512), w43_lds_bytes(), kpx_stream(stream), g);
    else if (tile_stats && bn_beta) hipLaunchKernelGGL((conv_wino43_kernel<2, false>), dim3(blocks), dim3(512), w43_lds_bytes(), kpx_stream(stream), g);
    else if (tile_stats) hipLaunchKernelGGL((conv_wino43_kernel<1, false>), dim3(blocks), dim3(512), w43_lds_bytes(), kpx_stream(stream), g);
    else hipLaunchKernelGGL((conv_wino43_kernel<0, false>), dim3(blocks), dim3(512), w43_lds_bytes(), kpx_stream(stream), g);
    return kpx_launch_status();
}
extern "C" int kpx_conv3x3_wino43_f32(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                                      float* out, int Nn, int ldout, int act, void* stream) {
    return w43_launch(in, N, H, W, K, ldin, U, bias, out, Nn, ldout, act, nullptr, stream);
}
// 4 x 16-pixel strips per tensor: the unit of the statistics slab (kpx_bn_stats_from_tiles_f32 with tile_pixels = 64)
extern "C" size_t kpx_conv3x3_wino43_stats_tiles(int N, int H, int W) {
    return (H % 16 || W % 32 || N <= 0) ? 0 : (size_t)N * (H / 16) * (W / 32) * 8;          // (0: no statistics for packed 16 x 16 images)
}
extern "C" int kpx_conv3x3_wino43_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                                            float* out, int Nn, int ldout, int act, float* tile_stats, void* stream) {
    if (!tile_stats) return KPX_EINVAL;
    return w43_launch(in, N, H, W, K, ldin, U, bias, out, Nn, ldout, act, tile_stats, stream);
}

// The same with the two epilogue options VGG19 uses (reference models/networks/vgg.py:45-55): mask_y (or NULL) -- the output is zeroed
// where mask_y <= 0, i.e. the ReLU backward of the tensor this data gradient belongs to; pool_y (or NULL) -- the 2x2 max-pool of the
// activated output is written as well ([N,H/2,W/2,Nn], pixel stride ld_pool).  Needs Nn % 64 == 0 and 16-B aligned rows everywhere.
extern "C" int kpx_conv3x3_wino43_ex_f32(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                                         float* out, int Nn, int ldout, int act, const float* mask_y, int ld_mask, float* pool_y, int ld_pool, void* stream) {
    if (Nn % 64 || ldout % 4 || (((uintptr_t)out) & 15) || (mask_y && (ld_mask % 4 || ld_mask < Nn || (((uintptr_t)mask_y) & 15))) ||
        (pool_y && (ld_pool % 4 || ld_pool < Nn || (((uintptr_t)pool_y) & 15) || (H & 1) || (W & 1))))
        return KPX_EINVAL;
    return w43_launch(in, N, H, W, K, ldin, U, bias, out, Nn, ldout, act, nullptr, stream, mask_y, ld_mask, pool_y, ld_pool);
}

// Data gradient towards a ReLU'd batch norm's output (the tensor `bn_y` this convolution read in the forward pass): out = dgrad * [bn_y > 0]
// and tile_stats[strip][2][Nn] = per 4 x 16-pixel strip sum(out), sum(out * (bn_y - beta)) -- kpx_bn_train_bwd_f32 takes them instead of
// its own reduction pass (tf.gradients of tf.contrib.layers.batch_norm + tf.nn.relu, models/networks/layers.py:13-14 behind
// models/networks/__init__.py:10-24).  Nn a multiple of 64, H % 16 == 0, W % 32 == 0, 16-B aligned rows.
extern "C" int kpx_conv3x3_wino43_bnbwd_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* U,
                                                  float* out, int Nn, int ldout, const float* bn_y, int ld_bn_y, const float* bn_beta,
                                                  float* tile_stats, void* stream) {
    if (!tile_stats || !bn_y || !bn_beta || Nn % 64 || ldout % 4 || (((uintptr_t)out) & 15) || ld_bn_y % 4 || ld_bn_y < Nn || (((uintptr_t)bn_y) & 15) ||
        (((uintptr_t)bn_beta) & 15) || W == 16)
        return KPX_EINVAL;
    return w43_launch(in, N, H, W, K, ldin, U, nullptr, out, Nn, ldout, KPX_ACT_NONE, tile_stats, stream, bn_y, ld_bn_y, nullptr, 0, bn_beta);
}
