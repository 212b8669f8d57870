// Fused Winograd F(2x2, 3x3) convolution for gfx950 (fp32, v_mfma_f32_32x32x2_f32): 16 multiplies per 2x2 output tile instead
// of 36, i.e. 2.25x fewer MFMAs than the direct implicit GEMM for the 3x3 stride-1 SAME layers that dominate the path
// (translator, VGG19, encoders; reference models/networks/__init__.py:13,22,50..., models/networks/vgg.py:51).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// One workgroup (8 wavefronts) owns 8x8 tiles = 16x16 output pixels of one image x 32 or 64 output channels and ALL 16 Winograd
// points: the input and output transforms happen in LDS, so neither the transformed input (4x the input) nor the transformed
// output ever touches HBM.  Per 8-channel chunk the 18x18-pixel raw patch is staged in LDS, every thread computes one row of
// B^T d B for one tile into V[point][tile][channel] (16-B halves XOR-swizzled for conflict-free ds_read_b128), and the wavefronts
// multiply V with filter fragments they load straight from the pre-transformed, fragment-ordered U (see conv_wino_v2_kernel).
// The same kernel serves dgrad with filters transformed from the flipped / transposed weights.  The filter transform is its own
// entry point (kpx_wino_filter_transform[_batch]_f32): constant filters (VGG19) are transformed once, trainable ones once per
// optimiser update in ONE launch, instead of once per convolution call.
#include "kpx_common.h"
#include "kpx_env.h"
#include <stdlib.h>
#include <type_traits>
#include <atomic>

struct WinoGeom {
    const float* x; float* y; const float* U; const float* bias;
    int N, H, W, Cin, ldx, Cout, ldy, act;      // Cin / Cout: gathered / produced channels (real counts)
    int pack;                                   // 1: H = W = 8, one workgroup = 4 consecutive images as a 2x2 mosaic
    int Kp, Np;                                 // U is [16][Kp][Np]: Kp = Cin rounded up to 8, Np = Cout rounded up to 32 (zero padded)
    int tiles_y, tiles_x, nt;          // 16x16-pixel blocks per image, cout tiles of 32
    int stagger;                       // v2: wavefronts 4-7 run the MFMA half of a chunk first
    float* stats;                      // optional [N * tiles][2][Cout]: per-tile, per-channel sum and sum of squares of the OUTPUT (batch-norm statistics)
    const float* mask_y; int ld_mask;  // optional (data gradient feeding a ReLU'd batch norm): y of that batch norm at the output's pixels / channels ...
    const float* bn_beta;              // ... and its beta: stats become sum(dz), sum(dz * (y - beta)) with dz = out * [y > 0]
};

static __device__ __attribute__((aligned(16))) float wino_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// U[p][c][n] = sum_{r,q} G[i][r] g[r][q][c][n] G[j][q], p = 4*i + j.   dgrad: g'[r][q][c'][n'] = w[2-r][2-q][n'][c'].
// Fragment layout for the v2 kernel: Uf[p][kc = Kp/8][nb = Np/32][lh 2][li 32][j 4] = U[p][c = 8 kc + 4 lh + j][n = 32 nb + li], i.e. the
// B operand of v_mfma_f32_32x32x2_f32 for (point p, 8-channel chunk kc, 32-cout block nb) is ONE coalesced 16-B load per lane (1 KB per
// wavefront), element j feeding the MFMA of k-pair j.  The filter never passes through LDS.
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_filter_transform_frag_kernel(const float* __restrict__ w, int Cin, int Cout, int Kp, int Np, float* __restrict__ Uf) {
    const int K = DGRAD ? Cout : Cin, Nn = DGRAD ? Cin : Cout;
    const int KC = Kp >> 3, NB = Np >> 5;
    const size_t total = (size_t)Kp * Np;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        // idx enumerates the fragment order (kc, nb, lh, li, j) so that the 16 stores of a wavefront are 16 contiguous 256-B runs
        const int j = (int)(idx & 3), li = (int)((idx >> 2) & 31), lh = (int)((idx >> 7) & 1);
        const size_t blk = idx >> 8;
        const int nb = (int)(blk % NB), kc = (int)(blk / NB);
        const int c = 8 * kc + 4 * lh + j, n = 32 * nb + li;
        const bool real = c < K && n < Nn;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                g[r][q] = !real ? 0.f : (DGRAD ? w[((size_t)((2 - r) * 3 + (2 - q)) * Cin + n) * Cout + c] : w[((size_t)(r * 3 + q) * Cin + c) * Cout + n]);
        float t[4][3];                                   // G g
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            t[0][q] = g[0][q];
            t[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
            t[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
            t[3][q] = g[2][q];
        }
        const size_t pstride = (size_t)KC * NB * 256;
        float* o = Uf + blk * 256 + (idx & 255);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                    // (G g) G^T
            o[(size_t)(i * 4 + 0) * pstride] = t[i][0];
            o[(size_t)(i * 4 + 1) * pstride] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
            o[(size_t)(i * 4 + 2) * pstride] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
            o[(size_t)(i * 4 + 3) * pstride] = t[i][2];
        }
    }
}

// One ds_read_b128 of the transform serves lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (MI355X LDS), i.e. tiles
// {0,1,6,7} of one tile row + {2,3,4,5} of the next: with this pixel order and a row stride of 160 floats (two rows = 0 mod 64
// banks) those 16 lanes hit 16 distinct 16-B bank groups for every patch column c.
__device__ __forceinline__ int w8_pos(int p) { return p < 8 ? p : (p == 16 ? 8 : (p == 17 ? 17 : p + 1)); }
__device__ __forceinline__ int w8_pix(int pos) { return pos < 8 ? pos : (pos == 8 ? 16 : (pos == 17 ? 17 : pos - 1)); }

// ---- v2 forward / dgrad kernel ----------------------------------------------------------------------------------------------------
// Same tile (16x16 output pixels x 32*MODE output channels x all 16 Winograd points per workgroup of 8 wavefronts) and the same
// conflict-free raw / V images as above, restructured so that the matrix pipe is the only thing on the critical path:
//   * the transformed filters never pass through LDS: each lane loads its B fragments straight from the fragment-ordered Uf
//     (one 16-B load per point and chunk, L2-resident);
//   * raw and V are double buffered: while chunk k is multiplied, chunk k+1 is transformed and chunk k+2 is staged, by the same
//     wavefronts, as independent instruction streams inside one basic block -- ONE barrier per 8-channel chunk instead of three;
//   * wavefront w owns Winograd row a = w & 3 (points 4a .. 4a+3) of one 32-cout half (MODE 2) / one 32-tile half (MODE 1), so the
//     sum over the point columns of A^T M A happens in registers and the single-pass LDS epilogue moves half the accumulator bytes.
#define W2_RAW 3200
#define W2_V 8192
#define W2_MAIN (2 * W2_RAW + 2 * W2_V)

#ifdef KPX_WINO_STAMP      // diagnostic build only (profiles/wino_stamps.sh): s_memtime stamps of wavefronts 0 and 4 of the first 64 workgroups
__device__ unsigned long long* wino_dbg = nullptr;
extern "C" int kpx_debug_wino_stamps(unsigned long long* buf) { return -(int)hipMemcpyToSymbol(HIP_SYMBOL(wino_dbg), &buf, sizeof(buf)); }
#define W2_STAMP(slot) do { if (dbg) dbg[(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W2_STAMP(slot) do { } while (0)
#endif

// STATS 0: plain; 1: per-tile sum / sum of squares of the output (batch-norm statistics); 2: batch-norm backward sums (g.mask_y, g.bn_beta)
template <int MODE, int STATS>
__global__ __launch_bounds__(512, 2) void conv_wino_v2_kernel(const WinoGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = 32 * MODE, NTG = MODE;
    float* const rawb = smem;
    float* const Vb = smem + 2 * W2_RAW;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wa = wave & 3, hsel = wave >> 2;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int ntc = g.Np / NC;
    const int nti = L % ntc; L /= ntc;
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = g.pack ? 4 * L : L / g.tiles_y;       // packed: images n .. n+3
    const int oy0 = by * 16, ox0 = bx * 16, n0 = nti * NC;

    // raw patch units: u = t + 512*i < 720 (800 packed): position u>>1 (rows x 20 positions), 16-B half u&1
    const float* rp[2]; bool rok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = t + 512 * i, q = u >> 1, py = q / 20, ps = q - py * 20;
        int iy, ix, ni = n;
        bool ok;
        if (g.pack) {
            const int sy = py / 10, sx = ps / 10;
            iy = py - 10 * sy - 1; ix = ps - 10 * sx - 1; ni = n + 2 * sy + sx;
            ok = u < 800 && (unsigned)iy < 8u && (unsigned)ix < 8u;
        } else {
            iy = oy0 - 1 + py; ix = ox0 - 1 + w8_pix(ps);
            ok = u < 720 && ps < 18 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        }
        if (g.Cin <= 4 && (u & 1)) ok = false;
        rok[i] = ok;
        rp[i] = ok ? g.x + ((size_t)(ni * g.H + iy) * g.W + ix) * g.ldx + (u & 1) * 4 : wino_zero16;
    }
    // B fragments: points 4*wa + b, 32-cout block nb of this wavefront
    const int KC = g.Kp >> 3, NB = g.Np >> 5;
    const int nb = MODE == 2 ? nti * 2 + hsel : nti;
    const size_t ub_pstride = (size_t)KC * NB * 256, ub_step = (size_t)NB * 256;
    const float* ubp = g.U + ((size_t)(4 * wa) * KC * NB + nb) * 256 + lane * 4;
    // transform item: 16-B half, tile, row of V = B^T d B
    const int tslot = t & 1, ttile = (t >> 1) & 63, vrow = t >> 7;
    const int tty = ttile >> 3, ttx = ttile & 7;
    const int ra = vrow == 0 ? 0 : (vrow == 2 ? 2 : 1), rb = vrow == 0 ? 2 : (vrow == 1 ? 2 : (vrow == 2 ? 1 : 3));
    const float sgn = vrow == 1 ? 1.f : -1.f;
    int trd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        trd[c] = g.pack ? ((tty >> 2) * 10 + 2 * (tty & 3) + ra) * 160 + ((ttx >> 2) * 10 + 2 * (ttx & 3) + c) * 8 + tslot * 4
                        : (2 * tty + ra) * 160 + w8_pos(2 * ttx + c) * 8 + tslot * 4;
    const int trb = (rb - ra) * 160;
    const int vwr = (vrow * 4) * 512 + ttile * 8 + ((tslot ^ ((ttile >> 3) & 1)) << 2);
    int a_rd[NTG];
#pragma unroll
    for (int i = 0; i < NTG; ++i) {
        const int tile = (MODE == 2 ? i : hsel) * 32 + li;
        a_rd[i] = (4 * wa) * 512 + tile * 8 + ((lh ^ ((tile >> 3) & 1)) << 2);
    }

    f32x16 acc[4][NTG];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < NTG; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][i][r] = 0.f;

    const int nchunks = g.Kp / 8;
    const int ktail = g.Cin - (nchunks - 1) * 8 - tslot * 4;      // valid channels of this thread's 16-B half in the LAST chunk (<4: pad)
    f32x4 rr[2], ub[4];
    auto load_raw = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) { rr[i] = *reinterpret_cast<const f32x4*>(rp[i]); if (rok[i]) rp[i] += 8; }
    };
    // second staging unit of threads >= 288 does not exist (720 / 800 units): they re-store their first unit (same address, same
    // value) so that the chunk body stays ONE basic block the scheduler can interleave with the MFMAs
    const int st2 = t < 288 ? (t + 512) * 4 : t * 4;
    auto stage = [&](float* rawW) {
        *reinterpret_cast<f32x4*>(&rawW[t * 4]) = rr[0];
        *reinterpret_cast<f32x4*>(&rawW[st2]) = t < 288 ? rr[1] : rr[0];
    };
    auto transform = [&](const float* rawR, float* Vw, bool mask_tail) {
        f32x4 tr[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&rawR[trd[c]]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&rawR[trd[c] + trb]);
            tr[c] = a + sgn * b;
        }
        if (mask_tail) {                                 // channels >= Cin of a padded last chunk must not reach V
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j >= ktail) tr[c][j] = 0.f;
        }
        *reinterpret_cast<f32x4*>(&Vw[vwr]) = tr[0] - tr[2];
        *reinterpret_cast<f32x4*>(&Vw[vwr + 512]) = tr[1] + tr[2];
        *reinterpret_cast<f32x4*>(&Vw[vwr + 1024]) = tr[2] - tr[1];
        *reinterpret_cast<f32x4*>(&Vw[vwr + 1536]) = tr[1] - tr[3];
    };
    // the B fragment of point b is refreshed in place (next chunk) right after its last MFMA of this chunk: 16 registers, and a
    // full chunk of MFMA time to cover the L2 latency
    auto mfma = [&](const float* Vr, bool refill) {
        f32x4 av[2][NTG];                                  // A fragments of point b+1 are read under the MFMAs of point b
#pragma unroll
        for (int i = 0; i < NTG; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(&Vr[a_rd[i]]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b < 3) {
#pragma unroll
                for (int i = 0; i < NTG; ++i) av[(b + 1) & 1][i] = *reinterpret_cast<const f32x4*>(&Vr[a_rd[i] + (b + 1) * 512]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NTG; ++i)
                    acc[b][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][i][j], ub[b][j], acc[b][i], 0, 0, 0);
            if (refill) ub[b] = *reinterpret_cast<const f32x4*>(ubp + (size_t)b * ub_pstride);
        }
        ubp += ub_step;
    };
    const bool late = g.stagger && __builtin_amdgcn_readfirstlane(t) >= 256;      // wave-uniform by construction
#ifdef KPX_WINO_STAMP
    unsigned long long* dbg = (wino_dbg && lane == 0 && (wave & 3) == 0 && blockIdx.x < 64) ? wino_dbg + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 256 : nullptr;
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // one chunk: multiply chunk k while chunk k+1 is transformed and chunk k+2 staged.  FULL: k + 3 < nchunks (no conditions)
    auto body = [&](int k, auto full_tag, auto parity_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int PAR = decltype(parity_tag)::value;       // 0 / 1: k & 1 known at compile time (LDS offsets become immediates); 2: runtime
        const int cur = PAR == 2 ? (k & 1) : PAR;
        float* const rawC = rawb + cur * W2_RAW;
        float* const rawN = rawb + (cur ^ 1) * W2_RAW;
        float* const Vc = Vb + cur * W2_V;
        float* const Vn = Vb + (cur ^ 1) * W2_V;
        W2_STAMP(8 + 4 * k);
        __syncthreads();                                 // V[cur] (chunk k) and raw[cur^1] (chunk k+1) are complete
        W2_STAMP(9 + 4 * k);
        // The two wavefronts of a SIMD (w, w+4) run the halves of the chunk in opposite order: while one transforms / stages, the
        // other owns the matrix pipe, and the late transformer works under its partner's last MFMAs (MI355X: stagger by wave >= 4).
        if (late) mfma(Vc, FULL || k + 1 < nchunks);
        if (late) W2_STAMP(10 + 4 * k);
        if (FULL || k + 1 < nchunks) transform(rawN, Vn, !FULL && k + 1 == nchunks - 1 && ktail < 4);
        if (FULL || k + 2 < nchunks) stage(rawC);        // chunk k+2 (raw[cur] was consumed by the transform of iteration k-1)
        if (FULL || k + 3 < nchunks) load_raw();         // chunk k+3
        if (!late) W2_STAMP(10 + 4 * k);
        if (!late) mfma(Vc, FULL || k + 1 < nchunks);
        W2_STAMP(11 + 4 * k);
    };

    // prologue: chunk 0 staged + transformed, chunk 1 staged, chunk 2 in flight
    load_raw();
#pragma unroll
    for (int b = 0; b < 4; ++b) ub[b] = *reinterpret_cast<const f32x4*>(ubp + (size_t)b * ub_pstride);
    ubp += ub_step;
    stage(rawb);
    if (nchunks > 1) load_raw();
    __syncthreads();
    transform(rawb, Vb, nchunks == 1 && ktail < 4);
    if (nchunks > 1) stage(rawb + W2_RAW);
    if (nchunks > 2) load_raw();
    int k = 0;
    for (; k + 4 < nchunks; k += 2) {
        body(k, std::true_type{}, std::integral_constant<int, 0>{});
        body(k + 1, std::true_type{}, std::integral_constant<int, 1>{});
    }
    for (; k < nchunks; ++k) body(k, std::false_type{}, std::integral_constant<int, 2>{});

    // epilogue, one pass: the sum over the point columns b of A^T M A in registers (P[a][j] = sum_b M[a][b] A[b][j]),
    // P[4 a][2 j][64 tiles][NC couts] through LDS, then Y[i][j] = sum_a A[a][i] P[a][j] + bias + activation, 16-B stores.
#ifdef KPX_WINO_STAMP
    if (dbg) { dbg[2] = __builtin_amdgcn_s_memtime(); }
#endif
    __syncthreads();
    float* const P = smem;
#pragma unroll
    for (int i = 0; i < NTG; ++i) {
        const int tg = MODE == 2 ? i : hsel;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][i][r], m1 = acc[1][i][r], m2 = acc[2][i][r], m3 = acc[3][i][r];
            const int tile = tg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int col = (MODE == 2 ? hsel * 32 : 0) + li;
            P[((wa * 2 + 0) * 64 + tile) * NC + col] = m0 + m1 + m2;
            P[((wa * 2 + 1) * 64 + tile) * NC + col] = m1 - m2 - m3;
        }
    }
    // (data gradient feeding a ReLU'd batch norm) that batch norm's output at this thread's pixels: issued now, so the loads fly
    // under the barrier and the P reads instead of stalling the statistics code at the end of the epilogue
    const bool mask_vec = STATS == 2 && (g.ld_mask & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.mask_y) & 15) == 0);
    f32x4 ym[MODE][4];
    if (STATS == 2) {
#pragma unroll
        for (int it = 0; it < MODE; ++it) {
            const int idx = t + 512 * it;
            const int tile = idx / (NC / 4), cq = idx - tile * (NC / 4), c0 = n0 + cq * 4;
            const int oy = oy0 + 2 * (tile >> 3), ox = ox0 + 2 * (tile & 7);
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float* my = g.mask_y + ((size_t)(n * g.H + oy + (px >> 1)) * g.W + ox + (px & 1)) * g.ld_mask + c0;
                ym[it][px] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (mask_vec && c0 + 3 < g.Cout) ym[it][px] = *reinterpret_cast<const f32x4*>(my);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c0 + e < g.Cout) ym[it][px][e] = my[e];
                }
            }
        }
    }
    __syncthreads();
    const bool vec_ok = (g.ldy & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.y) & 15) == 0);
    f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};      // this thread's 4 couts: sum / sum of squares over its pixels
#pragma unroll
    for (int it = 0; it < MODE; ++it) {
        const int idx = t + 512 * it;
        const int tile = idx / (NC / 4), cq = idx - tile * (NC / 4);
        f32x4 pv[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j) pv[a][j] = *reinterpret_cast<const f32x4*>(&P[((a * 2 + j) * 64 + tile) * NC + cq * 4]);
        const int c0 = n0 + cq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c0 + e < g.Cout) bv[e] = g.bias[c0 + e];
        }
        f32x4 bvec = {0.f, 0.f, 0.f, 0.f};
        if (STATS == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c0 + e < g.Cout) bvec[e] = g.bn_beta[c0 + e];
        }
        const int ty = tile >> 3, tx = tile & 7;
        const int on = g.pack ? n + 2 * (ty >> 2) + (tx >> 2) : n;
        const int oy = g.pack ? 2 * (ty & 3) : oy0 + 2 * ty, ox = g.pack ? 2 * (tx & 3) : ox0 + 2 * tx;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                f32x4 v = (dy == 0 ? pv[0][dx] + pv[1][dx] + pv[2][dx] : pv[1][dx] - pv[2][dx] - pv[3][dx]) + bv;
                if (g.act == KPX_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (g.act == KPX_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
                }
                if (STATS == 2) {
                    const f32x4 yv = ym[it][dy * 2 + dx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dz = yv[e] > 0.f ? v[e] : 0.f;
                        st_s[e] += dz; st_q[e] += dz * (yv[e] - bvec[e]);
                    }
                } else if (STATS == 1) { st_s += v; st_q += v * v; }
                float* o = g.y + ((size_t)(on * g.H + oy + dy) * g.W + ox + dx) * g.ldy + c0;
                if (vec_ok && c0 + 3 < g.Cout) *reinterpret_cast<f32x4*>(o) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c0 + e < g.Cout) o[e] = v[e];
                }
            }
    }
    if (STATS != 0) {
        // batch-norm statistics of this 16x16-pixel tile: the 512 per-thread partials go through LDS once and 2*NC threads add them
        // in a fixed order (threads cq, cq + NC/4, ...), so the slab -- and everything derived from it -- is bitwise reproducible.
        __syncthreads();                                 // P has been consumed by every thread
        float* const S = smem;                           // [512][8]
        *reinterpret_cast<f32x4*>(&S[t * 8]) = st_s;
        *reinterpret_cast<f32x4*>(&S[t * 8 + 4]) = st_q;
        __syncthreads();
        if (t < 2 * NC) {
            const int stat = t / NC, c = t - stat * NC, cq = c >> 2, e = c & 3;
            float r = 0.f;
#pragma unroll 8
            for (int m = 0; m < 512 / (NC / 4); ++m) r += S[(cq + (NC / 4) * m) * 8 + stat * 4 + e];
            const size_t tile_id = ((size_t)n * g.tiles_y + by) * g.tiles_x + bx;
            if (n0 + c < g.Cout) g.stats[(tile_id * 2 + stat) * g.Cout + n0 + c] = r;
        }
    }
#ifdef KPX_WINO_STAMP
    if (dbg) { dbg[3] = __builtin_amdgcn_s_memtime(); dbg[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

static inline int w2_lds_bytes(int mode) { const int main = W2_MAIN * 4, epi = 4 * 2 * 64 * 32 * mode * 4; return main > epi ? main : epi; }

static std::atomic<unsigned long long> wino_attr_mask{0};

// shape / alignment eligibility (stride-1 3x3 SAME only); K = channels of the gathered tensor, Nn = produced channels
extern "C" int kpx_conv3x3_wino_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr) {
    if (kpx_env()->no_wino || N <= 0 || K <= 0 || Nn <= 0) return 0;       // (bench.py flips KPX_NO_WINO and calls kpx_reload_env() to time the direct kernel)
    // K is padded to a multiple of 8 (the pad channels must exist in the row: ldin >= Kp) and Nn to a multiple of 32
    const bool shape = (H % 16 == 0 && W % 16 == 0) || (H == 8 && W == 8 && N % 4 == 0);      // 8x8 images are packed four to a workgroup
    const int kmin = 4, nmin = 4;
    const bool kfit = ldin >= ((K + 7) & ~7) || K == 4;          // K = 4: only the lower 16-B half of the chunk is ever loaded
    return shape && K >= kmin && Nn >= nmin && (K >= 16 || K == 4 || K == 8) && kfit && (ldin % 4 == 0) && (((uintptr_t)in_ptr) & 15) == 0;
}
extern "C" __attribute__((visibility("hidden"))) int kpx_wino_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr) {
    return kpx_conv3x3_wino_eligible(N, H, W, K, Nn, ldin, in_ptr);
}

extern "C" size_t kpx_wino_u_bytes(int Cin, int Cout) {           // U[16][K padded to 8][Nn padded to 32], either direction
    const size_t a = (size_t)((Cin + 7) & ~7) * ((Cout + 31) & ~31), b = (size_t)((Cout + 7) & ~7) * ((Cin + 31) & ~31);
    return 16 * 4 * (a > b ? a : b);
}

extern "C" int kpx_wino_filter_transform_f32(const float* w_hwio, int Cin, int Cout, int dgrad, float* U, void* stream) {
    if (!w_hwio || !U || Cin <= 0 || Cout <= 0) return KPX_EINVAL;
    const int K = dgrad ? Cout : Cin, Nn = dgrad ? Cin : Cout;
    const int Kp = (K + 7) & ~7, Np = (Nn + 31) & ~31;
    const size_t pairs = (size_t)Kp * Np;
    size_t nb = (pairs + 255) / 256; if (nb > 1024) nb = 1024;
    hipStream_t s = kpx_stream(stream);
    if (dgrad) hipLaunchKernelGGL(wino_filter_transform_frag_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, w_hwio, Cin, Cout, Kp, Np, U);
    else hipLaunchKernelGGL(wino_filter_transform_frag_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, w_hwio, Cin, Cout, Kp, Np, U);
    return kpx_launch_status();
}

// Many filters in one launch: `descs` is a DEVICE array of n KpxWinoDesc (include/kpx.h); blockIdx.y = filter.
__global__ __launch_bounds__(256) void wino_filter_transform_batch_kernel(const KpxWinoDesc* __restrict__ descs) {
    const KpxWinoDesc d = descs[blockIdx.y];
    const int Cin = d.cin, Cout = d.cout;
    const bool dg = d.dgrad != 0;
    const int K = dg ? Cout : Cin, Nn = dg ? Cin : Cout;
    const int Kp = (K + 7) & ~7, Np = (Nn + 31) & ~31, KC = Kp >> 3, NB = Np >> 5;
    const size_t total = (size_t)Kp * Np, pstride = (size_t)KC * NB * 256;
    const float* __restrict__ w = d.w;
    float* __restrict__ Uf = d.u;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx & 3), li = (int)((idx >> 2) & 31), lh = (int)((idx >> 7) & 1);
        const size_t blk = idx >> 8;
        const int nb = (int)(blk % NB), kc = (int)(blk / NB);
        const int c = 8 * kc + 4 * lh + j, n = 32 * nb + li;
        const bool real = c < K && n < Nn;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                g[r][q] = !real ? 0.f : (dg ? w[((size_t)((2 - r) * 3 + (2 - q)) * Cin + n) * Cout + c] : w[((size_t)(r * 3 + q) * Cin + c) * Cout + n]);
        float t[4][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            t[0][q] = g[0][q];
            t[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
            t[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
            t[3][q] = g[2][q];
        }
        float* o = Uf + blk * 256 + (idx & 255);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[(size_t)(i * 4 + 0) * pstride] = t[i][0];
            o[(size_t)(i * 4 + 1) * pstride] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
            o[(size_t)(i * 4 + 2) * pstride] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
            o[(size_t)(i * 4 + 3) * pstride] = t[i][2];
        }
    }
}
extern "C" int kpx_wino_filter_transform_batch_f32(const void* descs_dev, int n, void* stream) {
    if (!descs_dev || n <= 0 || n > 65535) return KPX_EINVAL;
    hipLaunchKernelGGL(wino_filter_transform_batch_kernel, dim3(64, (unsigned)n), dim3(256), 0, kpx_stream(stream), (const KpxWinoDesc*)descs_dev);
    return kpx_launch_status();
}

// forward: in = x (K = Cin), out = y (Nn = Cout);  dgrad: in = dy (K = Cout), out = dx (Nn = Cin); U: fragment-ordered filters for (K, Nn)
static int wino_launch(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                       float* out, int Nn, int ldout, int act, float* tile_stats, void* stream,
                       const float* mask_y = nullptr, int ld_mask = 0, const float* bn_beta = nullptr);
extern "C" int kpx_conv3x3_wino_f32(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                                    float* out, int Nn, int ldout, int act, void* stream) {
    return wino_launch(in, N, H, W, K, ldin, U, bias, out, Nn, ldout, act, nullptr, stream);
}
extern "C" size_t kpx_conv3x3_wino_stats_tiles(int N, int H, int W) {
    return (H % 16 || W % 16 || N <= 0) ? 0 : (size_t)N * (H / 16) * (W / 16);
}
extern "C" int kpx_conv3x3_wino_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                                          float* out, int Nn, int ldout, int act, float* tile_stats, void* stream) {
    if (!tile_stats || H % 16 || W % 16) return KPX_EINVAL;       // (8x8 images are packed four to a workgroup: no per-image tiles)
    return wino_launch(in, N, H, W, K, ldin, U, bias, out, Nn, ldout, act, tile_stats, stream);
}
extern "C" int kpx_conv3x3_wino_bnbwd_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* U,
                                                float* out, int Nn, int ldout, const float* bn_y, int ld_bn_y, const float* bn_beta,
                                                float* tile_stats, void* stream) {
    if (!tile_stats || !bn_y || !bn_beta || ld_bn_y < Nn || H % 16 || W % 16) return KPX_EINVAL;
    return wino_launch(in, N, H, W, K, ldin, U, nullptr, out, Nn, ldout, KPX_ACT_NONE, tile_stats, stream, bn_y, ld_bn_y, bn_beta);
}
static int wino_launch(const float* in, int N, int H, int W, int K, int ldin, const float* U, const float* bias,
                       float* out, int Nn, int ldout, int act, float* tile_stats, void* stream,
                       const float* mask_y, int ld_mask, const float* bn_beta) {
    if (!in || !U || !out || ldin < (K == 4 ? 4 : K) || ldout < Nn || act < 0 || act > 2 || !kpx_conv3x3_wino_eligible(N, H, W, K, Nn, ldin, in)) return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    if (kpx_first_use_on_device(&wino_attr_mask)) {
        hipError_t e = hipSuccess;
#define W2_ATTR(M, S) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_v2_kernel<M, S>), hipFuncAttributeMaxDynamicSharedMemorySize, w2_lds_bytes(M))
        W2_ATTR(1, 0); W2_ATTR(2, 0); W2_ATTR(1, 1); W2_ATTR(2, 1); W2_ATTR(1, 2); W2_ATTR(2, 2);
#undef W2_ATTR
        if (e != hipSuccess) return -(int)e;
    }
    WinoGeom g{};
    g.x = in; g.y = out; g.U = U; g.bias = bias; g.stats = tile_stats;
    g.mask_y = mask_y; g.ld_mask = ld_mask; g.bn_beta = bn_beta;
    g.N = N; g.H = H; g.W = W; g.Cin = K; g.ldx = ldin; g.Cout = Nn; g.ldy = ldout; g.act = act;
    g.Kp = (K + 7) & ~7; g.Np = (Nn + 31) & ~31;
    g.pack = (H == 8 && W == 8) ? 1 : 0;
    g.tiles_y = g.pack ? 1 : H / 16; g.tiles_x = g.pack ? 1 : W / 16; g.nt = g.Np / 32;
    const unsigned blocks = (unsigned)((size_t)(g.pack ? N / 4 : N) * g.tiles_y * g.tiles_x * g.nt);
    // 64 output channels per workgroup when that still fills the 256 CUs, else 32
    const bool wide = blocks / 2 >= 256 && g.Np % 64 == 0;
    g.stagger = 1;                                       // wavefronts 4-7 run the MFMA half of a chunk first (worth 1-2 %, A/B in round 2)
    const int st = !tile_stats ? 0 : (mask_y ? 2 : 1);
#define W2_GO(M, S) hipLaunchKernelGGL((conv_wino_v2_kernel<M, S>), dim3(M == 2 ? blocks / 2 : blocks), dim3(512), w2_lds_bytes(M), s, g)
    if (wide) { if (st == 0) W2_GO(2, 0); else if (st == 1) W2_GO(2, 1); else W2_GO(2, 2); }
    else { if (st == 0) W2_GO(1, 0); else if (st == 1) W2_GO(1, 1); else W2_GO(1, 2); }
#undef W2_GO
    return kpx_launch_status();
}

// used by kpx_conv2d_fwd_f32 / kpx_conv2d_dgrad_f32 (conv_igemm.hip) when the caller did not pre-transform: transform into the
// workspace, then run
extern "C" __attribute__((visibility("hidden"))) int kpx_wino_conv3x3(const float* in, int N, int H, int W, int K, int ldin, const float* w_hwio, int Cin, int Cout, int dgrad,
                                const float* bias, int act, float* out, int Nn, int ldout, float* U_ws, hipStream_t s) {
    int rc = kpx_wino_filter_transform_f32(w_hwio, Cin, Cout, dgrad, U_ws, (void*)s);
    if (rc) return rc;
    return kpx_conv3x3_wino_f32(in, N, H, W, K, ldin, U_ws, bias, out, Nn, ldout, act, (void*)s);
}

// ------------------------------------------------------------------------------------------ Winograd weight gradient
// dU[p][c][n] = sum over 2x2-output tiles of V[p][tile][c] * dM[p][tile][n]   (V = B^T d B of the input patch, dM = A dY A^T of the
// 2x2 output-gradient tile), then dg = G^T dU G: 16 multiplies per tile instead of 36, like the forward.  GEMM view per point:
// M = Cin, N = Cout, K = tiles.  One workgroup (8 wavefronts) owns a 64 x 64 (c, n) block for all 16 points and a contiguous range
// of tile chunks (split-K over the batch); wave w holds points 2w, 2w+1 as 2 x (2 x 2) accumulators of 32x32.  A chunk is 4 x 2
// tiles = 8 x 4 output pixels: its 10 x 6 input patch and 8 x 4 dy pixels (64 channels each) are staged in LDS, transformed into
// V[16][8][64] and D[16][8][64], and multiplied with ds_read_b32 operands (K = tile index).  The epilogue applies G^T . G and
// writes a partial HWIO slab per split; the fixed-order wgrad_reduce_kernel sums the slabs (bitwise reproducible, no atomics).
struct WinoWgradGeom {
    const float* x; const float* dy; float* out;      // out: [S][9][Cin][Cout] partial slabs (or dw itself when S == 1)
    int N, H, W, Cin, ldx, Cout, lddy;                // Cin, Cout: real channel counts (Cout a multiple of 4); blocks may overhang them
    int cit, cot, S, cps, total_chunks, chy, chx;     // channel blocks, splits, chunks per split, chunks per image column / row
    size_t slab;
};

// CIT x COT = 32-channel MFMA tiles per workgroup block: (2,2) = 64 x 64, (2,1) = 64 x 32 (few output channels), (1,2) = 32 x 64
//
// Round 4: ONE barrier per chunk.  The first version staged the raw patch in LDS, transformed it, multiplied, with three barriers per
// chunk and one workgroup per CU: the matrix pipe idled during staging and transform (SQ_VALU_MFMA_BUSY 0.49).  Now every thread loads
// the eight x pixels (two patch rows x four columns) and four dy pixels of ITS transform item straight from global memory into registers
// (neighbouring items re-read shared pixels from L1 / L2: 12 instead of 3 16-B loads per thread and chunk, no raw stage, no LDS reads in
// the transform), V / D are double buffered (131 KB), and an iteration is: barrier, transform chunk k+1 from registers into the other
// buffer, issue the loads of chunk k+2, multiply chunk k -- wavefronts 4-7 multiply first and transform afterwards, so that one of the
// two wavefronts of a SIMD always has MFMAs to issue (the scheme of conv_wino_v2_kernel).
template <int CIT, int COT>
__global__ __launch_bounds__(512, 2) void conv_wino_wgrad_kernel(const WinoWgradGeom g) {
    constexpr int CI = 32 * CIT, CO = 32 * COT, SI = CI / 4, SO = CO / 4;     // channels and 16-B slots per pixel
    constexpr int VF = 16 * 8 * CI, DF = 16 * 8 * CO, BUF = VF + DF;
    extern __shared__ __attribute__((aligned(16))) float smem[];              // [2][V[16 points][8 tiles][CI] | D[16 points][8 tiles][CO]]

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int split = L % g.S; L /= g.S;
    const int cot = L % g.cot, cit = L / g.cot;
    const int c0 = cit * CI, n0 = cot * CO;

    // transform items: (16-B slot, tile 0..7, row 0..3 of the 4x4 transformed tile); the row is uniform per wavefront
    const int vsl = (t % SI) * 4, vtile = (t / SI) & 7, vrow = t / (8 * SI);         // x side
    const int esl = (t % SO) * 4, etile = (t / SO) & 7, erow = t / (8 * SO);         // dy side
    const bool vthr = t < 32 * SI, ethr = t < 32 * SO;
    const bool vact = vthr && c0 + vsl < g.Cin, eact = ethr && n0 + esl < g.Cout;
    const int xtail = g.Cin - (c0 + vsl);                                             // valid channels of this 16-B unit (>= 4: all)
    const int ra = vrow == 0 ? 0 : (vrow == 2 ? 2 : 1), rb = vrow == 0 ? 2 : (vrow == 1 ? 2 : (vrow == 2 ? 1 : 3));
    const float sgn = vrow == 1 ? 1.f : -1.f;
    const int pya = 2 * (vtile >> 2) + ra - 1, pyb = 2 * (vtile >> 2) + rb - 1, px0 = 2 * (vtile & 3) - 1;   // patch rows / first column, relative to the chunk origin
    const int vwr = (vrow * 4) * 8 * CI + vtile * CI + vsl;
    const float d0 = erow == 3 ? 0.f : 1.f, d1 = erow == 0 ? 0.f : (erow == 1 ? 1.f : -1.f);      // row i of A: t = d0*dY[0] + d1*dY[1]
    const int dwr = VF + (erow * 4) * 8 * CO + etile * CO + esl;
    // Operands through buffer descriptors over the whole tensors: voffset = this item's pixels relative to the chunk origin (fixed for the
    // life of the thread), soffset = the chunk origin (scalar).  An out-of-image pixel / out-of-range channel unit carries the offset
    // 0x80000000 and reads as zero.  The x descriptor starts one row + one pixel BEFORE the tensor so that the patch's halo offsets
    // (row -1, column -1) stay non-negative; those positions are only ever addressed with the out-of-range offset.
    const int dyo = ((2 * (etile >> 2) * g.W + 2 * (etile & 3)) * g.lddy + n0 + esl) * 4;   // bytes: this item's dy pixel (0,0) inside the chunk
    const int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x) - (size_t)(g.W + 1) * g.ldx, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dy), 0, 0x7fffffff, 0x00020000);
    int vxa[4], vxb[4], vd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        vxa[c] = vact ? (((pya + 1) * g.W + px0 + c + 1) * g.ldx + c0 + vsl) * 4 : OOB;
        vxb[c] = vact ? (((pyb + 1) * g.W + px0 + c + 1) * g.ldx + c0 + vsl) * 4 : OOB;
    }
    vd[0] = eact ? dyo : OOB; vd[1] = eact ? dyo + g.lddy * 4 : OOB; vd[2] = eact ? dyo + g.W * g.lddy * 4 : OOB; vd[3] = eact ? dyo + (g.W + 1) * g.lddy * 4 : OOB;
    const bool top_a = pya < 0, bot_a = pya > 3, top_b = pyb < 0, bot_b = pyb > 3, left0 = px0 < 0, right3 = px0 + 3 > 7;
    const int p0 = 2 * wave;

    f32x16 acc[2][CIT][COT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < CIT; ++b)
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

    const int ch_begin = split * g.cps;
    int ch_end = ch_begin + g.cps; if (ch_end > g.total_chunks) ch_end = g.total_chunks;
    const int nch = ch_end - ch_begin;
    // chunk cursor (scalar): image n, chunk row cy, chunk column cx of the NEXT chunk to load
    // (readfirstlane: integer division runs on the vector ALU, and a cursor left in vector registers would make every buffer load's scalar
    //  offset a 13-instruction readfirstlane loop)
    const int lq = ch_begin / g.chx;
    int lcx = __builtin_amdgcn_readfirstlane(ch_begin - lq * g.chx);
    int lcy = __builtin_amdgcn_readfirstlane(lq % g.chy), ln = __builtin_amdgcn_readfirstlane(lq / g.chy);
    f32x4 rxa[4], rxb[4], rd[4];
    auto load_next = [&]() {
        const int pix = (ln * g.H + lcy * 4) * g.W + lcx * 8;             // chunk origin (scalar); byte offsets < 2^31 (checked by the planner)
        const int sx = pix * g.ldx * 4, sd = pix * g.lddy * 4;
        const bool s_top = lcy == 0, s_bot = lcy + 1 == g.chy, s_left = lcx == 0, s_right = lcx + 1 == g.chx;
        // (a branch-free interior fast path -- plain offsets when no flag is set -- measured SLOWER: 0.368 vs 0.357 ms, A/B in one session)
        const bool inv_a = (top_a && s_top) || (bot_a && s_bot), inv_b = (top_b && s_top) || (bot_b && s_bot);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool inv_c = (c == 0 && left0 && s_left) || (c == 3 && right3 && s_right);
            rxa[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (inv_a || inv_c) ? OOB : vxa[c], sx, 0));
            rxb[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (inv_b || inv_c) ? OOB : vxb[c], sx, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) rd[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsd, vd[i], sd, 0));
        if (++lcx == g.chx) { lcx = 0; if (++lcy == g.chy) { lcy = 0; ++ln; } }
    };
    auto transform = [&](float* buf) {
        if (vthr) {
            f32x4 tr[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) tr[c] = rxa[c] + sgn * rxb[c];
            if (g.Cin & 3) {                 // a 16-B unit straddling Cin: the row's pad channels must not count
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 1; j < 4; ++j) if (j >= xtail) tr[c][j] = 0.f;
            }
            *reinterpret_cast<f32x4*>(&buf[vwr]) = tr[0] - tr[2];
            *reinterpret_cast<f32x4*>(&buf[vwr + 8 * CI]) = tr[1] + tr[2];
            *reinterpret_cast<f32x4*>(&buf[vwr + 16 * CI]) = tr[2] - tr[1];
            *reinterpret_cast<f32x4*>(&buf[vwr + 24 * CI]) = tr[1] - tr[3];
        }
        if (ethr) {
            const f32x4 t0 = d0 * rd[0] + d1 * rd[2], t1 = d0 * rd[1] + d1 * rd[3];
            *reinterpret_cast<f32x4*>(&buf[dwr]) = t0;
            *reinterpret_cast<f32x4*>(&buf[dwr + 8 * CO]) = t0 + t1;
            *reinterpret_cast<f32x4*>(&buf[dwr + 16 * CO]) = t0 - t1;
            *reinterpret_cast<f32x4*>(&buf[dwr + 24 * CO]) = -t1;
        }
    };
    auto mfma = [&](const float* buf) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const float* Vp = buf + (p0 + pt) * 8 * CI + lh * CI + li;
            const float* Dp = buf + VF + (p0 + pt) * 8 * CO + lh * CO + li;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float a[CIT], b[COT];
#pragma unroll
                for (int i = 0; i < CIT; ++i) a[i] = Vp[s * 2 * CI + i * 32];
#pragma unroll
                for (int j = 0; j < COT; ++j) b[j] = Dp[s * 2 * CO + j * 32];
#pragma unroll
                for (int i = 0; i < CIT; ++i)
#pragma unroll
                    for (int j = 0; j < COT; ++j)
                        acc[pt][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[pt][i][j], 0, 0, 0);
            }
        }
    };
    const bool late = wave >= 4;                              // wave-uniform
    if (nch > 0) {
        load_next();
        transform(smem);
        if (nch > 1) load_next();
    }
    for (int k = 0; k < nch; ++k) {
        float* const cur = smem + (k & 1) * BUF;
        float* const nxt = smem + ((k & 1) ^ 1) * BUF;
        __syncthreads();                                     // buffer `cur` (chunk k) is complete; everybody is done reading `nxt` (chunk k-1)
        if (late) mfma(cur);
        if (k + 1 < nch) transform(nxt);                     // the registers hold chunk k+1
        if (k + 2 < nch) load_next();                        // chunk k+2: in flight under this iteration's MFMAs
        if (!late) mfma(cur);
    }
    __syncthreads();                                         // the epilogue's M overlays the buffers

    // dg = G^T dU G per (c, n), one 32 x 32 quarter at a time through LDS: M[16][32 c][32 n]
    float* Ms = smem;
    float* out = g.out + (size_t)split * g.slab;
    const int oc = t & 31;
#pragma unroll
    for (int q = 0; q < CIT * COT; ++q) {
        const int ci = q / COT, co = q % COT;
        if (q) __syncthreads();
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ms[((p0 + pt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[pt][ci][co][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cl = (t >> 5) + 16 * i;
            float m[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = Ms[(p * 32 + cl) * 32 + oc];
            // rows of G^T: [1,.5,.5,0], [0,.5,-.5,0], [0,.5,.5,1]
            float h[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[0][j] = m[j] + 0.5f * (m[4 + j] + m[8 + j]);
                h[1][j] = 0.5f * (m[4 + j] - m[8 + j]);
                h[2][j] = 0.5f * (m[4 + j] + m[8 + j]) + m[12 + j];
            }
            const int c = c0 + ci * 32 + cl, n = n0 + co * 32 + oc;
            if (c < g.Cin && n < g.Cout) {
                float* o = out + (size_t)c * g.Cout + n;
                const size_t tap = (size_t)g.Cin * g.Cout;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    o[(r * 3 + 0) * tap] = h[r][0] + 0.5f * (h[r][1] + h[r][2]);
                    o[(r * 3 + 1) * tap] = 0.5f * (h[r][1] - h[r][2]);
                    o[(r * 3 + 2) * tap] = 0.5f * (h[r][1] + h[r][2]) + h[r][3];
                }
            }
        }
    }
}

// block shape for a channel count: 64 when the count is a multiple of 64 or large, else 32
static inline int ww_tile(int C) { return (C % 64 == 0 || C > 96) ? 2 : 1; }

// splits for the Winograd wgrad (0 = shape not handled): ~256 workgroups (one per CU; measured best of 128..1024), >= 8 chunks per split, slabs <= 128 MB
extern "C" __attribute__((visibility("hidden"))) int kpx_wino_wgrad_splits(int N, int H, int W, int Cin, int Cout) {
    if (kpx_env()->no_wino) return 0;
    const int comin = 4;
    if (H % 4 || W % 8 || Cout % 4 || Cin < 32 || Cout < comin) return 0;         // (a Cin that is not a multiple of 4 needs ldx >= Cin rounded up, checked by the caller)
    const int ti = ww_tile(Cin), to = ww_tile(Cout);
    if (ti == 1 && to == 1) return 0;                     // 32 x 32 blocks: too few MFMAs per barrier, the direct kernels do better
    const long tc = (long)N * (H / 4) * (W / 8), tiles = (long)((Cin + 32 * ti - 1) / (32 * ti)) * ((Cout + 32 * to - 1) / (32 * to));
    const long target = 256;
    long S = target / tiles;                              // floor: tiles * S workgroups must fit ONE round of the chip (158 -> 256 channels: 12 tiles,
    if (S < 1) S = 1;                                     // 22 splits = 264 workgroups ran as two rounds, 0.309 ms; 21 splits = 252: one round)
    if (S > tc / 8) S = tc / 8;
    const long cap = (128L << 20) / ((long)9 * Cin * Cout * 4);
    if (S > cap) S = cap;
    if (S < 1) S = 1;
    if (tc < 64 || tiles * S < 128) return 0;             // too little work to fill the chip: the direct kernels do better
    return (int)S;
}

extern "C" __attribute__((visibility("hidden"))) int kpx_wino_wgrad3x3(const float* x, int N, int H, int W, int Cin, int ldx, const float* dy, int Cout, int lddy,
                                                                   float* slabs, int S, hipStream_t s) {
    static std::atomic<unsigned long long> attr_mask{0};
    // V + D, double buffered (the epilogue's M[16][32][32], 64 KB, overlays them)
    const int lds22 = 2 * (16 * 8 * 64 + 16 * 8 * 64) * 4, lds21 = 2 * (16 * 8 * 64 + 16 * 8 * 32) * 4, lds12 = 2 * (16 * 8 * 32 + 16 * 8 * 64) * 4;
    if (kpx_first_use_on_device(&attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_wgrad_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds22);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_wgrad_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds21 > 65536 ? lds21 : 65536);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_wgrad_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds12 > 65536 ? lds12 : 65536);
        if (e != hipSuccess) return -(int)e;
    }
    const int ti = ww_tile(Cin), to = ww_tile(Cout);
    WinoWgradGeom g{};
    g.x = x; g.dy = dy; g.out = slabs;
    g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.ldx = ldx; g.Cout = Cout; g.lddy = lddy;
    g.cit = (Cin + 32 * ti - 1) / (32 * ti); g.cot = (Cout + 32 * to - 1) / (32 * to); g.S = S;
    g.chy = H / 4; g.chx = W / 8; g.total_chunks = N * g.chy * g.chx;
    g.cps = (g.total_chunks + S - 1) / S;
    g.slab = (size_t)9 * Cin * Cout;
    const dim3 grid((unsigned)(g.cit * g.cot * S));
    // the epilogue's M[16][32][32] (64 KB) overlays the main-loop buffers: at least 64 KB
    if (ti == 2 && to == 2) hipLaunchKernelGGL((conv_wino_wgrad_kernel<2, 2>), grid, dim3(512), lds22, s, g);
    else if (ti == 2) hipLaunchKernelGGL((conv_wino_wgrad_kernel<2, 1>), grid, dim3(512), lds21 > 65536 ? lds21 : 65536, s, g);
    else hipLaunchKernelGGL((conv_wino_wgrad_kernel<1, 2>), grid, dim3(512), lds12 > 65536 ? lds12 : 65536, s, g);
    return kpx_launch_status();
}
