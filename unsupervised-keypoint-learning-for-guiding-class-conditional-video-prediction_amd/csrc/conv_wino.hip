// Fused Winograd F(2x2, 3x3) convolution for gfx950 (fp32, v_mfma_f32_32x32x2_f32): 16 multiplies per 2x2 output tile instead
// of 36, i.e. 2.25x fewer MFMAs than the direct implicit GEMM for the 3x3 stride-1 SAME layers that dominate the path
// (translator, VGG19, encoders; reference models/networks/__init__.py:13,22,50..., models/networks/vgg.py:51).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// One workgroup (8 wavefronts) owns 8x8 tiles = 16x16 output pixels of one image x 32 output channels and ALL 16 Winograd
// points: the input and output transforms happen in LDS, so neither the transformed input (4x the input) nor the transformed
// output ever touches HBM.  Per 16-channel chunk: the 18x18-pixel raw patch and the 16x16x32 slice of the pre-transformed
// filters U are staged in LDS, every thread transforms half a tile (B^T d B) into V[point][tile][channel] (16-B slots
// XOR-swizzled for conflict-free ds_read_b128), then wave w multiplies points 2w and 2w+1 (4 accumulators of 32x32).
// After the K loop the accumulators go through LDS once more for A^T M A + bias + activation.
// The same kernel serves dgrad with filters transformed from the flipped / transposed weights.
#include "kpx_common.h"
#include <stdlib.h>

struct WinoGeom {
    const float* x; float* y; const float* U; const float* bias;
    int N, H, W, Cin, ldx, Cout, ldy, act;
    int tiles_y, tiles_x, nt;          // 16x16-pixel blocks per image, cout tiles of 32
};

static __device__ __attribute__((aligned(16))) float wino_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// U[p][c][n] = sum_{r,q} G[i][r] g[r][q][c][n] G[j][q], p = 4*i + j.   dgrad: g'[r][q][c'][n'] = w[2-r][2-q][n'][c'].
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_filter_transform_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ U) {
    // forward: K = Cin, Nn = Cout;  dgrad: K = Cout (channels of dy), Nn = Cin
    const int K = DGRAD ? Cout : Cin, Nn = DGRAD ? Cin : Cout;
    const size_t total = (size_t)K * Nn;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx / Nn), n = (int)(idx - (size_t)c * Nn);
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                g[r][q] = DGRAD ? w[((size_t)((2 - r) * 3 + (2 - q)) * Cin + n) * Cout + c] : w[((size_t)(r * 3 + q) * Cin + c) * Cout + n];
        float t[4][3];                                   // G g
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            t[0][q] = g[0][q];
            t[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
            t[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
            t[3][q] = g[2][q];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                    // (G g) G^T
            const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
            U[((size_t)(i * 4 + 0) * K + c) * Nn + n] = u0;
            U[((size_t)(i * 4 + 1) * K + c) * Nn + n] = u1;
            U[((size_t)(i * 4 + 2) * K + c) * Nn + n] = u2;
            U[((size_t)(i * 4 + 3) * K + c) * Nn + n] = u3;
        }
    }
}

#define WINO_RAW 5184      // 324 pixels x 16 channels
#define WINO_U 8192        // 16 points x 16 channels x 32 couts
#define WINO_V 16384       // 16 points x 64 tiles x 16 channels (re-used as M[16][32 tiles][32 couts] in the epilogue)
#define WINO_LDS_BYTES ((WINO_RAW + WINO_U + WINO_V) * 4)

__global__ __launch_bounds__(512) void conv_wino_kernel(const WinoGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* raw = smem;                     // [324][16]
    float* Us = smem + WINO_RAW;           // [16][16][32]
    float* Vs = smem + WINO_RAW + WINO_U;  // [16][64][16] swizzled ; epilogue: [16][32][32]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int nti = L % g.nt; L /= g.nt;
    const int bx = L % g.tiles_x; L /= g.tiles_x;
    const int by = L % g.tiles_y;
    const int n = L / g.tiles_y;
    const int oy0 = by * 16, ox0 = bx * 16, n0 = nti * 32;

    // ---- fixed per-thread global-load units
    const float* rp[3];                    // raw patch: unit u = t + 512*i < 1296 : pixel u>>2 (18x18), 16-B slot u&3
    bool rok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = t + 512 * i, px = u >> 2, py = px / 18, pxx = px - py * 18;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
        rok[i] = u < 1296 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        rp[i] = g.x + ((size_t)(n * g.H + (rok[i] ? iy : 0)) * g.W + (rok[i] ? ix : 0)) * g.ldx + (u & 3) * 4;
    }
    const float* up[4];                    // U slice: unit u = t + 512*i < 2048 : point u>>7, channel (u>>3)&15, cout slot u&7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = t + 512 * i;
        up[i] = g.U + ((size_t)(u >> 7) * g.Cin + ((u >> 3) & 15)) * g.Cout + n0 + (u & 7) * 4;
    }
    // ---- fixed transform item: tile (8x8 grid), 16-B channel slot, half (rows 0-1 / 2-3 of B^T d B)
    const int tslot = t & 3, thalf = (t >> 2) & 1, ttile = t >> 3;
    const int tty = ttile >> 3, ttx = ttile & 7;
    const int traw = ((2 * tty + thalf) * 18 + 2 * ttx) * 16 + tslot * 4;       // first patch row this item reads
    // ---- fixed MFMA read addresses
    int a_rd[2][2];                        // [tile group][u]
#pragma unroll
    for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile = tg * 32 + li;
            a_rd[tg][u] = tile * 16 + ((((2 * u + lh)) ^ ((tile >> 2) & 3)) << 2);
        }
    const int p0 = 2 * wave;

    f32x16 acc[2][2];                      // [point][tile group]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 rr[3], ru[4];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) rr[i] = *reinterpret_cast<const f32x4*>(rok[i] ? rp[i] + c0 : wino_zero16);
#pragma unroll
        for (int i = 0; i < 4; ++i) ru[i] = *reinterpret_cast<const f32x4*>(up[i] + (size_t)c0 * g.Cout);
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) if (t + 512 * i < 1296) *reinterpret_cast<f32x4*>(&raw[(t + 512 * i) * 4]) = rr[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&Us[(t + 512 * i) * 4]) = ru[i];
    };

    const int nchunks = g.Cin / 16;
    load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        store_chunk();
        __syncthreads();
        // ---- input transform: this thread produces rows (2*thalf, 2*thalf+1) of V = B^T d B for its tile and 4 channels
        {
            f32x4 d[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = *reinterpret_cast<const f32x4*>(&raw[traw + (r * 18 + c) * 16]);
            f32x4 tr0[4], tr1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (thalf == 0) { tr0[c] = d[0][c] - d[2][c]; tr1[c] = d[1][c] + d[2][c]; }      // rows 0,1 from patch rows 0,1,2
                else            { tr0[c] = d[1][c] - d[0][c]; tr1[c] = d[0][c] - d[2][c]; }      // rows 2,3 from patch rows 1,2,3
            }
            const int vbase = ttile * 16 + ((tslot ^ ((ttile >> 2) & 3)) << 2);
            const int prow = thalf * 2;
#pragma unroll
            for (int rrw = 0; rrw < 2; ++rrw) {
                const f32x4* s = rrw ? tr1 : tr0;
                const int p = (prow + rrw) * 4;
                *reinterpret_cast<f32x4*>(&Vs[(p + 0) * 1024 + vbase]) = s[0] - s[2];
                *reinterpret_cast<f32x4*>(&Vs[(p + 1) * 1024 + vbase]) = s[1] + s[2];
                *reinterpret_cast<f32x4*>(&Vs[(p + 2) * 1024 + vbase]) = s[2] - s[1];
                *reinterpret_cast<f32x4*>(&Vs[(p + 3) * 1024 + vbase]) = s[1] - s[3];
            }
        }
        if (ch + 1 < nchunks) load_chunk((ch + 1) * 16);      // in flight during the MFMA phase
        __syncthreads();
        // ---- 16 batched GEMMs: wave w owns points 2w, 2w+1
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const float* Vp = Vs + (p0 + pt) * 1024;
            const float* Up = Us + (p0 + pt) * 512;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(&Vp[a_rd[0][u]]);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(&Vp[a_rd[1][u]]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float b = Up[(4 * (2 * u + lh) + j) * 32 + li];
                    acc[pt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b, acc[pt][0], 0, 0, 0);
                    acc[pt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b, acc[pt][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- output transform, one tile group (32 tiles = 4 tile rows) at a time through LDS: M[16][32][32]
    const int oc = t & 31;
    const float bv = g.bias ? g.bias[n0 + oc] : 0.f;
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        if (tg) __syncthreads();
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Vs[((p0 + pt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[pt][tg][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int tl = (t >> 5) + 16 * i;              // local tile 0..31 of this group
            float m[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = Vs[(p * 32 + tl) * 32 + oc];
            float s0[4], s1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { s0[c] = m[c] + m[4 + c] + m[8 + c]; s1[c] = m[4 + c] - m[8 + c] - m[12 + c]; }
            float yv[2][2];
            yv[0][0] = s0[0] + s0[1] + s0[2]; yv[0][1] = s0[1] - s0[2] - s0[3];
            yv[1][0] = s1[0] + s1[1] + s1[2]; yv[1][1] = s1[1] - s1[2] - s1[3];
            const int tile = tg * 32 + tl, ty = tile >> 3, tx = tile & 7;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float v = yv[dy][dx] + bv;
                    if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                    g.y[((size_t)(n * g.H + oy0 + 2 * ty + dy) * g.W + ox0 + 2 * tx + dx) * g.ldy + n0 + oc] = v;
                }
        }
    }
}

static bool wino_attr_set = false;

// shape / alignment eligibility (stride-1 3x3 SAME only); K = channels of the gathered tensor, Nn = produced channels
extern "C" int kpx_wino_eligible(int H, int W, int K, int Nn, int ldin, const void* in_ptr) {
    if (getenv("KPX_NO_WINO")) return 0;
    return (H % 16 == 0) && (W % 16 == 0) && (K % 16 == 0) && (Nn % 32 == 0) && K >= 32 && (ldin % 4 == 0) && (((uintptr_t)in_ptr) & 15) == 0;
}

// forward: in = x (K = Cin), out = y (Nn = Cout);  dgrad: in = dy (K = Cout), out = dx (Nn = Cin), w always HWIO [3][3][Cin][Cout]
extern "C" int kpx_wino_conv3x3(const float* in, int N, int H, int W, int K, int ldin, const float* w_hwio, int Cin, int Cout, int dgrad,
                                const float* bias, int act, float* out, int Nn, int ldout, float* U_ws, hipStream_t s) {
    const size_t pairs = (size_t)K * Nn;
    size_t nb = (pairs + 255) / 256; if (nb > 1024) nb = 1024;
    if (dgrad) hipLaunchKernelGGL(wino_filter_transform_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, w_hwio, Cin, Cout, U_ws);
    else hipLaunchKernelGGL(wino_filter_transform_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, w_hwio, Cin, Cout, U_ws);
    int rc = kpx_launch_status();
    if (rc) return rc;
    if (!wino_attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES);
        if (e != hipSuccess) return -(int)e;
        wino_attr_set = true;
    }
    WinoGeom g{};
    g.x = in; g.y = out; g.U = U_ws; g.bias = bias;
    g.N = N; g.H = H; g.W = W; g.Cin = K; g.ldx = ldin; g.Cout = Nn; g.ldy = ldout; g.act = act;
    g.tiles_y = H / 16; g.tiles_x = W / 16; g.nt = Nn / 32;
    const unsigned blocks = (unsigned)((size_t)N * g.tiles_y * g.tiles_x * g.nt);
    hipLaunchKernelGGL(conv_wino_kernel, dim3(blocks), dim3(512), WINO_LDS_BYTES, s, g);
    return kpx_launch_status();
}
