// fp32 implicit-GEMM convolution for gfx950 (MI355X): forward, dgrad and wgrad on
// v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 157 TF peak).
//
// Replaces tf.layers.conv2d / tf.nn.conv2d of the reference (models/networks/layers.py:6-9,
// models/networks/vgg.py:51) and their TF-generated gradients.
//
// One "gather conv" kernel serves forward and dgrad:
//   out[n, a*osy+oy0, b*osx+ox0, j] = act( bias[j] + sum_{tr,tq,c} in[n, a*isy+tr*ity+iy0, b*isx+tq*itx+ix0, c] * B[tap(tr,tq)][c][j] )
// forward : in = x, B = HWIO weights as stored ([c][j] row-major),   a,b = output pixel, input step = stride
// dgrad   : in = dy, B = the same HWIO weights read transposed ([j][c]), one launch per stride-parity class of
//           dx pixels so that no MFMA is spent on the zeros of a dilated gradient.
// GEMM view: M = N*Ha*Wa pixels, N = output channels, K = taps*Cin; the A tile (pixels x 16 channels) is gathered
// straight from NHWC (channels contiguous -> 16-B loads), staged in LDS [m][16+4] and read back as ds_read_b128;
// the B tile is staged [k][n] and read as ds_read_b32.  Within a 16-wide K chunk lane-half h consumes channels
// {8u+4h+j}: a K permutation shared by A and B, so the product is unchanged.
#include "kpx_common.h"
#include "kpx_env.h"
#include <stdlib.h>
#include <stdio.h>

#include "conv_geom.h"

// Tile geometry: BM x BN outputs per workgroup, WM x WN wavefronts each owning (BM/WM) x (BN/WN) as 32x32 MFMA
// accumulators, K chunks of 32 channels.  LDS (double buffered): A [BM][32] with the 16-B slot index XOR-swizzled by
// ((row>>1)&7) -- conflict-free for both the 8-lane ds_write_b128 groups (one row = 128 contiguous bytes = one full cache
// line of the pixel's channels) and the 16-lane ds_read_b128 groups; B [32][BN], for dgrad (weights read transposed) the
// column index is XORed with ((k>>2)&7)<<2 so the scalar transposing writes spread over all banks.
// 16 zero bytes in the code object: invalid (out-of-image / out-of-range) 16-B units are loaded from here, so the
// TAIL=false kernels need no per-element select between the load and the LDS write.
__device__ __attribute__((aligned(16))) float kpx_zero16[4] = {0.f, 0.f, 0.f, 0.f};          // (each translation unit that needs it has its own)

template <int V> struct kpx_ic { static constexpr int value = V; };

template <int BM, int BN, int WM, int WN, bool BT, bool VEC, bool MERGE, bool TAIL>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(const ConvGeom g) {
    constexpr int NT = WM * WN * 64;
    constexpr int BK = 32;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int RA = (BM * 8) / NT;              // A float4 units per thread (BM rows x 8 slots)
    constexpr int RB = (BN * 8 + NT - 1) / NT;     // B float4 units per thread (32 x BN/4, or BN rows x 8 slots when BT)
    constexpr int BU = BN * 8;
    constexpr int ASZ = BM * BK, BSZ = BK * BN;    // floats per LDS stage
    static_assert(TM >= 1 && TN >= 1 && RA >= 1 && (BM * 8) % NT == 0, "tile/wave shape");
    __shared__ __attribute__((aligned(16))) float As[2 * ASZ];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const ConvClass k = g.cls[blockIdx.y];
    const int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    if (L >= k.mt * g.nt) return;                  // this class has fewer tiles than the widest one
    const int m0 = (L / g.nt) * BM, n0 = (L % g.nt) * BN;
    const int HW = k.Ha * k.Wa;
    const int wrow = wm * TM * 32, wcol = wn * TN * 32;

    // ---- everything that does not change over the K loop is computed once per thread: pointers of the A / B units at
    // tap (0,0) / channel 0, their validity as bit masks over the taps, and all LDS addresses.  Per chunk only wave-uniform
    // (scalar) offsets are added, so the loop carries ~2 VALU per MFMA instead of ~6.
    const int kq = t & 7;
    const float* a_ptr[RA];
    unsigned a_vr[RA], a_vq[RA];
    int a_iw0[RA], a_st[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int row = (t >> 3) + (NT / 8) * i;
        const int m = m0 + row;
        const bool ok = m < k.M;
        const int mm = ok ? m : 0;
        const int n = mm / HW, rem = mm - n * HW, a = rem / k.Wa, b = rem - a * k.Wa;
        const int ih0 = a * g.isy + k.iy0, iw0 = b * g.isx + k.ix0;
        a_ptr[i] = g.x + (ptrdiff_t)(n * g.Hi * g.Wi + ih0 * g.Wi + iw0) * g.ldx + kq * 4;
        unsigned vr = 0, vq = 0;
        for (int r = 0; r < k.Tr; ++r) if (ok && (unsigned)(ih0 + r * g.ity) < (unsigned)g.Hi) vr |= 1u << r;
        for (int q = 0; q < k.Tq; ++q) if (MERGE || (unsigned)(iw0 + q * g.itx) < (unsigned)g.Wi) vq |= 1u << q;
        a_vr[i] = vr; a_vq[i] = vq; a_iw0[i] = iw0;
        a_st[i] = row * BK + ((kq ^ ((row >> 1) & 7)) << 2);
    }
    const float* b_ptr[RB];
    bool b_ok[RB];
    int b_k[RB], b_n[RB], b_st[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int u = t + NT * i;
        if (!BT) {
            const int kr = u / (BN / 4), n4 = u % (BN / 4), n = n0 + n4 * 4;
            b_k[i] = kr; b_n[i] = n; b_ok[i] = u < BU && n < g.Cout;
            b_ptr[i] = g.w + (size_t)kr * g.ldw + n;
            b_st[i] = kr * BN + n4 * 4;
        } else {
            const int nn = u >> 3, ks = u & 7, n = n0 + nn;
            b_k[i] = ks * 4; b_n[i] = n; b_ok[i] = u < BU && n < g.Cout;
            b_ptr[i] = g.w + (size_t)n * g.ldw + ks * 4;
            b_st[i] = (ks * 4) * BN + (nn ^ (ks << 2));
        }
    }
    int a_rd[TM][4], b_rd[TN][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ks = 2 * u + lh;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wrow + i * 32 + li;
            a_rd[i][u] = row * BK + ((ks ^ ((row >> 1) & 7)) << 2);
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = wcol + n * 32 + li;
            b_rd[n][u] = (ks * 4) * BN + (BT ? (col ^ (ks << 2)) : col);
        }
    }

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
    if (g.ksplit > 1) {                                 // this workgroup's slice of the K loop
        const int cps = (nchunks + g.ksplit - 1) / g.ksplit;
        const int cb = blockIdx.z * cps, ce = min(nchunks, cb + cps);
        const int tap0 = cb / nck;
        c0 = (cb - tap0 * nck) * BK; tr = tap0 / k.Tq; tq = tap0 - tr * k.Tq;
        nchunks = ce > cb ? ce - cb : 0;
    }
    f32x4 ra[RA], rb[RB];
    unsigned am[RA], bm[RB];

    auto tail4 = [](int c, int lim) -> unsigned {      // bit j set iff c + j < lim
        const int r = lim - c;
        return r >= 4 ? 15u : (r <= 0 ? 0u : (15u >> (4 - r)));
    };
    auto load4 = [&](const float* p, unsigned mask, const float* safe) -> f32x4 {
        f32x4 v;
        if (VEC) {
            v = *reinterpret_cast<const f32x4*>(mask ? p : safe);
        } else {
            v[0] = *((mask & 1u) ? p : safe);
            v[1] = *((mask & 2u) ? p + 1 : safe);
            v[2] = *((mask & 4u) ? p + 2 : safe);
            v[3] = *((mask & 8u) ? p + 3 : safe);
        }
        return v;
    };
    // Loads go out unconditionally: an invalid unit reads the 16 zero bytes of kpx_zero16 (or, with TAIL, a clamped
    // address that is masked when the registers are written to LDS), so nothing in the MFMA phase depends on them.
    auto load_chunk = [&]() {
        const int a_off = ((tr * g.ity) * g.Wi + tq * g.itx) * g.ldx + c0;          // wave-uniform
        const int tap = (k.wr0 + tr * g.wrs) * g.KW + (k.wq0 + tq * g.wqs);
        const size_t b_off = (size_t)tap * g.wts + (BT ? (size_t)c0 : (size_t)c0 * g.ldw);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int c = c0 + kq * 4;
            const bool v = (((a_vr[i] >> tr) & (a_vq[i] >> tq)) & 1u) != 0 && c < g.Cin;
            const float* p = a_ptr[i] + a_off;
            unsigned mask = v ? 15u : 0u;
            if (TAIL) {
                if (MERGE) {
                    mask = 0;
                    if (v) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (c + j < g.Cin && (unsigned)(a_iw0[i] + (c + j) / g.merge) < (unsigned)g.Wi) mask |= 1u << j;
                    }
                } else if (v) {
                    mask = tail4(c, g.Cin);
                }
            }
            am[i] = mask;
            ra[i] = load4(p, mask, TAIL ? g.x : kpx_zero16);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool v = b_ok[i] && c0 + b_k[i] < g.Cin;
            unsigned mask = v ? 15u : 0u;
            if (TAIL && v) mask = BT ? tail4(c0 + b_k[i], g.Cin) : tail4(b_n[i], g.Cout);
            bm[i] = mask;
            rb[i] = load4(b_ptr[i] + b_off, mask, TAIL ? g.w : kpx_zero16);
        }
        c0 += BK;
        if (c0 >= g.Cin) {
            c0 = 0;
            if (++tq == k.Tq) { tq = 0; ++tr; }
        }
    };
    auto masked = [](f32x4 v, unsigned m) -> f32x4 {
        if (!TAIL) return v;
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (m >> j) & 1u ? v[j] : 0.f;
        return r;
    };
    auto store_chunk = [&](int buf) {                  // buf is a compile-time constant at every call site
#pragma unroll
        for (int i = 0; i < RA; ++i)
            *reinterpret_cast<f32x4*>(&As[buf * ASZ + a_st[i]]) = masked(ra[i], am[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (t + NT * i < BU) {
                const f32x4 v = masked(rb[i], bm[i]);
                if (!BT) {
                    *reinterpret_cast<f32x4*>(&Bs[buf * BSZ + b_st[i]]) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) Bs[buf * BSZ + b_st[i] + j * BN] = v[j];
                }
            }
        }
    };
    // one group = 8 k-steps: lane-half lh consumes channels 4*(2u+lh) .. +3 of the chunk (a K permutation shared by A and B).
    // The fragments of group u+1 are read from LDS before the MFMAs of group u are issued (register double buffering), so
    // only the first group of a chunk exposes the LDS latency.
    struct Frag { f32x4 a[TM]; float b[4][TN]; };
    auto read_frag = [&](int buf, int u, Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const f32x4*>(&As[buf * ASZ + a_rd[i][u]]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < TN; ++n) f.b[j][n] = Bs[buf * BSZ + b_rd[n][u] + j * BN];
    };
    auto mfma_frag = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][j], f.b[j][n], acc[i][n], 0, 0, 0);
    };
    // The next chunk's global loads are issued after the first group and written to the other LDS buffer after the third,
    // so their address arithmetic, the wait and the ds_writes sit between MFMAs instead of in the gap around the barrier.
    auto chunk = [&](auto bufc, bool more) {
        constexpr int buf = decltype(bufc)::value;
        Frag f0, f1;
        read_frag(buf, 0, f0);
        read_frag(buf, 1, f1);
        mfma_frag(f0);
        if (more) load_chunk();
        read_frag(buf, 2, f0);
        mfma_frag(f1);
        read_frag(buf, 3, f1);
        mfma_frag(f0);
        if (more) store_chunk(buf ^ 1);
        mfma_frag(f1);
        __syncthreads();
    };

    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    int ch = 0;
    for (; ch + 1 < nchunks; ch += 2) {                // unrolled by two: the LDS buffer index is an immediate
        chunk(kpx_ic<0>{}, true);
        chunk(kpx_ic<1>{}, ch + 2 < nchunks);
    }
    if (ch < nchunks) chunk(kpx_ic<0>{}, false);

    // epilogue: the output pixel of every tile row goes through LDS (the A buffers are free now)
    int* rowpix = reinterpret_cast<int*>(&As[0]);
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
    if (g.ksplit > 1) {                                 // raw partial sums; bias / activation happen in splitk_reduce_kernel
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
                const int row = wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int pix = rowpix[row];
                if (cok && pix >= 0) {
                    float v = acc[i][n][r] + bv;
                    if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                    else if (g.act == KPX_ACT_TANH) v = tanhf(v);
                    if (g.mul_y) v *= kpx_act_grad_from_y(g.mul_y[(size_t)pix * g.ld_mul + col], g.mul_act);
                    g.y[(size_t)pix * g.ldy + col] = v;
                }
            }
        }
    }
}

// y[pix][c] = act( bias[c] + sum_s ws[s][pix][c] ), fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, size_t slab, int S, size_t npix, int C,
                                                            const float* __restrict__ bias, int act, float* __restrict__ y, int ldy,
                                                            const float* __restrict__ mul_y, int ld_mul, int mul_act, int io16 = 0) {
    const int C4 = C >> 2;                              // C % 4 == 0 is a precondition of the split-K path
    const size_t total = npix * (size_t)C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / C4;
        const int c = (int)(i - pix * C4) * 4;
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + pix * C + c);
        for (int s2 = 1; s2 < S; ++s2) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(ws + (size_t)s2 * slab + pix * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += b[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = a[j] + (bias ? bias[c + j] : 0.f);
            if (act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
            else if (act == KPX_ACT_TANH) v = tanhf(v);
            a[j] = v;
        }
        if (mul_y) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float my = (io16 & 4) ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(mul_y)[pix * ld_mul + c + j] << 16) : mul_y[pix * ld_mul + c + j];
                a[j] *= kpx_act_grad_from_y(my, mul_act);
            }
        }
        if (io16 & 2) {                                  // bf16 output (the bf16 configuration): four channels = 8 bytes
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 o = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
            *reinterpret_cast<bf16x4*>(reinterpret_cast<unsigned short*>(y) + pix * ldy + c) = o;
        } else *reinterpret_cast<f32x4*>(y + pix * ldy + c) = a;
    }
}

// Split-K plan (shape only): used when the 128x128 tiling would leave most CUs idle (small M, long K).
static int conv_splitk_plan(long M, int Cout, long nchunks_total) {
    if (Cout % 4 != 0 || Cout < 64) return 1;
    const long tiles = ((M + 127) / 128) * ((Cout + 127) / 128);
    const long max_tiles = 256;                          // a full round of 128x128 tiles: splitting only adds the reduce pass
    if (tiles >= max_tiles) return 1;
    long S = 512 / tiles;
    if (S > 8) S = 8;
    while (S > 1 && nchunks_total / S < 6) --S;         // keep >= 6 chunks (192 channels-taps) per split
    if ((double)S * (double)M * Cout * 4.0 > (double)((size_t)256 << 20)) return 1;
    return (int)(S < 1 ? 1 : S);
}

template <bool BT, bool VEC, bool MERGE, bool TAIL>
static int launch_gather_conv_v(ConvGeom g, hipStream_t s) {
    if (g.ncls <= 0) { g.ncls = 1; g.cls[0] = ConvClass{g.Ha, g.Wa, g.oy0, g.ox0, g.Tr, g.Tq, g.iy0, g.ix0, g.wr0, g.wq0, 0, 0}; }
    g.M = 0;
    for (int i = 0; i < g.ncls; ++i) { g.cls[i].M = g.N * g.cls[i].Ha * g.cls[i].Wa; if (g.cls[i].M > g.M) g.M = g.cls[i].M; }
    if (g.M <= 0 || g.Cout <= 0) return 0;
    if (g.ksplit > 1) {                                 // decided by the entry point (needs the caller's workspace)
        const int BMs = 128, BNs = g.Cout > 64 ? 128 : 64;
        g.nt = (g.Cout + BNs - 1) / BNs;
        int mtmax = 0;
        for (int i = 0; i < g.ncls; ++i) { g.cls[i].mt = (g.cls[i].M + BMs - 1) / BMs; if (g.cls[i].mt > mtmax) mtmax = g.cls[i].mt; }
        g.mt = mtmax;
        const dim3 nblk((unsigned)(g.mt * g.nt), (unsigned)g.ncls, (unsigned)g.ksplit);
        if (BNs == 128) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 4, BT, VEC, MERGE, TAIL>), nblk, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 4, 2, BT, VEC, MERGE, TAIL>), nblk, dim3(512), 0, s, g);
        int rc = kpx_launch_status();
        if (rc) return rc;
        const size_t npix = (size_t)g.N * g.Ho * g.Wo;
        size_t nb = (npix * (g.Cout / 4) + 255) / 256; if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float*)g.ws, g.ws_slab, g.ksplit, npix, g.Cout,
                           g.bias, g.act, g.y, g.ldy, g.mul_y, g.ld_mul, g.mul_act);
        return kpx_launch_status();
    }
    // N tile: smallest padded width, ties -> wider tile
    int BN = 128;
    {
        auto pad = [&](int b) { return (g.Cout + b - 1) / b * b; };
        if (pad(64) < pad(BN)) BN = 64;
        if (pad(32) < pad(BN)) BN = 32;
    }
    auto blocks = [&](int bm, int bn) {
        long b = 0;
        for (int i = 0; i < g.ncls; ++i) b += (long)((g.cls[i].M + bm - 1) / bm) * ((g.Cout + bn - 1) / bn);
        return b;
    };
    // pick the tile by how well its workgroup count fills 256 CUs x resident workgroups per CU (LDS-limited)
    int BM = 128;
    if (BN != 32) {
        struct Cand { int bm, bn, occ; double eff; };
        const Cand c128[] = {{128, 128, 2, 1.0}, {64, 128, 3, 0.92}, {128, 64, 3, 0.92}, {64, 64, 5, 0.85}};
        const Cand c64[] = {{128, 64, 3, 1.0}, {64, 64, 5, 0.92}};
        const Cand* cs = BN == 128 ? c128 : c64;
        const int nc = BN == 128 ? 4 : 2;
        double best = -1;
        for (int i = 0; i < nc; ++i) {
            const long nb = blocks(cs[i].bm, cs[i].bn), slots = 256L * cs[i].occ;
            const long rounds = (nb + slots - 1) / slots;
            const double score = cs[i].eff * (double)nb / (double)(rounds * slots);
            if (score > best) { best = score; BM = cs[i].bm; BN = cs[i].bn; }
        }
    }
    g.nt = (g.Cout + BN - 1) / BN;
    int mtmax = 0;
    for (int i = 0; i < g.ncls; ++i) { g.cls[i].mt = (g.cls[i].M + BM - 1) / BM; if (g.cls[i].mt > mtmax) mtmax = g.cls[i].mt; }
    g.mt = mtmax;
    const dim3 nblk((unsigned)(g.mt * g.nt), (unsigned)g.ncls);
#define KPX_LAUNCH(bm, bn, wm, wn) \
    hipLaunchKernelGGL((conv_igemm_kernel<bm, bn, wm, wn, BT, VEC, MERGE, TAIL>), nblk, dim3((wm) * (wn) * 64), 0, s, g)
    if (BM == 128 && BN == 128) KPX_LAUNCH(128, 128, 2, 4);        // 8 waves of 64x32
    else if (BM == 64 && BN == 128) KPX_LAUNCH(64, 128, 2, 4);     // 8 waves of 32x32 (small M, wide N)
    else if (BM == 128 && BN == 64) KPX_LAUNCH(128, 64, 4, 2);     // 8 waves of 32x32
    else if (BM == 64 && BN == 64) KPX_LAUNCH(64, 64, 2, 2);       // 4 waves of 32x32
    else KPX_LAUNCH(128, 32, 4, 1);                                // 4 waves of 32x32
#undef KPX_LAUNCH
    return kpx_launch_status();
}

// conv_gemm3.hip: the same gather convolution on the bf16 matrix pipe with fp32 operands split into three bf16 terms (fp32-equivalent)
extern "C" __attribute__((visibility("hidden"))) int kpx_gemm3_eligible(const ConvGeom* g);
extern "C" __attribute__((visibility("hidden"))) int kpx_gemm3_launch(ConvGeom g, int bt, int terms, hipStream_t s);
extern "C" __attribute__((visibility("hidden"))) int kpx_wgrad3_eligible(const WgradGeom* g);
extern "C" __attribute__((visibility("hidden"))) int kpx_wgrad3_launch(WgradGeom g, int bm, int terms, hipStream_t s);

template <bool BT>
static int launch_gather_conv(const ConvGeom& g, hipStream_t s) {
    if (kpx_gemm3_eligible(&g)) {
        int rc = kpx_gemm3_launch(g, BT ? 1 : 0, g.terms == 1 ? 1 : 3, s);
        if (rc || g.ksplit <= 1) return rc;
        const size_t npix = (size_t)g.N * g.Ho * g.Wo;
        size_t nb = (npix * (g.Cout / 4) + 255) / 256; if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float*)g.ws, g.ws_slab, g.ksplit, npix, g.Cout,
                           g.bias, g.act, g.y, g.ldy, g.mul_y, g.ld_mul, g.mul_act, g.io16);
        return kpx_launch_status();
    }
    if (g.io16) return KPX_EINVAL;                       // bf16 tensors: only the kernel above reads them
    if (g.merge) return launch_gather_conv_v<false, false, true, true>(g, s);
    if (g.vecA && g.vecB) {
        if (g.Cin % 4 == 0 && g.Cout % 4 == 0) return launch_gather_conv_v<BT, true, false, false>(g, s);
        return launch_gather_conv_v<BT, true, false, true>(g, s);
    }
    return launch_gather_conv_v<BT, false, false, true>(g, s);
}

// Tiny-Cout / long-K forward (img_discr D_logit: 3x3x2048 -> 1, reference networks/__init__.py:150): one wavefront
// per output pixel, lanes stride over the contiguous channel axis, shuffle-tree reduction.  An MFMA tile would idle
// 31/32 of its columns and serialise 18432-deep K loops on a handful of workgroups.
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const ConvGeom g) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= g.M) return;
    const int HW = g.Ha * g.Wa;
    const int n = m / HW, rem = m - n * HW, a = rem / g.Wa, b = rem - a * g.Wa;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tr = 0; tr < g.Tr; ++tr) {
        const int ih = a * g.isy + g.iy0 + tr * g.ity;
        if ((unsigned)ih >= (unsigned)g.Hi) continue;
        for (int tq = 0; tq < g.Tq; ++tq) {
            const int iw = b * g.isx + g.ix0 + tq * g.itx;
            if ((unsigned)iw >= (unsigned)g.Wi) continue;
            const float* xp = g.x + ((size_t)(n * g.Hi + ih) * g.Wi + iw) * g.ldx;
            const float* wp = g.w + (size_t)((g.wr0 + tr * g.wrs) * g.KW + g.wq0 + tq * g.wqs) * g.wts;
            for (int c = lane * 4; c < g.Cin; c += 256) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + c);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    for (int o = 0; o < g.Cout; ++o) acc[o] = fmaf(xv[j], wp[(size_t)(c + j) * g.ldw + o], acc[o]);
            }
        }
    }
    const int pix = (n * g.Ho + a * g.osy + g.oy0) * g.Wo + b * g.osx + g.ox0;
    for (int o = 0; o < g.Cout; ++o) {
        float v = kpx_wave_sum(acc[o]);
        if (lane == 0) {
            v += g.bias ? g.bias[o] : 0.f;
            if (g.act == KPX_ACT_RELU) v = fmaxf(v, 0.f);
            else if (g.act == KPX_ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
            else if (g.act == KPX_ACT_TANH) v = tanhf(v);
            g.y[(size_t)pix * g.ldy + o] = v;
        }
    }
}
static int launch_small_cout(ConvGeom g, hipStream_t s) {
    g.M = g.N * g.Ha * g.Wa;
    hipLaunchKernelGGL(conv_small_cout_kernel, dim3((unsigned)((g.M + 3) / 4)), dim3(256), 0, s, g);
    return kpx_launch_status();
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int dgrad_splitk_plan(int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride) {
    const long Mcls = (long)N * ((Hi + stride - 1) / stride) * ((Wi + stride - 1) / stride);      // largest parity class
    const long chunks = (long)(KH / stride > 0 ? KH / stride : 1) * (KW / stride > 0 ? KW / stride : 1) * ((Cout + 31) / 32);   // smallest class
    if (Cin % 4 != 0) return 1;
    int S = conv_splitk_plan(Mcls * stride * stride, Cin, chunks);   // tiles are counted over all classes of the launch
    return S;
}

// conv_wino.hip: fused Winograd F(2x2,3x3) for the stride-1 3x3 SAME layers
extern "C" __attribute__((visibility("hidden"))) int kpx_wino_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr);
extern "C" __attribute__((visibility("hidden"))) int kpx_wino_conv3x3(const float* in, int N, int H, int W, int K, int ldin, const float* w_hwio, int Cin, int Cout, int dgrad,
                                const float* bias, int act, float* out, int Nn, int ldout, float* U_ws, hipStream_t s);
// conv_rgb.hip: LDS-resident first-layer kernel (Cin <= 4, stride 1)
extern "C" __attribute__((visibility("hidden"))) int kpx_conv_rgb_fwd(const float* x, int N, int Hi, int Wi, int Cin, const float* w, int KH, int KW,
                                                                  const float* bias, float* y, int Ho, int Wo, int Cout, int ldy,
                                                                  int stride, int pad_t, int pad_l, int act, hipStream_t s, int y16 = 0);
extern "C" __attribute__((visibility("hidden"))) int kpx_conv_rgb_dgrad(const float* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                                                                    float* dx, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l, hipStream_t s, int dy16 = 0);
extern "C" int kpx_conv3x3_c16_eligible(int N, int H, int W, int K, int Nn, int ldin, int ldout, const void* in_ptr);
extern "C" int kpx_conv3x3_c16_f32(const float* in, int N, int H, int W, int K, int ldin, const float* w_hwio, int dgrad, const float* bias,
                                   float* out, int ldout, int act, float* tile_stats, void* stream);
extern "C" __attribute__((visibility("hidden"))) int kpx_conv_few_fwd(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const float* w, int KH, int KW,
                                                                  const float* bias, float* y, int Ho, int Wo, int Cout, int ldy,
                                                                  int stride, int pad_t, int pad_l, int act, hipStream_t s);
extern "C" __attribute__((visibility("hidden"))) int kpx_wino_wgrad_splits(int N, int H, int W, int Cin, int Cout);
extern "C" __attribute__((visibility("hidden"))) int kpx_wsmall_splits(int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride);
extern "C" __attribute__((visibility("hidden"))) int kpx_wsmall_launch(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const float* dy, int Ho, int Wo, int Cout, int lddy,
                                                                   int KH, int KW, int stride, int pad_t, int pad_l, float* slabs, int S, hipStream_t s, int dy16 = 0);
extern "C" __attribute__((visibility("hidden"))) int kpx_wino_wgrad3x3(const float* x, int N, int H, int W, int Cin, int ldx, const float* dy, int Cout, int lddy,
                                                                   float* slabs, int S, hipStream_t s);
static inline size_t wino_ws_bytes(int Cin, int Cout) {          // U[16][K padded to 8][Nn padded to 32] for either direction
    const size_t a = (size_t)((Cin + 7) & ~7) * ((Cout + 31) & ~31), b = (size_t)((Cout + 7) & ~7) * ((Cin + 31) & ~31);
    return 16 * 4 * (a > b ? a : b);
}

extern "C" size_t kpx_conv2d_fwd_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
    if (Cin % 4 != 0) return 0;
    const int S = conv_splitk_plan((long)N * Ho * Wo, Cout, (long)KH * KW * ((Cin + 31) / 32));
    size_t b = S > 1 ? (size_t)S * N * Ho * Wo * Cout * 4 : 0;
    if (KH == 3 && KW == 3 && b < wino_ws_bytes(Cin, Cout)) b = wino_ws_bytes(Cin, Cout);     // stride unknown here: upper bound
    return b;
}

extern "C" size_t kpx_conv2d_dgrad_workspace_bytes(int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride) {
    const int S = dgrad_splitk_plan(N, Hi, Wi, Cin, Cout, KH, KW, stride);
    size_t b = S > 1 ? (size_t)S * N * Hi * Wi * Cin * 4 : 0;
    if (KH == 3 && KW == 3 && stride == 1 && b < wino_ws_bytes(Cin, Cout)) b = wino_ws_bytes(Cin, Cout);
    return b;
}

static int fwd_impl(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const float* w, int KH, int KW, const float* bias,
                    float* y, int Ho, int Wo, int Cout, int ldy, int stride, int pad_t, int pad_l, int act, int arith, int io16,
                    void* workspace, size_t workspace_bytes, void* stream);
extern "C" int kpx_conv2d_fwd_f32(const float* x, int N, int Hi, int Wi, int Cin, int ldx,
                                  const float* w, int KH, int KW, const float* bias,
                                  float* y, int Ho, int Wo, int Cout, int ldy,
                                  int stride, int pad_t, int pad_l, int act, int arith, void* workspace, size_t workspace_bytes, void* stream) {
    return fwd_impl(x, N, Hi, Wi, Cin, ldx, w, KH, KW, bias, y, Ho, Wo, Cout, ldy, stride, pad_t, pad_l, act, arith, 0, workspace, workspace_bytes, stream);
}
// bf16 configuration: x bf16 (pixel stride ldx elements, a multiple of 8), y bf16 (y_f32 = 0) or fp32; only shapes the bf16-pipe gather kernel
// takes (KPX_EINVAL otherwise: the caller converts and uses the fp32 entry)
extern "C" int kpx_conv2d_fwd_bf16(const void* x, int N, int Hi, int Wi, int Cin, int ldx, const float* w, int KH, int KW, const float* bias,
                                   void* y, int y_f32, int Ho, int Wo, int Cout, int ldy, int stride, int pad_t, int pad_l, int act,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (ldx % 8 || !aligned16(x) || !aligned16(y) || (!y_f32 && ldy % 4)) return KPX_EINVAL;
    return fwd_impl((const float*)x, N, Hi, Wi, Cin, ldx, w, KH, KW, bias, (float*)y, Ho, Wo, Cout, ldy, stride, pad_t, pad_l, act, KPX_ARITH_BF16, 1 | (y_f32 ? 0 : 2),
                    workspace, workspace_bytes, stream);
}
static int fwd_impl(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const float* w, int KH, int KW, const float* bias,
                    float* y, int Ho, int Wo, int Cout, int ldy, int stride, int pad_t, int pad_l, int act, int arith, int io16,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !w || !y || N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 ||
        KH <= 0 || KW <= 0 || stride <= 0 || ldx < Cin || ldy < Cout || act < 0 || act > 3 || arith < 0 || arith > 1)
        return KPX_EINVAL;
    if (io16) goto gather;                               // bf16 tensors: the specialised fp32 kernels below do not read them
    if (KH == 3 && KW == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && Ho == Hi && Wo == Wi && act != KPX_ACT_TANH &&
        kpx_conv3x3_c16_eligible(N, Hi, Wi, Cin, Cout, ldx, ldy, x))                 // exactly 16 produced channels: 16x16x4 MFMA blocks (conv_c16.hip)
        return kpx_conv3x3_c16_f32(x, N, Hi, Wi, Cin, ldx, w, 0, bias, y, ldy, act, nullptr, stream);
    if (Cout <= 4 && Cin <= 128 && (size_t)N * Ho * Wo >= 65536) {            // few produced channels over a large image: VALU kernel (conv_rgb.hip)
        const int rc = kpx_conv_few_fwd(x, N, Hi, Wi, Cin, ldx, w, KH, KW, bias, y, Ho, Wo, Cout, ldy, stride, pad_t, pad_l, act, kpx_stream(stream));
        if (rc != -2) return rc;
    }
    if (KH == 3 && KW == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && Ho == Hi && Wo == Wi && act != KPX_ACT_TANH && aligned16(w) &&
        workspace && workspace_bytes >= wino_ws_bytes(Cin, Cout) && kpx_wino_eligible(N, Hi, Wi, Cin, Cout, ldx, x))
        return kpx_wino_conv3x3(x, N, Hi, Wi, Cin, ldx, w, Cin, Cout, 0, bias, act, y, Cout, ldy, (float*)workspace, kpx_stream(stream));
    if (Cin <= 4 && ldx == Cin && stride <= 2 && Cout <= 64 && Ho * Wo >= 256) {       // image-input layers (stride 2: img_discr conv_0): patch + filter resident in LDS
        const int rc = kpx_conv_rgb_fwd(x, N, Hi, Wi, Cin, w, KH, KW, bias, y, Ho, Wo, Cout, ldy, stride, pad_t, pad_l, act, kpx_stream(stream));
        if (rc != -2) return rc;
    }
gather:
    ConvGeom g{};
    g.x = x; g.y = y; g.w = w; g.bias = bias; g.io16 = io16;
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.ldx = ldx;
    g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.ldy = ldy;
    g.Ha = Ho; g.Wa = Wo; g.osy = 1; g.oy0 = 0; g.osx = 1; g.ox0 = 0;
    g.isy = stride; g.iy0 = -pad_t; g.isx = stride; g.ix0 = -pad_l;
    g.Tr = KH; g.Tq = KW; g.ity = 1; g.itx = 1;
    g.wr0 = 0; g.wrs = 1; g.wq0 = 0; g.wqs = 1; g.KW = KW;
    g.wts = Cin * Cout; g.ldw = Cout; g.act = act; g.terms = arith == KPX_ARITH_BF16 ? 1 : 3;
    g.vecA = (ldx % 4 == 0) && aligned16(x);
    g.vecB = (Cout % 4 == 0) && aligned16(w);
    if (Cin % 4 != 0 && ldx == Cin && KW > 1 && KW * Cin <= 64) {      // image inputs (Cin = 3): merge each filter row
        g.merge = Cin; g.Tq = 1; g.KW = 1; g.Cin = KW * Cin; g.wts = KW * Cin * Cout; g.vecA = 0;
    }
    if (io16 && !kpx_gemm3_eligible(&g)) return KPX_EINVAL;
    if (!io16 && Cout <= 4 && Cin % 4 == 0 && Cin >= 256 && g.vecA && g.vecB == (Cout == 4))
        return launch_small_cout(g, kpx_stream(stream));
    if (g.vecA && g.vecB && !g.merge && Cin % 4 == 0) {        // small-M / long-K layers (the discriminator's 10x10 .. 4x4 maps): split K over workgroups
        const int S = conv_splitk_plan((long)N * Ho * Wo, Cout, (long)KH * KW * ((Cin + 31) / 32));
        g.ws_slab = (size_t)N * Ho * Wo * Cout;
        if (S > 1 && workspace && workspace_bytes >= (size_t)S * g.ws_slab * 4) { g.ksplit = S; g.ws = (float*)workspace; }
    }
    return launch_gather_conv<false>(g, kpx_stream(stream));
}

extern "C" int kpx_act_bwd_f32(const float* dy, const float* y, float* dz, size_t n, int act, void* stream);

// dx = dgrad(dy) [* act_in'(y_in)]: the optional factor is the activation backward of the tensor dx is the gradient of (y_in = its ACTIVATED
// value, same shape as dx).  Fused into the epilogue where the layer runs on the gather kernels; a second pass over dx otherwise.
static int dgrad_impl(const float* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                      float* dx, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l, int arith,
                      const float* y_in, int ld_y_in, int act_in, void* workspace, size_t workspace_bytes, void* stream, int io16 = 0);

extern "C" int kpx_conv2d_dgrad_f32(const float* dy, int N, int Ho, int Wo, int Cout, int lddy,
                                    const float* w, int KH, int KW,
                                    float* dx, int Hi, int Wi, int Cin, int lddx,
                                    int stride, int pad_t, int pad_l, int arith, void* workspace, size_t workspace_bytes, void* stream) {
    return dgrad_impl(dy, N, Ho, Wo, Cout, lddy, w, KH, KW, dx, Hi, Wi, Cin, lddx, stride, pad_t, pad_l, arith, nullptr, 0, KPX_ACT_NONE,
                      workspace, workspace_bytes, stream);
}

extern "C" int kpx_conv2d_dgrad_act_f32(const float* dy, int N, int Ho, int Wo, int Cout, int lddy,
                                        const float* w, int KH, int KW,
                                        float* dx, int Hi, int Wi, int Cin, int lddx,
                                        int stride, int pad_t, int pad_l, int arith,
                                        const float* y_in, int ld_y_in, int act_in, void* workspace, size_t workspace_bytes, void* stream) {
    if (!y_in || ld_y_in < Cin || (act_in != KPX_ACT_RELU && act_in != KPX_ACT_LRELU)) return KPX_EINVAL;
    return dgrad_impl(dy, N, Ho, Wo, Cout, lddy, w, KH, KW, dx, Hi, Wi, Cin, lddx, stride, pad_t, pad_l, arith, y_in, ld_y_in, act_in,
                      workspace, workspace_bytes, stream);
}

static inline void launch_wgrad_reduce(const float* ws, float* dw, size_t n, int S, hipStream_t s);
// bf16 configuration, image-input layers (the images stay fp32; what the layer produces / receives is bf16).  KPX_EINVAL for shapes the
// LDS-resident image kernels (conv_rgb.hip, conv_wsmall.hip) do not take: the caller then converts and uses the fp32 entries.
//   forward:  x fp32 [N, Hi, Wi, Cin <= 4] contiguous -> y bf16 (pixel stride ldy elements)
extern "C" int kpx_conv_image_fwd_bf16(const float* x, int N, int Hi, int Wi, int Cin, const float* w, int KH, int KW, const float* bias,
                                       void* y, int Ho, int Wo, int Cout, int ldy, int stride, int pad_t, int pad_l, int act, void* stream) {
    if (!x || !w || !y || N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cin > 4 || Cout <= 0 || Cout > 64 || Ho * Wo < 256 || stride < 1 || stride > 2 ||
        ldy < Cout || act < 0 || act > 3)
        return KPX_EINVAL;
    const int rc = kpx_conv_rgb_fwd(x, N, Hi, Wi, Cin, w, KH, KW, bias, (float*)y, Ho, Wo, Cout, ldy, stride, pad_t, pad_l, act, kpx_stream(stream), 1);
    return rc == -2 ? KPX_EINVAL : rc;
}
//   data gradient:  dy bf16 (pixel stride lddy elements, a multiple of 8) -> dx fp32 [N, Hi, Wi, Cin] (pixel stride lddx)
extern "C" int kpx_conv_image_dgrad_bf16(const void* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                                         float* dx, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l, void* stream) {
    if (!dy || !w || !dx || N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cin > 4 || Cout <= 0 || Ho <= 0 || Wo <= 0 || lddy < Cout || lddx < Cin)
        return KPX_EINVAL;
    const int rc = kpx_conv_rgb_dgrad((const float*)dy, N, Ho, Wo, Cout, lddy, w, KH, KW, dx, Hi, Wi, Cin, lddx, stride, pad_t, pad_l, kpx_stream(stream), 1);
    return rc == -2 ? KPX_EINVAL : rc;
}
//   weight gradient:  x fp32 image, dy bf16 (pixel stride lddy elements, a multiple of 4) -> dw fp32 [KH, KW, Cin, Cout]
//   (workspace: kpx_conv2d_wgrad_workspace_bytes)
extern "C" int kpx_conv_image_wgrad_bf16(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const void* dy, int Ho, int Wo, int Cout, int lddy,
                                         float* dw, int KH, int KW, int stride, int pad_t, int pad_l, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cin > 4 || Cout <= 0 || Ho <= 0 || Wo <= 0 || ldx < Cin || lddy < Cout)
        return KPX_EINVAL;
    const bool off32_ok = (size_t)N * Hi * Wi * (size_t)ldx * 4 < 0x60000000ull && (size_t)N * Ho * Wo * (size_t)lddy * 4 < 0x60000000ull;
    if (!off32_ok || Cout % 4 || lddy % 4 || !aligned16(dy)) return KPX_EINVAL;
    const int Ss = kpx_wsmall_splits(N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride);
    const size_t slab = (size_t)KH * KW * Cin * Cout;
    if (Ss < 1 || (Ss > 1 && (!workspace || workspace_bytes < (size_t)Ss * slab * 4))) return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    int rc = kpx_wsmall_launch(x, N, Hi, Wi, Cin, ldx, (const float*)dy, Ho, Wo, Cout, lddy, KH, KW, stride, pad_t, pad_l, Ss > 1 ? (float*)workspace : dw, Ss, s, 1);
    if (rc) return rc;
    if (Ss > 1) { launch_wgrad_reduce((const float*)workspace, dw, slab, Ss, s); rc = kpx_launch_status(); }
    return rc;
}

// bf16 configuration: dy bf16, dx bf16 (dx_f32 = 0) or fp32 (the gradient towards an fp32 tensor); y_in (optional, with act_in): the bf16
// ACTIVATED tensor dx is the gradient of -- its activation backward is applied in the epilogue.  KPX_EINVAL for shapes the bf16-pipe gather
// kernel does not take.
extern "C" int kpx_conv2d_dgrad_bf16(const void* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                                     void* dx, int dx_f32, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l,
                                     const void* y_in, int ld_y_in, int act_in, void* workspace, size_t workspace_bytes, void* stream) {
    if (lddy % 8 || !aligned16(dy) || !aligned16(dx) || (!dx_f32 && lddx % 4)) return KPX_EINVAL;
    if (y_in && (ld_y_in < Cin || (act_in != KPX_ACT_RELU && act_in != KPX_ACT_LRELU))) return KPX_EINVAL;
    return dgrad_impl((const float*)dy, N, Ho, Wo, Cout, lddy, w, KH, KW, (float*)dx, Hi, Wi, Cin, lddx, stride, pad_t, pad_l, KPX_ARITH_BF16,
                      (const float*)y_in, ld_y_in, y_in ? act_in : KPX_ACT_NONE, workspace, workspace_bytes, stream, 1 | (dx_f32 ? 0 : 2) | (y_in ? 4 : 0));
}

// the specialised data-gradient kernels have no factor in their epilogues: one pass over dx afterwards (contiguous tensors only)
static int dgrad_act_pass(int rc, float* dx, int N, int Hi, int Wi, int Cin, int lddx, const float* y_in, int ld_y_in, int act_in, void* stream) {
    if (rc || !y_in) return rc;
    if (lddx != Cin || ld_y_in != Cin) return KPX_EINVAL;
    return kpx_act_bwd_f32(dx, y_in, dx, (size_t)N * Hi * Wi * Cin, act_in, stream);
}

static int dgrad_impl(const float* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                      float* dx, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l, int arith,
                      const float* y_in, int ld_y_in, int act_in, void* workspace, size_t workspace_bytes, void* stream, int io16) {
    if (!dy || !w || !dx || N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 ||
        KH <= 0 || KW <= 0 || stride <= 0 || lddy < Cout || lddx < Cin || arith < 0 || arith > 1)
        return KPX_EINVAL;
    if (stride > 2) return KPX_EINVAL;             // at most 4 parity classes per launch (the path has strides 1 and 2)
    if (io16) goto gather;                         // bf16 tensors: only the bf16-pipe gather kernel reads them
    if (KH == 3 && KW == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && Ho == Hi && Wo == Wi &&
        kpx_conv3x3_c16_eligible(N, Hi, Wi, Cout, Cin, lddy, lddx, dy))
        return dgrad_act_pass(kpx_conv3x3_c16_f32(dy, N, Hi, Wi, Cout, lddy, w, 1, nullptr, dx, lddx, KPX_ACT_NONE, nullptr, stream),
                              dx, N, Hi, Wi, Cin, lddx, y_in, ld_y_in, act_in, stream);
    if (Cin <= 4) {                                // gradient towards an image: VALU kernel (conv_rgb.hip)
        const int rc = kpx_conv_rgb_dgrad(dy, N, Ho, Wo, Cout, lddy, w, KH, KW, dx, Hi, Wi, Cin, lddx, stride, pad_t, pad_l, kpx_stream(stream));
        if (rc != -2) return dgrad_act_pass(rc, dx, N, Hi, Wi, Cin, lddx, y_in, ld_y_in, act_in, stream);
    }
    if (KH == 3 && KW == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && Ho == Hi && Wo == Wi &&
        workspace && workspace_bytes >= wino_ws_bytes(Cin, Cout) && kpx_wino_eligible(N, Hi, Wi, Cout, Cin, lddy, dy))
        return dgrad_act_pass(kpx_wino_conv3x3(dy, N, Hi, Wi, Cout, lddy, w, Cin, Cout, 1, nullptr, KPX_ACT_NONE, dx, Cin, lddx, (float*)workspace, kpx_stream(stream)),
                              dx, N, Hi, Wi, Cin, lddx, y_in, ld_y_in, act_in, stream);
gather:
    ConvGeom g{};
    g.x = dy; g.y = dx; g.w = w; g.bias = nullptr; g.io16 = io16;
    g.mul_y = y_in; g.ld_mul = ld_y_in; g.mul_act = act_in;
    g.N = N; g.Hi = Ho; g.Wi = Wo; g.Cin = Cout; g.ldx = lddy;       // "input" of the gather = dy
    g.Ho = Hi; g.Wo = Wi; g.Cout = Cin; g.ldy = lddx;                // "output" = dx
    g.osy = stride; g.osx = stride;
    g.isy = 1; g.ity = -1; g.isx = 1; g.itx = -1;
    g.wrs = stride; g.wqs = stride; g.KW = KW;
    g.wts = Cin * Cout; g.ldw = Cout; g.act = KPX_ACT_NONE; g.terms = arith == KPX_ARITH_BF16 ? 1 : 3;
    g.vecA = (lddy % 4 == 0) && aligned16(dy);
    g.vecB = (Cout % 4 == 0) && aligned16(w);
    g.ncls = 0;
    for (int ph = 0; ph < stride; ++ph) {
        for (int pw = 0; pw < stride; ++pw) {
            if (ph >= Hi || pw >= Wi) continue;
            ConvClass c{};
            c.Ha = (Hi - ph + stride - 1) / stride; c.Wa = (Wi - pw + stride - 1) / stride;
            c.oy0 = ph; c.ox0 = pw;
            const int r0 = (ph + pad_t) % stride, q0 = (pw + pad_l) % stride;
            c.Tr = r0 < KH ? (KH - r0 + stride - 1) / stride : 0;
            c.Tq = q0 < KW ? (KW - q0 + stride - 1) / stride : 0;
            if (c.Tr == 0 || c.Tq == 0) { c.Tr = 0; c.Tq = 0; }
            c.iy0 = (ph + pad_t - r0) / stride;
            c.ix0 = (pw + pad_l - q0) / stride;
            c.wr0 = r0; c.wq0 = q0;
            g.cls[g.ncls++] = c;
        }
    }
    if (io16 && !kpx_gemm3_eligible(&g)) return KPX_EINVAL;
    if (g.vecA && g.vecB && Cout % 4 == 0) {
        const int S = dgrad_splitk_plan(N, Hi, Wi, Cin, Cout, KH, KW, stride);
        g.ws_slab = (size_t)N * Hi * Wi * Cin;
        if (S > 1 && workspace && workspace_bytes >= (size_t)S * g.ws_slab * 4) { g.ksplit = S; g.ws = (float*)workspace; }
    }
    return launch_gather_conv<true>(g, kpx_stream(stream));
}

// ------------------------------------------------------------------------------------------ wgrad
// dw[tap][c][k] = sum_p x[p shifted by tap][c] * dy[p][k]: GEMM with M = Cin, N = Cout, K = pixels.
// Both operands are pixel-major in memory, so the LDS tiles are plain [pixel][channel] copies of NHWC
// and the MFMA operands are ds_read_b32 (32 consecutive channels per half-wave, conflict-free).
// The pixel range is split over blocks (split-K); partial slabs are summed by wgrad_reduce_kernel in a
// fixed order, so the result is bitwise reproducible (no float atomics).

template <int BMc, int BNk, int WM, int WN, bool VEC, bool MERGE>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_kernel(const WgradGeom g) {
    constexpr int NT = WM * WN * 64;
    constexpr int BKP = 32;                          // pixels per K chunk
    constexpr int TM = BMc / (WM * 32), TN = BNk / (WN * 32);
    constexpr int RA = (BKP * BMc / 4) / NT, RB = (BKP * BNk / 4) / NT;     // float4 units per thread
    static_assert(RA >= 1 && RB >= 1 && TM >= 1 && TN >= 1, "tile/wave shape");
    __shared__ __attribute__((aligned(16))) float As[2][BKP * BMc];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKP * BNk];

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
    const int cbase = cti * BMc, kbase = kti * BNk;
    const int pbeg = split * g.pps;
    const int pend = min(g.P, pbeg + g.pps);

    // per-thread loader coordinates: A unit -> (pixel-in-chunk, channel group)
    int apx[RA], ac[RA], an[RA], aho[RA], awo[RA], aq[RA][4];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int u = t + NT * i;
        apx[i] = u / (BMc / 4);
        ac[i] = cbase + (u % (BMc / 4)) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) aq[i][j] = MERGE ? (ac[i] + j) / g.merge : 0;
        const int p = pbeg + apx[i];
        an[i] = p / (g.Ho * g.Wo);
        const int rem = p - an[i] * g.Ho * g.Wo;
        aho[i] = rem / g.Wo;
        awo[i] = rem - aho[i] * g.Wo;
    }
    int bpx[RB], bk[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int u = t + NT * i;
        bpx[i] = u / (BNk / 4);
        bk[i] = kbase + (u % (BNk / 4)) * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[RA], rb[RB];
    unsigned am[RA], bm[RB];
    int p0 = pbeg;
    auto tail4 = [](int c, int lim) -> unsigned {
        const int r_ = lim - c;
        return r_ >= 4 ? 15u : (r_ <= 0 ? 0u : (15u >> (4 - r_)));
    };
    // unconditional loads from a clamped address, masked when written to LDS (see conv_igemm_kernel)
    auto load_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int p = p0 + apx[i];
            const int ih = aho[i] * g.stride + r - g.pad_t, iw = awo[i] * g.stride + q - g.pad_l;
            const int c = ac[i];
            unsigned mask = 0;
            const float* ptr = g.x + ((ptrdiff_t)(an[i] * g.Hi + ih) * g.Wi + iw) * g.ldx + c;
            if (p < pend && (unsigned)ih < (unsigned)g.Hi) {
                if (MERGE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c + j < g.Cin && (unsigned)(iw + aq[i][j]) < (unsigned)g.Wi) mask |= 1u << j;
                } else if ((unsigned)iw < (unsigned)g.Wi) {
                    mask = tail4(c, g.Cin);
                }
            }
            am[i] = mask;
            f32x4 v;
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(mask ? ptr : g.x);
            } else {
                v[0] = *((mask & 1u) ? ptr : g.x);
                v[1] = *((mask & 2u) ? ptr + 1 : g.x);
                v[2] = *((mask & 4u) ? ptr + 2 : g.x);
                v[3] = *((mask & 8u) ? ptr + 3 : g.x);
            }
            ra[i] = v;
            awo[i] += BKP;
            while (awo[i] >= g.Wo) {
                awo[i] -= g.Wo;
                if (++aho[i] == g.Ho) { aho[i] = 0; ++an[i]; }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int p = p0 + bpx[i];
            const int k = bk[i];
            const unsigned mask = p < pend ? tail4(k, g.Cout) : 0u;
            const float* ptr = g.dy + (size_t)p * g.lddy + k;
            bm[i] = mask;
            f32x4 v;
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(mask ? ptr : g.dy);
            } else {
                v[0] = *((mask & 1u) ? ptr : g.dy);
                v[1] = *((mask & 2u) ? ptr + 1 : g.dy);
                v[2] = *((mask & 4u) ? ptr + 2 : g.dy);
                v[3] = *((mask & 8u) ? ptr + 3 : g.dy);
            }
            rb[i] = v;
        }
        p0 += BKP;
    };
    auto masked = [](f32x4 v, unsigned m) -> f32x4 {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (m >> j) & 1u ? v[j] : 0.f;
        return o;
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[buf][(t + NT * i) * 4]) = masked(ra[i], am[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][(t + NT * i) * 4]) = masked(rb[i], bm[i]);
    };

    const int nchunks = pend > pbeg ? (pend - pbeg + BKP - 1) / BKP : 0;
    if (nchunks > 0) { load_chunk(); store_chunk(0); }
    __syncthreads();
    const int wrow = wm * TM * 32, wcol = wn * TN * 32;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk();
        const float* Ab = As[buf];
        const float* Bb = Bs[buf];
#pragma unroll
        for (int s2 = 0; s2 < BKP / 2; ++s2) {
            const int kp = 2 * s2 + lh;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ab[kp * BMc + wrow + i * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bb[kp * BNk + wcol + j * 32 + li];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_chunk(buf ^ 1);
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

// Fast path of the per-tap wgrad for Wo % 32 == 0 (every generator layer at 32/64/128 pixels): a K chunk is 32 consecutive
// output pixels of ONE image row, so (n, ho, row validity, base addresses) are wave-uniform scalars and each thread only adds
// a fixed offset; invalid units read the zero page; chunk loop unrolled by two for immediate LDS buffer offsets.
struct WgradTapGeom {
    const float* x; const float* dy; float* out;
    int N, Hi, Wi, Cin, ldx;
    int Ho, Wo, Cout, lddy;
    int KH, KW, stride, pad_t, pad_l;
    int total_chunks, cpb, S, ct, kt;
    size_t slab;
};

template <int BMc, int BNk, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_tap_rows_kernel(const WgradTapGeom g) {
    constexpr int NT = WM * WN * 64, CW = 32;
    constexpr int TM = BMc / (WM * 32), TN = BNk / (WN * 32);
    constexpr int RA = (CW * BMc / 4) / NT, RB = (CW * BNk / 4) / NT;
    constexpr int ASZ = CW * BMc, BSZ = CW * BNk;
    static_assert(RA >= 1 && RB >= 1 && TM >= 1 && TN >= 1, "tile/wave shape");
    __shared__ __attribute__((aligned(16))) float As[2 * ASZ];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];

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
    const int cbase = cti * BMc, kbase = kti * BNk;
    const int wpr = g.Wo / CW;
    const int cbeg = split * g.cpb, cend = min(g.total_chunks, cbeg + g.cpb);
    const int nch = cend > cbeg ? cend - cbeg : 0;

    // fixed per-thread parts
    int a_off[RA], a_px[RA];
    bool a_cok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int u = t + NT * i;
        const int px = u / (BMc / 4), c = cbase + (u % (BMc / 4)) * 4;
        a_px[i] = px * g.stride;
        a_off[i] = px * g.stride * g.ldx + c;
        a_cok[i] = c < g.Cin;
    }
    int b_off[RB];
    bool b_kok[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int u = t + NT * i;
        const int px = u / (BNk / 4), kk = kbase + (u % (BNk / 4)) * 4;
        b_off[i] = px * g.lddy + kk;
        b_kok[i] = kk < g.Cout;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // wave-uniform chunk cursor
    int wc = cbeg % wpr, ho = (cbeg / wpr) % g.Ho, n = cbeg / (wpr * g.Ho);
    f32x4 ra[RA], rb[RB];
    auto load_chunk = [&]() {
        const int ih = ho * g.stride + r - g.pad_t;
        const int iwb = wc * CW * g.stride + q - g.pad_l;
        const bool rowok = (unsigned)ih < (unsigned)g.Hi;
        const float* xb = g.x + ((ptrdiff_t)(n * g.Hi + ih) * g.Wi + iwb) * g.ldx;
        const float* yb = g.dy + ((size_t)(n * g.Ho + ho) * g.Wo + wc * CW) * g.lddy;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const bool v = rowok && a_cok[i] && (unsigned)(iwb + a_px[i]) < (unsigned)g.Wi;
            ra[i] = *reinterpret_cast<const f32x4*>(v ? xb + a_off[i] : kpx_zero16);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rb[i] = *reinterpret_cast<const f32x4*>(b_kok[i] ? yb + b_off[i] : kpx_zero16);
        if (++wc == wpr) { wc = 0; if (++ho == g.Ho) { ho = 0; ++n; } }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[buf * ASZ + (t + NT * i) * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bs[buf * BSZ + (t + NT * i) * 4]) = rb[i];
    };
    const int wrow = wm * TM * 32, wcol = wn * TN * 32;
    const int a_rd = lh * BMc + wrow + li, b_rd = lh * BNk + wcol + li;
    auto mfma_steps = [&](int buf, int s0, int s1) {
#pragma unroll
        for (int s2 = s0; s2 < s1; ++s2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[buf * ASZ + a_rd + 2 * s2 * BMc + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[buf * BSZ + b_rd + 2 * s2 * BNk + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    auto chunk = [&](auto bufc, bool more) {
        constexpr int buf = decltype(bufc)::value;
        mfma_steps(buf, 0, 4);
        if (more) load_chunk();
        mfma_steps(buf, 4, 12);
        if (more) store_chunk(buf ^ 1);
        mfma_steps(buf, 12, 16);
        __syncthreads();
    };
    if (nch > 0) { load_chunk(); store_chunk(0); }
    __syncthreads();
    int ch = 0;
    for (; ch + 1 < nch; ch += 2) {
        chunk(kpx_ic<0>{}, true);
        chunk(kpx_ic<1>{}, ch + 2 < nch);
    }
    if (ch < nch) chunk(kpx_ic<0>{}, false);

    float* out = g.out + (size_t)split * g.slab + (size_t)tap * g.Cin * g.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int kk = kbase + wcol + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = cbase + wrow + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (c < g.Cin && kk < g.Cout) out[(size_t)c * g.Cout + kk] = acc[i][j][e];
            }
    }
}

// dw[i] = sum_s ws[s][i] in a fixed order.  Two shapes of the same sum:
//  * few slabs (S <= 16: the discriminator's large filters, 8-34 M elements): a thread owns four consecutive elements (16-B accesses) and walks
//    the slabs in order with all loads of a group of eight in flight -- the 64-elements-per-workgroup form below moved 168 MB at 1.5 TB/s;
//  * many slabs: a workgroup owns 256 consecutive elements (four per lane), its 4 wavefronts stride over the slabs (1 KB per slab and wave),
//    8 loads in flight per lane, then a 4-way LDS combine in fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_few_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t n4, size_t n, int S) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + i * 4);
        int k = 1;
        for (; k + 7 < S; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(ws + (size_t)(k + j) * n + i * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; k < S; ++k) acc += *reinterpret_cast<const f32x4*>(ws + (size_t)k * n + i * 4);
        *reinterpret_cast<f32x4*>(dw + i * 4) = acc;
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t n, int S) {
    constexpr int W = VEC ? 4 : 1;
    typedef float vecw __attribute__((ext_vector_type(W)));
    const int e = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + e) * W;
    vecw acc = 0.f;
    if (i < n) {
        int k = sg;
        for (; k + 28 < S; k += 32) {
            vecw v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const vecw*>(ws + (size_t)(k + 4 * j) * n + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; k < S; k += 4) acc += *reinterpret_cast<const vecw*>(ws + (size_t)k * n + i);
    }
    __shared__ vecw sm[4][64];
    sm[sg][e] = acc;
    __syncthreads();
    if (sg == 0 && i < n) *reinterpret_cast<vecw*>(dw + i) = (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
}

static inline void launch_wgrad_reduce(const float* ws, float* dw, size_t n, int S, hipStream_t s) {
    const bool vec = n % 4 == 0 && (((uintptr_t)ws | (uintptr_t)dw) & 15) == 0;
    if (vec && S <= 16) {
        size_t nb = (n / 4 + 255) / 256; if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(wgrad_reduce_few_kernel, dim3((unsigned)nb), dim3(256), 0, s, ws, dw, n / 4, n, S);
    } else if (vec) {
        hipLaunchKernelGGL(wgrad_reduce_kernel<true>, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, s, ws, dw, n, S);
    } else {
        hipLaunchKernelGGL(wgrad_reduce_kernel<false>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, ws, dw, n, S);
    }
}

// ------------------------------------------------------------------------------------------ wgrad, small channel counts
// High-resolution layers with <= 64 input and <= 32 output channels (pose_encoder tail, translator head, encoder conv_2):
// the generic kernel above gives every filter tap its own workgroups, so x and dy are re-read KH*KW times and most of each
// 64x64 MFMA tile is padding.  Here one workgroup walks chunks of 32 consecutive output pixels of one image row, stages
// the KH x (32+KW-1) input halo and the 32 dy pixels ONCE in LDS, and its 4 wavefronts split the taps (wave w owns taps
// w, w+4, w+8), each tap a 32*CT x 32 accumulator.  Partial slabs per workgroup, same fixed-order reduce.
struct WgradRowsGeom {
    const float* x; const float* dy; float* out;
    int N, Hi, Wi, Cin, ldx;
    int Ho, Wo, Cout, lddy;
    int KH, KW, pad_t, pad_l;
    int total_chunks, cpb, ct;
    size_t slab;
};

template <int CT>
__global__ __launch_bounds__(256) void conv_wgrad_rows_kernel(const WgradRowsGeom g) {
    constexpr int CW = 32, HALO = CW + 2, CinT = 32 * CT, CoutT = 32;
    constexpr int XU = 3 * HALO * (CinT / 4);          // x float4 units per chunk (3 rows max)
    constexpr int RX = (XU + 255) / 256;
    constexpr int RY = (CW * CoutT / 4) / 256;         // = 1
    __shared__ __attribute__((aligned(16))) float xs[2][3 * HALO * CinT];
    __shared__ __attribute__((aligned(16))) float dys[2][CW * CoutT];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int L = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int cti = L % g.ct; L /= g.ct;
    const int split = L;
    const int cbase = cti * CinT;
    const int T = g.KH * g.KW;
    const int wpr = g.Wo / CW;                          // chunks per output row
    const int cbeg = split * g.cpb, cend = min(g.total_chunks, cbeg + g.cpb);

    f32x16 acc[3][CT];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[sl][i][e] = 0.f;

    f32x4 rx[RX], ry[RY];
    unsigned mx[RX], my[RY];
    auto load_chunk = [&](int chunk) {
        const int wc = chunk % wpr, rest = chunk / wpr, ho = rest % g.Ho, n = rest / g.Ho;
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const int u = t + 256 * i;
            const int c4 = u % (CinT / 4), col = (u / (CinT / 4)) % HALO, r = u / ((CinT / 4) * HALO);
            const int ih = ho + r - g.pad_t, iw = wc * CW + col - g.pad_l, c = cbase + c4 * 4;
            unsigned mask = 0;
            if (u < XU && r < g.KH && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi && c < g.Cin) {
                const int rem = g.Cin - c;
                mask = rem >= 4 ? 15u : (15u >> (4 - rem));
            }
            const float* p = g.x + ((ptrdiff_t)(n * g.Hi + ih) * g.Wi + iw) * g.ldx + c;
            mx[i] = mask;
            rx[i] = *reinterpret_cast<const f32x4*>(mask ? p : g.x);
        }
#pragma unroll
        for (int i = 0; i < RY; ++i) {
            const int u = t + 256 * i;
            const int k4 = u % (CoutT / 4), px = u / (CoutT / 4);
            const int k = k4 * 4;
            unsigned mask = 0;
            if (k < g.Cout) { const int rem = g.Cout - k; mask = rem >= 4 ? 15u : (15u >> (4 - rem)); }
            const float* p = g.dy + ((size_t)(n * g.Ho + ho) * g.Wo + wc * CW + px) * g.lddy + k;
            my[i] = mask;
            ry[i] = *reinterpret_cast<const f32x4*>(mask ? p : g.dy);
        }
    };
    auto masked = [](f32x4 v, unsigned m) -> f32x4 {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (m >> j) & 1u ? v[j] : 0.f;
        return o;
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const int u = t + 256 * i;
            if (u < XU) *reinterpret_cast<f32x4*>(&xs[buf][u * 4]) = masked(rx[i], mx[i]);
        }
#pragma unroll
        for (int i = 0; i < RY; ++i) *reinterpret_cast<f32x4*>(&dys[buf][(t + 256 * i) * 4]) = masked(ry[i], my[i]);
    };

    const int nch = cend > cbeg ? cend - cbeg : 0;
    if (nch > 0) { load_chunk(cbeg); store_chunk(0); }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nch;
        if (more) load_chunk(cbeg + ch + 1);
        const float* X = xs[buf];
        const float* Y = dys[buf];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int tap = wave + 4 * sl;
            if (tap < T) {
                const int r = tap / g.KW, q = tap - r * g.KW;
                const float* Xr = X + (r * HALO + q) * CinT;
#pragma unroll
                for (int s2 = 0; s2 < CW / 2; ++s2) {
                    const int kp = 2 * s2 + lh;
                    const float b = Y[kp * CoutT + li];
#pragma unroll
                    for (int i = 0; i < CT; ++i)
                        acc[sl][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xr[kp * CinT + i * 32 + li], b, acc[sl][i], 0, 0, 0);
                }
            }
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    float* out = g.out + (size_t)split * g.slab;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        const int tap = wave + 4 * sl;
        if (tap < T) {
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = cbase + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (c < g.Cin && li < g.Cout) out[((size_t)tap * g.Cin + c) * g.Cout + li] = acc[sl][i][e];
                }
        }
    }
}

// Same idea for the image-input convs (Cin = 3, 7x7): with dense NHWC the KW*Cin floats under one filter row are
// contiguous, so filter row r is ONE tap whose operand for output pixel p is the float window seg[r][p*Cin .. p*Cin+KW*Cin).
// The workgroup stages KH raw row segments of (32+KW-1)*Cin floats; wave w owns filter rows w and w+4.
template <int DUMMY>
__global__ __launch_bounds__(256) void conv_wgrad_rows_merged_kernel(const WgradRowsGeom g) {
    constexpr int CW = 32, SEG = 128, CoutT = 32, KHMAX = 8;
    constexpr int RX = KHMAX * SEG / 256;              // 4 scalar units per thread
    __shared__ float xs[2][KHMAX * SEG];
    __shared__ __attribute__((aligned(16))) float dys[2][CW * CoutT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int split = kpx_xcd_remap(blockIdx.x, gridDim.x);
    const int KWC = g.KW * g.Cin;                       // merged channels per filter row (<= 32)
    const int segf = (CW + g.KW - 1) * g.Cin;           // valid floats per staged row (<= SEG)
    const int wpr = g.Wo / CW;
    const int cbeg = split * g.cpb, cend = min(g.total_chunks, cbeg + g.cpb);

    f32x16 acc[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[sl][e] = 0.f;

    int xr[RX], xf[RX], xcol[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int u = t + 256 * i;
        xr[i] = u / SEG; xf[i] = u % SEG; xcol[i] = xf[i] / g.Cin;
    }
    float rx[RX];
    f32x4 ry;
    unsigned my;
    auto load_chunk = [&](int chunk) {
        const int wc = chunk % wpr, rest = chunk / wpr, ho = rest % g.Ho, n = rest / g.Ho;
        const int iw0 = wc * CW - g.pad_l;
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const int ih = ho + xr[i] - g.pad_t;
            const bool ok = xr[i] < g.KH && xf[i] < segf && (unsigned)ih < (unsigned)g.Hi && (unsigned)(iw0 + xcol[i]) < (unsigned)g.Wi;
            const float* p = g.x + ((ptrdiff_t)(n * g.Hi + ih) * g.Wi + iw0) * g.Cin + xf[i];
            const float v = *(ok ? p : g.x);
            rx[i] = ok ? v : 0.f;
        }
        {
            const int k4 = t % (CoutT / 4), px = t / (CoutT / 4), k = k4 * 4;
            my = 0;
            if (k < g.Cout) { const int rem = g.Cout - k; my = rem >= 4 ? 15u : (15u >> (4 - rem)); }
            const float* p = g.dy + ((size_t)(n * g.Ho + ho) * g.Wo + wc * CW + px) * g.lddy + k;
            ry = *reinterpret_cast<const f32x4*>(my ? p : g.dy);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RX; ++i) xs[buf][t + 256 * i] = rx[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (my >> j) & 1u ? ry[j] : 0.f;
        *reinterpret_cast<f32x4*>(&dys[buf][t * 4]) = o;
    };
    const int nch = cend > cbeg ? cend - cbeg : 0;
    if (nch > 0) { load_chunk(cbeg); store_chunk(0); }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nch;
        if (more) load_chunk(cbeg + ch + 1);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int r = wave + 4 * sl;
            if (r < g.KH) {
                const float* Xr = &xs[buf][r * SEG];
#pragma unroll
                for (int s2 = 0; s2 < CW / 2; ++s2) {
                    const int kp = 2 * s2 + lh;
                    acc[sl] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xr[kp * g.Cin + li], dys[buf][kp * CoutT + li], acc[sl], 0, 0, 0);
                }
            }
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* out = g.out + (size_t)split * g.slab;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int r = wave + 4 * sl;
        if (r < g.KH) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (c < KWC && li < g.Cout) out[((size_t)r * KWC + c) * g.Cout + li] = acc[sl][e];
            }
        }
    }
}

static bool wgrad_rows_merged_plan(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int* S, int* cpb) {
    if (Cin % 4 == 0 || KW * Cin > 32 || (32 + KW - 1) * Cin > 128 || KH > 8 || KW < 2 || Wo % 32 != 0 || Cout > 32 ||
        (long)N * Ho * Wo < 65536)
        return false;
    const long total = (long)N * Ho * (Wo / 32);
    long s = total < 768 ? total : 768;
    *cpb = (int)((total + s - 1) / s);
    *S = (int)((total + *cpb - 1) / *cpb);
    return true;
}

// shape-only eligibility + split plan of the rows kernel (alignment / stride are checked by the caller)
static bool wgrad_rows_plan(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int* S, int* cpb, int* CT) {
    if (KH > 3 || KW > 3 || KH * KW < 2 || Wo % 32 != 0 || Cout > 32 || Cin > 128 || (long)N * Ho * Wo < 65536) return false;
    *CT = Cin > 32 ? 2 : 1;
    const int ct = (Cin + 32 * *CT - 1) / (32 * *CT);
    const long total = (long)N * Ho * (Wo / 32);
    long want = 768 / ct;
    if (want < 1) want = 1;
    long s = total < want ? total : want;
    const size_t slab_bytes = (size_t)KH * KW * Cin * Cout * 4;
    const size_t cap = (size_t)256 << 20;
    if ((size_t)s * slab_bytes > cap) s = (long)(cap / slab_bytes);
    if (s < 1) s = 1;
    *cpb = (int)((total + s - 1) / s);
    *S = (int)((total + *cpb - 1) / *cpb);
    return true;
}

static void wgrad_tiles(int Cin, int Cout, int& bm, int& bn) {
    auto pick = [](int c) { return ((c + 127) / 128 * 128 == (c + 63) / 64 * 64) ? 128 : 64; };
    bm = pick(Cin); bn = pick(Cout);
    if (bm != bn) { bm = 64; bn = 64; }   // only the square tiles are instantiated
}

static inline bool wgrad_merge(int Cin, int ldx, int KW) { return Cin % 4 != 0 && ldx == Cin && KW > 1 && KW * Cin <= 64; }

static int wgrad_splits(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
    int bm, bn;
    wgrad_tiles(Cin, Cout, bm, bn);
    long tiles = (long)KH * KW * ((Cin + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    if (wgrad_merge(Cin, Cin, KW)) tiles = (long)KH * ((Cout + bn - 1) / bn);
    const long P = (long)N * Ho * Wo;
    const long target = bm == 128 ? 2560 : 4096;   // measured optimum: several short rounds balance better than one long one
    long S = target / tiles;                            // floor: never split a layer that already has enough tiles
    const long maxS_pix = P / 512 > 0 ? P / 512 : 1;   // >= 16 chunks of 32 pixels per split
    if (S > maxS_pix) S = maxS_pix;
    const size_t slab_bytes = (size_t)KH * KW * Cin * Cout * 4;
    const size_t cap = (size_t)256 << 20;
    if ((size_t)S * slab_bytes > cap) S = (long)(cap / slab_bytes);
    if (S < 1) S = 1;
    return (int)S;
}

// bf16 configuration: x and dy bf16 (pixel strides multiples of 8 elements), dw fp32; the bf16-pipe weight-gradient kernel of conv_gemm3.hip
// with the generic split plan (workspace: kpx_conv2d_wgrad_workspace_bytes).  KPX_EINVAL for shapes it does not take.
extern "C" int kpx_conv2d_wgrad_bf16(const void* x, int N, int Hi, int Wi, int Cin, int ldx, const void* dy, int Ho, int Wo, int Cout, int lddy,
                                     float* dw, int KH, int KW, int stride, int pad_t, int pad_l, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
        ldx < Cin || lddy < Cout || ldx % 8 || lddy % 8 || !aligned16(x) || !aligned16(dy))
        return KPX_EINVAL;
    WgradGeom g{};
    g.x = (const float*)x; g.dy = (const float*)dy; g.io16 = 1;
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.ldx = ldx;
    g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.lddy = lddy;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad_t = pad_t; g.pad_l = pad_l;
    g.P = N * Ho * Wo;
    g.S = wgrad_splits(N, Ho, Wo, Cin, Cout, KH, KW);
    g.slab = (size_t)KH * KW * Cin * Cout;
    if (g.S > 1 && (!workspace || workspace_bytes < (size_t)g.S * g.slab * 4)) return KPX_EINVAL;
    g.pps = ((g.P + g.S - 1) / g.S + 31) / 32 * 32;
    g.out = g.S > 1 ? (float*)workspace : dw;
    int bm, bn;
    wgrad_tiles(Cin, Cout, bm, bn);
    g.ct = (Cin + bm - 1) / bm; g.kt = (Cout + bn - 1) / bn;
    g.vecA = 1; g.vecB = 1; g.terms = 1;
    if (!kpx_wgrad3_eligible(&g)) return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    int rc = kpx_wgrad3_launch(g, bm, 1, s);
    if (rc) return rc;
    if (g.S > 1) { launch_wgrad_reduce((const float*)workspace, dw, g.slab, g.S, s); rc = kpx_launch_status(); }
    return rc;
}

extern "C" size_t kpx_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
    const int S = wgrad_splits(N, Ho, Wo, Cin, Cout, KH, KW);
    size_t need = S > 1 ? (size_t)S * KH * KW * Cin * Cout * 4 : 0;
    if (KH == 3 && KW == 3) {                      // Winograd wgrad (stride / padding unknown here: upper bound)
        const int Sw = kpx_wino_wgrad_splits(N, Ho, Wo, Cin, Cout);
        const size_t nw = (size_t)Sw * 9 * Cin * Cout * 4;
        if (Sw > 1 && nw > need) need = nw;
    }
    for (int st = 1; st <= 2; ++st) {             // tiny-filter family (conv_wsmall.hip); the stride is unknown here: both
        const int Ss = kpx_wsmall_splits(N, Ho * st, Wo * st, Cin, Ho, Wo, Cout, KH, KW, st);
        const size_t ns = (size_t)Ss * KH * KW * Cin * Cout * 4;
        if (Ss > 1 && ns > need) need = ns;
    }
    int S2, cpb, CT;
    if (wgrad_rows_plan(N, Ho, Wo, Cin, Cout, KH, KW, &S2, &cpb, &CT)) {
        const size_t n2 = (size_t)S2 * KH * KW * Cin * Cout * 4;
        if (n2 > need) need = n2;
    }
    if (wgrad_rows_merged_plan(N, Ho, Wo, Cin, Cout, KH, KW, &S2, &cpb)) {
        const size_t n2 = (size_t)S2 * KH * KW * Cin * Cout * 4;
        if (n2 > need) need = n2;
    }
    return need;
}

extern "C" int kpx_conv2d_wgrad_f32(const float* x, int N, int Hi, int Wi, int Cin, int ldx,
                                    const float* dy, int Ho, int Wo, int Cout, int lddy,
                                    float* dw, int KH, int KW, int stride, int pad_t, int pad_l, int arith,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 ||
        KH <= 0 || KW <= 0 || stride <= 0 || ldx < Cin || lddy < Cout || arith < 0 || arith > 1)
        return KPX_EINVAL;
    {   // tiny filters over very many pixels (image-input layers, 16 -> 16 at full resolution, the 64 -> 4 head): conv_wsmall.hip
        // (16-B staging units: dy always, x when it has a multiple of 4 channels)
        // the kernel addresses x and dy with 32-bit byte offsets at their PIXEL STRIDES (a channel slice of a wider buffer has ldx > Cin)
        const bool off32_ok = (size_t)N * Hi * Wi * (size_t)ldx * 4 < 0x60000000ull && (size_t)N * Ho * Wo * (size_t)lddy * 4 < 0x60000000ull;
        const bool units_ok = off32_ok && Cout % 4 == 0 && lddy % 4 == 0 && aligned16(dy) && (Cin % 4 != 0 || (ldx % 4 == 0 && aligned16(x)));
        const int Ss = units_ok ? kpx_wsmall_splits(N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride) : 0;
        const size_t slab = (size_t)KH * KW * Cin * Cout;
        if (Ss >= 1 && (Ss == 1 || (workspace && workspace_bytes >= (size_t)Ss * slab * 4))) {
            hipStream_t s = kpx_stream(stream);
            int rc = kpx_wsmall_launch(x, N, Hi, Wi, Cin, ldx, dy, Ho, Wo, Cout, lddy, KH, KW, stride, pad_t, pad_l, Ss > 1 ? (float*)workspace : dw, Ss, s);
            if (rc) return rc;
            if (Ss > 1) { launch_wgrad_reduce((const float*)workspace, dw, slab, Ss, s); rc = kpx_launch_status(); }
            return rc;
        }
    }
    // (the bf16x3 weight gradient of conv_gemm3.hip runs where the generic fp32 kernel would, and on the strided layers: measured ahead of the
    //  specialised fp32 kernels it loses -- 18.65 vs 17.9 ms per step in round 3)
    const int g3_first = 0;
    if (g3_first < 2 && KH == 3 && KW == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && Ho == Hi && Wo == Wi && ldx % 4 == 0 && lddy % 4 == 0 &&
        ldx >= ((Cin + 3) & ~3) && aligned16(x) && aligned16(dy) &&
        (size_t)N * Hi * Wi * (size_t)(ldx > lddy ? ldx : lddy) * 4 < 0x60000000ull) {       // (the kernel addresses x / dy with 32-bit byte offsets)
        const int Sw = kpx_wino_wgrad_splits(N, Hi, Wi, Cin, Cout);
        const size_t slab = (size_t)9 * Cin * Cout;
        if (Sw >= 1 && (Sw == 1 || (workspace && workspace_bytes >= (size_t)Sw * slab * 4))) {
            hipStream_t s = kpx_stream(stream);
            int rc = kpx_wino_wgrad3x3(x, N, Hi, Wi, Cin, ldx, dy, Cout, lddy, Sw > 1 ? (float*)workspace : dw, Sw, s);
            if (rc) return rc;
            if (Sw > 1) { launch_wgrad_reduce((const float*)workspace, dw, slab, Sw, s); rc = kpx_launch_status(); }
            return rc;
        }
    }
    {
        int S2, cpb;
        if (!g3_first && stride == 1 && ldx == Cin && lddy % 4 == 0 && aligned16(dy) && wgrad_rows_merged_plan(N, Ho, Wo, Cin, Cout, KH, KW, &S2, &cpb)) {
            WgradRowsGeom r{};
            r.x = x; r.dy = dy;
            r.N = N; r.Hi = Hi; r.Wi = Wi; r.Cin = Cin; r.ldx = ldx;
            r.Ho = Ho; r.Wo = Wo; r.Cout = Cout; r.lddy = lddy;
            r.KH = KH; r.KW = KW; r.pad_t = pad_t; r.pad_l = pad_l;
            r.total_chunks = N * Ho * (Wo / 32); r.cpb = cpb; r.ct = 1;
            r.slab = (size_t)KH * KW * Cin * Cout;
            if (S2 > 1 && (!workspace || workspace_bytes < (size_t)S2 * r.slab * 4)) return KPX_EINVAL;
            r.out = S2 > 1 ? (float*)workspace : dw;
            hipStream_t s = kpx_stream(stream);
            hipLaunchKernelGGL((conv_wgrad_rows_merged_kernel<0>), dim3((unsigned)S2), dim3(256), 0, s, r);
            int rc = kpx_launch_status();
            if (rc) return rc;
            if (S2 > 1) { launch_wgrad_reduce((const float*)workspace, dw, r.slab, S2, s); rc = kpx_launch_status(); }
            return rc;
        }
    }
    {
        int S2, cpb, CT;
        const bool vec = (ldx % 4 == 0) && (lddy % 4 == 0) && (Cin % 4 == 0) && aligned16(x) && aligned16(dy);
        if (!g3_first && stride == 1 && vec && wgrad_rows_plan(N, Ho, Wo, Cin, Cout, KH, KW, &S2, &cpb, &CT)) {
            WgradRowsGeom r{};
            r.x = x; r.dy = dy;
            r.N = N; r.Hi = Hi; r.Wi = Wi; r.Cin = Cin; r.ldx = ldx;
            r.Ho = Ho; r.Wo = Wo; r.Cout = Cout; r.lddy = lddy;
            r.KH = KH; r.KW = KW; r.pad_t = pad_t; r.pad_l = pad_l;
            r.total_chunks = N * Ho * (Wo / 32); r.cpb = cpb; r.ct = (Cin + 32 * CT - 1) / (32 * CT);
            r.slab = (size_t)KH * KW * Cin * Cout;
            if (S2 > 1 && (!workspace || workspace_bytes < (size_t)S2 * r.slab * 4)) return KPX_EINVAL;
            r.out = S2 > 1 ? (float*)workspace : dw;
            hipStream_t s = kpx_stream(stream);
            const dim3 grid((unsigned)(S2 * r.ct));
            if (CT == 2) hipLaunchKernelGGL((conv_wgrad_rows_kernel<2>), grid, dim3(256), 0, s, r);
            else hipLaunchKernelGGL((conv_wgrad_rows_kernel<1>), grid, dim3(256), 0, s, r);
            int rc = kpx_launch_status();
            if (rc) return rc;
            if (S2 > 1) {
                launch_wgrad_reduce((const float*)workspace, dw, r.slab, S2, s);
                rc = kpx_launch_status();
            }
            return rc;
        }
    }
    WgradGeom g{};
    g.x = x; g.dy = dy;
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.ldx = ldx;
    g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.lddy = lddy;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad_t = pad_t; g.pad_l = pad_l;
    g.P = N * Ho * Wo;
    g.S = wgrad_splits(N, Ho, Wo, Cin, Cout, KH, KW);
    g.slab = (size_t)KH * KW * Cin * Cout;
    if (g.S > 1 && (!workspace || workspace_bytes < (size_t)g.S * g.slab * 4)) return KPX_EINVAL;
    g.pps = ((g.P + g.S - 1) / g.S + 31) / 32 * 32;
    g.out = g.S > 1 ? (float*)workspace : dw;
    int bm, bn;
    wgrad_tiles(Cin, Cout, bm, bn);
    g.ct = (Cin + bm - 1) / bm; g.kt = (Cout + bn - 1) / bn;
    g.vecA = (ldx % 4 == 0) && aligned16(x);
    g.vecB = (lddy % 4 == 0) && aligned16(dy);
    g.terms = arith == KPX_ARITH_BF16 ? 1 : 3;
    // (strided layers: the bf16x3 kernel below beats the fp32 tap-rows kernel -- encoder conv_3 at N = 64: 0.152 vs 0.198 ms; the stride-1
    //  layers that reach this point are faster on tap-rows / rows, and the 3x3 stride-1 layers on the Winograd weight gradient)
    const bool g3_strided = stride > 1 && g.vecA && g.vecB && kpx_wgrad3_eligible(&g);
    if (!g3_first && !g3_strided && g.vecA && g.vecB && Cin % 4 == 0 && Cout % 4 == 0 && Wo % 32 == 0 && !wgrad_merge(Cin, ldx, KW)) {
        // chunk-aligned split: same S as the generic plan, but in units of 32-pixel row chunks
        WgradTapGeom r{};
        r.x = x; r.dy = dy;
        r.N = N; r.Hi = Hi; r.Wi = Wi; r.Cin = Cin; r.ldx = ldx;
        r.Ho = Ho; r.Wo = Wo; r.Cout = Cout; r.lddy = lddy;
        r.KH = KH; r.KW = KW; r.stride = stride; r.pad_t = pad_t; r.pad_l = pad_l;
        r.total_chunks = N * Ho * (Wo / 32);
        r.cpb = (r.total_chunks + g.S - 1) / g.S;
        r.S = (r.total_chunks + r.cpb - 1) / r.cpb;             // <= g.S, so the workspace query still covers it
        r.ct = g.ct; r.kt = g.kt; r.slab = g.slab;
        r.out = r.S > 1 ? (float*)workspace : dw;
        hipStream_t s = kpx_stream(stream);
        const dim3 grid((unsigned)(r.S * KH * KW * r.ct * r.kt));
        if (bm == 128) hipLaunchKernelGGL((conv_wgrad_tap_rows_kernel<128, 128, 2, 4>), grid, dim3(512), 0, s, r);
        else hipLaunchKernelGGL((conv_wgrad_tap_rows_kernel<64, 64, 2, 2>), grid, dim3(256), 0, s, r);
        int rc = kpx_launch_status();
        if (rc) return rc;
        if (r.S > 1) { launch_wgrad_reduce((const float*)workspace, dw, r.slab, r.S, s); rc = kpx_launch_status(); }
        return rc;
    }
    int taps = KH * KW;
    if (wgrad_merge(Cin, ldx, KW)) {     // image inputs (Cin = 3): one tap per filter row, KW*Cin merged channels
        g.merge = Cin; g.Cin = KW * Cin; g.KW = 1; g.vecA = 0; taps = KH;
        g.ct = (g.Cin + bm - 1) / bm;
    }
    hipStream_t s = kpx_stream(stream);
    const dim3 grid((unsigned)(g.S * taps * g.ct * g.kt));
    const bool vec = g.vecA && g.vecB && !g.merge;
    if (vec && kpx_wgrad3_eligible(&g)) {                // bf16x3 (fp32-equivalent) weight gradient on the bf16 matrix pipe (conv_gemm3.hip)
        int rc = kpx_wgrad3_launch(g, bm, g.terms == 1 ? 1 : 3, s);
        if (rc) return rc;
        if (g.S > 1) { launch_wgrad_reduce((const float*)workspace, dw, g.slab, g.S, s); rc = kpx_launch_status(); }
        return rc;
    }
    if (g.merge) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, false, true>), grid, dim3(256), 0, s, g);
    else if (bm == 128 && vec) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 4, true, false>), grid, dim3(512), 0, s, g);
    else if (bm == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 4, false, false>), grid, dim3(512), 0, s, g);
    else if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, true, false>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, false, false>), grid, dim3(256), 0, s, g);
    int rc = kpx_launch_status();
    if (rc) return rc;
    if (g.S > 1) {
        launch_wgrad_reduce((const float*)workspace, dw, g.slab, g.S, s);
        rc = kpx_launch_status();
    }
    return rc;
}
