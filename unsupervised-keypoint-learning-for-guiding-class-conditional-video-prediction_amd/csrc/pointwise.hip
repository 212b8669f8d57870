// HBM-bound NHWC kernels of the detector_translator path: batch-norm (stats / apply / backward), activation
// backward, bilinear x2 (TF-1.12 legacy sampling), channel-slice copy, head blend, VGG input transform, max-pool.
// All are streaming kernels: 16-B per-lane accesses along the contiguous channel axis, grid-stride loops capped
// at 2048 blocks, per-channel reductions accumulated in fp64 and finished by a second tiny kernel (no atomics,
// bitwise reproducible).
#include "kpx_common.h"

#define KPX_MAX_BLOCKS 2048
#define KPX_RED_BLOCKS 1024

static inline unsigned grid_for(size_t work_items) {
    size_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > KPX_MAX_BLOCKS) b = KPX_MAX_BLOCKS;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------ utilities
__global__ __launch_bounds__(256) void fill_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
extern "C" int kpx_fill_f32(float* p, size_t n, float value, void* stream) {
    if (!p) return KPX_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, kpx_stream(stream), p, n, value);
    return kpx_launch_status();
}

__global__ __launch_bounds__(256) void u8_to_unit_kernel(const unsigned char* __restrict__ src, size_t n, float* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = (float)((double)src[i] / 255.0);          // numpy's float64 division, then tf.data's float32 cast
        dst[i] = __fsub_rn(__fmul_rn(v, 2.0f), 1.0f);              // map_fn in float32 (no fma contraction)
    }
}
extern "C" int kpx_u8_to_unit_f32(const unsigned char* src, size_t n, float* dst, void* stream) {
    if (!src || !dst) return KPX_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(u8_to_unit_kernel, dim3(grid_for(n)), dim3(256), 0, kpx_stream(stream), src, n, dst);
    return kpx_launch_status();
}

__global__ __launch_bounds__(256) void axpy_kernel(float* y, const float* x, size_t n, float a) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = fmaf(a, x[i], y[i]);
}
extern "C" int kpx_axpy_f32(float* y, const float* x, size_t n, float a, void* stream) {
    if (!y || !x) return KPX_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, kpx_stream(stream), y, x, n, a);
    return kpx_launch_status();
}

template <bool VEC>
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* src, int lds_, float* dst, int ldd, size_t P, int C) {
    const int G = VEC ? C / 4 : C;
    const size_t total = P * (size_t)G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / G;
        const int c = (int)(i - p * G);
        if (VEC) *reinterpret_cast<f32x4*>(dst + p * ldd + c * 4) = *reinterpret_cast<const f32x4*>(src + p * lds_ + c * 4);
        else dst[p * ldd + c] = src[p * lds_ + c];
    }
}
extern "C" int kpx_copy_channels_f32(const float* src, int ldsrc, float* dst, int lddst, size_t P, int C, void* stream) {
    if (!src || !dst || C <= 0 || ldsrc < C || lddst < C) return KPX_EINVAL;
    if (P == 0) return 0;
    const bool vec = (C % 4 == 0) && (ldsrc % 4 == 0) && (lddst % 4 == 0) && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    if (vec) hipLaunchKernelGGL(copy_channels_kernel<true>, dim3(grid_for(P * (C / 4))), dim3(256), 0, kpx_stream(stream), src, ldsrc, dst, lddst, P, C);
    else hipLaunchKernelGGL(copy_channels_kernel<false>, dim3(grid_for(P * C)), dim3(256), 0, kpx_stream(stream), src, ldsrc, dst, lddst, P, C);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ activation backward
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dy, const float* y, float* dz, size_t n, int act) {
    const float neg = act == KPX_ACT_RELU ? 0.f : (act == KPX_ACT_LRELU ? 0.01f : 1.f);
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = act == KPX_ACT_TANH ? g[j] * (1.0f - v[j] * v[j]) : (v[j] > 0.f ? g[j] : g[j] * neg);
        reinterpret_cast<f32x4*>(dz)[i] = g;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dz[i] = act == KPX_ACT_TANH ? dy[i] * (1.0f - y[i] * y[i]) : (y[i] > 0.f ? dy[i] : dy[i] * neg);
}
extern "C" int kpx_act_bwd_f32(const float* dy, const float* y, float* dz, size_t n, int act, void* stream) {
    if (!dy || !y || !dz || act < 0 || act > 3 || (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dz) & 15)) return KPX_EINVAL;
    if (n == 0) return 0;
    if (act == KPX_ACT_NONE && dz == dy) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, kpx_stream(stream), dy, y, dz, n, act);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ per-channel reductions
// MODE 0: sum x            (bias gradient)
// MODE 1: sum x, sum x^2   (batch-norm statistics)
// MODE 2: sum dz, sum dz*xhat with dz = dy*[act'(y)], y = (x-mean)*invstd*gamma+beta   (batch-norm backward)
struct RedArgs {
    const float* x; int ldx; const float* dy; int lddy;
    size_t P; int C;
    const float* mean; const float* invstd; const float* gamma; const float* beta; int act;
    double* part;       // [gridDim.x][2][C]
    // blockIdx.z = group g of a batched launch (weight-sharing calls with statistics of their own): x, dy advance by P pixels, mean /
    // invstd by C, the partials by part_gstride doubles per group (all 0 / unused for the plain single-group entries)
    size_t part_gstride;
};

template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const RedArgs a) {
    constexpr int W = VEC ? 4 : 1;
    const int G = VEC ? a.C / 4 : a.C;               // channel groups
    const int Gb = G < 256 ? G : 256;                // groups handled per blockIdx.y slice
    const int PPB = 256 / Gb;
    const int t = threadIdx.x;
    const int cgl = t % Gb, prow = t / Gb;
    const int cg = blockIdx.y * Gb + cgl;
    const bool active = prow < PPB && cg < G;
    const int grp = blockIdx.z;
    const float* const gx = a.x + (size_t)grp * a.P * a.ldx;
    const float* const gdy = MODE == 2 ? a.dy + (size_t)grp * a.P * a.lddy : nullptr;
    double s0[W], s1[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { s0[j] = 0.0; s1[j] = 0.0; }
    float mu[W], is[W], ga[W], be[W];
    if (MODE == 2 && active) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int c = cg * W + j;
            mu[j] = a.mean[grp * a.C + c]; is[j] = a.invstd[grp * a.C + c]; ga[j] = a.gamma[c]; be[j] = a.beta[c];
        }
    }
    if (active) {
        // four pixels per trip: eight independent 16-B loads in flight per thread (the loop is latency-bound otherwise)
        constexpr int U = VEC ? 4 : 1;
        const size_t step = (size_t)gridDim.x * PPB;
        for (size_t p0 = (size_t)blockIdx.x * PPB + prow; p0 < a.P; p0 += step * U) {
            float xv[U][W], gv[U][W];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t p = p0 + u * step;
                const bool in = p < a.P;
                if (VEC) {
                    const f32x4 v = in ? *reinterpret_cast<const f32x4*>(gx + p * a.ldx + cg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < W; ++j) xv[u][j] = v[j];
                    if (MODE == 2) {
                        const f32x4 d = in ? *reinterpret_cast<const f32x4*>(gdy + p * a.lddy + cg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < W; ++j) gv[u][j] = d[j];
                    }
                } else {
                    xv[u][0] = gx[p * a.ldx + cg];
                    if (MODE == 2) gv[u][0] = gdy[p * a.lddy + cg];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (p0 + u * step >= a.P) break;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    if (MODE == 0) s0[j] += (double)xv[u][j];
                    if (MODE == 1) { s0[j] += (double)xv[u][j]; s1[j] += (double)xv[u][j] * (double)xv[u][j]; }
                    if (MODE == 2) {
                        const float xh = (xv[u][j] - mu[j]) * is[j];
                        const float yv = fmaf(xh, ga[j], be[j]);
                        const float dz = (a.act == KPX_ACT_RELU && !(yv > 0.f)) ? 0.f : gv[u][j];
                        s0[j] += (double)dz; s1[j] += (double)dz * (double)xh;
                    }
                }
            }
        }
    }
    __shared__ double sm[2][256 * W];
#pragma unroll
    for (int j = 0; j < W; ++j) { sm[0][t * W + j] = s0[j]; sm[1][t * W + j] = s1[j]; }
    __syncthreads();
    if (t < Gb && blockIdx.y * Gb + t < G) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            double r0 = 0.0, r1 = 0.0;
            for (int pr = 0; pr < PPB; ++pr) { r0 += sm[0][(pr * Gb + t) * W + j]; r1 += sm[1][(pr * Gb + t) * W + j]; }
            const int c = (blockIdx.y * Gb + t) * W + j;
            double* const gp = a.part + (size_t)grp * a.part_gstride;
            gp[((size_t)blockIdx.x * 2 + 0) * a.C + c] = r0;
            gp[((size_t)blockIdx.x * 2 + 1) * a.C + c] = r1;
        }
    }
}

static int launch_chan_reduce(int mode, RedArgs a, int* nb_out, hipStream_t s, int groups = 1) {
    const bool vec = (a.C % 4 == 0) && (a.ldx % 4 == 0) && (((uintptr_t)a.x) & 15) == 0 &&
                     (mode != 2 || ((a.lddy % 4 == 0) && (((uintptr_t)a.dy) & 15) == 0));
    const int G = vec ? a.C / 4 : a.C;
    const int Gb = G < 256 ? G : 256;
    const int PPB = 256 / Gb;
    size_t nb = a.P / ((size_t)PPB * 8);
    if (nb < 1) nb = 1;
    if (nb > KPX_RED_BLOCKS) nb = KPX_RED_BLOCKS;
    *nb_out = (int)nb;
    const dim3 grid((unsigned)nb, (unsigned)((G + Gb - 1) / Gb), (unsigned)groups), block(256);
#define KPX_RED(M) \
    do { if (vec) hipLaunchKernelGGL((chan_reduce_kernel<M, true>), grid, block, 0, s, a); \
         else hipLaunchKernelGGL((chan_reduce_kernel<M, false>), grid, block, 0, s, a); } while (0)
    if (mode == 0) KPX_RED(0); else if (mode == 1) KPX_RED(1); else KPX_RED(2);
#undef KPX_RED
    return kpx_launch_status();
}

// partial sums [KPX_RED_BLOCKS][2][C] doubles, then 2*C floats where bn_bwd parks the finished channel sums
extern "C" size_t kpx_chan_reduce_scratch_bytes(int C) {
    const size_t c = (size_t)(C > 0 ? C : 1);
    return (size_t)KPX_RED_BLOCKS * 2 * c * sizeof(double) + ((2 * c * sizeof(float) + 15) & ~(size_t)15);
}

// 256 threads per channel: the (up to 1024) block partials are summed in a fixed order -- thread t takes partials t, t+256, ..;
// shuffle tree per wavefront, then the four wave sums in LDS order -- so the result does not depend on timing.  (One wavefront per
// channel took 7-10 us of dependent-load latency per launch, 120 launches per step.)
__device__ __forceinline__ void kpx_sum_partials(const double* part, int nb, int C, int c, double& s, double& q) {
    double a0 = 0.0, a1 = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) {
        a0 += part[((size_t)b * 2) * C + c];
        a1 += part[((size_t)b * 2 + 1) * C + c];
    }
    a0 = kpx_wave_sum_d(a0);
    a1 = kpx_wave_sum_d(a1);
    __shared__ double sm[2][4];
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = a0; sm[1][threadIdx.x >> 6] = a1; }
    __syncthreads();
    s = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    q = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
}

__global__ __launch_bounds__(256) void chan_sum_finalize_kernel(const double* part, int nb, int C, float* out) {
    const int c = blockIdx.x;
    double s, q;
    kpx_sum_partials(part, nb, C, c, s, q);
    if (threadIdx.x == 0) out[c] = (float)s;
}
extern "C" int kpx_chan_sum_f32(const float* x, size_t P, int C, int ldx, float* sum_out, void* scratch, void* stream) {
    if (!x || !sum_out || !scratch || C <= 0 || ldx < C || P == 0) return KPX_EINVAL;
    RedArgs a{}; a.x = x; a.ldx = ldx; a.P = P; a.C = C; a.part = (double*)scratch;
    int nb; int rc = launch_chan_reduce(0, a, &nb, kpx_stream(stream));
    if (rc) return rc;
    hipLaunchKernelGGL(chan_sum_finalize_kernel, dim3(C), dim3(256), 0, kpx_stream(stream), (const double*)scratch, nb, C, sum_out);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ batch norm
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* part, int nb, int C, double count, float eps,
                                                               float* mean, float* invstd, float* var_biased,
                                                               float* mm, float* mv, float decay) {
    const int c = blockIdx.x;
    double s, q;
    kpx_sum_partials(part, nb, C, c, s, q);
    if (threadIdx.x != 0) return;
    const double m = s / count;
    double v = q / count - m * m;
    if (v < 0.0) v = 0.0;
    const float mf = (float)m, vf = (float)v;
    mean[c] = mf;
    if (var_biased) var_biased[c] = vf;
    invstd[c] = 1.0f / sqrtf(vf + eps);
    if (mm && mv) {   // TF fused batch norm: moving -= (moving - batch) * (1 - decay), variance Bessel-corrected
        const float one_minus = 1.0f - decay;
        const float unb = (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
        mm[c] = mm[c] - (mm[c] - mf) * one_minus;
        mv[c] = mv[c] - (mv[c] - unb) * one_minus;
    }
}
extern "C" int kpx_bn_stats_f32(const float* x, size_t P, int C, int ldx, float eps,
                                float* mean, float* invstd, float* var_biased,
                                float* moving_mean, float* moving_var, float decay, void* scratch, void* stream) {
    if (!x || !mean || !invstd || !scratch || C <= 0 || ldx < C || P == 0) return KPX_EINVAL;
    RedArgs a{}; a.x = x; a.ldx = ldx; a.P = P; a.C = C; a.part = (double*)scratch;
    int nb; int rc = launch_chan_reduce(1, a, &nb, kpx_stream(stream));
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(256), 0, kpx_stream(stream), (const double*)scratch, nb, C,
                       (double)P, eps, mean, invstd, var_biased, moving_mean, moving_var, decay);
    return kpx_launch_status();
}

// Batch-norm backward sums from per-tile sums written by the epilogue of the data-gradient kernel that PRODUCED dy
// (kpx_conv3x3_wino_bnbwd_stats_f32): tile_stats[tile][2][C] = sum(dz), sum(dz * (y - beta)); x_hat = (y - beta) / gamma wherever dz != 0.
__global__ __launch_bounds__(256) void bn_bwd_from_tiles_kernel(const float* __restrict__ ts, size_t tile0, size_t ntiles, int C,
                                                               const float* __restrict__ gamma, float* dgamma, float* dbeta, float* sums, int accumulate) {
    const int c = blockIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (size_t b = threadIdx.x; b < ntiles; b += 256) {
        a0 += (double)ts[((tile0 + b) * 2) * C + c];
        a1 += (double)ts[((tile0 + b) * 2 + 1) * C + c];
    }
    a0 = kpx_wave_sum_d(a0);
    a1 = kpx_wave_sum_d(a1);
    __shared__ double sm[2][4];
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = a0; sm[1][threadIdx.x >> 6] = a1; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double s = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    const double ga = (double)gamma[c];
    const double q = ga != 0.0 ? ((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3])) / ga : 0.0;
    if (accumulate) { dbeta[c] += (float)s; dgamma[c] += (float)q; } else { dbeta[c] = (float)s; dgamma[c] = (float)q; }
    sums[c] = (float)s; sums[C + c] = (float)q;
}

// Batch statistics from the per-tile sums a convolution epilogue wrote (kpx_conv3x3_wino_stats_f32): tile_stats[tile][2][C] floats.
// One workgroup per channel adds the tiles [tile0, tile0 + ntiles) in a fixed order in fp64 (same tree as kpx_sum_partials).
__global__ __launch_bounds__(256) void bn_stats_from_tiles_kernel(const float* __restrict__ ts, size_t tile0, size_t ntiles, int C, double count, float eps,
                                                                 float* mean, float* invstd, float* var_biased, float* mm, float* mv, float decay) {
    const int c = blockIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (size_t b = threadIdx.x; b < ntiles; b += 256) {
        a0 += (double)ts[((tile0 + b) * 2) * C + c];
        a1 += (double)ts[((tile0 + b) * 2 + 1) * C + c];
    }
    a0 = kpx_wave_sum_d(a0);
    a1 = kpx_wave_sum_d(a1);
    __shared__ double sm[2][4];
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = a0; sm[1][threadIdx.x >> 6] = a1; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double s = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]), q = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
    const double m = s / count;
    double v = q / count - m * m;
    if (v < 0.0) v = 0.0;
    const float mf = (float)m, vf = (float)v;
    mean[c] = mf;
    if (var_biased) var_biased[c] = vf;
    invstd[c] = 1.0f / sqrtf(vf + eps);
    if (mm && mv) {
        const float one_minus = 1.0f - decay;
        const float unb = (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
        mm[c] = mm[c] - (mm[c] - mf) * one_minus;
        mv[c] = mv[c] - (mv[c] - unb) * one_minus;
    }
}
extern "C" int kpx_bn_stats_from_tiles_f32(const float* tile_stats, size_t tile0, size_t ntiles, int tile_pixels, int C, float eps,
                                           float* mean, float* invstd, float* var_biased,
                                           float* moving_mean, float* moving_var, float decay, void* stream) {
    if (!tile_stats || !mean || !invstd || C <= 0 || ntiles == 0 || tile_pixels <= 0) return KPX_EINVAL;
    hipLaunchKernelGGL(bn_stats_from_tiles_kernel, dim3(C), dim3(256), 0, kpx_stream(stream), tile_stats, tile0, ntiles, C,
                       (double)ntiles * (double)tile_pixels, eps, mean, invstd, var_biased, moving_mean, moving_var, decay);
    return kpx_launch_status();
}

__global__ void bn_invstd_kernel(const float* var, int C, float eps, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = 1.0f / sqrtf(var[c] + eps);
}
extern "C" int kpx_bn_invstd_f32(const float* var, int C, float eps, float* invstd, void* stream) {
    if (!var || !invstd || C <= 0) return KPX_EINVAL;
    hipLaunchKernelGGL(bn_invstd_kernel, dim3((C + 63) / 64), dim3(64), 0, kpx_stream(stream), var, C, eps, invstd);
    return kpx_launch_status();
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, size_t P, int C, int ldx, const float* mean, const float* invstd,
                                                       const float* gamma, const float* beta, float* y, int ldy, int act) {
    constexpr int W = VEC ? 4 : 1;
    const int G = C / W;
    const size_t total = P * (size_t)G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / G;
        const int c0 = (int)(i - p * G) * W;
        float v[W];
        if (VEC) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * ldx + c0);
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] = xv[j];
        } else v[0] = x[p * ldx + c0];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float xh = (v[j] - mean[c0 + j]) * invstd[c0 + j];
            float r = fmaf(xh, gamma[c0 + j], beta[c0 + j]);
            if (act == KPX_ACT_RELU) r = fmaxf(r, 0.f);
            v[j] = r;
        }
        if (VEC) {
            f32x4 o = {v[0], v[W > 1 ? 1 : 0], v[W > 2 ? 2 : 0], v[W > 3 ? 3 : 0]};
            *reinterpret_cast<f32x4*>(y + p * ldy + c0) = o;
        } else y[p * ldy + c0] = v[0];
    }
}
// Strip mapping for the 16-B-vector case: a thread keeps ONE channel group (4 channels) for its whole life, so the per-channel
// parameters are read once into registers instead of six scalar loads per element, and walks pixels four at a time (independent
// loads in flight).  256 threads = Gb channel groups x PPB pixels; blockIdx.y slices channel counts beyond 1024.
__global__ __launch_bounds__(256) void bn_apply_strip_kernel(const float* __restrict__ x, size_t P, int C, int ldx, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, int ldy, int act) {
    const int G = C >> 2, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    const int t = threadIdx.x, cgl = t % Gb, prow = t / Gb;
    const int cg = blockIdx.y * Gb + cgl;
    if (prow >= PPB || cg >= G) return;
    // blockIdx.z = group of a batched launch: P pixels and C statistics further on (gridDim.z = 1 for the plain entry)
    x += (size_t)blockIdx.z * P * ldx; y += (size_t)blockIdx.z * P * ldy; mean += blockIdx.z * C; invstd += blockIdx.z * C;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cg * 4), is = *reinterpret_cast<const f32x4*>(invstd + cg * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cg * 4), be = *reinterpret_cast<const f32x4*>(beta + cg * 4);
    const size_t step = (size_t)gridDim.x * PPB;
    for (size_t p0 = (size_t)blockIdx.x * PPB + prow; p0 < P; p0 += step * 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t p = p0 + u * step; if (p < P) v[u] = *reinterpret_cast<const f32x4*>(x + p * ldx + cg * 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = p0 + u * step;
            if (p >= P) break;
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (v[u][j] - mu[j]) * is[j];
                r[j] = fmaf(xh, ga[j], be[j]);
                if (act == KPX_ACT_RELU) r[j] = fmaxf(r[j], 0.f);
            }
            *reinterpret_cast<f32x4*>(y + p * ldy + cg * 4) = r;
        }
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_strip_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, size_t P, int C,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int act, const float* __restrict__ sums, float inv_count,
                                                                 float* __restrict__ dx, int lddx) {
    const int G = C >> 2, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    const int t = threadIdx.x, cgl = t % Gb, prow = t / Gb;
    const int cg = blockIdx.y * Gb + cgl;
    if (prow >= PPB || cg >= G) return;
    dy += (size_t)blockIdx.z * P * lddy; x += (size_t)blockIdx.z * P * ldx; dx += (size_t)blockIdx.z * P * lddx;       // group of a batched launch
    mean += blockIdx.z * C; invstd += blockIdx.z * C; sums += blockIdx.z * 2 * C;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cg * 4), is = *reinterpret_cast<const f32x4*>(invstd + cg * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cg * 4), be = *reinterpret_cast<const f32x4*>(beta + cg * 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + cg * 4), s1 = *reinterpret_cast<const f32x4*>(sums + C + cg * 4);
    f32x4 gi, m0, m1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { gi[j] = ga[j] * is[j]; m0[j] = s0[j] * inv_count; m1[j] = s1[j] * inv_count; }
    const size_t step = (size_t)gridDim.x * PPB;
    for (size_t p0 = (size_t)blockIdx.x * PPB + prow; p0 < P; p0 += step * 4) {
        f32x4 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = p0 + u * step;
            if (p < P) { xv[u] = *reinterpret_cast<const f32x4*>(x + p * ldx + cg * 4); gv[u] = *reinterpret_cast<const f32x4*>(dy + p * lddy + cg * 4); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = p0 + u * step;
            if (p >= P) break;
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[u][j] - mu[j]) * is[j];
                const float yv = fmaf(xh, ga[j], be[j]);
                const float dz = (act == KPX_ACT_RELU && !(yv > 0.f)) ? 0.f : gv[u][j];
                r[j] = gi[j] * (dz - m0[j] - xh * m1[j]);
            }
            *reinterpret_cast<f32x4*>(dx + p * lddx + cg * 4) = r;
        }
    }
}
static inline dim3 strip_grid(size_t P, int C, int groups = 1) {
    const int G = C >> 2, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    size_t nb = (P + (size_t)PPB * 4 - 1) / ((size_t)PPB * 4);
    if (nb < 1) nb = 1;
    if (nb > KPX_MAX_BLOCKS) nb = KPX_MAX_BLOCKS;
    return dim3((unsigned)nb, (unsigned)((G + Gb - 1) / Gb), (unsigned)groups);
}

extern "C" int kpx_bn_apply_f32(const float* x, size_t P, int C, int ldx, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, float* y, int ldy, int act, void* stream) {
    if (!x || !y || !mean || !invstd || !gamma || !beta || C <= 0 || ldx < C || ldy < C || act < 0 || act > 1) return KPX_EINVAL;
    if (P == 0) return 0;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(bn_apply_strip_kernel, strip_grid(P, C), dim3(256), 0, kpx_stream(stream), x, P, C, ldx, mean, invstd, gamma, beta, y, ldy, act);
    else hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for(P * C)), dim3(256), 0, kpx_stream(stream), x, P, C, ldx, mean, invstd, gamma, beta, y, ldy, act);
    return kpx_launch_status();
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* part, int nb, int C, float* dgamma, float* dbeta, float* sums, int accumulate) {
    const int c = blockIdx.x;
    double s, q;
    kpx_sum_partials(part, nb, C, c, s, q);
    if (threadIdx.x != 0) return;
    if (accumulate) { dbeta[c] += (float)s; dgamma[c] += (float)q; } else { dbeta[c] = (float)s; dgamma[c] = (float)q; }
    sums[c] = (float)s; sums[C + c] = (float)q;
}
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, int lddy, const float* x, int ldx, size_t P, int C,
                                                           const float* mean, const float* invstd, const float* gamma, const float* beta,
                                                           int act, const float* sums, float inv_count, float* dx, int lddx) {
    constexpr int W = VEC ? 4 : 1;
    const int G = C / W;
    const size_t total = P * (size_t)G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / G;
        const int c0 = (int)(i - p * G) * W;
        float xv[W], gv[W];
        if (VEC) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + p * ldx + c0);
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + p * lddy + c0);
#pragma unroll
            for (int j = 0; j < W; ++j) { xv[j] = a[j]; gv[j] = d[j]; }
        } else { xv[0] = x[p * ldx + c0]; gv[0] = dy[p * lddy + c0]; }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int c = c0 + j;
            const float xh = (xv[j] - mean[c]) * invstd[c];
            const float yv = fmaf(xh, gamma[c], beta[c]);
            const float dz = (act == KPX_ACT_RELU && !(yv > 0.f)) ? 0.f : gv[j];
            xv[j] = gamma[c] * invstd[c] * (dz - sums[c] * inv_count - xh * (sums[C + c] * inv_count));
        }
        if (VEC) {
            f32x4 o = {xv[0], xv[W > 1 ? 1 : 0], xv[W > 2 ? 2 : 0], xv[W > 3 ? 3 : 0]};
            *reinterpret_cast<f32x4*>(dx + p * lddx + c0) = o;
        } else dx[p * lddx + c0] = xv[0];
    }
}
extern "C" int kpx_bn_bwd_f32(const float* dy, int lddy, const float* x, int ldx, size_t P, int C,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                              float* dx, int lddx, float* dgamma, float* dbeta, int accumulate, void* scratch, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !scratch || C <= 0 ||
        ldx < C || lddy < C || lddx < C || act < 0 || act > 1 || P == 0)
        return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    RedArgs a{}; a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.P = P; a.C = C;
    a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.act = act; a.part = (double*)scratch;
    int nb; int rc = launch_chan_reduce(2, a, &nb, s);
    if (rc) return rc;
    // the per-channel sums are parked (as floats) behind the partials in the scratch buffer
    float* sums = reinterpret_cast<float*>((double*)scratch + (size_t)KPX_RED_BLOCKS * 2 * C);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, (const double*)scratch, nb, C, dgamma, dbeta, sums, accumulate);
    rc = kpx_launch_status();
    if (rc) return rc;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) &&
                     ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0;
    const float inv_count = (float)(1.0 / (double)P);
    if (vec) hipLaunchKernelGGL(bn_bwd_apply_strip_kernel, strip_grid(P, C), dim3(256), 0, s, dy, lddy, x, ldx, P, C, mean, invstd, gamma, beta, act, sums, inv_count, dx, lddx);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(P * C)), dim3(256), 0, s, dy, lddy, x, ldx, P, C, mean, invstd, gamma, beta, act, sums, inv_count, dx, lddx);
    return kpx_launch_status();
}

/* kpx_bn_bwd_f32 with the two channel reductions taken from per-tile sums (see bn_bwd_from_tiles_kernel) instead of a pass over (dy, x). */
extern "C" int kpx_bn_bwd_from_tiles_f32(const float* dy, int lddy, const float* x, int ldx, size_t P, int C,
                                         const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                                         float* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                                         const float* tile_stats, size_t tile0, size_t ntiles, void* scratch, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !scratch || !tile_stats || C <= 0 ||
        ldx < C || lddy < C || lddx < C || act != KPX_ACT_RELU || P == 0 || ntiles == 0)
        return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    float* sums = reinterpret_cast<float*>((double*)scratch + (size_t)KPX_RED_BLOCKS * 2 * C);
    hipLaunchKernelGGL(bn_bwd_from_tiles_kernel, dim3(C), dim3(256), 0, s, tile_stats, tile0, ntiles, C, gamma, dgamma, dbeta, sums, accumulate);
    int rc = kpx_launch_status();
    if (rc) return rc;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) &&
                     ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0;
    const float inv_count = (float)(1.0 / (double)P);
    if (vec) hipLaunchKernelGGL(bn_bwd_apply_strip_kernel, strip_grid(P, C), dim3(256), 0, s, dy, lddy, x, ldx, P, C, mean, invstd, gamma, beta, act, sums, inv_count, dx, lddx);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(P * C)), dim3(256), 0, s, dy, lddy, x, ldx, P, C, mean, invstd, gamma, beta, act, sums, inv_count, dx, lddx);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ inference: batch norm folded into the conv
// relu(gamma * (conv(x, w) + b - mean) * rsqrt(var + eps) + beta) = relu(conv(x, w * s) + (b - mean) * s + beta), s = gamma * rsqrt(var + eps):
// with the MOVING statistics (is_training = False: KeypointModel / FinalModel, reference keypoint_model.py:48-50, final_model.py:62,68,95) the
// normalisation is a constant per-channel affine map, folded once per checkpoint into the filter and the bias -- the inference networks then
// run conv + ReLU epilogue and never make the batch-norm pass over the activation.
__global__ __launch_bounds__(256) void bn_fold_conv_kernel(const float* __restrict__ w, const float* __restrict__ bias, size_t rows, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mm,
                                                          const float* __restrict__ mv, float eps, float* __restrict__ w_out, float* __restrict__ b_out) {
    const size_t total = rows * (size_t)C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float sc = gamma[c] * (1.0f / sqrtf(mv[c] + eps));
        w_out[i] = w[i] * sc;
        if (i < (size_t)C) b_out[c] = ((bias ? bias[c] : 0.f) - mm[c]) * sc + beta[c];
    }
}
extern "C" int kpx_bn_fold_conv_f32(const float* w, const float* bias, size_t rows, int C, const float* gamma, const float* beta,
                                    const float* moving_mean, const float* moving_var, float eps, float* w_out, float* b_out, void* stream) {
    if (!w || !gamma || !beta || !moving_mean || !moving_var || !w_out || !b_out || rows == 0 || C <= 0) return KPX_EINVAL;
    hipLaunchKernelGGL(bn_fold_conv_kernel, dim3(grid_for(rows * C)), dim3(256), 0, kpx_stream(stream), w, bias, rows, C, gamma, beta,
                       moving_mean, moving_var, eps, w_out, b_out);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ batched batch norm
// The weight-sharing calls of the path (the two pose_encoder calls of one pair, reference detector_translator_model.py:166-167) run as ONE
// batch with `groups` sets of statistics: consecutive groups of P pixels, mean / invstd [groups][C], moving statistics updated once per
// group IN ORDER, gamma / beta gradients summed over the groups in order.  One launch per phase for all groups (blockIdx.z), one finalize
// workgroup per channel looping over the groups -- the same partial sums in the same order as `groups` separate calls: same bits.
// (up to four groups side by side: segment sg = threadIdx.x / 256 of a 256 x min(groups, 4)-thread workgroup sums group g0 + sg exactly as the
//  single-group kernels do -- thread t of the segment takes partials t, t + 256, ..; shuffle tree per wavefront; the four wave sums in LDS
//  order -- and thread 0 then finishes the groups IN ORDER: the moving statistics / gradient sums see the same numbers in the same order)
__global__ __launch_bounds__(1024) void bn_stats_finalize_groups_kernel(const double* part, size_t part_gstride, int nb, const float* __restrict__ ts, size_t tiles_per_group,
                                                                       int groups, int C, double count, float eps, float* mean, float* invstd,
                                                                       float* mm, float* mv, float decay) {
    const int c = blockIdx.x, sg = threadIdx.x >> 8, t = threadIdx.x & 255, nseg = blockDim.x >> 8;
    __shared__ double sm[4][2][4];
    for (int g0 = 0; g0 < groups; g0 += nseg) {
        const int g = g0 + sg;
        double a0 = 0.0, a1 = 0.0;
        if (g < groups) {
            if (ts) {                                      // per-tile sums from a convolution epilogue (floats)
                const size_t t0 = (size_t)g * tiles_per_group;
                for (size_t b = t; b < tiles_per_group; b += 256) {
                    a0 += (double)ts[((t0 + b) * 2) * C + c];
                    a1 += (double)ts[((t0 + b) * 2 + 1) * C + c];
                }
            } else {                                       // block partials of chan_reduce_kernel<1> (doubles)
                const double* gp = part + (size_t)g * part_gstride;
                for (int b = t; b < nb; b += 256) {
                    a0 += gp[((size_t)b * 2) * C + c];
                    a1 += gp[((size_t)b * 2 + 1) * C + c];
                }
            }
        }
        a0 = kpx_wave_sum_d(a0);
        a1 = kpx_wave_sum_d(a1);
        __syncthreads();                                   // (sm is reused per batch of groups)
        if ((t & 63) == 0) { sm[sg][0][t >> 6] = a0; sm[sg][1][t >> 6] = a1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int j = 0; j < nseg && g0 + j < groups; ++j) {
                const double s = (sm[j][0][0] + sm[j][0][1]) + (sm[j][0][2] + sm[j][0][3]), q = (sm[j][1][0] + sm[j][1][1]) + (sm[j][1][2] + sm[j][1][3]);
                const double m = s / count;
                double v = q / count - m * m;
                if (v < 0.0) v = 0.0;
                const float mf = (float)m, vf = (float)v;
                mean[(g0 + j) * C + c] = mf;
                invstd[(g0 + j) * C + c] = 1.0f / sqrtf(vf + eps);
                if (mm && mv) {
                    const float one_minus = 1.0f - decay;
                    const float unb = (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
                    mm[c] = mm[c] - (mm[c] - mf) * one_minus;
                    mv[c] = mv[c] - (mv[c] - unb) * one_minus;
                }
            }
        }
    }
}
extern "C" int kpx_bn_train_fwd_f32(const float* x, size_t P, int groups, int C, int ldx, const float* tile_stats, size_t tiles_per_group,
                                    float eps, const float* gamma, const float* beta, float* mean, float* invstd,
                                    float* moving_mean, float* moving_var, float decay, float* y, int ldy, int act, void* scratch, void* stream) {
    if (!x || !y || !gamma || !beta || !mean || !invstd || !scratch || groups <= 0 || groups > 65535 || C <= 0 || ldx < C || ldy < C || act < 0 || act > 1 || P == 0)
        return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    const size_t gstride = (size_t)KPX_RED_BLOCKS * 2 * C;
    int nb = 0;
    if (!tile_stats) {
        RedArgs a{}; a.x = x; a.ldx = ldx; a.P = P; a.C = C; a.part = (double*)scratch; a.part_gstride = gstride;
        int rc = launch_chan_reduce(1, a, &nb, s, groups);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(bn_stats_finalize_groups_kernel, dim3(C), dim3(256 * (groups < 4 ? groups : 4)), 0, s, (const double*)scratch, gstride, nb, tile_stats, tiles_per_group,
                       groups, C, (double)P, eps, mean, invstd, moving_mean, moving_var, decay);
    int rc = kpx_launch_status();
    if (rc) return rc;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    if (vec) {
        hipLaunchKernelGGL(bn_apply_strip_kernel, strip_grid(P, C, groups), dim3(256), 0, s, x, P, C, ldx, mean, invstd, gamma, beta, y, ldy, act);
        return kpx_launch_status();
    }
    for (int g = 0; g < groups; ++g) {
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for(P * C)), dim3(256), 0, s, x + (size_t)g * P * ldx, P, C, ldx, mean + g * C, invstd + g * C,
                           gamma, beta, y + (size_t)g * P * ldy, ldy, act);
        if ((rc = kpx_launch_status())) return rc;
    }
    return 0;
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_groups_kernel(const double* part, size_t part_gstride, int nb, const float* __restrict__ ts, size_t tiles_per_group,
                                                                     const float* __restrict__ gamma, int groups, int C,
                                                                     float* dgamma, float* dbeta, float* sums, int accumulate) {
    const int c = blockIdx.x, sg = threadIdx.x >> 8, t = threadIdx.x & 255, nseg = blockDim.x >> 8;
    __shared__ double sm[4][2][4];
    for (int g0 = 0; g0 < groups; g0 += nseg) {
        const int g = g0 + sg;
        double a0 = 0.0, a1 = 0.0;
        if (g < groups) {
            if (ts) {                                      // per-tile sums from the data-gradient epilogue that produced dy: sum(dz), sum(dz * (y - beta))
                const size_t t0 = (size_t)g * tiles_per_group;
                for (size_t b = t; b < tiles_per_group; b += 256) {
                    a0 += (double)ts[((t0 + b) * 2) * C + c];
                    a1 += (double)ts[((t0 + b) * 2 + 1) * C + c];
                }
            } else {
                const double* gp = part + (size_t)g * part_gstride;
                for (int b = t; b < nb; b += 256) {
                    a0 += gp[((size_t)b * 2) * C + c];
                    a1 += gp[((size_t)b * 2 + 1) * C + c];
                }
            }
        }
        a0 = kpx_wave_sum_d(a0);
        a1 = kpx_wave_sum_d(a1);
        __syncthreads();
        if ((t & 63) == 0) { sm[sg][0][t >> 6] = a0; sm[sg][1][t >> 6] = a1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int j = 0; j < nseg && g0 + j < groups; ++j) {
                const double s = (sm[j][0][0] + sm[j][0][1]) + (sm[j][0][2] + sm[j][0][3]);
                double q = (sm[j][1][0] + sm[j][1][1]) + (sm[j][1][2] + sm[j][1][3]);
                if (ts) { const double ga = (double)gamma[c]; q = ga != 0.0 ? q / ga : 0.0; }      // x_hat = (y - beta) / gamma wherever dz != 0
                if (accumulate || g0 + j > 0) { dbeta[c] += (float)s; dgamma[c] += (float)q; } else { dbeta[c] = (float)s; dgamma[c] = (float)q; }
                sums[(size_t)(g0 + j) * 2 * C + c] = (float)s; sums[(size_t)(g0 + j) * 2 * C + C + c] = (float)q;
            }
        }
    }
}
extern "C" int kpx_bn_train_bwd_f32(const float* dy, int lddy, const float* x, int ldx, size_t P, int groups, int C,
                                    const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                                    float* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                                    const float* tile_stats, size_t tiles_per_group, void* scratch, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !scratch || groups <= 0 || groups > 65535 || C <= 0 ||
        ldx < C || lddy < C || lddx < C || act < 0 || act > 1 || P == 0 || (tile_stats && (act != KPX_ACT_RELU || tiles_per_group == 0)))
        return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    const size_t gstride = (size_t)KPX_RED_BLOCKS * 2 * C;
    int nb = 0, rc = 0;
    if (!tile_stats) {                                   // (with tile_stats the data-gradient epilogue that produced dy already reduced the two sums)
        RedArgs a{}; a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.P = P; a.C = C;
        a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.act = act; a.part = (double*)scratch; a.part_gstride = gstride;
        rc = launch_chan_reduce(2, a, &nb, s, groups);
        if (rc) return rc;
    }
    float* sums = reinterpret_cast<float*>((double*)scratch + (size_t)groups * gstride);          // [groups][2][C] behind the partials
    hipLaunchKernelGGL(bn_bwd_finalize_groups_kernel, dim3(C), dim3(256 * (groups < 4 ? groups : 4)), 0, s, (const double*)scratch, gstride, nb, tile_stats, tiles_per_group,
                       gamma, groups, C, dgamma, dbeta, sums, accumulate);
    if ((rc = kpx_launch_status())) return rc;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0;
    const float inv_count = (float)(1.0 / (double)P);
    if (vec) {
        hipLaunchKernelGGL(bn_bwd_apply_strip_kernel, strip_grid(P, C, groups), dim3(256), 0, s, dy, lddy, x, ldx, P, C, mean, invstd, gamma, beta, act, sums, inv_count, dx, lddx);
        return kpx_launch_status();
    }
    for (int g = 0; g < groups; ++g) {
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(P * C)), dim3(256), 0, s, dy + (size_t)g * P * lddy, lddy, x + (size_t)g * P * ldx, ldx, P, C,
                           mean + g * C, invstd + g * C, gamma, beta, act, sums + (size_t)g * 2 * C, inv_count, dx + (size_t)g * P * lddx, lddx);
        if ((rc = kpx_launch_status())) return rc;
    }
    return 0;
}
// scratch of the two entries above: per group the partials of kpx_chan_reduce_scratch_bytes + 2 C floats
extern "C" size_t kpx_bn_train_scratch_bytes(int C, int groups) {
    const size_t c = (size_t)(C > 0 ? C : 1), g = (size_t)(groups > 0 ? groups : 1);
    return g * ((size_t)KPX_RED_BLOCKS * 2 * c * sizeof(double) + ((2 * c * sizeof(float) + 15) & ~(size_t)15));
}

// ------------------------------------------------------------------------------------------ bilinear x2 (legacy TF sampling)
template <bool VEC>
__global__ __launch_bounds__(256) void resize2x_fwd_kernel(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy) {
    constexpr int V = VEC ? 4 : 1;
    const int G = C / V, W2 = 2 * W, H2 = 2 * H;
    const size_t total = (size_t)N * H2 * W2 * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % G) * V;
        size_t p = i / G;
        const int ox = (int)(p % W2); p /= W2;
        const int oy = (int)(p % H2);
        const int n = (int)(p / H2);
        const int iy = oy >> 1, ix = ox >> 1;
        const int iy1 = min(iy + 1, H - 1), ix1 = min(ix + 1, W - 1);
        const float ty = (oy & 1) ? 0.5f : 0.f, tx = (ox & 1) ? 0.5f : 0.f;
        const float* r0 = x + ((size_t)(n * H + iy) * W) * ldx + c;
        const float* r1 = x + ((size_t)(n * H + iy1) * W) * ldx + c;
        float* o = y + ((size_t)(n * H2 + oy) * W2 + ox) * ldy + c;
        if (VEC) {
            const f32x4 tl = *reinterpret_cast<const f32x4*>(r0 + (size_t)ix * ldx), tr = *reinterpret_cast<const f32x4*>(r0 + (size_t)ix1 * ldx);
            const f32x4 bl = *reinterpret_cast<const f32x4*>(r1 + (size_t)ix * ldx), br = *reinterpret_cast<const f32x4*>(r1 + (size_t)ix1 * ldx);
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float top = tl[j] + (tr[j] - tl[j]) * tx, bot = bl[j] + (br[j] - bl[j]) * tx;
                r[j] = top + (bot - top) * ty;
            }
            *reinterpret_cast<f32x4*>(o) = r;
        } else {
            const float tl = r0[(size_t)ix * ldx], tr = r0[(size_t)ix1 * ldx], bl = r1[(size_t)ix * ldx], br = r1[(size_t)ix1 * ldx];
            const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
            *o = top + (bot - top) * ty;
        }
    }
}
// One thread per INPUT pixel and channel quad: its four neighbours (loaded once) make the 2 x 2 output pixels it owns -- the same expressions
// per output as resize2x_fwd_kernel (same bits), a quarter of the loads and of the index arithmetic, 32-bit indices.  (The per-output-pixel
// kernel with its 64-bit divisions moved 2.4 TB/s on the rollout's [256,32,32,256] -> [256,64,64,256] launches: 10 % of a run.)
__global__ __launch_bounds__(256) void resize2x_fwd_quad_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx, float* __restrict__ y, int ldy) {
    const unsigned G = (unsigned)C >> 2, total = (unsigned)N * H * W * G;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cq = i % G; unsigned p = i / G;
        const unsigned ix = p % (unsigned)W; p /= (unsigned)W;
        const unsigned iy = p % (unsigned)H, n = p / (unsigned)H;
        const unsigned iy1 = iy + 1 < (unsigned)H ? iy + 1 : iy, ix1 = ix + 1 < (unsigned)W ? ix + 1 : ix;
        const float* r0 = x + ((size_t)(n * H + iy) * W) * ldx + cq * 4;
        const float* r1 = x + ((size_t)(n * H + iy1) * W) * ldx + cq * 4;
        const f32x4 tl = *reinterpret_cast<const f32x4*>(r0 + (size_t)ix * ldx), tr = *reinterpret_cast<const f32x4*>(r0 + (size_t)ix1 * ldx);
        const f32x4 bl = *reinterpret_cast<const f32x4*>(r1 + (size_t)ix * ldx), br = *reinterpret_cast<const f32x4*>(r1 + (size_t)ix1 * ldx);
        float* o = y + ((size_t)(n * 2 * H + 2 * iy) * (2 * W) + 2 * ix) * ldy + cq * 4;
        const size_t rs = (size_t)2 * W * ldy;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float ty = a ? 0.5f : 0.f, tx = b ? 0.5f : 0.f;
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float top = tl[j] + (tr[j] - tl[j]) * tx, bot = bl[j] + (br[j] - bl[j]) * tx;
                    r[j] = top + (bot - top) * ty;
                }
                *reinterpret_cast<f32x4*>(o + a * rs + (size_t)b * ldy) = r;
            }
    }
}
extern "C" int kpx_resize2x_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || ldx < C || ldy < C) return KPX_EINVAL;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    if (vec && (size_t)N * H * W * (C / 4) < 0x7fffffffu) {
        const size_t quads = (size_t)N * H * W * (C / 4);
        size_t nb = (quads + 255) / 256; if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL(resize2x_fwd_quad_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), x, N, H, W, C, ldx, y, ldy);
        return kpx_launch_status();
    }
    const size_t items = (size_t)N * 4 * H * W * (vec ? C / 4 : C);
    if (vec) hipLaunchKernelGGL(resize2x_fwd_kernel<true>, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), x, N, H, W, C, ldx, y, ldy);
    else hipLaunchKernelGGL(resize2x_fwd_kernel<false>, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), x, N, H, W, C, ldx, y, ldy);
    return kpx_launch_status();
}

// dx[i] gathers out rows {2i-1 (w .5, i>=1), 2i (w 1), 2i+1 (w .5, or 1 at the clamped last row)} x the same in x.
template <bool VEC>
__global__ __launch_bounds__(256) void resize2x_bwd_kernel(const float* dy, int N, int H, int W, int C, int lddy, float* dx, int lddx) {
    constexpr int V = VEC ? 4 : 1;
    const int G = C / V, W2 = 2 * W, H2 = 2 * H;
    const size_t total = (size_t)N * H * W * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % G) * V;
        size_t p = i / G;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H);
        const int n = (int)(p / H);
        float wy[3], wx[3];
        wy[0] = iy >= 1 ? 0.5f : 0.f; wy[1] = 1.f; wy[2] = iy == H - 1 ? 1.f : 0.5f;
        wx[0] = ix >= 1 ? 0.5f : 0.f; wx[1] = 1.f; wx[2] = ix == W - 1 ? 1.f : 0.5f;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int oy = 2 * iy - 1 + a;
            if (oy < 0) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int ox = 2 * ix - 1 + b;
                if (ox < 0) continue;
                const float wgt = wy[a] * wx[b];
                const float* q = dy + ((size_t)(n * H2 + oy) * W2 + ox) * lddy + c;
                if (VEC) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] = fmaf(wgt, v[j], acc[j]);
                } else acc[0] = fmaf(wgt, *q, acc[0]);
            }
        }
        float* o = dx + ((size_t)(n * H + iy) * W + ix) * lddx + c;
        if (VEC) {
            f32x4 r = {acc[0], acc[V > 1 ? 1 : 0], acc[V > 2 ? 2 : 0], acc[V > 3 ? 3 : 0]};
            *reinterpret_cast<f32x4*>(o) = r;
        } else *o = acc[0];
    }
}
extern "C" int kpx_resize2x_bwd_f32(const float* dy, int N, int H, int W, int C, int lddy, float* dx, int lddx, void* stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || lddy < C || lddx < C) return KPX_EINVAL;
    const bool vec = (C % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0;
    const size_t items = (size_t)N * H * W * (vec ? C / 4 : C);
    if (vec) hipLaunchKernelGGL(resize2x_bwd_kernel<true>, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), dy, N, H, W, C, lddy, dx, lddx);
    else hipLaunchKernelGGL(resize2x_bwd_kernel<false>, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), dy, N, H, W, C, lddy, dx, lddx);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ translator heads + blend
__global__ __launch_bounds__(256) void head_blend_fwd_kernel(const float* im, const float* raw4, size_t P, float* fin, float* crude, float* mask) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        const f32x4 r = reinterpret_cast<const f32x4*>(raw4)[p];
        const float m = 1.0f / (1.0f + expf(-r[3]));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            fin[p * 3 + j] = im[p * 3 + j] * m + r[j] * (1.0f - m);
            if (crude) crude[p * 3 + j] = r[j];
        }
        if (mask) mask[p] = m;
    }
}
extern "C" int kpx_head_blend_fwd_f32(const float* im, const float* raw4, size_t P, float* final_out, float* crude_out, float* mask_out, void* stream) {
    if (!im || !raw4 || !final_out || (((uintptr_t)raw4) & 15)) return KPX_EINVAL;
    if (P == 0) return 0;
    hipLaunchKernelGGL(head_blend_fwd_kernel, dim3(grid_for(P)), dim3(256), 0, kpx_stream(stream), im, raw4, P, final_out, crude_out, mask_out);
    return kpx_launch_status();
}
__global__ __launch_bounds__(256) void head_blend_bwd_kernel(const float* dfin, const float* im, const float* raw4, size_t P, float* draw4) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        const f32x4 r = reinterpret_cast<const f32x4*>(raw4)[p];
        const float m = 1.0f / (1.0f + expf(-r[3]));
        f32x4 d;
        float dm = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float g = dfin[p * 3 + j];
            d[j] = g * (1.0f - m);
            dm = fmaf(g, im[p * 3 + j] - r[j], dm);
        }
        d[3] = dm * m * (1.0f - m);
        reinterpret_cast<f32x4*>(draw4)[p] = d;
    }
}
extern "C" int kpx_head_blend_bwd_f32(const float* dfinal, const float* im, const float* raw4, size_t P, float* draw4, void* stream) {
    if (!dfinal || !im || !raw4 || !draw4 || ((((uintptr_t)raw4) | ((uintptr_t)draw4)) & 15)) return KPX_EINVAL;
    if (P == 0) return 0;
    hipLaunchKernelGGL(head_blend_bwd_kernel, dim3(grid_for(P)), dim3(256), 0, kpx_stream(stream), dfinal, im, raw4, P, draw4);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ VGG input transform
__global__ __launch_bounds__(256) void vgg_prep_fwd_kernel(const float* rgb, size_t P, float* bgr) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        const float r = (rgb[p * 3 + 0] + 1.0f) / 2.0f * 255.0f;
        const float g = (rgb[p * 3 + 1] + 1.0f) / 2.0f * 255.0f;
        const float b = (rgb[p * 3 + 2] + 1.0f) / 2.0f * 255.0f;
        bgr[p * 3 + 0] = b - 103.939f; bgr[p * 3 + 1] = g - 116.779f; bgr[p * 3 + 2] = r - 123.68f;
    }
}
extern "C" int kpx_vgg_prep_fwd_f32(const float* rgb, size_t P, float* bgr, void* stream) {
    if (!rgb || !bgr) return KPX_EINVAL;
    if (P == 0) return 0;
    hipLaunchKernelGGL(vgg_prep_fwd_kernel, dim3(grid_for(P)), dim3(256), 0, kpx_stream(stream), rgb, P, bgr);
    return kpx_launch_status();
}
__global__ __launch_bounds__(256) void vgg_prep_bwd_kernel(const float* dbgr, size_t P, float* drgb) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        drgb[p * 3 + 0] = dbgr[p * 3 + 2] * 127.5f;
        drgb[p * 3 + 1] = dbgr[p * 3 + 1] * 127.5f;
        drgb[p * 3 + 2] = dbgr[p * 3 + 0] * 127.5f;
    }
}
extern "C" int kpx_vgg_prep_bwd_f32(const float* dbgr, size_t P, float* drgb, void* stream) {
    if (!dbgr || !drgb) return KPX_EINVAL;
    if (P == 0) return 0;
    hipLaunchKernelGGL(vgg_prep_bwd_kernel, dim3(grid_for(P)), dim3(256), 0, kpx_stream(stream), dbgr, P, drgb);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ max-pool 2x2 s2 SAME
template <bool VEC>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* x, int N, int H, int W, int C, float* y) {
    constexpr int V = VEC ? 4 : 1;
    const int G = C / V, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % G) * V;
        size_t p = i / G;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float m[V];
#pragma unroll
        for (int j = 0; j < V; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int iy = 2 * oy + a, ix = 2 * ox + b;
                if (iy >= H || ix >= W) continue;
                const float* q = x + ((size_t)(n * H + iy) * W + ix) * C + c;
                if (VEC) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
                    for (int j = 0; j < V; ++j) m[j] = fmaxf(m[j], v[j]);
                } else m[0] = fmaxf(m[0], *q);
            }
        float* o = y + ((size_t)(n * Ho + oy) * Wo + ox) * C + c;
        if (VEC) { f32x4 r = {m[0], m[V > 1 ? 1 : 0], m[V > 2 ? 2 : 0], m[V > 3 ? 3 : 0]}; *reinterpret_cast<f32x4*>(o) = r; }
        else *o = m[0];
    }
}
extern "C" int kpx_maxpool2_fwd_f32(const float* x, int N, int H, int W, int C, float* y, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return KPX_EINVAL;
    const bool vec = (C % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    const size_t items = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (vec ? C / 4 : C);
    if (vec) hipLaunchKernelGGL(maxpool2_fwd_kernel<true>, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), x, N, H, W, C, y);
    else hipLaunchKernelGGL(maxpool2_fwd_kernel<false>, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), x, N, H, W, C, y);
    return kpx_launch_status();
}
// gradient goes to the first maximum of each window in row-major scan order (TF / Eigen argmax rule)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* dy, const float* x, int N, int H, int W, int C, float* dx) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t total = (size_t)N * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t p = i / C;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float best = -INFINITY; int bi = 0;
        float v[4]; bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
            ok[k] = iy < H && ix < W;
            v[k] = ok[k] ? x[((size_t)(n * H + iy) * W + ix) * C + c] : -INFINITY;
            if (ok[k] && v[k] > best) { best = v[k]; bi = k; }
        }
        const float g = dy[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
            if (ok[k]) dx[((size_t)(n * H + iy) * W + ix) * C + c] = (k == bi) ? g : 0.f;
        }
    }
}
// Gradient arriving at a VGG19 feature tensor y = relu(conv) (reference models/networks/vgg.py:43,45-55; detector_translator_model.py:274-289),
// in ONE pass instead of four (max-pool backward, L1 backward, their sum, ReLU backward):
//   d[pix][c] = [y_pred > 0] * ( (pix is the first maximum of its 2x2 window ? dy_pooled : 0) + g * sign(y_pred - y_gt) )
// f = [gt half ; pred half] of the feature ([2B,H,W,C]); dy_pooled = gradient w.r.t. the pooled tensor [B,Ho,Wo,C] or NULL for the last
// feature (no pool behind it); g = gscale_host * *gscale_dev.  Four channels per thread.
__global__ __launch_bounds__(256) void vgg_feat_bwd_kernel(const float* __restrict__ f, size_t half, const float* gdev, float ghost,
                                                           const float* __restrict__ dyp, int B, int H, int W, int C, float* __restrict__ d) {
    const float g = ghost * (gdev ? *gdev : 1.0f);
    const int C4 = C >> 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const float* fp = f + half;                                    // pred half
    if (!dyp) {
        const size_t total = half >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const f32x4 yp = *reinterpret_cast<const f32x4*>(fp + 4 * i), yg = *reinterpret_cast<const f32x4*>(f + 4 * i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float df = yg[e] - yp[e]; const float l1 = df > 0.f ? -g : (df < 0.f ? g : 0.f); o[e] = yp[e] > 0.f ? l1 : 0.f; }
            *reinterpret_cast<f32x4*>(d + 4 * i) = o;
        }
        return;
    }
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        size_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        f32x4 v[4], gt[4]; bool ok[4]; size_t off[4];
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
            ok[k] = iy < H && ix < W;
            off[k] = ((size_t)(n * H + iy) * W + ix) * C + c;
            if (ok[k]) { v[k] = *reinterpret_cast<const f32x4*>(fp + off[k]); gt[k] = *reinterpret_cast<const f32x4*>(f + off[k]); }
            else { v[k] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; gt[k] = v[k]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) if (ok[k] && v[k][e] > best[e]) { best[e] = v[k][e]; bi[e] = k; }      // first maximum in scan order
        }
        const f32x4 gp = *reinterpret_cast<const f32x4*>(dyp + (((size_t)(n * Ho + oy) * Wo + ox) * C + c));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float df = gt[k][e] - v[k][e];
                const float l1 = df > 0.f ? -g : (df < 0.f ? g : 0.f);
                o[e] = v[k][e] > 0.f ? ((bi[e] == k ? gp[e] : 0.f) + l1) : 0.f;
            }
            *reinterpret_cast<f32x4*>(d + off[k]) = o;
        }
    }
}
extern "C" int kpx_vgg_feat_bwd_f32(const float* f, size_t half, const float* gscale_dev, float gscale_host, const float* dy_pooled,
                                    int B, int H, int W, int C, float* d, void* stream) {
    if (!f || !d || half == 0 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || half != (size_t)B * H * W * C ||
        ((((uintptr_t)f) | ((uintptr_t)d) | ((uintptr_t)dy_pooled)) & 15))
        return KPX_EINVAL;
    const size_t items = dy_pooled ? (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4) : half / 4;
    hipLaunchKernelGGL(vgg_feat_bwd_kernel, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), f, half, gscale_dev, gscale_host, dy_pooled, B, H, W, C, d);
    return kpx_launch_status();
}
extern "C" int kpx_maxpool2_bwd_f32(const float* dy, const float* x, int N, int H, int W, int C, float* dx, void* stream) {
    if (!dy || !x || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return KPX_EINVAL;
    const size_t items = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), dy, x, N, H, W, C, dx);
    return kpx_launch_status();
}

// ================================================================================================ bf16 STORAGE variants (BASELINE configs[2])
// The same streaming kernels for bf16 tensors in HBM (layers.py:13-14 batch norm, networks/__init__.py:63,98 resize, vgg.py:25-45 pools,
// detector_translator_model.py:274-289 feature L1): 16-B accesses = EIGHT channels per lane, arithmetic in fp32, per-channel reductions in
// fp64 through the same block partials and fixed-order finalize kernels as the fp32 entries (statistics, moving averages, gamma / beta
// gradients stay fp32).  Every entry wants C, the pixel strides multiples of 8 and 16-byte aligned pointers (KPX_EINVAL otherwise) unless
// it says otherwise.
typedef unsigned int pw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

__device__ __forceinline__ float pw_bf(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ unsigned short pw_to_bf(float v) { const __bf16 b = (__bf16)v; return __builtin_bit_cast(unsigned short, b); }
__device__ __forceinline__ void pw_ld8(const bf16_t* p, float* v) {
    const pw_u32x4 r = *reinterpret_cast<const pw_u32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __builtin_bit_cast(float, r[j] << 16); v[2 * j + 1] = __builtin_bit_cast(float, r[j] & 0xffff0000u); }
}
__device__ __forceinline__ unsigned pw_pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void pw_st8(bf16_t* p, const float* v) {
    const pw_u32x4 r = {pw_pack2(v[0], v[1]), pw_pack2(v[2], v[3]), pw_pack2(v[4], v[5]), pw_pack2(v[6], v[7])};
    *reinterpret_cast<pw_u32x4*>(p) = r;
}
static inline bool pw_al16(const void* a, const void* b = nullptr, const void* c = nullptr) { return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0; }

// ---- casts / copies between channel slices: dst[p][0:C] = src[p][0:C] (any C; vector path when everything is 8-aligned)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_channels_kernel(const TS* __restrict__ src, int lds_, TD* __restrict__ dst, int ldd, size_t P, int C, int vec) {
    if (vec) {
        const int G = C >> 3;
        const size_t total = P * (size_t)G;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t p = i / G; const int c = (int)(i - p * G) * 8;
            float v[8];
            if (sizeof(TS) == 2) pw_ld8(reinterpret_cast<const bf16_t*>(src) + p * lds_ + c, v);
            else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + p * lds_ + c), b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + p * lds_ + c + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
            }
            if (sizeof(TD) == 2) pw_st8(reinterpret_cast<bf16_t*>(dst) + p * ldd + c, v);
            else {
                float* o = reinterpret_cast<float*>(dst) + p * ldd + c;
                *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        }
        return;
    }
    const size_t total = P * (size_t)C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / C; const int c = (int)(i - p * C);
        float v;
        if (sizeof(TS) == 2) v = pw_bf(reinterpret_cast<const bf16_t*>(src)[p * lds_ + c]); else v = reinterpret_cast<const float*>(src)[p * lds_ + c];
        if (sizeof(TD) == 2) reinterpret_cast<bf16_t*>(dst)[p * ldd + c] = pw_to_bf(v); else reinterpret_cast<float*>(dst)[p * ldd + c] = v;
    }
}
// few fp32 channels (C <= 8) into a contiguous 8-channel bf16 tensor, the channels C .. 7 written as zeros: one 16-byte store per pixel
__global__ __launch_bounds__(256) void cast_pad8_kernel(const float* __restrict__ src, int lds_, bf16_t* __restrict__ dst, size_t P, int C) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = j < C ? src[p * lds_ + j] : 0.f;
        pw_st8(dst + p * 8, v);
    }
}
// kind: 0 = f32 -> bf16, 1 = bf16 -> f32, 2 = bf16 -> bf16 (channel-slice copy: tf.concat),
//       3 = f32 -> bf16 with the destination's channels C .. 7 zeroed (C <= 8, lddst = 8: the 4-channel gradient of the translator's head as an
//           operand of the bf16 3x3 kernels, which gather whole 8-channel groups)
extern "C" int kpx_cast_channels(const void* src, int ldsrc, void* dst, int lddst, size_t P, int C, int kind, void* stream) {
    if (!src || !dst || C <= 0 || ldsrc < C || lddst < C || kind < 0 || kind > 3) return KPX_EINVAL;
    if (P == 0) return 0;
    if (kind == 3) {
        if (C > 8 || lddst != 8 || (((uintptr_t)dst) & 15)) return KPX_EINVAL;
        hipLaunchKernelGGL(cast_pad8_kernel, dim3(grid_for(P)), dim3(256), 0, kpx_stream(stream), (const float*)src, ldsrc, (bf16_t*)dst, P, C);
        return kpx_launch_status();
    }
    const int sa = kind == 0 ? 4 : 2, da = kind == 1 ? 4 : 2;
    const int vec = (C % 8 == 0) && (((size_t)ldsrc * sa) % 16 == 0) && (((size_t)lddst * da) % 16 == 0) && pw_al16(src, dst);
    const unsigned nb = grid_for(vec ? P * (C / 8) : P * C);
    hipStream_t s = kpx_stream(stream);
    if (kind == 0) hipLaunchKernelGGL((cast_channels_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, s, (const float*)src, ldsrc, (bf16_t*)dst, lddst, P, C, vec);
    else if (kind == 1) hipLaunchKernelGGL((cast_channels_kernel<bf16_t, float>), dim3(nb), dim3(256), 0, s, (const bf16_t*)src, ldsrc, (float*)dst, lddst, P, C, vec);
    else hipLaunchKernelGGL((cast_channels_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)src, ldsrc, (bf16_t*)dst, lddst, P, C, vec);
    return kpx_launch_status();
}

// ---- per-channel reductions over bf16 tensors (modes as chan_reduce_kernel); DYF32: the gradient operand of mode 2 is fp32
template <int MODE, bool DYF32>
__global__ __launch_bounds__(256) void chan_reduce_bf16_kernel(const RedArgs a) {
    const int G = a.C >> 3, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    const int t = threadIdx.x, cgl = t % Gb, prow = t / Gb;
    const int cg = blockIdx.y * Gb + cgl;
    const bool active = prow < PPB && cg < G;
    const int grp = blockIdx.z;
    const bf16_t* const gx = reinterpret_cast<const bf16_t*>(a.x) + (size_t)grp * a.P * a.ldx;
    const bf16_t* const gdy = MODE == 2 && !DYF32 ? reinterpret_cast<const bf16_t*>(a.dy) + (size_t)grp * a.P * a.lddy : nullptr;
    const float* const gdf = MODE == 2 && DYF32 ? a.dy + (size_t)grp * a.P * a.lddy : nullptr;
    double s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.0; s1[j] = 0.0; }
    float mu[8], is[8], ga[8], be[8];
    if (MODE == 2 && active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = cg * 8 + j; mu[j] = a.mean[grp * a.C + c]; is[j] = a.invstd[grp * a.C + c]; ga[j] = a.gamma[c]; be[j] = a.beta[c]; }
    }
    if (active) {
        const size_t step = (size_t)gridDim.x * PPB;
        for (size_t p0 = (size_t)blockIdx.x * PPB + prow; p0 < a.P; p0 += step * 2) {
            float xv[2][8], gv[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const size_t p = p0 + u * step;
                if (p < a.P) {
                    pw_ld8(gx + p * a.ldx + cg * 8, xv[u]);
                    if (MODE == 2) {
                        if (DYF32) {
                            const f32x4 d0 = *reinterpret_cast<const f32x4*>(gdf + p * a.lddy + cg * 8), d1 = *reinterpret_cast<const f32x4*>(gdf + p * a.lddy + cg * 8 + 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { gv[u][j] = d0[j]; gv[u][4 + j] = d1[j]; }
                        } else pw_ld8(gdy + p * a.lddy + cg * 8, gv[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (p0 + u * step >= a.P) break;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (MODE == 0) s0[j] += (double)xv[u][j];
                    if (MODE == 1) { s0[j] += (double)xv[u][j]; s1[j] += (double)xv[u][j] * (double)xv[u][j]; }
                    if (MODE == 2) {
                        const float xh = (xv[u][j] - mu[j]) * is[j];
                        const float yv = fmaf(xh, ga[j], be[j]);
                        const float dz = (a.act == KPX_ACT_RELU && !(yv > 0.f)) ? 0.f : gv[u][j];
                        s0[j] += (double)dz; s1[j] += (double)dz * (double)xh;
                    }
                }
            }
        }
    }
    __shared__ double sm[2][256 * 8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sm[0][t * 8 + j] = s0[j]; sm[1][t * 8 + j] = s1[j]; }
    __syncthreads();
    if (t < Gb && blockIdx.y * Gb + t < G) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double r0 = 0.0, r1 = 0.0;
            for (int pr = 0; pr < PPB; ++pr) { r0 += sm[0][(pr * Gb + t) * 8 + j]; r1 += sm[1][(pr * Gb + t) * 8 + j]; }
            const int c = (blockIdx.y * Gb + t) * 8 + j;
            double* const gp = a.part + (size_t)grp * a.part_gstride;
            gp[((size_t)blockIdx.x * 2 + 0) * a.C + c] = r0;
            gp[((size_t)blockIdx.x * 2 + 1) * a.C + c] = r1;
        }
    }
}
static int launch_chan_reduce_bf16(int mode, bool dyf32, RedArgs a, int* nb_out, hipStream_t s, int groups = 1) {
    const int G = a.C / 8, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    size_t nb = a.P / ((size_t)PPB * 8);
    if (nb < 1) nb = 1;
    if (nb > KPX_RED_BLOCKS) nb = KPX_RED_BLOCKS;
    *nb_out = (int)nb;
    const dim3 grid((unsigned)nb, (unsigned)((G + Gb - 1) / Gb), (unsigned)groups), block(256);
    if (mode == 0) hipLaunchKernelGGL((chan_reduce_bf16_kernel<0, false>), grid, block, 0, s, a);
    else if (mode == 1) hipLaunchKernelGGL((chan_reduce_bf16_kernel<1, false>), grid, block, 0, s, a);
    else if (dyf32) hipLaunchKernelGGL((chan_reduce_bf16_kernel<2, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((chan_reduce_bf16_kernel<2, false>), grid, block, 0, s, a);
    return kpx_launch_status();
}
// bias gradient of a bf16 gradient tensor: sum over pixels, fp32 out (same scratch as kpx_chan_sum_f32)
extern "C" int kpx_chan_sum_bf16(const void* x, size_t P, int C, int ldx, float* sum_out, void* scratch, void* stream) {
    if (!x || !sum_out || !scratch || C <= 0 || C % 8 || ldx % 8 || ldx < C || P == 0 || !pw_al16(x)) return KPX_EINVAL;
    RedArgs a{}; a.x = (const float*)x; a.ldx = ldx; a.P = P; a.C = C; a.part = (double*)scratch;
    int nb; int rc = launch_chan_reduce_bf16(0, false, a, &nb, kpx_stream(stream));
    if (rc) return rc;
    hipLaunchKernelGGL(chan_sum_finalize_kernel, dim3(C), dim3(256), 0, kpx_stream(stream), (const double*)scratch, nb, C, sum_out);
    return kpx_launch_status();
}

// ---- batch norm, train mode, all weight-sharing groups per launch (as kpx_bn_train_fwd_f32 / _bwd_f32)
template <bool OUTF32>
__global__ __launch_bounds__(256) void bn_apply_strip_bf16_kernel(const bf16_t* __restrict__ x, size_t P, int C, int ldx, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  void* __restrict__ yv, int ldy, int act) {
    const int G = C >> 3, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    const int t = threadIdx.x, cgl = t % Gb, prow = t / Gb;
    const int cg = blockIdx.y * Gb + cgl;
    if (prow >= PPB || cg >= G) return;
    x += (size_t)blockIdx.z * P * ldx; mean += blockIdx.z * C; invstd += blockIdx.z * C;
    float sc[8], sh[8];                                   // y = x * sc + sh with sc = invstd * gamma, sh = beta - mean * sc: the fp32 kernel's value up to one rounding
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = cg * 8 + j; sc[j] = invstd[c] * gamma[c]; sh[j] = fmaf(-mean[c], sc[j], beta[c]); }
    const size_t step = (size_t)gridDim.x * PPB;
    for (size_t p0 = (size_t)blockIdx.x * PPB + prow; p0 < P; p0 += step * 4) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t p = p0 + u * step; if (p < P) pw_ld8(x + p * ldx + cg * 8, v[u]); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = p0 + u * step;
            if (p >= P) break;
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { r[j] = fmaf(v[u][j], sc[j], sh[j]); if (act == KPX_ACT_RELU) r[j] = fmaxf(r[j], 0.f); }
            if (OUTF32) {
                float* o = reinterpret_cast<float*>(yv) + ((size_t)blockIdx.z * P + p) * ldy + cg * 8;
                *reinterpret_cast<f32x4*>(o) = f32x4{r[0], r[1], r[2], r[3]}; *reinterpret_cast<f32x4*>(o + 4) = f32x4{r[4], r[5], r[6], r[7]};
            } else pw_st8(reinterpret_cast<bf16_t*>(yv) + ((size_t)blockIdx.z * P + p) * ldy + cg * 8, r);
        }
    }
}
static inline dim3 strip_grid8(size_t P, int C, int groups) {
    const int G = C >> 3, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    size_t nb = (P + (size_t)PPB * 4 - 1) / ((size_t)PPB * 4);
    if (nb < 1) nb = 1;
    if (nb > KPX_MAX_BLOCKS) nb = KPX_MAX_BLOCKS;
    return dim3((unsigned)nb, (unsigned)((G + Gb - 1) / Gb), (unsigned)groups);
}
// x bf16 [groups*P, C]; y bf16 (y_f32 = 0) or fp32 (y_f32 = 1); tile_stats as kpx_bn_train_fwd_f32 (kpx_conv3x3_bf16s writes them)
extern "C" int kpx_bn_train_fwd_bf16(const void* x, size_t P, int groups, int C, int ldx, const float* tile_stats, size_t tiles_per_group,
                                     float eps, const float* gamma, const float* beta, float* mean, float* invstd,
                                     float* moving_mean, float* moving_var, float decay, void* y, int ldy, int y_f32, int act, void* scratch, void* stream) {
    if (!x || !y || !gamma || !beta || !mean || !invstd || !scratch || groups <= 0 || groups > 65535 || C <= 0 || C % 8 || ldx % 8 || ldy % 8 || ldx < C || ldy < C ||
        act < 0 || act > 1 || P == 0 || !pw_al16(x, y))
        return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    const size_t gstride = (size_t)KPX_RED_BLOCKS * 2 * C;
    int nb = 0;
    if (!tile_stats) {
        RedArgs a{}; a.x = (const float*)x; a.ldx = ldx; a.P = P; a.C = C; a.part = (double*)scratch; a.part_gstride = gstride;
        int rc = launch_chan_reduce_bf16(1, false, a, &nb, s, groups);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(bn_stats_finalize_groups_kernel, dim3(C), dim3(256 * (groups < 4 ? groups : 4)), 0, s, (const double*)scratch, gstride, nb, tile_stats, tiles_per_group,
                       groups, C, (double)P, eps, mean, invstd, moving_mean, moving_var, decay);
    int rc = kpx_launch_status();
    if (rc) return rc;
    if (y_f32) hipLaunchKernelGGL(bn_apply_strip_bf16_kernel<true>, strip_grid8(P, C, groups), dim3(256), 0, s, (const bf16_t*)x, P, C, ldx, mean, invstd, gamma, beta, y, ldy, act);
    else hipLaunchKernelGGL(bn_apply_strip_bf16_kernel<false>, strip_grid8(P, C, groups), dim3(256), 0, s, (const bf16_t*)x, P, C, ldx, mean, invstd, gamma, beta, y, ldy, act);
    return kpx_launch_status();
}

template <bool DYF32>
__global__ __launch_bounds__(256) void bn_bwd_apply_strip_bf16_kernel(const void* __restrict__ dyv, int lddy, const bf16_t* __restrict__ x, int ldx, size_t P, int C,
                                                                      const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, int act, const float* __restrict__ sums, float inv_count,
                                                                      bf16_t* __restrict__ dx, int lddx) {
    const int G = C >> 3, Gb = G < 256 ? G : 256, PPB = 256 / Gb;
    const int t = threadIdx.x, cgl = t % Gb, prow = t / Gb;
    const int cg = blockIdx.y * Gb + cgl;
    if (prow >= PPB || cg >= G) return;
    x += (size_t)blockIdx.z * P * ldx; dx += (size_t)blockIdx.z * P * lddx;
    mean += blockIdx.z * C; invstd += blockIdx.z * C; sums += blockIdx.z * 2 * C;
    float mu[8], is[8], ga[8], be[8], gi[8], m0[8], m1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
        gi[j] = ga[j] * is[j]; m0[j] = sums[c] * inv_count; m1[j] = sums[C + c] * inv_count;
    }
    const size_t step = (size_t)gridDim.x * PPB;
    for (size_t p0 = (size_t)blockIdx.x * PPB + prow; p0 < P; p0 += step * 2) {
        float xv[2][8], gv[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t p = p0 + u * step;
            if (p < P) {
                pw_ld8(x + p * ldx + cg * 8, xv[u]);
                if (DYF32) {
                    const float* q = reinterpret_cast<const float*>(dyv) + ((size_t)blockIdx.z * P + p) * lddy + cg * 8;
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(q), d1 = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { gv[u][j] = d0[j]; gv[u][4 + j] = d1[j]; }
                } else pw_ld8(reinterpret_cast<const bf16_t*>(dyv) + ((size_t)blockIdx.z * P + p) * lddy + cg * 8, gv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t p = p0 + u * step;
            if (p >= P) break;
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (xv[u][j] - mu[j]) * is[j];
                const float yv = fmaf(xh, ga[j], be[j]);
                const float dz = (act == KPX_ACT_RELU && !(yv > 0.f)) ? 0.f : gv[u][j];
                r[j] = gi[j] * (dz - m0[j] - xh * m1[j]);
            }
            pw_st8(dx + p * lddx + cg * 8, r);
        }
    }
}
// dy bf16 (dy_f32 = 0) or fp32 (dy_f32 = 1: the batch norm whose output stayed fp32), x bf16 (the batch norm's input) -> dx bf16
extern "C" int kpx_bn_train_bwd_bf16(const void* dy, int lddy, int dy_f32, const void* x, int ldx, size_t P, int groups, int C,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                                     void* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                                     const float* tile_stats, size_t tiles_per_group, void* scratch, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !scratch || groups <= 0 || groups > 65535 || C <= 0 || C % 8 ||
        ldx % 8 || lddy % 8 || lddx % 8 || ldx < C || lddy < C || lddx < C || act < 0 || act > 1 || P == 0 || !pw_al16(dy, x, dx) ||
        (tile_stats && (act != KPX_ACT_RELU || tiles_per_group == 0)))
        return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    const size_t gstride = (size_t)KPX_RED_BLOCKS * 2 * C;
    int nb = 0, rc = 0;
    RedArgs a{}; a.x = (const float*)x; a.ldx = ldx; a.dy = (const float*)dy; a.lddy = lddy; a.P = P; a.C = C;
    a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.act = act; a.part = (double*)scratch; a.part_gstride = gstride;
    if (!tile_stats) {                                   // (with tile_stats the data-gradient epilogue that produced dy already reduced the two sums)
        rc = launch_chan_reduce_bf16(2, dy_f32 != 0, a, &nb, s, groups);
        if (rc) return rc;
    }
    float* sums = reinterpret_cast<float*>((double*)scratch + (size_t)groups * gstride);
    hipLaunchKernelGGL(bn_bwd_finalize_groups_kernel, dim3(C), dim3(256 * (groups < 4 ? groups : 4)), 0, s, (const double*)scratch, gstride, nb, tile_stats, tiles_per_group,
                       gamma, groups, C, dgamma, dbeta, sums, accumulate);
    if ((rc = kpx_launch_status())) return rc;
    const float inv_count = (float)(1.0 / (double)P);
    if (dy_f32) hipLaunchKernelGGL(bn_bwd_apply_strip_bf16_kernel<true>, strip_grid8(P, C, groups), dim3(256), 0, s, dy, lddy, (const bf16_t*)x, ldx, P, C, mean, invstd, gamma, beta, act, sums, inv_count, (bf16_t*)dx, lddx);
    else hipLaunchKernelGGL(bn_bwd_apply_strip_bf16_kernel<false>, strip_grid8(P, C, groups), dim3(256), 0, s, dy, lddy, (const bf16_t*)x, ldx, P, C, mean, invstd, gamma, beta, act, sums, inv_count, (bf16_t*)dx, lddx);
    return kpx_launch_status();
}

// ---- activation backward: dz = dy * act'(y) on bf16 tensors (n multiple of 8)
__global__ __launch_bounds__(256) void act_bwd_bf16_kernel(const bf16_t* dy, const bf16_t* y, bf16_t* dz, size_t n8, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        float g[8], v[8];
        pw_ld8(dy + 8 * i, g); pw_ld8(y + 8 * i, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] *= kpx_act_grad_from_y(v[j], act);
        pw_st8(dz + 8 * i, g);
    }
}
extern "C" int kpx_act_bwd_bf16(const void* dy, const void* y, void* dz, size_t n, int act, void* stream) {
    if (!dy || !y || !dz || n % 8 || !pw_al16(dy, y, dz)) return KPX_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, kpx_stream(stream), (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dz, n / 8, act);
    return kpx_launch_status();
}

// ---- bilinear x2 (legacy TF sampling), one thread per INPUT pixel and channel octet (as resize2x_fwd_quad_kernel)
__global__ __launch_bounds__(256) void resize2x_fwd_bf16_kernel(const bf16_t* __restrict__ x, int N, int H, int W, int C, int ldx, bf16_t* __restrict__ y, int ldy) {
    const unsigned G = (unsigned)C >> 3, total = (unsigned)N * H * W * G;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cq = i % G; unsigned p = i / G;
        const unsigned ix = p % (unsigned)W; p /= (unsigned)W;
        const unsigned iy = p % (unsigned)H, n = p / (unsigned)H;
        const unsigned iy1 = iy + 1 < (unsigned)H ? iy + 1 : iy, ix1 = ix + 1 < (unsigned)W ? ix + 1 : ix;
        const bf16_t* r0 = x + ((size_t)(n * H + iy) * W) * ldx + cq * 8;
        const bf16_t* r1 = x + ((size_t)(n * H + iy1) * W) * ldx + cq * 8;
        float tl[8], tr[8], bl[8], br[8];
        pw_ld8(r0 + (size_t)ix * ldx, tl); pw_ld8(r0 + (size_t)ix1 * ldx, tr); pw_ld8(r1 + (size_t)ix * ldx, bl); pw_ld8(r1 + (size_t)ix1 * ldx, br);
        bf16_t* o = y + ((size_t)(n * 2 * H + 2 * iy) * (2 * W) + 2 * ix) * ldy + cq * 8;
        const size_t rs = (size_t)2 * W * ldy;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float ty = a ? 0.5f : 0.f, tx = b ? 0.5f : 0.f;
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float top = tl[j] + (tr[j] - tl[j]) * tx, bot = bl[j] + (br[j] - bl[j]) * tx; r[j] = top + (bot - top) * ty; }
                pw_st8(o + a * rs + (size_t)b * ldy, r);
            }
    }
}
extern "C" int kpx_resize2x_fwd_bf16(const void* x, int N, int H, int W, int C, int ldx, void* y, int ldy, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || ldx % 8 || ldy % 8 || ldx < C || ldy < C || !pw_al16(x, y) || (size_t)N * H * W * (C / 8) >= 0x7fffffffu) return KPX_EINVAL;
    size_t nb = ((size_t)N * H * W * (C / 8) + 255) / 256; if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(resize2x_fwd_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), (const bf16_t*)x, N, H, W, C, ldx, (bf16_t*)y, ldy);
    return kpx_launch_status();
}
__global__ __launch_bounds__(256) void resize2x_bwd_bf16_kernel(const bf16_t* __restrict__ dy, int N, int H, int W, int C, int lddy, bf16_t* __restrict__ dx, int lddx) {
    const int G = C >> 3, W2 = 2 * W, H2 = 2 * H;
    const size_t total = (size_t)N * H * W * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % G) * 8;
        size_t p = i / G;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H);
        const int n = (int)(p / H);
        float wy[3], wx[3];
        wy[0] = iy >= 1 ? 0.5f : 0.f; wy[1] = 1.f; wy[2] = iy == H - 1 ? 1.f : 0.5f;
        wx[0] = ix >= 1 ? 0.5f : 0.f; wx[1] = 1.f; wx[2] = ix == W - 1 ? 1.f : 0.5f;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int oy = 2 * iy - 1 + a;
            if (oy < 0) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int ox = 2 * ix - 1 + b;
                if (ox < 0) continue;
                const float wgt = wy[a] * wx[b];
                float v[8];
                pw_ld8(dy + ((size_t)(n * H2 + oy) * W2 + ox) * lddy + c, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(wgt, v[j], acc[j]);
            }
        }
        pw_st8(dx + ((size_t)(n * H + iy) * W + ix) * lddx + c, acc);
    }
}
extern "C" int kpx_resize2x_bwd_bf16(const void* dy, int N, int H, int W, int C, int lddy, void* dx, int lddx, void* stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || lddy % 8 || lddx % 8 || lddy < C || lddx < C || !pw_al16(dy, dx)) return KPX_EINVAL;
    hipLaunchKernelGGL(resize2x_bwd_bf16_kernel, dim3(grid_for((size_t)N * H * W * (C / 8))), dim3(256), 0, kpx_stream(stream), (const bf16_t*)dy, N, H, W, C, lddy, (bf16_t*)dx, lddx);
    return kpx_launch_status();
}

// ---- VGG19: 2x2 max-pool (even H, W), the one-pass feature gradient, feature L1
__global__ __launch_bounds__(256) void maxpool2_fwd_bf16_kernel(const bf16_t* __restrict__ x, int N, int H, int W, int C, bf16_t* __restrict__ y) {
    const int G = C >> 3, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % G) * 8;
        size_t p = i / G;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float m[8], v[8];
        pw_ld8(x + ((size_t)(n * H + 2 * oy) * W + 2 * ox) * C + c, m);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            pw_ld8(x + ((size_t)(n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * C + c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
        }
        pw_st8(y + ((size_t)(n * Ho + oy) * Wo + ox) * C + c, m);
    }
}
extern "C" int kpx_maxpool2_fwd_bf16(const void* x, int N, int H, int W, int C, void* y, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2 || C <= 0 || C % 8 || !pw_al16(x, y)) return KPX_EINVAL;
    hipLaunchKernelGGL(maxpool2_fwd_bf16_kernel, dim3(grid_for((size_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, kpx_stream(stream), (const bf16_t*)x, N, H, W, C, (bf16_t*)y);
    return kpx_launch_status();
}
// d[pix][c] = [y_pred > 0] * ( (pix is the first maximum of its 2x2 window ? dy_pooled : 0) + g * sign(y_pred - y_gt) )   (as vgg_feat_bwd_kernel)
__global__ __launch_bounds__(256) void vgg_feat_bwd_bf16_kernel(const bf16_t* __restrict__ f, size_t half, const float* gdev, float ghost,
                                                                const bf16_t* __restrict__ dyp, int B, int H, int W, int C, bf16_t* __restrict__ d) {
    const float g = ghost * (gdev ? *gdev : 1.0f);
    const int C8 = C >> 3, Ho = H / 2, Wo = W / 2;
    const bf16_t* fp = f + half;
    if (!dyp) {
        const size_t total = half >> 3;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            float yp[8], yg[8], o[8];
            pw_ld8(fp + 8 * i, yp); pw_ld8(f + 8 * i, yg);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float df = yg[e] - yp[e]; const float l1 = df > 0.f ? -g : (df < 0.f ? g : 0.f); o[e] = yp[e] > 0.f ? l1 : 0.f; }
            pw_st8(d + 8 * i, o);
        }
        return;
    }
    const size_t total = (size_t)B * Ho * Wo * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        size_t p = i / C8;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float v[4][8], gt[4][8], best[8], gp[8]; size_t off[4]; int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            off[k] = ((size_t)(n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * C + c;
            pw_ld8(fp + off[k], v[k]); pw_ld8(f + off[k], gt[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (v[k][e] > best[e]) { best[e] = v[k][e]; bi[e] = k; }
        }
        pw_ld8(dyp + (((size_t)(n * Ho + oy) * Wo + ox) * C + c), gp);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float df = gt[k][e] - v[k][e];
                const float l1 = df > 0.f ? -g : (df < 0.f ? g : 0.f);
                o[e] = v[k][e] > 0.f ? ((bi[e] == k ? gp[e] : 0.f) + l1) : 0.f;
            }
            pw_st8(d + off[k], o);
        }
    }
}
extern "C" int kpx_vgg_feat_bwd_bf16(const void* f, size_t half, const float* gscale_dev, float gscale_host, const void* dy_pooled,
                                     int B, int H, int W, int C, void* d, void* stream) {
    if (!f || !d || half == 0 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || half != (size_t)B * H * W * C || (dy_pooled && (H % 2 || W % 2)) || !pw_al16(f, d, dy_pooled))
        return KPX_EINVAL;
    const size_t items = dy_pooled ? (size_t)B * (H / 2) * (W / 2) * (C / 8) : half / 8;
    hipLaunchKernelGGL(vgg_feat_bwd_bf16_kernel, dim3(grid_for(items)), dim3(256), 0, kpx_stream(stream), (const bf16_t*)f, half, gscale_dev, gscale_host, (const bf16_t*)dy_pooled, B, H, W, C, (bf16_t*)d);
    return kpx_launch_status();
}
// mean |f[0:half] - f[half:2 half]| of a bf16 feature tensor [gt ; pred] (half a multiple of 8): partials in fp64, out fp32
__global__ __launch_bounds__(256) void l1_pair_partial_bf16_kernel(const bf16_t* __restrict__ f, size_t half, double* __restrict__ part) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half / 8; i += (size_t)gridDim.x * 256) {
        float a[8], b[8];
        pw_ld8(f + 8 * i, a); pw_ld8(f + half + 8 * i, b);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += fabsf(a[j] - b[j]);
        s += (double)t;
    }
    s = kpx_wave_sum_d(s);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void l1_pair_finalize_bf16_kernel(const double* part, int nb, double count, float* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
    s = kpx_wave_sum_d(s);
    if (threadIdx.x == 0) *out = (float)(s / count);
}
extern "C" int kpx_l1_pair_fwd_bf16(const void* f, size_t half, float* loss_out, void* scratch, void* stream) {
    if (!f || !loss_out || !scratch || half == 0 || half % 8 || !pw_al16(f)) return KPX_EINVAL;
    size_t nb = (half / 8 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    hipStream_t s = kpx_stream(stream);
    hipLaunchKernelGGL(l1_pair_partial_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)f, half, (double*)scratch);
    int rc = kpx_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(l1_pair_finalize_bf16_kernel, dim3(1), dim3(64), 0, s, (const double*)scratch, (int)nb, (double)half, loss_out);
    return kpx_launch_status();
}
