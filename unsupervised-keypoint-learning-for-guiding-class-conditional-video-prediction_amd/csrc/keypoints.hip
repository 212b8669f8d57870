// Key-point head (separable softmax expectation) and Gaussian heat-map renderer.
//
// Head: model_utils.get_coord applied twice + tf.stack (utils/model.py:63-70, networks/__init__.py:68-72):
//   y: mean over W -> softmax over H -> sum(p * linspace(-1,1,H));  x: mean over H -> softmax over W -> ...
// The logits [B,H,W,K] are read ONCE (TF reads them twice, once per axis): stage 1 streams row stripes and emits
// row sums + per-stripe column sums; stage 2 gives one wavefront to each (b,k,axis) profile and does max / sum /
// expectation with wave shuffles.
// Renderer: model_utils.get_gaussian_maps (utils/model.py:49-60) written straight in NHWC (the reference builds BKHW
// and transposes): a pure write stream, 16 B per lane.
// linspace follows tf.linspace in fp32 (start + step*i, two roundings) via __fmul_rn/__fadd_rn so that hipcc's
// default fp-contraction cannot fuse it.
#include "kpx_common.h"
#include "kpx_env.h"
#include <stdlib.h>

#define KP_RS 8   // rows per stripe in stage 1

__device__ __forceinline__ float kpx_linspace(int i, int n) {
    const float step = __fdiv_rn(2.0f, (float)(n - 1));
    return __fadd_rn(-1.0f, __fmul_rn(step, (float)i));
}

__global__ __launch_bounds__(256) void kp_stage1_kernel(const float* __restrict__ x, int H, int W, int K,
                                                        float* __restrict__ rowsum, float* __restrict__ colpart, int nstripes) {
    const int b = blockIdx.x, stripe = blockIdx.y;
    const int T = (256 / K) * K, t = threadIdx.x;
    const int WK = W * K;
    const int h0 = stripe * KP_RS;
    const float* xb = x + ((size_t)b * H + h0) * WK;
    float racc[KP_RS];
#pragma unroll
    for (int r = 0; r < KP_RS; ++r) racc[r] = 0.f;
    if (t < T) {
        for (int e = t; e < WK; e += T) {          // T % K == 0, so this thread's channel k = t % K is fixed
            float v[KP_RS];
#pragma unroll
            for (int r = 0; r < KP_RS; ++r) v[r] = (h0 + r < H) ? xb[(size_t)r * WK + e] : 0.f;
            float c = 0.f;
#pragma unroll
            for (int r = 0; r < KP_RS; ++r) { c += v[r]; racc[r] += v[r]; }
            colpart[((size_t)b * nstripes + stripe) * WK + e] = c;
        }
    }
    __shared__ float sm[KP_RS][256];
#pragma unroll
    for (int r = 0; r < KP_RS; ++r) sm[r][t] = (t < T) ? racc[r] : 0.f;
    __syncthreads();
    if (t < K) {
        const int groups = T / K;
#pragma unroll
        for (int r = 0; r < KP_RS; ++r) {
            if (h0 + r >= H) break;
            float s = 0.f;
            for (int g2 = 0; g2 < groups; ++g2) s += sm[r][g2 * K + t];
            rowsum[((size_t)b * H + h0 + r) * K + t] = s;
        }
    }
}

// one wavefront per (b, k, axis)
__global__ __launch_bounds__(256) void kp_stage2_kernel(const float* __restrict__ rowsum, const float* __restrict__ colpart,
                                                        int B, int H, int W, int K, int nstripes,
                                                        float* __restrict__ mu, float* __restrict__ prob_y, float* __restrict__ prob_x) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wid >= B * K * 2) return;
    const int axis = wid & 1;                 // 0 -> x (profile over W), 1 -> y (profile over H)
    const int k = (wid >> 1) % K, b = (wid >> 1) / K;
    const int n = axis ? H : W;
    const int WK = W * K;
    // pass 1: means (kept in registers: up to 8 per lane -> n <= 512) and max
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        float m = -INFINITY;
        if (i < n) {
            if (axis) m = rowsum[((size_t)b * H + i) * K + k] / (float)W;
            else {
                float s = 0.f;
                for (int st = 0; st < nstripes; ++st) s += colpart[((size_t)b * nstripes + st) * WK + (size_t)i * K + k];
                m = s / (float)H;
            }
        }
        v[j] = m;
        mx = fmaxf(mx, m);
    }
    mx = kpx_wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        v[j] = (i < n) ? expf(v[j] - mx) : 0.f;
        se += v[j];
    }
    se = kpx_wave_sum(se);
    const float inv = 1.0f / se;                // tf.nn.softmax: e * (1/sum)
    float ex = 0.f;
    float* pout = axis ? prob_y : prob_x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < n) {
            const float p = v[j] * inv;
            pout[((size_t)b * n + i) * K + k] = p;
            ex += p * kpx_linspace(i, n);
        }
    }
    ex = kpx_wave_sum(ex);
    if (lane == 0) mu[((size_t)b * K + k) * 2 + axis] = ex;
}

extern "C" size_t kpx_keypoint_head_scratch_bytes(int B, int H, int W, int K) {
    const size_t nstripes = (size_t)(H + KP_RS - 1) / KP_RS;
    return ((size_t)B * H * K + (size_t)B * nstripes * W * K) * sizeof(float);
}

extern "C" int kpx_keypoint_head_fwd_f32(const float* logits, int B, int H, int W, int K,
                                         float* mu, float* prob_y, float* prob_x, void* scratch, void* stream) {
    if (!logits || !mu || !prob_y || !prob_x || !scratch || B <= 0 || H < 2 || W < 2 || K <= 0 || K > 256 || H > 512 || W > 512)
        return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    const int nstripes = (H + KP_RS - 1) / KP_RS;
    float* rowsum = (float*)scratch;
    float* colpart = rowsum + (size_t)B * H * K;
    hipLaunchKernelGGL(kp_stage1_kernel, dim3(B, nstripes), dim3(256), 0, s, logits, H, W, K, rowsum, colpart, nstripes);
    int rc = kpx_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(kp_stage2_kernel, dim3((B * K * 2 + 3) / 4), dim3(256), 0, s, (const float*)rowsum, (const float*)colpart,
                       B, H, W, K, nstripes, mu, prob_y, prob_x);
    return kpx_launch_status();
}

// dlogits[b,h,w,k] = dRow[b,h,k]/W + dCol[b,w,k]/H, dRow[h] = dmu_y * p_y[h] * (lin_H[h] - mu_y), dCol likewise
__global__ __launch_bounds__(256) void kp_bwd_kernel(const float* __restrict__ dmu, const float* __restrict__ mu,
                                                     const float* __restrict__ prob_y, const float* __restrict__ prob_x,
                                                     int B, int H, int W, int K, float* __restrict__ dl) {
    const size_t total = (size_t)B * H * W * K;
    const float iw = 1.0f / (float)W, ih = 1.0f / (float)H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % K);
        size_t p = i / K;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int b = (int)(p / H);
        const size_t bk = ((size_t)b * K + k) * 2;
        const float drow = dmu[bk + 1] * prob_y[((size_t)b * H + h) * K + k] * (kpx_linspace(h, H) - mu[bk + 1]);
        const float dcol = dmu[bk + 0] * prob_x[((size_t)b * W + w) * K + k] * (kpx_linspace(w, W) - mu[bk + 0]);
        dl[i] = drow * iw + dcol * ih;
    }
}
extern "C" int kpx_keypoint_head_bwd_f32(const float* dmu, const float* mu, const float* prob_y, const float* prob_x,
                                         int B, int H, int W, int K, float* dlogits, void* stream) {
    if (!dmu || !mu || !prob_y || !prob_x || !dlogits || B <= 0 || H < 2 || W < 2 || K <= 0) return KPX_EINVAL;
    const size_t total = (size_t)B * H * W * K;
    size_t nb = (total + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(kp_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), dmu, mu, prob_y, prob_x, B, H, W, K, dlogits);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ 1x1 head folded into the key-point head
// pose_encoder ends in a 1x1 convolution C -> K (networks/__init__.py:54) whose output, the logits [B,H,W,K], is consumed ONLY by the two
// axis means of get_coord (utils/model.py:63-70).  Both are linear, so they commute:
//   mean_w(x W + b)[b,h,k] = (sum_w x[b,h,w,:]) W[:,k] / W_ + b[k]
// and the logits are never formed: stage 1 runs on the C-channel activation x itself, the [C,K] projection is applied to the H + W profile
// rows of each image.  The backward is separable for the same reason:
//   dlogits[b,h,w,k] = dRow[b,h,k]/W_ + dCol[b,w,k]/H_                      (kp_bwd_kernel above)
//   dx[b,h,w,c]      = R[b,h,c] + Cc[b,w,c],   R = (dRow/W_) W^T, Cc = (dCol/H_) W^T            -> one write pass, nothing read
//   dW[c,k]          = sum_{b,h} xs_y[b,h,c] dRow[b,h,k]/W_ + sum_{b,w} xs_x[b,w,c] dCol[b,w,k]/H_  (xs_* = the forward's row / column sums)
//   db[k]            = sum_{b,h} dRow[b,h,k] + sum_{b,w} dCol[b,w,k]            (analytically 0: softmax ignores a per-channel constant)
// Against conv + head this removes the logits write + two reads, the dlogits write + three reads and the 1x1 weight-gradient pass.
// Rounding differs from the unfused order only in where the fp32 sums are taken (sum over pixels first, project second).

// grid (B, 2): axis 1 -> y profile (rows, from xs_y), axis 0 -> x profile (columns: stripe partials are summed here and kept as xs_x)
__global__ __launch_bounds__(256) void kp_proj_stage2_kernel(const float* __restrict__ xs_y, const float* __restrict__ colpart,
                                                             float* __restrict__ xs_x, const float* __restrict__ wk,
                                                             const float* __restrict__ bias, int H, int W, int C, int K, int nstripes,
                                                             float* __restrict__ mu, float* __restrict__ prob_y, float* __restrict__ prob_x) {
    extern __shared__ float kp_sm[];            // [n][C] profile sums, then [C][K] weights
    const int b = blockIdx.x, axis = blockIdx.y;
    const int n = axis ? H : W, other = axis ? W : H;
    float* S = kp_sm;
    float* wl = kp_sm + (size_t)n * C;
    const int nC = n * C, WC = W * C;
    for (int e = threadIdx.x; e < nC; e += 256) {
        float s;
        if (axis) s = xs_y[(size_t)b * nC + e];
        else {
            s = 0.f;
            const float* cp = colpart + (size_t)b * nstripes * WC + e;
            int st = 0;
            for (; st + 7 < nstripes; st += 8) {               // eight stripe loads in flight (the plain loop was a chain of dependent L2 loads)
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = cp[(size_t)(st + j) * WC];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
            for (; st < nstripes; ++st) s += cp[(size_t)st * WC];
            if (blockIdx.z == 0) xs_x[(size_t)b * nC + e] = s;
        }
        S[e] = s;
    }
    for (int e = threadIdx.x; e < C * K; e += 256) wl[e] = wk[e];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_other = (float)other;
    // blockIdx.z splits the K profiles of this (image, axis): workgroup z takes k = 4 z + wave, 4 (z + gridDim.z) + wave, ..
    for (int k = blockIdx.z * 4 + wave; k < K; k += 4 * gridDim.z) {
        const float bk = bias ? bias[k] : 0.f;
        float v[8];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane + 64 * j;
            float m = -INFINITY;
            if (i < n) {
                float d = 0.f;
                for (int c = 0; c < C; ++c) d = fmaf(S[i * C + c], wl[c * K + k], d);
                m = d / inv_other + bk;
            }
            v[j] = m;
            mx = fmaxf(mx, m);
        }
        mx = kpx_wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane + 64 * j;
            v[j] = (i < n) ? expf(v[j] - mx) : 0.f;
            se += v[j];
        }
        se = kpx_wave_sum(se);
        const float inv = 1.0f / se;
        float ex = 0.f;
        float* pout = axis ? prob_y : prob_x;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane + 64 * j;
            if (i < n) {
                const float p = v[j] * inv;
                pout[((size_t)b * n + i) * K + k] = p;
                ex += p * kpx_linspace(i, n);
            }
        }
        ex = kpx_wave_sum(ex);
        if (lane == 0) mu[((size_t)b * K + k) * 2 + axis] = ex;
    }
}

static bool kp_proj_dims_ok(int B, int H, int W, int C, int K) {
    const int n = H > W ? H : W;
    return B > 0 && H >= 2 && W >= 2 && H <= 512 && W <= 512 && C > 0 && C <= 256 && K > 0 && K <= 64 && (C % 4) == 0 &&
           ((size_t)n * C + (size_t)C * K) * sizeof(float) <= 64 * 1024 && ((size_t)n * K + (size_t)C * K) * sizeof(float) <= 64 * 1024;
}

extern "C" int kpx_keypoint_head_proj_eligible(int B, int H, int W, int C, int K) { return kp_proj_dims_ok(B, H, W, C, K) ? 1 : 0; }

extern "C" size_t kpx_keypoint_head_proj_scratch_bytes(int B, int H, int W, int C, int K) {
    const size_t nstripes = (size_t)(H + KP_RS - 1) / KP_RS;
    const size_t fwd = (size_t)B * nstripes * W * C;                                   // stripe column sums
    const size_t bwd = (size_t)B * (H + W) * C + (size_t)B * 2 * ((size_t)C * K + K);  // R, Cc, per-(image, axis) dW / db partials
    return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

extern "C" int kpx_keypoint_head_proj_fwd_f32(const float* x, const float* wk, const float* bias, int B, int H, int W, int C, int K,
                                              float* mu, float* prob_y, float* prob_x, float* xs_y, float* xs_x, void* scratch, void* stream) {
    if (!x || !wk || !mu || !prob_y || !prob_x || !xs_y || !xs_x || !scratch || !kp_proj_dims_ok(B, H, W, C, K)) return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    const int nstripes = (H + KP_RS - 1) / KP_RS;
    float* colpart = (float*)scratch;
    hipLaunchKernelGGL(kp_stage1_kernel, dim3(B, nstripes), dim3(256), 0, s, x, H, W, C, xs_y, colpart, nstripes);
    int rc = kpx_launch_status();
    if (rc) return rc;
    const int n = H > W ? H : W;
    const int kz = (K + 3) / 4 < 4 ? (K + 3) / 4 : 4;          // 128 workgroups alone (B = 64) left the chip idle: 45 us -> split the K profiles 4 ways
    hipLaunchKernelGGL(kp_proj_stage2_kernel, dim3(B, 2, kz), dim3(256), ((size_t)n * C + (size_t)C * K) * sizeof(float), s,
                       (const float*)xs_y, (const float*)colpart, xs_x, wk, bias, H, W, C, K, nstripes, mu, prob_y, prob_x);
    return kpx_launch_status();
}

// grid (B, 2): d[i,k] = dmu * p * (lin - mu) / other;  RC[b,i,c] = sum_k d[i,k] W[c,k];  partial dW[c,k] = sum_i xs[b,i,c] d[i,k];
// partial db[k] = other * sum_i d[i,k]
__global__ __launch_bounds__(256) void kp_proj_bwd_small_kernel(const float* __restrict__ dmu, const float* __restrict__ mu,
                                                                const float* __restrict__ prob_y, const float* __restrict__ prob_x,
                                                                const float* __restrict__ xs_y, const float* __restrict__ xs_x,
                                                                const float* __restrict__ wk, int H, int W, int C, int K,
                                                                float* __restrict__ Rb, float* __restrict__ Cb, float* __restrict__ part) {
    extern __shared__ float kp_sm[];            // [n][K] d, then [C][K] weights
    const int b = blockIdx.x, axis = blockIdx.y, B = gridDim.x;
    const int n = axis ? H : W, other = axis ? W : H;
    float* d = kp_sm;
    float* wl = kp_sm + (size_t)n * K;
    const float* p = axis ? prob_y : prob_x;
    const float io = 1.0f / (float)other;
    for (int e = threadIdx.x; e < n * K; e += 256) {
        const int i = e / K, k = e - i * K;
        const size_t bk = ((size_t)b * K + k) * 2 + axis;
        d[e] = dmu[bk] * p[(size_t)b * n * K + e] * (kpx_linspace(i, n) - mu[bk]) * io;
    }
    for (int e = threadIdx.x; e < C * K; e += 256) wl[e] = wk[e];
    __syncthreads();
    float* out = axis ? Rb + (size_t)b * H * C : Cb + (size_t)b * W * C;
    for (int e = threadIdx.x; e < n * C; e += 256) {
        const int i = e / C, c = e - i * C;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(d[i * K + k], wl[c * K + k], s);
        out[e] = s;
    }
    const float* xs = (axis ? xs_y : xs_x) + (size_t)b * n * C;
    float* pp = part + ((size_t)b * 2 + axis) * ((size_t)C * K + K);
    for (int e = threadIdx.x; e < C * K + K; e += 256) {
        float s = 0.f;
        if (e < C * K) {
            const int c = e / K, k = e - c * K;
            int i = 0;
            for (; i + 7 < n; i += 8) {                      // eight row / column sums of x in flight per thread
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = xs[(i + j) * C + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(v[j], d[(i + j) * K + k], s);
            }
            for (; i < n; ++i) s = fmaf(xs[i * C + c], d[i * K + k], s);
        } else {
            const int k = e - C * K;
            for (int i = 0; i < n; ++i) s += d[i * K + k];
            s *= (float)other;
        }
        pp[e] = s;
    }
    (void)B;
}

// dW / db = sum over the 2B partial rows in a fixed order (deterministic): segment sg = threadIdx.x / 64 of a 256-thread workgroup sums rows
// sg, sg + 4, .. of 64 elements, the four segment sums are combined in LDS order; accumulate != 0 adds into the destination
// (one thread per element walking all 128 rows was a 30 us chain of dependent L2 loads)
__global__ __launch_bounds__(256) void kp_proj_bwd_reduce_kernel(const float* __restrict__ part, int rows, int CK, int K,
                                                                 float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const int n = CK + K;
    float s = 0.f;
    if (e < n) {
        int r = sg;
        for (; r + 12 < rows; r += 16) {
            const float v0 = part[(size_t)r * n + e], v1 = part[(size_t)(r + 4) * n + e], v2 = part[(size_t)(r + 8) * n + e], v3 = part[(size_t)(r + 12) * n + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; r < rows; r += 4) s += part[(size_t)r * n + e];
    }
    __shared__ float sm[4][64];
    sm[sg][el] = s;
    __syncthreads();
    if (sg == 0 && e < n) {
        const float tot = (sm[0][el] + sm[1][el]) + (sm[2][el] + sm[3][el]);
        if (e < CK) { if (dw) dw[e] = accumulate ? dw[e] + tot : tot; }
        else if (db) db[e - CK] = accumulate ? db[e - CK] + tot : tot;
    }
}

// dx[b,h,w,:] = R[b,h,:] + Cc[b,w,:]: pure write stream, 16 B per lane.  grid (slabs of rows, B); a thread owns one (w, channel quad)
// column for the whole slab when W*C/4 is a multiple of the block size (the Cc term stays in registers).
__global__ __launch_bounds__(256) void kp_proj_dx_kernel(const float* __restrict__ Rb, const float* __restrict__ Cb, int H, int W, int C,
                                                         int rows_per_block, float* __restrict__ dx) {
    const int b = blockIdx.y, h0 = blockIdx.x * rows_per_block;
    const int cq = C >> 2, rowq = W * cq;            // float4 per image row
    const float4* R4 = (const float4*)(Rb + (size_t)b * H * C);
    const float4* C4 = (const float4*)(Cb + (size_t)b * W * C);
    float4* out = (float4*)(dx + (size_t)b * H * W * C);
    for (int q0 = threadIdx.x; q0 < rowq; q0 += 256) {
        const float4 cc = C4[q0];
        const int c4 = q0 % cq;
        for (int r = 0; r < rows_per_block; ++r) {
            const int h = h0 + r;
            if (h >= H) break;
            const float4 rr = R4[h * cq + c4];
            float4 o;
            o.x = rr.x + cc.x; o.y = rr.y + cc.y; o.z = rr.z + cc.z; o.w = rr.w + cc.w;
            out[(size_t)h * rowq + q0] = o;
        }
    }
}

extern "C" int kpx_keypoint_head_proj_bwd_f32(const float* dmu, const float* mu, const float* prob_y, const float* prob_x,
                                              const float* xs_y, const float* xs_x, const float* wk, int B, int H, int W, int C, int K,
                                              float* dx, float* dw, float* db, int accumulate, void* scratch, void* stream) {
    if (!dmu || !mu || !prob_y || !prob_x || !xs_y || !xs_x || !wk || !scratch || !kp_proj_dims_ok(B, H, W, C, K)) return KPX_EINVAL;
    hipStream_t s = kpx_stream(stream);
    float* Rb = (float*)scratch;
    float* Cb = Rb + (size_t)B * H * C;
    float* part = Cb + (size_t)B * W * C;
    const int n = H > W ? H : W;
    hipLaunchKernelGGL(kp_proj_bwd_small_kernel, dim3(B, 2), dim3(256), ((size_t)n * K + (size_t)C * K) * sizeof(float), s,
                       dmu, mu, prob_y, prob_x, xs_y, xs_x, wk, H, W, C, K, Rb, Cb, part);
    int rc = kpx_launch_status();
    if (rc) return rc;
    if (dw || db) {
        hipLaunchKernelGGL(kp_proj_bwd_reduce_kernel, dim3((C * K + K + 63) / 64), dim3(256), 0, s, (const float*)part, 2 * B, C * K, K,
                           dw, db, accumulate);
        rc = kpx_launch_status();
        if (rc) return rc;
    }
    if (dx) {
        const int rpb = 8;
        hipLaunchKernelGGL(kp_proj_dx_kernel, dim3((H + rpb - 1) / rpb, B), dim3(256), 0, s, (const float*)Rb, (const float*)Cb, H, W, C, rpb, dx);
        rc = kpx_launch_status();
    }
    return rc;
}

// ------------------------------------------------------------------------------------------ Gaussian maps
// ONE definition of a heat-map element for every renderer and for the backward: the reference's rounding sequence
// square(y - mu_y) + square(x - mu_x), times float32(inv_std**2), negate, exp (utils/model.py:56-59); exp is the hardware
// exp2 path (__expf: |err| <= ~2 ulp of a result in [0,1], inside the 2e-6 absolute budget).  Because both renderers call
// this function, a map element has the same bits whether it is rendered contiguously or into a channel slice.
__device__ __forceinline__ float kpx_gauss(float yv, float xv, float my, float mx, float inv2) {
    const float dy = __fsub_rn(yv, my), dx = __fsub_rn(xv, mx);
    const float dist = __fmul_rn(__fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dx, dx)), inv2);
    return __expf(-dist);
}

// contiguous output (ldy == K): pure write stream.  The H*W*K floats of one image are covered as float4 by blockIdx.x
// slices of `per_block` float4 (a multiple of 256: every wave writes 1 KB contiguous per store instruction, no partial
// iterations); index arithmetic is 32-bit with multiply-shift divisions; streaming (non-temporal) stores.
__global__ __launch_bounds__(256) void gauss_fwd_flat_kernel(const float* __restrict__ mu, int K, int H, int W, float inv2,
                                                             float* __restrict__ out, int per_block) {
    const int b = blockIdx.y;
    __shared__ float smx[256], smy[256], sxs[512], sys[512];
    for (int i = threadIdx.x; i < K; i += 256) { smx[i] = mu[((size_t)b * K + i) * 2]; smy[i] = mu[((size_t)b * K + i) * 2 + 1]; }
    for (int i = threadIdx.x; i < W; i += 256) sxs[i] = kpx_linspace(i, W);
    for (int i = threadIdx.x; i < H; i += 256) sys[i] = kpx_linspace(i, H);
    __syncthreads();
    const float invK = 1.0f / (float)K, invW = 1.0f / (float)W;
    const int n4 = (H * W * K) >> 2;                          // (H*W*K) % 4 == 0 is checked by the launcher
    f32x4* ob = reinterpret_cast<f32x4*>(out + (size_t)b * H * W * K);
    const int i0 = blockIdx.x * per_block;
    const int i1 = min(i0 + per_block, n4);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const int f = i * 4;
        int pix = (int)(((float)f + 0.5f) * invK);            // exact after the +-1 fix-up for f < 2^23
        int k = f - pix * K;
        if (k < 0) { --pix; k += K; } else if (k >= K) { ++pix; k -= K; }
        int h = (int)(((float)pix + 0.5f) * invW);
        int w = pix - h * W;
        if (w < 0) { --h; w += W; } else if (w >= W) { ++h; w -= W; }
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = kpx_gauss(sys[h], sxs[w], smy[k], smx[k], inv2);
            if (++k == K) { k = 0; if (++w == W) { w = 0; ++h; } }
        }
        __builtin_nontemporal_store(r, ob + i);
    }
}
// contiguous output, K >= 4, no LDS and no barrier: the workgroup has T = K*m threads, so a thread's float4 index advances by a
// multiple of K floats per iteration and its four channels k0..k0+3 (mod K) never change -- their (mu_x, mu_y) live in registers,
// loaded once.  blockIdx.x covers `iters` consecutive T-float4 slabs of image blockIdx.y; streaming (non-temporal) 16-B stores.
// WFIX: the float4 index advances by a whole number of image rows per iteration (4*T/K is a multiple of W, e.g. K = 15, W = 128,
// T = 480), so a thread's four pixels keep their COLUMN as well: (x - mu_x)^2 is loop-invariant and an element costs six VALU ops + exp.
// The rounding sequence is kpx_gauss's, operation for operation (same bits as the strided renderer).
template <bool NT, bool WFIX>
__global__ __launch_bounds__(512) void gauss_fwd_reg_kernel(const float* __restrict__ mu, int K, int H, int W, float inv2,
                                                            float* __restrict__ out, int iters) {
    const int b = blockIdx.y, T = blockDim.x;
    const int n4 = (H * W * K) >> 2;
    int i = blockIdx.x * iters * T + threadIdx.x;
    const int f = i * 4;
    int pix = f / K;
    const int k0 = f - pix * K;
    int h = pix / W, w = pix - h * W;
    float mx[4], my[4]; int carry[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int kk = k0 + j;
        carry[j] = kk >= K;
        if (carry[j]) kk -= K;
        mx[j] = mu[((size_t)b * K + kk) * 2]; my[j] = mu[((size_t)b * K + kk) * 2 + 1];
    }
    const int step_pix = (4 * T) / K;
    f32x4* ob = reinterpret_cast<f32x4*>(out + (size_t)b * H * W * K);
    if (WFIX) {
        float dx2[4]; int rowc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int wj = w + carry[j];
            rowc[j] = wj >= W;
            if (rowc[j]) wj -= W;
            const float dx = __fsub_rn(kpx_linspace(wj, W), mx[j]);
            dx2[j] = __fmul_rn(dx, dx);
        }
        const int step_rows = step_pix / W;
        for (int it = 0; it < iters; ++it) {
            if (i < n4) {
                const float y0 = kpx_linspace(h, H), y1 = kpx_linspace(h + 1, H);
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dy = __fsub_rn(rowc[j] ? y1 : y0, my[j]);
                    const float dist = __fmul_rn(__fadd_rn(__fmul_rn(dy, dy), dx2[j]), inv2);
                    r[j] = __expf(-dist);
                }
                if (NT) __builtin_nontemporal_store(r, ob + i); else ob[i] = r;
            }
            i += T;
            h += step_rows;
        }
        return;
    }
    for (int it = 0; it < iters; ++it) {
        if (i < n4) {
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int wj = w + carry[j], hj = h;
                if (wj >= W) { wj -= W; ++hj; }
                r[j] = kpx_gauss(kpx_linspace(hj, H), kpx_linspace(wj, W), my[j], mx[j], inv2);
            }
            if (NT) __builtin_nontemporal_store(r, ob + i); else ob[i] = r;
        }
        i += T;
        w += step_pix;
        while (w >= W) { w -= W; ++h; }
    }
}
// strided output (channel slice of a wider concat buffer)
__global__ __launch_bounds__(256) void gauss_fwd_strided_kernel(const float* __restrict__ mu, int B, int K, int H, int W, float inv2,
                                                                float* __restrict__ out, int ldy) {
    const size_t total = (size_t)B * H * W * K;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e % K);
        const size_t pixg = e / K;
        const int w = (int)(pixg % W);
        const size_t t2 = pixg / W;
        const int h = (int)(t2 % H), b = (int)(t2 / H);
        const float mx = mu[((size_t)b * K + k) * 2], my = mu[((size_t)b * K + k) * 2 + 1];
        out[pixg * ldy + k] = kpx_gauss(kpx_linspace(h, H), kpx_linspace(w, W), my, mx, inv2);
    }
}
extern "C" int kpx_gaussian_maps_fwd_f32(const float* mu, int B, int K, int H, int W, double inv_std, float* maps, int ldy, void* stream) {
    if (!mu || !maps || B <= 0 || K <= 0 || H < 2 || W < 2 || ldy < K) return KPX_EINVAL;
    const float inv2 = (float)(inv_std * inv_std);      // python: inv_std ** 2 in float64, then cast (utils/model.py:58)
    const size_t total = (size_t)B * H * W * K;
    hipStream_t s = kpx_stream(stream);
    const bool flat = ldy == K && (((uintptr_t)maps) & 15) == 0 && ((size_t)H * W * K) % 4 == 0 && (size_t)H * W * K < (1u << 23);
    if (flat && K >= 4 && K <= 512) {
        int m = 1, best = 0;                                  // T = K*m threads: fullest last wavefront among T in [128, 512]
        bool wfix = false;
        for (int mm = 1; K * mm <= 512; ++mm) {
            const int T = K * mm;
            const bool wf = (4 * mm) % W == 0;                // whole rows per iteration: the cheaper loop body
            const int fill = T * 1000 / (((T + 63) / 64) * 64) + (T >= 192 ? 1000 : 0) + (wf ? 500 : 0);
            if (fill >= best) { best = fill; m = mm; wfix = wf; }
        }
        const int T = K * m, n4 = (int)(((size_t)H * W * K) >> 2);
        // Stores per thread: ~11 on the fixed-column path, ~6 on the general one (twice the VALU per element needs twice the wavefronts to
        // keep HBM fed), never fewer than ~512 workgroups.  Measured (nine / three rotating outputs): [64,128,128,15] 0.67 of the 8 TB/s spec at
        // 512-768 workgroups, 0.35 at 12 288; [32,256,256,40] 0.57 at 768, 0.72 at 12 288, 0.68 at 16 384.
        int G = (n4 + T * (wfix ? 11 : 6) - 1) / (T * (wfix ? 11 : 6));
        if (G * B < 512) G = (512 + B - 1) / B;
        if (G < 1) G = 1;
        int iters = (n4 + T * G - 1) / (T * G); if (iters < 1) iters = 1;
        G = (n4 + T * iters - 1) / (T * iters);
        const int nt = 1;                                 // non-temporal stores
        const dim3 grid((unsigned)G, (unsigned)B), block((unsigned)T);
        if (wfix) {
            if (nt) hipLaunchKernelGGL((gauss_fwd_reg_kernel<true, true>), grid, block, 0, s, mu, K, H, W, inv2, maps, iters);
            else hipLaunchKernelGGL((gauss_fwd_reg_kernel<false, true>), grid, block, 0, s, mu, K, H, W, inv2, maps, iters);
        } else {
            if (nt) hipLaunchKernelGGL((gauss_fwd_reg_kernel<true, false>), grid, block, 0, s, mu, K, H, W, inv2, maps, iters);
            else hipLaunchKernelGGL((gauss_fwd_reg_kernel<false, false>), grid, block, 0, s, mu, K, H, W, inv2, maps, iters);
        }
    } else if (flat && K <= 256 && W <= 512 && H <= 512) {
        const int n4 = (int)(((size_t)H * W * K) >> 2);
        int per_block = 1024;                                 // 4 float4 per thread; fewer when that would leave CUs idle
        while (per_block > 256 && (long)B * ((n4 + per_block - 1) / per_block) < 2048) per_block >>= 1;
        hipLaunchKernelGGL(gauss_fwd_flat_kernel, dim3((unsigned)((n4 + per_block - 1) / per_block), (unsigned)B), dim3(256), 0, s, mu, K, H, W, inv2, maps, per_block);
    } else {
        size_t nb = (total + 255) / 256; if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(gauss_fwd_strided_kernel, dim3((unsigned)nb), dim3(256), 0, s, mu, B, K, H, W, inv2, maps, ldy);
    }
    return kpx_launch_status();
}

// dmu_x[b,k] = sum_{h,w} dmaps * g * 2*inv2*(x_w - mu_x); dmu_y likewise.  One block per image.
__global__ __launch_bounds__(256) void gauss_bwd_kernel(const float* __restrict__ dmaps, int lddy, const float* __restrict__ mu,
                                                        int K, int H, int W, float inv2, float* __restrict__ dmu) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int T = (256 / K) * K, groups = T / K;
    const int k = t % K, g0 = t / K;
    float sx = 0.f, sy = 0.f;
    if (t < T) {
        const float mx = mu[((size_t)b * K + k) * 2], my = mu[((size_t)b * K + k) * 2 + 1];
        for (int pix = g0; pix < H * W; pix += groups) {
            const int h = pix / W, w = pix - h * W;
            const float yv = kpx_linspace(h, H), xv = kpx_linspace(w, W);
            const float g = kpx_gauss(yv, xv, my, mx, inv2);
            const float d = dmaps[((size_t)b * H * W + pix) * lddy + k] * g * (2.0f * inv2);
            sx = fmaf(d, xv - mx, sx);
            sy = fmaf(d, yv - my, sy);
        }
    }
    __shared__ float sm[2][256];
    sm[0][t] = sx; sm[1][t] = sy;
    __syncthreads();
    if (t < K) {
        float ax = 0.f, ay = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) { ax += sm[0][g2 * K + t]; ay += sm[1][g2 * K + t]; }
        dmu[((size_t)b * K + t) * 2] = ax;
        dmu[((size_t)b * K + t) * 2 + 1] = ay;
    }
}
extern "C" int kpx_gaussian_maps_bwd_f32(const float* dmaps, int lddy, const float* mu, int B, int K, int H, int W, double inv_std,
                                         float* dmu, void* stream) {
    if (!dmaps || !mu || !dmu || B <= 0 || K <= 0 || K > 256 || H < 2 || W < 2 || lddy < K) return KPX_EINVAL;
    const float inv2 = (float)(inv_std * inv_std);
    hipLaunchKernelGGL(gauss_bwd_kernel, dim3(B), dim3(256), 0, kpx_stream(stream), dmaps, lddy, mu, K, H, W, inv2, dmu);
    return kpx_launch_status();
}
