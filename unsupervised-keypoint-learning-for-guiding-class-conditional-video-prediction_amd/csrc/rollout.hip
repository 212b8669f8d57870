// Kernels used only by the evaluate.py / FinalModel rollout: LSTM cell pointwise part, batch tiling,
// and the tiled head blend with clip_by_value.  The dense layers of vae_decoder (fully_connected, LSTMCell matmul, to_coord)
// run on the implicit-GEMM conv kernel as 1x1 convolutions over a [B,1,1,In] tensor.
#include "kpx_common.h"

// tf.nn.rnn_cell.LSTMCell (reference models/networks/layers.py:17-21, no peepholes, forget_bias = 1.0):
//   gates = [x, h] @ kernel + bias, split as i, j, f, o ;  c' = c*sigmoid(f + forget_bias) + sigmoid(i)*tanh(j) ;  h' = tanh(c')*sigmoid(o)
__global__ __launch_bounds__(256) void lstm_pointwise_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                             float forget_bias, float* __restrict__ c_out, float* __restrict__ h_out,
                                                             int B, int U) {
    const int total = B * U;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int b = idx / U, u = idx - b * U;
        const float* gr = gates + (size_t)b * 4 * U;
        const float i = gr[u], j = gr[U + u], f = gr[2 * U + u], o = gr[3 * U + u];
        const float sig_i = 1.0f / (1.0f + expf(-i)), sig_f = 1.0f / (1.0f + expf(-(f + forget_bias))), sig_o = 1.0f / (1.0f + expf(-o));
        const float c = c_prev[idx] * sig_f + sig_i * tanhf(j);
        c_out[idx] = c;
        h_out[idx] = tanhf(c) * sig_o;
    }
}
extern "C" int kpx_lstm_pointwise_f32(const float* gates, const float* c_prev, float forget_bias, float* c_out, float* h_out,
                                      int B, int U, void* stream) {
    if (!gates || !c_prev || !c_out || !h_out || B <= 0 || U <= 0) return KPX_EINVAL;
    int nb = (B * U + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(nb), dim3(256), 0, kpx_stream(stream), gates, c_prev, forget_bias, c_out, h_out, B, U);
    return kpx_launch_status();
}

// tf.expand_dims + tf.tile([1,T,1,1,1]) + reshape (reference models/final_model.py:58-66,85-87):
//   dst[((b*T + t)*pix + p)*lddst + c] = src[(b*pix + p)*ldsrc + c]
__global__ __launch_bounds__(256) void tile_batch_kernel(const float* __restrict__ src, int ldsrc, int B, int T, int pix, int C4,
                                                         float* __restrict__ dst, int lddst) {
    const size_t total = (size_t)B * T * pix * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        size_t r = i / C4;
        const int p = (int)(r % pix); r /= pix;
        const int b = (int)(r / T);
        *reinterpret_cast<f32x4*>(dst + (r * pix + p) * lddst + c) = *reinterpret_cast<const f32x4*>(src + ((size_t)b * pix + p) * ldsrc + c);
    }
}
__global__ __launch_bounds__(256) void tile_batch_scalar_kernel(const float* __restrict__ src, int ldsrc, int B, int T, int pix, int C,
                                                                float* __restrict__ dst, int lddst) {
    const size_t total = (size_t)B * T * pix * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int p = (int)(r % pix); r /= pix;
        const int b = (int)(r / T);
        dst[(r * pix + p) * lddst + c] = src[((size_t)b * pix + p) * ldsrc + c];
    }
}
extern "C" int kpx_tile_batch_f32(const float* src, int ldsrc, int B, int T, int pix, int C, float* dst, int lddst, void* stream) {
    if (!src || !dst || B <= 0 || T <= 0 || pix <= 0 || C <= 0 || ldsrc < C || lddst < C) return KPX_EINVAL;
    const bool vec = (C % 4 == 0) && (ldsrc % 4 == 0) && (lddst % 4 == 0) && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const size_t items = (size_t)B * T * pix * (vec ? C / 4 : C);
    size_t nb = (items + 255) / 256; if (nb > 2048) nb = 2048;
    if (vec) hipLaunchKernelGGL(tile_batch_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), src, ldsrc, B, T, pix, C / 4, dst, lddst);
    else hipLaunchKernelGGL(tile_batch_scalar_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), src, ldsrc, B, T, pix, C, dst, lddst);
    return kpx_launch_status();
}

// final = tiled_im*mask + crude*(1-mask), then clip_by_value(crude, -1, 1), clip_by_value(final, -1, 1)
// (reference models/final_model.py:95-99); frame f uses image f / T.
__global__ __launch_bounds__(256) void head_blend_tiled_kernel(const float* __restrict__ im, const float* __restrict__ raw4, size_t P, int HW, int T,
                                                               int clip, float* __restrict__ fin, float* __restrict__ crude, float* __restrict__ mask) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        const size_t f = p / HW;
        const size_t ip = (f / T) * HW + (p - f * HW);
        const f32x4 r = reinterpret_cast<const f32x4*>(raw4)[p];
        const float m = 1.0f / (1.0f + expf(-r[3]));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = im[ip * 3 + j] * m + r[j] * (1.0f - m), c = r[j];
            if (clip) { v = fminf(fmaxf(v, -1.0f), 1.0f); c = fminf(fmaxf(c, -1.0f), 1.0f); }
            fin[p * 3 + j] = v;
            if (crude) crude[p * 3 + j] = c;
        }
        if (mask) mask[p] = m;
    }
}
extern "C" int kpx_head_blend_tiled_fwd_f32(const float* im, const float* raw4, size_t P, int HW, int T, int clip,
                                            float* final_out, float* crude_out, float* mask_out, void* stream) {
    if (!im || !raw4 || !final_out || HW <= 0 || T <= 0 || (((uintptr_t)raw4) & 15)) return KPX_EINVAL;
    if (P == 0) return 0;
    size_t nb = (P + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(head_blend_tiled_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), im, raw4, P, HW, T, clip, final_out, crude_out, mask_out);
    return kpx_launch_status();
}

// ------------------------------------------------------------------------------------------ stage-2 training (motion generator)
// Backward of one LSTMCell step (models/networks/layers.py:17-21; forward above).  Everything is recomputed from the saved gate
// pre-activations and c_prev:  dh' and dc' (from the next step) -> dgates [B,4U] (i,j,f,o order) and dc_prev.
__global__ __launch_bounds__(256) void lstm_pointwise_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                                 const float* __restrict__ dh, const float* __restrict__ dc_in, float forget_bias,
                                                                 float* __restrict__ dgates, float* __restrict__ dc_prev, int B, int U) {
    const int total = B * U;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int b = idx / U, u = idx - b * U;
        const float* gr = gates + (size_t)b * 4 * U;
        const float i = gr[u], j = gr[U + u], f = gr[2 * U + u], o = gr[3 * U + u];
        const float si = 1.0f / (1.0f + expf(-i)), sf = 1.0f / (1.0f + expf(-(f + forget_bias))), so = 1.0f / (1.0f + expf(-o));
        const float tj = tanhf(j), cp = c_prev ? c_prev[idx] : 0.f;
        const float c = cp * sf + si * tj, tc = tanhf(c);
        const float dhv = dh[idx];
        const float dct = (dc_in ? dc_in[idx] : 0.f) + dhv * so * (1.0f - tc * tc);
        float* dg = dgates + (size_t)b * 4 * U;
        dg[u] = dct * tj * si * (1.0f - si);
        dg[U + u] = dct * si * (1.0f - tj * tj);
        dg[2 * U + u] = dct * cp * sf * (1.0f - sf);
        dg[3 * U + u] = dhv * tc * so * (1.0f - so);
        dc_prev[idx] = dct * sf;
    }
}
extern "C" int kpx_lstm_pointwise_bwd_f32(const float* gates, const float* c_prev, const float* dh, const float* dc_in, float forget_bias,
                                          float* dgates, float* dc_prev, int B, int U, void* stream) {
    if (!gates || !dh || !dgates || !dc_prev || B <= 0 || U <= 0) return KPX_EINVAL;
    int nb = (B * U + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3(nb), dim3(256), 0, kpx_stream(stream), gates, c_prev, dh, dc_in, forget_bias, dgates, dc_prev, B, U);
    return kpx_launch_status();
}

// Reparameterisation + KL term of the sequence VAE (models/motion_generator_model.py:146, :291-293) on logit = [mu | stddev] [B,2V]:
//   z = mu + stddev * eps ;  kl = mean_b( 0.5 * sum_v( mu^2 + s^2 - log(1e-8 + s^2) - 1 ) )
__global__ __launch_bounds__(256) void vae_sample_kl_fwd_kernel(const float* __restrict__ logit, const float* __restrict__ eps, float* __restrict__ z,
                                                                float* __restrict__ kl, int B, int V) {
    double s = 0.0;
    for (int idx = threadIdx.x; idx < B * V; idx += 256) {
        const int b = idx / V, v = idx - b * V;
        const float mu = logit[(size_t)b * 2 * V + v], sd = logit[(size_t)b * 2 * V + V + v];
        z[idx] = fmaf(sd, eps[idx], mu);
        s += (double)(mu * mu + sd * sd - logf(1e-8f + sd * sd) - 1.0f);
    }
    s = kpx_wave_sum_d(s);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *kl = (float)(0.5 * (sm[0] + sm[1] + sm[2] + sm[3]) / (double)B);
}
extern "C" int kpx_vae_sample_kl_fwd_f32(const float* logit, const float* eps, float* z, float* kl_out, int B, int V, void* stream) {
    if (!logit || !eps || !z || !kl_out || B <= 0 || V <= 0) return KPX_EINVAL;
    hipLaunchKernelGGL(vae_sample_kl_fwd_kernel, dim3(1), dim3(256), 0, kpx_stream(stream), logit, eps, z, kl_out, B, V);
    return kpx_launch_status();
}
// dlogit from dz [B,V] and the scalar gradient of the KL term (device scalar times host scale)
__global__ __launch_bounds__(256) void vae_sample_kl_bwd_kernel(const float* __restrict__ logit, const float* __restrict__ eps, const float* __restrict__ dz,
                                                                const float* gkl_dev, float gkl_host, float* __restrict__ dlogit, int B, int V) {
    const float gk = gkl_host * (gkl_dev ? *gkl_dev : 1.0f) / (float)B;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * V; idx += gridDim.x * 256) {
        const int b = idx / V, v = idx - b * V;
        const float mu = logit[(size_t)b * 2 * V + v], sd = logit[(size_t)b * 2 * V + V + v];
        const float dzv = dz ? dz[idx] : 0.f;
        dlogit[(size_t)b * 2 * V + v] = dzv + gk * mu;
        dlogit[(size_t)b * 2 * V + V + v] = dzv * eps[idx] + gk * (sd - sd / (1e-8f + sd * sd));
    }
}
extern "C" int kpx_vae_sample_kl_bwd_f32(const float* logit, const float* eps, const float* dz, const float* gkl_dev, float gkl_host,
                                         float* dlogit, int B, int V, void* stream) {
    if (!logit || !eps || !dlogit || B <= 0 || V <= 0) return KPX_EINVAL;
    int nb = (B * V + 255) / 256; if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(vae_sample_kl_bwd_kernel, dim3(nb), dim3(256), 0, kpx_stream(stream), logit, eps, dz, gkl_dev, gkl_host, dlogit, B, V);
    return kpx_launch_status();
}

// Whole-sequence LSTM layer (tf.nn.dynamic_rnn / the unrolled cell calls of networks/__init__.py:105-138), zero initial state.
// The time loop lives here, not in the host language: per step two channel copies ([x_t, h_{t-1}] -> xin_t), the gate GEMM on
// the conv kernel (a 1x1 "convolution" over [B,1,1,In+U]) and the gate math -- 4 launches per step issued back to back from C.
//   x [T,B,In] ; kernel [In+U, 4U] ; bias [4U] ; saved for the backward: xin [T,B,In+U], gates [T,B,4U], cs [T,B,U] ; out hs [T,B,U]
extern "C" int kpx_lstm_layer_fwd_f32(const float* x, int T, int B, int In, const float* kernel, const float* bias, int U,
                                      float* xin, float* gates, float* cs, float* hs, const float* zeros_bu,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !kernel || !bias || !xin || !gates || !cs || !hs || !zeros_bu || T <= 0 || B <= 0 || In <= 0 || U <= 0) return KPX_EINVAL;
    const int W = In + U;
    for (int s = 0; s < T; ++s) {
        const float* h_prev = s ? hs + (size_t)(s - 1) * B * U : zeros_bu;
        const float* c_prev = s ? cs + (size_t)(s - 1) * B * U : zeros_bu;
        float* xs = xin + (size_t)s * B * W;
        float* gs = gates + (size_t)s * B * 4 * U;
        int rc = kpx_copy_channels_f32(x + (size_t)s * B * In, In, xs, W, (size_t)B, In, stream);
        if (!rc) rc = kpx_copy_channels_f32(h_prev, U, xs + In, W, (size_t)B, U, stream);
        if (!rc) rc = kpx_conv2d_fwd_f32(xs, B, 1, 1, W, W, kernel, 1, 1, bias, gs, 1, 1, 4 * U, 4 * U, 1, 0, 0, KPX_ACT_NONE, KPX_ARITH_F32, workspace, workspace_bytes, stream);
        if (!rc) rc = kpx_lstm_pointwise_f32(gs, c_prev, 1.0f, cs + (size_t)s * B * U, hs + (size_t)s * B * U, B, U, stream);
        if (rc) return rc;
    }
    return 0;
}

// Backward through time of the layer above: dhs [T,B,U] -> dgates [T,B,4U] (for the single batched weight-gradient GEMM the caller
// runs afterwards) and, when dx != NULL, dx [T,B,In].  Scratch: dxin [B,In+U], dh [B,U], dc0 / dc1 [B,U] (dc0 must be zero-filled).
extern "C" int kpx_lstm_layer_bwd_f32(const float* dhs, int T, int B, int In, const float* kernel, int U,
                                      const float* gates, const float* cs, float* dgates, float* dx,
                                      float* dxin, float* dh, float* dc0, float* dc1,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!dhs || !kernel || !gates || !cs || !dgates || !dxin || !dh || !dc0 || !dc1 || T <= 0 || B <= 0 || In <= 0 || U <= 0) return KPX_EINVAL;
    const int W = In + U;
    float* dc_in = dc0; float* dc_out = dc1;
    for (int s = T - 1; s >= 0; --s) {
        const float* dh_s = dhs + (size_t)s * B * U;
        int rc = 0;
        if (s != T - 1) {                              // dh_s = dhs[s] + (dxin of step s+1)[:, In:]
            rc = kpx_copy_channels_f32(dxin + In, W, dh, U, (size_t)B, U, stream);
            if (!rc) rc = kpx_axpy_f32(dh, dh_s, (size_t)B * U, 1.0f, stream);
            dh_s = dh;
        }
        float* dg = dgates + (size_t)s * B * 4 * U;
        if (!rc) rc = kpx_lstm_pointwise_bwd_f32(gates + (size_t)s * B * 4 * U, s ? cs + (size_t)(s - 1) * B * U : nullptr, dh_s, dc_in, 1.0f,
                                                 dg, dc_out, B, U, stream);
        float* tmp = dc_in; dc_in = dc_out; dc_out = tmp;
        if (!rc && (s || dx)) {
            rc = kpx_conv2d_dgrad_f32(dg, B, 1, 1, 4 * U, 4 * U, kernel, 1, 1, dxin, 1, 1, W, W, 1, 0, 0, KPX_ARITH_F32, workspace, workspace_bytes, stream);
            if (!rc && dx) rc = kpx_copy_channels_f32(dxin, W, dx + (size_t)s * B * In, In, (size_t)B, In, stream);
        }
        if (rc) return rc;
    }
    return 0;
}
