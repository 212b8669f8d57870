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
