// Loss reductions (perceptual L1, sigmoid cross-entropy) and the fused TF-style Adam update.
//   L1   : detector_translator_model.py:280-284  mean(|f_gt - f_pred|) on the two batch halves of a VGG feature
//   xent : detector_translator_model.py:249-254,265-267  reduce_mean(sigmoid_cross_entropy_with_logits)
//   Adam : detector_translator_model.py:198-202  tf.train.AdamOptimizer(lr, 0.5, 0.999) -> ApplyAdam arithmetic
#include "kpx_common.h"
#include <string.h>

__global__ __launch_bounds__(256) void l1_pair_partial_kernel(const float* __restrict__ f, size_t half, double* __restrict__ part) {
    double s = 0.0;
    const size_t n4 = half / 4;
    const f32x4* a = reinterpret_cast<const f32x4*>(f);
    const f32x4* b = reinterpret_cast<const f32x4*>(f + half);
    const bool vec = (half % 4 == 0) && ((((uintptr_t)f) & 15) == 0);
    if (vec) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const f32x4 x = a[i], y = b[i];
            s += (double)(fabsf(x[0] - y[0]) + fabsf(x[1] - y[1])) + (double)(fabsf(x[2] - y[2]) + fabsf(x[3] - y[3]));
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (size_t)gridDim.x * 256)
            s += (double)fabsf(f[i] - f[half + i]);
    }
    s = kpx_wave_sum_d(s);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void l1_pair_finalize_kernel(const double* part, int nb, double count, float* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
    s = kpx_wave_sum_d(s);
    if (threadIdx.x == 0) *out = (float)(s / count);
}
extern "C" int kpx_l1_pair_fwd_f32(const float* f, size_t half, float* loss_out, void* scratch, void* stream) {
    if (!f || !loss_out || !scratch || half == 0) return KPX_EINVAL;
    size_t nb = (half / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    hipStream_t s = kpx_stream(stream);
    hipLaunchKernelGGL(l1_pair_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, f, half, (double*)scratch);
    int rc = kpx_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(l1_pair_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)scratch, (int)nb, (double)half, loss_out);
    return kpx_launch_status();
}

__global__ __launch_bounds__(256) void l1_pair_bwd_kernel(const float* __restrict__ f, size_t half, const float* gdev, float ghost, float* __restrict__ dpred) {
    const float g = ghost * (gdev ? *gdev : 1.0f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (size_t)gridDim.x * 256) {
        const float d = f[i] - f[half + i];                 // gt - pred; d|d|/dpred = -sign(d)
        dpred[i] = d > 0.f ? -g : (d < 0.f ? g : 0.f);
    }
}
extern "C" int kpx_l1_pair_bwd_f32(const float* f, size_t half, const float* gscale_dev, float gscale_host, float* dpred, void* stream) {
    if (!f || !dpred || half == 0) return KPX_EINVAL;
    size_t nb = (half + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(l1_pair_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), f, half, gscale_dev, gscale_host, dpred);
    return kpx_launch_status();
}

// sigmoid_cross_entropy_with_logits(z, x) = max(x,0) - x*z + log1p(exp(-|x|))
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ x, size_t n0, float z0, size_t n1, float z1, float* out) {
    double s0 = 0.0, s1 = 0.0;
    for (size_t i = threadIdx.x; i < n0 + n1; i += 256) {
        const float v = x[i], z = i < n0 ? z0 : z1;
        const float l = fmaxf(v, 0.f) - v * z + log1pf(expf(-fabsf(v)));
        if (i < n0) s0 += (double)l; else s1 += (double)l;
    }
    s0 = kpx_wave_sum_d(s0); s1 = kpx_wave_sum_d(s1);
    __shared__ double sm[2][4];
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s0; sm[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3], b = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
        const float m0 = n0 ? (float)(a / (double)n0) : 0.f, m1 = n1 ? (float)(b / (double)n1) : 0.f;
        out[0] = m0 + m1;
        out[1] = m0;
        out[2] = m1;
    }
}
extern "C" int kpx_sigmoid_xent_fwd_f32(const float* logits, size_t n0, float label0, size_t n1, float label1, float* loss_out, void* stream) {
    if (!logits || !loss_out || n0 + n1 == 0) return KPX_EINVAL;
    hipLaunchKernelGGL(xent_fwd_kernel, dim3(1), dim3(256), 0, kpx_stream(stream), logits, n0, label0, n1, label1, loss_out);
    return kpx_launch_status();
}
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ x, size_t n0, float z0, size_t n1, float z1,
                                                       const float* gdev, float ghost, float* __restrict__ dx) {
    const float g = ghost * (gdev ? *gdev : 1.0f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        const float sg = 1.0f / (1.0f + expf(-v));
        dx[i] = i < n0 ? (sg - z0) * (g / (float)n0) : (sg - z1) * (g / (float)n1);
    }
}
extern "C" int kpx_sigmoid_xent_bwd_f32(const float* logits, size_t n0, float label0, size_t n1, float label1,
                                        const float* gscale_dev, float gscale_host, float* dlogits, void* stream) {
    if (!logits || !dlogits || n0 + n1 == 0) return KPX_EINVAL;
    size_t nb = (n0 + n1 + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(xent_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), logits, n0, label0, n1, label1,
                       gscale_dev, gscale_host, dlogits);
    return kpx_launch_status();
}

// ApplyAdam (TF 1.12): m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= m*alpha/(sqrt(v)+eps)
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, size_t n, float alpha, float omb1, float omb2, float eps, float gs) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * gs;
            mm[j] = __fadd_rn(mm[j], __fmul_rn(__fsub_rn(gr, mm[j]), omb1));
            vv[j] = __fadd_rn(vv[j], __fmul_rn(__fsub_rn(__fmul_rn(gr, gr), vv[j]), omb2));
            pp[j] = __fsub_rn(pp[j], __fdiv_rn(__fmul_rn(mm[j], alpha), __fadd_rn(__fsqrt_rn(vv[j]), eps)));
        }
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gr = g[i] * gs;
        const float mn = __fadd_rn(m[i], __fmul_rn(__fsub_rn(gr, m[i]), omb1));
        const float vn = __fadd_rn(v[i], __fmul_rn(__fsub_rn(__fmul_rn(gr, gr), v[i]), omb2));
        m[i] = mn; v[i] = vn;
        p[i] = __fsub_rn(p[i], __fdiv_rn(__fmul_rn(mn, alpha), __fadd_rn(__fsqrt_rn(vn), eps)));
    }
}
// The same update with the step size read from DEVICE memory: a captured HIP graph replays the launch with frozen arguments, and alpha
// (learning-rate decay, bias correction) is the one number of the step that changes between replays.
__global__ __launch_bounds__(256) void adam_tf_dev_alpha_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                                float* __restrict__ v, size_t n, const float* __restrict__ alpha_dev,
                                                                float omb1, float omb2, float eps, float gs) {
    const float alpha = *alpha_dev;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * gs;
            mm[j] = __fadd_rn(mm[j], __fmul_rn(__fsub_rn(gr, mm[j]), omb1));
            vv[j] = __fadd_rn(vv[j], __fmul_rn(__fsub_rn(__fmul_rn(gr, gr), vv[j]), omb2));
            pp[j] = __fsub_rn(pp[j], __fdiv_rn(__fmul_rn(mm[j], alpha), __fadd_rn(__fsqrt_rn(vv[j]), eps)));
        }
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gr = g[i] * gs;
        const float mn = __fadd_rn(m[i], __fmul_rn(__fsub_rn(gr, m[i]), omb1));
        const float vn = __fadd_rn(v[i], __fmul_rn(__fsub_rn(__fmul_rn(gr, gr), v[i]), omb2));
        m[i] = mn; v[i] = vn;
        p[i] = __fsub_rn(p[i], __fdiv_rn(__fmul_rn(mn, alpha), __fadd_rn(__fsqrt_rn(vn), eps)));
    }
}
extern "C" int kpx_adam_tf_flat_dev_alpha_f32(float* p, const float* g, float* m, float* v, size_t n,
                                              const float* alpha_dev, float beta1, float beta2, float eps, float gscale, void* stream) {
    if (!p || !g || !m || !v || !alpha_dev || ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15)) return KPX_EINVAL;
    if (n == 0) return 0;
    size_t nb = (n / 4 + 255) / 256; if (nb < 1) nb = 1; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adam_tf_dev_alpha_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), p, g, m, v, n, alpha_dev,
                       1.0f - beta1, 1.0f - beta2, eps, gscale);
    return kpx_launch_status();
}

extern "C" int kpx_adam_tf_flat_f32(float* p, const float* g, float* m, float* v, size_t n,
                                    float alpha, float beta1, float beta2, float eps, float gscale, void* stream) {
    if (!p || !g || !m || !v || ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15)) return KPX_EINVAL;
    if (n == 0) return 0;
    size_t nb = (n / 4 + 255) / 256; if (nb < 1) nb = 1; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)nb), dim3(256), 0, kpx_stream(stream), p, g, m, v, n, alpha,
                       1.0f - beta1, 1.0f - beta2, eps, gscale);
    return kpx_launch_status();
}

extern "C" int kpx_abi_version(void) { return KPX_ABI_VERSION; }

// ------------------------------------------------------------------------------------------ host utility
// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slice-by-8: the checksum TensorFlow's V2 checkpoint bundles carry for
// every tensor and table block (tensor_bundle.cc / lib/io/format.cc); used by the bundle reader / writer (tf_bundle.py).
// Pure host code: no device pointer is touched.
static uint32_t kpx_crc_tab[8][256];
static bool kpx_crc_ready = false;
static void kpx_crc_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        kpx_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) kpx_crc_tab[t][i] = (kpx_crc_tab[t - 1][i] >> 8) ^ kpx_crc_tab[0][kpx_crc_tab[t - 1][i] & 0xff];
    kpx_crc_ready = true;
}
extern "C" unsigned int kpx_crc32c_host(unsigned int crc, const void* data, size_t n) {
    if (!kpx_crc_ready) kpx_crc_init();
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7)) { c = kpx_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t v; memcpy(&v, p, 8);
        v ^= c;
        c = kpx_crc_tab[7][v & 0xff] ^ kpx_crc_tab[6][(v >> 8) & 0xff] ^ kpx_crc_tab[5][(v >> 16) & 0xff] ^ kpx_crc_tab[4][(v >> 24) & 0xff] ^
            kpx_crc_tab[3][(v >> 32) & 0xff] ^ kpx_crc_tab[2][(v >> 40) & 0xff] ^ kpx_crc_tab[1][(v >> 48) & 0xff] ^ kpx_crc_tab[0][(v >> 56) & 0xff];
        p += 8; n -= 8;
    }
    while (n--) c = kpx_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}
