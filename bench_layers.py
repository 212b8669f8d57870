#!/usr/bin/env python3
"""Per-layer microbenchmark of the conv kernels on the shapes of one detector_translator train step
(B=32, 128x128, K=15).  Prints time, TFLOP/s and fraction of the 157.3 TF fp32-MFMA peak for
forward / dgrad / wgrad of every distinct layer, weighted by how often the step runs it.

    python bench_layers.py [--batch 32] [--dtype bf16]

--dtype bf16: the same table for the bf16 configuration (bf16 activation tensors; image-input layers take fp32 images and produce bf16, the
translator's 4-channel head and D_logit produce fp32) -- each layer on the kernel `ops` picks for it there, fraction against the bf16 peak.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from kpx_amd import ops  # noqa: E402

PEAK = 157.3


def layers(B, K=15):
    L = []   # (name, N, H, Cin, Cout, k, stride, pad, cin_ld, need_dgrad, need_wgrad, count_fwd, count_dgrad, count_wgrad)

    def enc(prefix, n, first_dgrad):
        chans = [(3, 32, 7, 1, 128), (32, 32, 3, 1, 128), (32, 64, 3, 2, 128), (64, 64, 3, 1, 64), (64, 128, 3, 2, 64),
                 (128, 128, 3, 1, 32), (128, 256, 3, 2, 32), (256, 256, 3, 1, 16)]
        for i, (ci, co, k, s, h) in enumerate(chans, 1):
            L.append(('%s/conv_%d' % (prefix, i), n, h, ci, co, k, s, 0, ci, 1, 0 if i == 1 else 1, 1))
    enc('image_encoder', B, False)
    enc('pose_encoder/enc', 2 * B, False)
    for name, h, ci, co in [('conv_1_0', 16, 256, 128), ('conv_1_1', 16, 128, 128), ('conv_2_0', 16, 128, 128), ('conv_2_1', 16, 128, 128),
                            ('conv_3_0', 32, 256, 64), ('conv_3_1', 32, 64, 64), ('conv_4_0', 32, 64, 64), ('conv_4_1', 32, 64, 64),
                            ('conv_5_0', 64, 128, 32), ('conv_5_1', 64, 32, 32), ('conv_6_0', 64, 32, 32), ('conv_6_1', 64, 32, 32),
                            ('conv_7_0', 128, 64, 16), ('conv_7_1', 128, 16, 16)]:
        L.append(('pose/' + name, 2 * B, h, ci, co, 3, 1, 0, ci, 1, 1, 1))
    L.append(('pose/conv_0', 2 * B, 128, 16, K, 1, 1, 0, 16, 1, 1, 1))
    cj = 128 + 2 * K
    for name, h, ci, co, ld in [('conv_1_0', 32, cj, 256, (cj + 3) // 4 * 4), ('conv_1_1', 32, 256, 256, 256), ('conv_2_0', 32, 256, 256, 256),
                                ('conv_2_1', 32, 256, 256, 256), ('conv_3_0', 64, 256, 128, 256), ('conv_3_1', 64, 128, 128, 128),
                                ('conv_4_0', 64, 128, 128, 128), ('conv_4_1', 64, 128, 128, 128), ('conv_5_0', 128, 128, 64, 128),
                                ('conv_5_1', 128, 64, 64, 64), ('conv_6_0+1', 128, 64, 4, 64)]:
        L.append(('translator/' + name, B, h, ci, co, 3, 1, 0, ld, 1, 1, 1))
    h, c, ch = 128, 3, 64
    for i in range(6):
        ho = -(-(h + 2) // 2)
        # D-run: 2B fwd/dgrad/wgrad ; G-run: B fwd/dgrad  -> benchmark at 2B, weight 1.5 / 1.5 / 1
        L.append(('img_discr/conv_%d' % i, 2 * B, h, c, ch, 4, 2, 1, c, 1.5, 1.5 if i else 0.5, 1))
        h, c, ch = ho, ch, ch * 2
    L.append(('img_discr/D_logit', 2 * B, h, c, 1, 3, 1, 1, c, 1.5, 1.5, 1))
    vgg = [('conv1_1', 128, 3, 64), ('conv1_2', 128, 64, 64), ('conv2_1', 64, 64, 128), ('conv2_2', 64, 128, 128), ('conv3_1', 32, 128, 256),
           ('conv3_2', 32, 256, 256), ('conv3_3', 32, 256, 256), ('conv3_4', 32, 256, 256), ('conv4_1', 16, 256, 512), ('conv4_2', 16, 512, 512),
           ('conv4_3', 16, 512, 512), ('conv4_4', 16, 512, 512), ('conv5_1', 8, 512, 512), ('conv5_2', 8, 512, 512), ('conv5_3', 8, 512, 512),
           ('conv5_4', 8, 512, 512)]
    for name, h, ci, co in vgg:
        L.append(('vgg/' + name, 2 * B, h, ci, co, 3, 1, 0, ci, 1, 0.5, 0))   # dgrad runs on B images only -> weight .5 at 2B
    return L


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--filter', default='', help='comma-separated substrings of layer names')
    ap.add_argument('--unregistered', action='store_true', help='3x3 layers through the plain C entry (filter transformed inside the call, F(2x2,3x3) only)')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    ops.set_compute_dtype(args.dtype)
    b16 = args.dtype == 'bf16'
    peak = 2500.0 if b16 else PEAK
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    flops_tot = 0.0
    print('%-36s %5s %4s %5s %5s k s | %9s %6s | %9s %6s | %9s %6s' % ('layer', 'N', 'H', 'Cin', 'Cout', 'fwd ms', 'TF', 'dgrad ms', 'TF', 'wgrad ms', 'TF'))
    for (name, n, h, ci, co, k, s, pad, ld, wf, wd, ww) in layers(args.batch):
        if args.filter and not any(f in name for f in args.filter.split(',')):
            continue
        if b16 and ci > 4 and ld % 8:
            ld = (ci + 31) // 32 * 32                     # (the bf16 joint buffer: 158 channels in a 160-wide pixel)
        x = torch.randn(n, h, h, ld, device=dev)
        if ld > ci:
            x[..., ci:] = 0
        adt = torch.bfloat16 if (b16 and ci > 4) else torch.float32            # images stay fp32
        ydt = torch.bfloat16 if (b16 and co > 4) else torch.float32            # the 4-channel head and D_logit stay fp32
        x = x.to(adt)
        w = torch.randn(k, k, ci, co, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        pt, _, ho = ops.same_pad(h + 2 * pad, k, s)
        pad_t = pad + pt
        y = torch.empty(n, ho, ho, co, dtype=ydt, device=dev)
        dy = torch.randn(n, ho, ho, co, device=dev).to(ydt)
        dx = torch.empty(n, h, h, ld, dtype=adt, device=dev)
        dw = torch.empty_like(w)
        flops = 2.0 * n * ho * ho * co * k * k * ci
        # the filter is registered under the layer's variable name, so that the 3x3 layers run on the kernel the train step picks for
        # them (pre-transformed filters; F(4x4,3x3) or F(2x2,3x3) by the layer's f43_fwd attribute as networks.py declares it)
        vname = name.replace('pose/', 'pose_encoder/').replace('pose_encoder/enc/', 'pose_encoder/encoder/')
        f43 = not vname.startswith(('pose_encoder', 'image_encoder', 'translator/conv_1', 'translator/conv_2'))
        keys = ops.register_constant_filter(w, vname, f43_fwd=f43) if (k == 3 and not args.unregistered) else []
        used = ops.conv_kernel_uses['wino43']
        try:
            tf = timeit(lambda: ops.conv_fwd_raw(x, ld, ci, w, b, y, co, s, pad_t, pad_t, 0))
            f43_f = ops.conv_kernel_uses['wino43'] > used
            used = ops.conv_kernel_uses['wino43']
            td = timeit(lambda: ops.conv_dgrad_raw(dy, co, w, dx, ld, ci, s, pad_t, pad_t)) if wd else 0.0
            f43_d = ops.conv_kernel_uses['wino43'] > used
        finally:
            ops.release_filters(keys)
        name = name + (' [F43 ' + ('f' if f43_f else '-') + ('d' if f43_d else '-') + ']' if (f43_f or f43_d) else '')
        tw = timeit(lambda: ops.conv_wgrad_raw(x, ld, ci, dy, co, dw, s, pad_t, pad_t)) if ww else 0.0
        tot['fwd'] += wf * tf; tot['dgrad'] += wd * td; tot['wgrad'] += ww * tw
        flops_tot += flops * (wf + wd + ww)
        f = lambda t: flops / (t * 1e-3) / 1e12 if t else 0.0
        print('%-36s %5d %4d %5d %5d %d %d | %9.3f %6.1f | %9.3f %6.1f | %9.3f %6.1f' % (name, n, h, ci, co, k, s, tf, f(tf), td, f(td), tw, f(tw)))
    t = sum(tot.values())
    print('weighted per-step totals: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms, all %.2f ms; %.1f TF = %.1f%% of the %s MFMA peak%s'
          % (tot['fwd'], tot['dgrad'], tot['wgrad'], t, flops_tot / (t * 1e-3) / 1e12, 100 * flops_tot / (t * 1e-3) / 1e12 / peak, 'bf16' if b16 else 'fp32',
             '; layers routed through fp32 kernels between conversions: %s' % dict(ops.fallback_uses) if b16 else ''))


if __name__ == '__main__':
    main()
