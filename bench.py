#!/usr/bin/env python3
"""Benchmark of the detector_translator train step (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus N --steps 10 --warmup 3            (N > 1: starts one child process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = the reference's D-run + G-run (models/detector_translator_model.py:79-117) on one synthetic batch of
image pairs: generator forward, discriminator update (Adam), generator update (VGG19 perceptual + adversarial loss,
Adam), BN moving-average update -- nothing skipped.  Workload at every N: BASELINE configs[1] per GPU (128x128, K=15,
fp32, batch 32 per GPU, weak scaling); gradients are all-reduced over RCCL once per bucket per update.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RES, K_PTS, BATCH = 128, 15, 32
# forward MACs per image pair -- (generator forward without / dead image_encoder conv_7/8 branch, one img_discr
# pass, VGG19 per image) at 128x128 K=15 and at the 256x256 K=40 generalisation (Appendix A, second table)
GMAC = {128: (11.370, 0.227, 1.384, 6.370), 256: (3.630 - 0.908 + 2 * 7.296 + 28.689, 0.908, 3.956, 25.480)}
# bench configurations: BASELINE.json configs[1] (the headline), [3] and [4]
CONFIGS = {'c1': dict(res=128, k=15, batch=32, idx=1, metric='detector_translator train frames/sec @128x128 K=15'),
           'c3': dict(res=256, k=40, batch=16, idx=3, metric='detector_translator train frames/sec @256x256 K=40'),
           'c4': dict(res=128, k=15, batch=64, idx=4, metric='evaluate.py rollout (detector -> motion generator -> translator) predicted frames/sec @128x128 K=15, 32 frames per image')}


def step_gmac_per_pair(res=128):
    """Algorithmic MACs of the restructured step per pair (shared generator forward; conv fwd = dgrad = wgrad):
    G fwd + bwd(dgrad+wgrad) ; D-run: 2 D fwd + 2 x (dgrad+wgrad) ; G-run: D fwd + D dgrad ; VGG: 2 fwd + 1 dgrad."""
    g_fwd, g_dead, d_pass, vgg_img = GMAC[res]
    gen = g_fwd * 3 + g_dead          # the dead image_encoder conv_7/8 branch runs forward only (BN moving stats)
    d_run = 2 * d_pass * 3
    g_adv = d_pass * 2
    vgg = vgg_img * 3
    return gen + d_run + g_adv + vgg


def time_kernel(fn, iters, warm=3):
    """Average device time of fn() in ms, HIP events on the stream the kernels are launched on (torch's current)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _time_conv_3_1(dev, pretransformed=False, name='translator/conv_3_1', f43_fwd=True, n=BATCH, h=64, c=128, act=0):
    """One 3x3 stride-1 c -> c layer on [n,h,h,c] (default: translator conv_3_1 at the bench batch).  f43_fwd: the layer attribute that selects
    F(4x4,3x3) or F(2x2,3x3) in the forward direction exactly as in the train step (ops.WINO43)."""
    from kpx_amd import ops
    x = torch.randn(n, h, h, c, device=dev)
    w = torch.randn(3, 3, c, c, device=dev) * 0.03
    b = torch.zeros(c, device=dev)
    y = torch.empty(n, h, h, c, device=dev)
    keys = ops.register_constant_filter(w, name, f43_fwd=f43_fwd) if pretransformed else []     # the train step runs the kernel on filters transformed once per update
    try:
        ms = time_kernel(lambda: ops.conv_fwd_raw(x, c, c, w, b, y, c, 1, 1, 1, act), iters=100 if n * h * h <= 64 * 128 * 128 else 30, warm=20 if n * h * h <= 64 * 128 * 128 else 6)
    finally:
        ops.release_filters(keys)
    return ms, 2.0 * 9 * c * c * h * h * n


PROFILE = 'r06'          # prefix of the committed rocprofv3 summaries under profiles/ the lines below point at (profiles/collect_r06.sh)


def _rocprof_avg_ms(kernel_substr, csv_name=PROFILE + '_roofline_only_kernel_stats.csv'):
    """Average kernel duration from the committed rocprofv3 --kernel-trace --stats summary of `bench.py --roofline-only` (same command, same
    kernels): the live HIP-event figure includes the launch-to-launch gap (~1.5-3 us), which matters for a 12 us kernel."""
    import csv
    try:
        for r in csv.DictReader(open(os.path.join(ROOT, 'profiles', csv_name))):
            if kernel_substr in r['Name']:
                return round(float(r['AverageNs']) / 1e6, 5), 'profiles/' + csv_name
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def _profile_stamp():
    """Commit the committed rocprofv3 summaries under profiles/ were collected at (profiles/<round>_commit.txt, written by the collect script)."""
    try:
        return open(os.path.join(ROOT, 'profiles', PROFILE + '_commit.txt')).read().strip()
    except OSError:
        return None


def _pmc_traffic(name):
    """HBM bytes per launch from a committed rocprofv3 PMC summary of this same kernel / shape (profiles/<name>); the
    counters cannot be collected inside a normal bench run, so the JSON line names the file they come from."""
    try:
        return int(json.load(open(os.path.join(ROOT, 'profiles', name)))['traffic_bytes_per_launch']), 'profiles/' + name
    except (OSError, KeyError, ValueError):
        return None, None


WINO43_PMC, WINO_PMC, DIRECT_PMC, RENDER_PMC = PROFILE + '_wino43_pmc.json', PROFILE + '_wino_pmc.json', PROFILE + '_direct_pmc.json', PROFILE + '_render_pmc.json'
GEMM3_PMC, BF16_PMC, WGRAD_PMC = PROFILE + '_gemm3_pmc.json', PROFILE + '_bf16_pmc.json', PROFILE + '_wgrad_pmc.json'
WGRAD16_PMC = PROFILE + '_wgrad_bf16_pmc.json'


def _roofline_wino(dev, name, kernel_substr, kernel_desc, reduction, pmc, f43_fwd=True, csv_name=None, bf16x3=None, **shape):
    """bf16x3: None = the kernel the train step uses for this launch (ops.WINO43B: the bf16x3 form of F(4x4,3x3) where it takes the shape);
    False = the fp32-MFMA F(4x4,3x3) kernel (ops.WINO43B off for this leg).  The bound is the pipe the kernel's multiplies run on: the fp32
    MFMA peak (157.3 TFLOP/s), or for the bf16x3 form the six-product bound of the bf16 pipe, 2 500 / 6 = 416.7 TFLOP/s of fp32-equivalent
    work -- `achieved` counts the multiplies the kernel EXECUTES (algorithmic / the Winograd reduction, stated separately) in both cases."""
    from kpx_amd import ops
    keep = ops.WINO43B
    if bf16x3 is False:
        ops.WINO43B = False
    try:
        used_b = ops.conv_kernel_uses['wino43b']
        ms, flops = _time_conv_3_1(dev, pretransformed=True, name=name, f43_fwd=f43_fwd, **shape)
        ran_b = ops.conv_kernel_uses['wino43b'] > used_b
    finally:
        ops.WINO43B = keep
    alg = flops / (ms * 1e-3) / 1e12
    ach = alg / reduction
    if ran_b:
        kernel_substr = 'conv_wino43b_kernel<0, false>'
        kernel_desc = kernel_desc.replace('conv_wino43_kernel<0, false> F(4x4,3x3)', 'conv_wino43b_kernel<0, false> F(4x4,3x3), transform-domain GEMMs as bf16x3 = fp32-equivalent on the bf16 pipe,')
        pmc = pmc.replace('_wino43_', '_wino43b_')
    peak = 416.7 if ran_b else 157.3
    traffic, src = _pmc_traffic(pmc)
    rp_ms, rp_src = _rocprof_avg_ms(kernel_substr, csv_name) if csv_name else _rocprof_avg_ms(kernel_substr)
    out = {'committed_profile': {'note': 'from files committed under profiles/, NOT measured in this run', 'collected_at_commit': _profile_stamp(),
                                 'rocprof_avg_launch_ms': rp_ms, 'rocprof_source': rp_src, 'traffic_source': src,
                                 'frac_at_rocprof_avg': round(flops / reduction / (rp_ms * 1e-3) / (peak * 1e12), 4) if rp_ms else None},
           'bound': 'mfma', 'kernel': kernel_desc,
           'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s' + (' (fp32-equivalent executed multiplies; bf16 dense peak 2500 / 6 products)' if ran_b else ''),
           'frac': round(ach / peak, 4),
           'traffic': traffic, 'traffic_source': src, 'avg_launch_ms': round(ms, 4),
           'flops_per_launch_executed': flops / reduction, 'flops_per_launch_algorithmic': flops, 'winograd_reduction': reduction,
           'algorithmic_tflops': round(alg, 2), 'algorithmic_frac': round(alg / peak, 4)}
    if ran_b:
        out['bf16_mfma_tflops_issued'] = round(6 * ach, 1)                 # six bf16 products per fp32-equivalent multiply
        out['frac_of_fp32_mfma_peak'] = round(ach / 157.3, 4)              # (what rounds 2-5 reported for the fp32-MFMA kernel on this launch: 0.47-0.48)
    return out


def roofline_conv(dev):
    """Dominant kernel: the fused Winograd F(4x4,3x3) conv (fp32 MFMA) on the translator's 3x3 128->128 layer at 64x64
    (conv_3_1 / 4_0 / 4_1, SURVEY Appendix A: 603 979 776 MAC per image), batch 32.  `achieved` / `frac` count the MFMA FLOPs the
    kernel EXECUTES (algorithmic / 4: 36 instead of 144 multiplies per 4x4 output tile) against the fp32 MFMA peak;
    `algorithmic_tflops` / `algorithmic_frac` count direct-convolution FLOPs and exceed 1."""
    return _roofline_wino(dev, 'translator/conv_3_1', 'conv_wino43_kernel<0, false>',
                          'conv_wino43_kernel<0, false> F(4x4,3x3) fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1), filter pre-transformed as in the train step',
                          4.0, WINO43_PMC)


def roofline_conv_f32mfma(dev):
    """The same launch on the fp32-MFMA F(4x4,3x3) kernel (csrc/conv_wino43.hip; the headline kernel of rounds 2-5, still the kernel of the
    packed 16x16 and ragged-channel launches): executed FLOPs against the fp32 MFMA peak."""
    return _roofline_wino(dev, 'translator/conv_3_1', 'conv_wino43_kernel<0, false>',
                          'conv_wino43_kernel<0, false> F(4x4,3x3) fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1), fp32 MFMA, filter pre-transformed',
                          4.0, WINO43_PMC, bf16x3=False)


def roofline_conv_c3(dev, batch=16):
    """--config c3 (BASELINE configs[3], 256x256, K=40, 16 pairs per GPU): its own dominant launch -- the translator's 128 -> 128 3x3 layers
    now run at 128x128 (conv_3_1 / 4_0 / 4_1 of the 256x256 network: 2 415 919 104 MAC per image), F(4x4,3x3), 16 x 8 x 4 x 2 = 1 024 workgroups."""
    return _roofline_wino(dev, 'translator/conv_3_1', 'conv_wino43_kernel<0, false>',
                          'conv_wino43_kernel<0, false> F(4x4,3x3) fwd 3x3 s1 128->128 @128x128 B=%d (translator conv_3_1 of the 256x256 K=40 network)' % batch,
                          4.0, PROFILE + '_wino43_c3_pmc.json', csv_name=PROFILE + '_roofline_only_c3_kernel_stats.csv', n=batch, h=128, c=128)


def roofline_conv_c4(dev, frames=256):
    """--config c4 (the rollout): the inference translator's dominant launch -- conv_3_1 (128 -> 128 at 64x64) on one slab of 256 predicted
    frames (FinalModel.frames_per_launch; 2 048 frames per run = eight such slabs), batch norm folded into the filter, ReLU epilogue."""
    return _roofline_wino(dev, 'translator/conv_3_1', 'conv_wino43_kernel<0, false>',
                          'conv_wino43_kernel<0, false> F(4x4,3x3) fwd 3x3 s1 128->128 @64x64 on a %d-frame slab of the rollout (BN folded, ReLU epilogue)' % frames,
                          4.0, PROFILE + '_wino43_c4_pmc.json', csv_name=PROFILE + '_roofline_only_c4_kernel_stats.csv', n=frames, h=64, c=128, act=1)


def roofline_wgrad(dev):
    """The slowest direction of the trainable 3x3 layers: the Winograd F(2x2,3x3) WEIGHT gradient of translator conv_3_0 (256 -> 128 at
    64x64, batch 32; 1 207 959 552 MAC per image).  `achieved` counts the MFMA FLOPs the kernel executes (16 instead of 36 multiplies per
    2x2 outputs: algorithmic / 2.25) over the time of the whole operator (conv_wino_wgrad_kernel + the fixed-order reduce of its split slabs)."""
    from kpx_amd import ops
    n, h, ci, co = BATCH, 64, 256, 128
    x = torch.randn(n, h, h, ci, device=dev)
    dy = torch.randn(n, h, h, co, device=dev)
    dw = torch.empty(3, 3, ci, co, device=dev)
    ms = time_kernel(lambda: ops.conv_wgrad_raw(x, ci, ci, dy, co, dw, 1, 1, 1), iters=50, warm=10)
    flops = 2.0 * 9 * ci * co * h * h * n
    ach = flops / 2.25 / (ms * 1e-3) / 1e12
    traffic, src = _pmc_traffic(WGRAD_PMC)
    rp_ms, rp_src = _rocprof_avg_ms('conv_wino_wgrad_kernel<2, 2>')
    return {'committed_profile': {'note': 'from files committed under profiles/, NOT measured in this run', 'collected_at_commit': _profile_stamp(),
                                  'rocprof_avg_launch_ms_main_kernel': rp_ms, 'rocprof_source': rp_src, 'traffic_source': src},
            'bound': 'mfma', 'kernel': 'conv_wino_wgrad_kernel<2,2> (+ slab reduce) wgrad 3x3 s1 256->128 @64x64 B=32 (translator conv_3_0)',
            'achieved': round(ach, 2), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(ach / 157.3, 4), 'traffic': traffic, 'traffic_source': src,
            'avg_launch_ms': round(ms, 4), 'flops_per_launch_executed': flops / 2.25, 'flops_per_launch_algorithmic': flops,
            'algorithmic_tflops': round(flops / (ms * 1e-3) / 1e12, 2)}


def roofline_conv_f23(dev):
    """The F(2x2,3x3) kernel on the same layer shape (what the key-point detector, the image encoder and the translator's 32x32 layers
    run): executed FLOPs = algorithmic / 2.25."""
    return _roofline_wino(dev, 'pose_encoder/same_shape', 'conv_wino_v2_kernel<2, 0>',
                          'conv_wino_v2_kernel<2, 0> F(2x2,3x3) fwd 3x3 s1 128->128 @64x64 B=32, filter pre-transformed', 2.25, WINO_PMC, f43_fwd=False)


def roofline_conv_direct(dev):
    """The fp32-MFMA direct implicit-GEMM kernel on the same layer (the fallback of the bf16x3 kernels: odd-channel / tiny-channel shapes):
    KPX_NO_WINO=1 and KPX_NO_GEMM3=1 (without the second switch the layer runs fp32-equivalent on the bf16 pipe and is not bound by this peak)."""
    from kpx_amd._lib import lib
    os.environ['KPX_NO_WINO'] = '1'
    os.environ['KPX_NO_GEMM3'] = '1'
    lib.kpx_reload_env()                    # the switches are parsed once; re-read them for this leg only
    try:
        ms, flops = _time_conv_3_1(dev)
    finally:
        del os.environ['KPX_NO_WINO']
        del os.environ['KPX_NO_GEMM3']
        lib.kpx_reload_env()
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = _pmc_traffic(DIRECT_PMC)
    return {'bound': 'mfma', 'kernel': 'conv_igemm_kernel<128,128,2,4> fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1, KPX_NO_WINO=1 KPX_NO_GEMM3=1)',
            'achieved': round(ach, 2), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(ach / 157.3, 4),
            'traffic': traffic, 'traffic_source': src, 'avg_launch_ms': round(ms, 4), 'flops_per_launch': flops}


def roofline_conv_bf16(dev, n=BATCH, h=64, c=128):
    """The bf16-storage 3x3 kernel (BASELINE configs[2]; csrc/conv_bf16s.hip) on the roofline layer: bf16 tensors in HBM, fp32 accumulate,
    filters prepared once.  Two bounds are reported: the dense bf16 matrix peak (2.5 PFLOP/s; `frac`) and the HBM time of the ALGORITHMIC
    bytes (bf16 x read once + bf16 y written once + the fp32 filter's bf16 copy)."""
    from kpx_amd import ops
    from kpx_amd._lib import lib, check
    x = torch.randn(n, h, h, c, device=dev).bfloat16()
    w = torch.randn(3, 3, c, c, device=dev) * 0.03
    b = torch.zeros(c, device=dev)
    y = torch.empty(n, h, h, c, dtype=torch.bfloat16, device=dev)
    wf = torch.empty(lib.kpx_conv3x3_bf16s_weights_bytes(c, c), dtype=torch.uint8, device=dev)
    check(lib.kpx_conv3x3_bf16s_prepare_f32(w.data_ptr(), c, c, 0, wf.data_ptr(), ops._stream()), 'prepare')
    ms = time_kernel(lambda: check(lib.kpx_conv3x3_bf16s(x.data_ptr(), n, h, h, c, c, wf.data_ptr(), b.data_ptr(), y.data_ptr(), c, c, 0, 1, None, 0, None, ops._stream()), 'conv'),
                     iters=100, warm=20)
    flops = 2.0 * 9 * c * c * h * h * n
    nbytes = 2 * n * h * h * c * 2 + 9 * c * c * 2
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = _pmc_traffic(BF16_PMC)
    rp_ms, rp_src = _rocprof_avg_ms('conv3x3_bf16s_kernel<2, 2, 4, 32, 0>', PROFILE + '_roofline_only_bf16_kernel_stats.csv')
    return {'committed_profile': {'note': 'from files committed under profiles/, NOT measured in this run', 'collected_at_commit': _profile_stamp(),
                                  'rocprof_avg_launch_ms': rp_ms, 'rocprof_source': rp_src, 'traffic_source': src},
            'bound': 'mfma', 'kernel': 'conv3x3_bf16s_kernel<2,2,4,32,0> (512 pixels x 128 couts per workgroup, LDS-DMA operands, filters prepared once) fwd 3x3 s1 '
                                       '%d->%d @%dx%d B=%d (translator conv_3_1), bf16 tensors' % (c, c, h, h, n),
            'achieved': round(ach, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(ach / 2500.0, 4), 'traffic': traffic, 'traffic_source': src,
            'avg_launch_ms': round(ms, 4), 'flops_per_launch': flops, 'bytes_per_launch_algorithmic': nbytes,
            'hbm_gbps_algorithmic': round(nbytes / (ms * 1e-3) / 1e9, 1), 'hbm_frac_algorithmic': round(nbytes / (ms * 1e-3) / 8e12, 4)}


def roofline_wgrad_bf16(dev):
    """The bf16 configuration's weight gradient of the same layer as roofline_wgrad (translator conv_3_0, 256 -> 128 at 64x64, batch 32):
    conv3x3_wgrad_bf16_kernel (bf16 x and dy, transposed LDS reads, fp32 accumulate) + the fixed-tree reduce of its split slabs, against the
    dense bf16 matrix peak.  The slabs (one round of 256 workgroups x 9 x 64 x 128 fp32) are traffic beyond the algorithmic bytes."""
    from kpx_amd import ops
    from kpx_amd._lib import lib, check
    n, h, ci, co = BATCH, 64, 256, 128
    x = torch.randn(n, h, h, ci, device=dev).bfloat16()
    dy = torch.randn(n, h, h, co, device=dev).bfloat16()
    dw = torch.empty(3, 3, ci, co, device=dev)
    nbytes = lib.kpx_conv3x3_wgrad_bf16_workspace_bytes(n, h, h, ci, co)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ms = time_kernel(lambda: check(lib.kpx_conv3x3_wgrad_bf16(x.data_ptr(), n, h, h, ci, ci, dy.data_ptr(), co, co, dw.data_ptr(), ws.data_ptr(), nbytes, ops._stream()), 'wgrad'),
                     iters=50, warm=10)
    flops = 2.0 * 9 * ci * co * h * h * n
    ach = flops / (ms * 1e-3) / 1e12
    alg = n * h * h * (ci + co) * 2 + 9 * ci * co * 4
    traffic, src = _pmc_traffic(WGRAD16_PMC)
    rp_ms, rp_src = _rocprof_avg_ms('conv3x3_wgrad_bf16_kernel<2, 4, 1, 1>', PROFILE + '_roofline_only_bf16_kernel_stats.csv')
    return {'committed_profile': {'note': 'from files committed under profiles/, NOT measured in this run', 'collected_at_commit': _profile_stamp(),
                                  'rocprof_avg_launch_ms_main_kernel': rp_ms, 'rocprof_source': rp_src, 'traffic_source': src},
            'bound': 'mfma', 'kernel': 'conv3x3_wgrad_bf16_kernel<2,4,1,1> (+ wgrad16_reduce_kernel) wgrad 3x3 s1 256->128 @64x64 B=32 (translator conv_3_0), bf16 tensors',
            'achieved': round(ach, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(ach / 2500.0, 4), 'traffic': traffic, 'traffic_source': src,
            'avg_launch_ms': round(ms, 4), 'flops_per_launch': flops, 'bytes_per_launch_algorithmic': alg, 'slab_bytes_per_launch': nbytes}


def roofline_conv_bf16x3(dev):
    """The fp32-equivalent bf16x3 implicit-GEMM kernel (csrc/conv_gemm3.hip) on the discriminator's conv_3 (4x4 stride 2, 256 -> 512 at
    18x18 -> 10x10, N = 64 = real + generated halves of a batch of 32): six bf16 MFMAs per 32x32x16 block, so the matrix-pipe bound is the
    bf16 peak / 6 = 417 TFLOP/s of fp32-equivalent work; `frac` is against that, `frac_of_fp32_mfma_peak` against the 157.3 TF a
    v_mfma_f32_32x32x2_f32 kernel is bound by (the kernel this one replaced reached 0.59 of it on this layer)."""
    from kpx_amd import ops
    n, h, ci, co = 2 * BATCH, 18, 256, 512
    x = torch.randn(n, h, h, ci, device=dev)
    w = torch.randn(4, 4, ci, co, device=dev) * 0.02
    b = torch.zeros(co, device=dev)
    y = torch.empty(n, 10, 10, co, device=dev)
    ms = time_kernel(lambda: ops.conv_fwd_raw(x, ci, ci, w, b, y, co, 2, 2, 2, 2), iters=50, warm=10)
    flops = 2.0 * n * 10 * 10 * co * 16 * ci
    ach = flops / (ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'kernel': 'conv_gemm3_kernel<128,128,2,4,false,3> (+ split-K slab reduce) fwd 4x4 s2 256->512 @18x18 N=64 (img_discr conv_3), bf16x3 = fp32-equivalent',
            'achieved': round(ach, 2), 'peak': 416.7, 'unit': 'TFLOP/s (fp32-equivalent; bf16 dense peak 2500 / 6 products)', 'frac': round(ach / 416.7, 4),
            'frac_of_fp32_mfma_peak': round(ach / 157.3, 4), 'traffic': _pmc_traffic(GEMM3_PMC)[0], 'traffic_source': _pmc_traffic(GEMM3_PMC)[1],
            'traffic_note': 'HBM bytes of the conv kernel alone (its split-K slabs included, their reduce kernel not)',
            'avg_launch_ms': round(ms, 4), 'flops_per_launch': flops}


def roofline_render(dev, res=RES, k_pts=K_PTS, batch=BATCH):
    """HBM-bound kernel the north star singles out: Gaussian heat-map render at [128,128,K=15], batch 32:
    algorithmic bytes = B*H*W*K*4 written (+ K*8 read) per launch (983 040 B per image).  The launches rotate over
    nine 62.9 MB outputs (566 MB > the 256 MB Infinity Cache) so that every launch's stores have to reach HBM.
    (--config c3: [32,256,256,40] = 335.5 MB per launch, three rotating outputs.)"""
    from kpx_amd import ops
    from kpx_amd._lib import lib, check
    RES, K_PTS = res, k_pts
    b = 2 * batch                      # current + future key-point maps of one batch of pairs
    mu = (torch.rand(b, K_PTS, 2, device=dev) * 2 - 1).contiguous()
    outs = [torch.empty(b, RES, RES, K_PTS, device=dev) for _ in range(9 if res == 128 else 3)]
    it = [0]

    def run():
        out = outs[it[0] % len(outs)]
        it[0] += 1
        check(lib.kpx_gaussian_maps_fwd_f32(mu.data_ptr(), b, K_PTS, RES, RES, 14.3, out.data_ptr(), K_PTS, ops._stream()), 'gauss')
    ms = time_kernel(run, iters=198, warm=18)
    nbytes = b * (RES * RES * K_PTS * 4 + K_PTS * 8)
    ach = nbytes / (ms * 1e-3) / 1e9
    if (res, k_pts, batch) == (128, 15, 32):
        traffic, src = _pmc_traffic(RENDER_PMC)
        rp_ms, rp_src = _rocprof_avg_ms('gauss_fwd')
    elif (res, k_pts, batch) == (256, 40, 16):
        traffic, src = _pmc_traffic(PROFILE + '_render_c3_pmc.json')
        rp_ms, rp_src = _rocprof_avg_ms('gauss_fwd', PROFILE + '_roofline_only_c3_kernel_stats.csv')
    else:
        traffic, src, rp_ms, rp_src = None, None, None, None
    return {'committed_profile': {'note': 'from files committed under profiles/, NOT measured in this run', 'collected_at_commit': _profile_stamp(),
                                  'rocprof_avg_launch_ms': rp_ms, 'rocprof_source': rp_src, 'traffic_source': src,
                                  'frac_at_rocprof_avg': round(nbytes / (rp_ms * 1e-3) / 8e12, 4) if rp_ms else None},
            'avg_launch_ms_note': 'HIP events over back-to-back launches include the launch-to-launch gap of this 12 us kernel',
            'bound': 'hbm', 'kernel': 'gauss_fwd kernel [%d,%d,%d,%d] (current+future maps of %d pairs), %d rotating outputs' % (b, RES, RES, K_PTS, batch, len(outs)), 'achieved': round(ach, 1),
            'peak': 8000.0, 'unit': 'GB/s', 'frac': round(ach / 8000.0, 4), 'traffic': traffic, 'traffic_source': src,
            'avg_launch_ms': round(ms, 5), 'bytes_per_launch': nbytes}


def host_cores():
    """Cores this process may actually use: min(affinity mask, cgroup-v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline():
    """The CPU restatement of the reference (oracle/restatement.py; NOT TF 1.12 -- it cannot be installed)
    timed on this box's host cores: full train steps of BASELINE configs[0] (B=4, 128x128, K=15, VGG19 perceptual loss)."""
    from oracle import restatement as R
    cores = host_cores()
    torch.set_num_threads(cores)
    b = 4
    st = R.TrainState(R.init_variables(K_PTS, res=RES, seed=1234), R.synthetic_vgg(seed=19))
    im, fut = R.synthetic_pair(b, res=RES)
    R.train_step(st, im, fut)                      # warm-up
    times = []
    t_end = time.time() + 25.0
    while len(times) < 5 and (time.time() < t_end or len(times) < 2):
        t0 = time.time()
        R.train_step(st, im, fut)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {'value': round(b / med, 3), 'unit': 'image pairs/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d full train steps (D+G update, VGG19 loss) of batch 4 at 128x128 K=15, median; torch-CPU fp32 '
                      'restatement of the reference, not TF 1.12' % len(times),
            'sec_per_step': round(med, 3)}


def _timed_train_leg(dev, config, dtype, steps, warmup):
    """ms per step / pairs per second of one more train configuration on a FRESH model in this process (the `also` legs: BASELINE configs[2]
    and [3] under the driver's clock, after the headline's timed region and roofline legs)."""
    import gc
    import kpx_amd
    from kpx_amd import ops as kops
    from kpx_amd.synthetic import synthetic_pair
    conf = CONFIGS[config]
    res, k, batch = conf['res'], conf['k'], conf['batch']
    kops.set_compute_dtype(dtype)
    try:
        cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': batch},
               'model': {'n_pts': k}, 'paths': {'log_dir': '/tmp/kpx_bench', 'vggnet': None}}
        vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19), device=dev)
        model = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=res)
        model.build()
        feed = {kk: torch.from_numpy(v).to(dev) for kk, v in synthetic_pair(batch, res=res, seed0=0, seed1=1).items()}
        for i in range(warmup):
            model.train_step(None, feed, i, batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            model.train_step(None, feed, warmup + i, batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        losses = model.loss_values()
        ok = bool(np.isfinite(losses['loss_D']) and np.isfinite(losses['loss_G']))
        leg = {'workload': 'Penn %dx%d K=%d detector_translator %s, batch=%d (BASELINE configs[%d])' % (res, res, k, 'fp32' if dtype == 'f32' else 'bf16 + VGG19 perceptual loss',
                                                                                               batch, conf['idx'] if dtype == 'f32' else 2),
               'ms_per_step': round(dt / steps * 1e3, 3), 'value': round(batch * steps / dt, 2), 'unit': 'image pairs/sec', 'steps': steps, 'warmup': warmup,
               'dtype': dtype, 'launch_mode': model.LAUNCH_MODES[model.launch_mode()].split(':')[0], 'losses_finite': ok}
        if dtype == 'bf16':
            leg['fp32_kernel_fallbacks'] = dict(kops.fallback_uses)
        del model, vgg, feed
        gc.collect()
        torch.cuda.empty_cache()
        if dtype == 'bf16':
            leg['roofline'] = roofline_conv_bf16(dev)
        else:
            kops.set_compute_dtype('f32')
            leg['roofline'] = roofline_conv_c3(dev, batch)
        return leg
    finally:
        kops.set_compute_dtype('f32')


def _timed_rollout_leg(dev, runs, warmup=1):
    """BASELINE configs[4] (bench_rollout's workload) on a fresh FinalModel in this process."""
    import gc
    import kpx_amd
    conf = CONFIGS['c4']
    res, k, b = conf['res'], conf['k'], conf['batch']
    cfg = {'model': {'n_pts': k, 'cell_info': [1024, 1024], 'vae_dim': 64, 'n_action': 9}, 'paths': {'log_dir': '/tmp/kpx_bench'}}
    fm = kpx_amd.FinalModel(cfg, device=dev, image_size=res, frames_per_launch=256)
    fm.build()
    rs = np.random.RandomState(7)
    im = torch.from_numpy((rs.randint(0, 256, size=(b, res, res, 3)).astype(np.float32) / 255.0 * 2 - 1).astype(np.float32)).to(dev)
    act = torch.from_numpy(np.eye(9, dtype=np.float32)[rs.randint(0, 9, size=b)]).to(dev)
    z = torch.from_numpy(rs.randn(b, 64).astype(np.float32)).to(dev)
    feed = {'image': im, 'action_code': act}
    for _ in range(warmup):
        out = fm.run(None, feed, z=z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(runs):
        out = fm.run(None, feed, z=z)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(out['pred_im_seq']).all())
    leg = {'workload': 'evaluate.py rollout 128x128 K=15: %d source images -> %d frames per run (BASELINE configs[4])' % (b, b * 32),
           'ms_per_step': round(dt / runs * 1e3, 3), 'value': round(b * 32 * runs / dt, 1), 'unit': 'predicted frames/sec', 'steps': runs, 'warmup': warmup,
           'dtype': 'f32', 'outputs_finite': ok}
    del fm, out, feed
    gc.collect()
    torch.cuda.empty_cache()
    leg['roofline'] = roofline_conv_c4(dev)
    return leg


def also_legs(dev):
    """The other BASELINE configurations under the same clock as the headline (N = 1 only, after everything the headline line needs has been
    measured): configs[2] (bf16), configs[3] (256x256, K=40) and configs[4] (the rollout).  A leg that fails reports its error; it can never
    change or suppress the headline fields."""
    also = {}
    for name, fn in (('bf16', lambda: _timed_train_leg(dev, 'c1', 'bf16', 10, 3)), ('c3', lambda: _timed_train_leg(dev, 'c3', 'f32', 5, 2)),
                     ('c4', lambda: _timed_rollout_leg(dev, 5, warmup=2))):
        t0 = time.perf_counter()
        try:
            also[name] = fn()
        except Exception as e:                             # noqa: BLE001 -- reported, never raised: the headline stands on its own
            also[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        also[name]['leg_wall_s'] = round(time.perf_counter() - t0, 1)
        only = {kk: v for kk, v in also[name].get('roofline', {}).items() if kk in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms')}
        if only:
            also[name]['roofline'] = only
    return also


# ---------------------------------------------------------------------------------------------- N > 1: ranks under a supervisor
# `bench.py --gpus N` must not be able to hang: a data-parallel step that deadlocks inside a collective (a captured graph with RCCL nodes that
# never completes, a rank that dies in the rendezvous) would leave the driver without a number.  So for N > 1 the ranks that do the work are
# always CHILD processes of a supervisor that never touches a GPU:
#   * `python bench.py --gpus N`                         -- this process supervises all N children;
#   * `python -m torch.distributed.run ... bench.py`     -- each launched process supervises ONE child (its own rank); the N supervisors agree
#                                                            through marker files in a shared directory (one node: the contract's --nnodes=1).
# An attempt = N fresh children with one KPX_DP_GRAPH form.  A child that exits non-zero, does not finish its warm-up steps inside
# KPX_BENCH_WARM_DEADLINE_S or its timed steps inside KPX_BENCH_RUN_DEADLINE_S makes every supervisor kill its children (fresh processes are the
# only safe restart: a process that has initialised the GPU must never exec) and start the next form: 'segments' (captured segments, the
# collectives enqueued from Python between them), then 'inline' without graphs (every launch eager).  KPX_DP_GRAPH=one (the whole step with
# both all-reduces in ONE captured graph) goes first only when asked for explicitly.  Rank 0's JSON line names the form that ran
# (`launch_mode_by_rank`) and the attempts before it (`dp_fallbacks`).  The reference has no counterpart (train.py:25-29: one session, one device).
DP_FORMS = (('segments', {}), ('inline', {'KPX_GRAPH': '0'}))


def dp_attempt_plan():
    first = os.environ.get('KPX_DP_GRAPH', '')
    plan = [f for f in DP_FORMS if f[0] != first]
    if first:
        plan.insert(0, (first, dict(dict(DP_FORMS).get(first, {}))))
    return plan


def _touch(path, text=''):
    tmp = path + '.tmp%d' % os.getpid()
    with open(tmp, 'w') as f:
        f.write(text)
    os.replace(tmp, path)


def supervise(my_ranks, world, child_argv, key, plan=None, warm_deadline=None, run_deadline=None, settle=60.0, log=sys.stderr):
    """Run the ranks `my_ranks` of a `world`-rank job as child processes, attempt by attempt (see above).  Returns (exit code, attempts)."""
    import shutil
    import socket
    import subprocess
    import tempfile
    plan = list(plan if plan is not None else dp_attempt_plan())
    warm_deadline = float(os.environ.get('KPX_BENCH_WARM_DEADLINE_S', '300')) if warm_deadline is None else warm_deadline
    run_deadline = float(os.environ.get('KPX_BENCH_RUN_DEADLINE_S', '180')) if run_deadline is None else run_deadline
    d = os.path.join(tempfile.gettempdir(), 'kpx_bench_%d_%s' % (os.getuid(), key))
    os.makedirs(d, exist_ok=True)
    lead = 0 in my_ranks
    for name in os.listdir(d):                       # markers of an earlier run that died with this key
        if any(name.endswith('_r%d.%s' % (r, sfx)) for r in my_ranks for sfx in ('warm', 'done', 'killed')) or (lead and name.startswith('a') and '_r' not in name):
            try:
                os.remove(os.path.join(d, name))
            except OSError:
                pass
    attempts = []

    def say(msg):
        print('[bench supervisor ranks %s] %s' % (','.join(map(str, my_ranks)), msg), file=log, flush=True)

    def wait_for(path, seconds):
        t_end = time.time() + seconds
        while not os.path.exists(path):
            if time.time() > t_end:
                return False
            time.sleep(0.02)
        return True
    rc = 1
    for k, (form, extra) in enumerate(plan):
        port_file = os.path.join(d, 'a%d.port' % k)
        if lead:
            with socket.socket() as sock:
                sock.bind(('127.0.0.1', 0))
                _touch(port_file, str(sock.getsockname()[1]))
        if not wait_for(port_file, settle + warm_deadline):
            say('no port for attempt %d: the supervisor of rank 0 is gone' % k)
            return 1, attempts
        port = open(port_file).read().strip()
        procs = {}
        for r in my_ranks:
            env = {kk: v for kk, v in os.environ.items() if not kk.startswith('TORCHELASTIC_')}      # (the agent's store is not reused: keys of a killed attempt would remain)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                       HSA_ENABLE_IPC_MODE_LEGACY='0', KPX_BENCH_CHILD='1', KPX_BENCH_DIR=d, KPX_BENCH_ATTEMPT=str(k), KPX_DP_GRAPH=form,
                       KPX_BENCH_FALLBACKS=json.dumps(attempts))
            env.update(extra)
            procs[r] = subprocess.Popen(list(child_argv), env=env)
        mark = lambda r, sfx: os.path.join(d, 'a%d_r%d.%s' % (k, r, sfx))       # noqa: E731
        abort_file = os.path.join(d, 'a%d.abort' % k)
        t0, t_warm, why = time.time(), None, None
        while why is None:
            codes = {r: p.poll() for r, p in procs.items()}
            bad = [r for r, c in codes.items() if c not in (None, 0)]
            if bad:
                why = 'rank %d exited with code %s' % (bad[0], codes[bad[0]])
            elif all(c == 0 for c in codes.values()):
                break
            elif os.path.exists(abort_file):
                why = 'aborted by another supervisor: ' + open(abort_file).read().strip()
            elif t_warm is None:
                if all(os.path.exists(mark(r, 'warm')) or codes[r] == 0 for r in my_ranks):
                    t_warm = time.time()
                elif time.time() - t0 > warm_deadline:
                    why = 'no completed warm-up step after %.0f s (form %r)' % (warm_deadline, form)
            elif time.time() - t_warm > run_deadline:
                why = 'timed steps not finished %.0f s after the warm-up (form %r)' % (run_deadline, form)
            time.sleep(0.05)
        if why is None and not lead:
            # my ranks are through; the attempt counts once rank 0 has printed the line (or fails with the others)
            t_end = time.time() + run_deadline
            while not os.path.exists(mark(0, 'done')) and why is None:
                if os.path.exists(abort_file):
                    why = 'aborted by another supervisor: ' + open(abort_file).read().strip()
                elif time.time() > t_end:
                    why = 'rank 0 never reported its result'
                time.sleep(0.05)
        if why is None and lead and not os.path.exists(mark(0, 'done')):
            why = 'rank 0 exited without a result'
        if os.path.exists(mark(0, 'done')):                      # rank 0 has printed the JSON line: the result stands, never another attempt
            for p in procs.values():                             # (stragglers, e.g. a rank stuck tearing the process group down, get 30 s)
                try:
                    p.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    p.kill()
            if why is not None:
                say('after the result was printed: ' + why)
            for r in my_ranks:
                _touch(mark(r, 'ack'))
            rc = 0
            break
        if not os.path.exists(abort_file):
            _touch(abort_file, why)
        say('attempt %d (KPX_DP_GRAPH=%s) failed: %s' % (k, form, why))
        for p in procs.values():
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5
        for p in procs.values():
            try:
                p.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                try:
                    p.wait(timeout=30)                           # (a rank stuck in an uninterruptible GPU wait must not block the supervisor for ever)
                except subprocess.TimeoutExpired:
                    say('a child survived SIGKILL for 30 s: giving up')
                    return 1, attempts + [{'form': form, 'failed': why + '; child unkillable'}]
        for r in my_ranks:
            _touch(mark(r, 'killed'))
        for r in range(world):                                   # every rank's child is gone before the GPUs are used again
            wait_for(mark(r, 'killed'), settle)
        if os.path.exists(mark(0, 'done')):                      # the abort raced with rank 0 printing its line: the result stands
            say('rank 0 reported its result while this attempt was being aborted: done')
            for r in my_ranks:
                _touch(mark(r, 'ack'))
            rc = 0
            break
        attempts.append({'form': form, 'failed': why})
    if lead and rc == 0:
        for r in range(world):                                   # the other supervisors read rank 0's marker: leave the directory until they have
            wait_for(os.path.join(d, 'a%d_r%d.ack' % (k, r)), 30.0)
        shutil.rmtree(d, ignore_errors=True)
    return rc, attempts


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this process supervises one child per GPU (it never initialises the GPU and never
    exec()s); rank 0's child prints the JSON line on the inherited stdout."""
    return supervise(list(range(n)), n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], 'pid%d' % os.getpid())[0]


def supervise_own_rank():
    """Under torch.distributed.run: this launched process supervises a child that does its rank's work."""
    world, rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK'])
    key = 'ppid%d_port%s' % (os.getppid(), os.environ.get('MASTER_PORT', '0'))
    return supervise([rank], world, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], key)[0]


def child_mark(suffix):
    """A supervised rank reports progress ('warm': warm-up steps done, 'done': result printed) to its supervisor."""
    if os.environ.get('KPX_BENCH_CHILD') == '1':
        _touch(os.path.join(os.environ['KPX_BENCH_DIR'], 'a%s_r%s.%s' % (os.environ['KPX_BENCH_ATTEMPT'], os.environ.get('RANK', '0'), suffix)))


def bench_rollout(args, conf, dev, rank, world, launched, backend):
    """BASELINE configs[4]: the evaluate.py pipeline (reference models/final_model.py:49-122) -- key-point detector + image encoder on the
    source image, vae_decoder (fc + 2 x LSTMCell(1024) x 32 steps + to_coord), translator on B*32 frames with inference-mode batch norm.
    One step = one FinalModel.run on `batch` source images = batch*32 predicted frames.  No collective: replicas only."""
    import kpx_amd
    res, k, b = conf['res'], conf['k'], args.batch
    cfg = {'model': {'n_pts': k, 'cell_info': [1024, 1024], 'vae_dim': 64, 'n_action': 9}, 'paths': {'log_dir': '/tmp/kpx_bench'}}
    fm = kpx_amd.FinalModel(cfg, device=dev, image_size=res, frames_per_launch=256)
    fm.build()
    rs = np.random.RandomState(7 + rank)
    im = torch.from_numpy((rs.randint(0, 256, size=(b, res, res, 3)).astype(np.float32) / 255.0 * 2 - 1).astype(np.float32)).to(dev)
    act = torch.from_numpy(np.eye(9, dtype=np.float32)[rs.randint(0, 9, size=b)]).to(dev)
    z = torch.from_numpy(rs.randn(b, 64).astype(np.float32)).to(dev)
    feed = {'image': im, 'action_code': act}

    def sync():
        if launched:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = fm.run(None, feed, z=z)
    sync()
    child_mark('warm')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fm.run(None, feed, z=z)
    t_enq = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    if launched:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(out['pred_im_seq']).all())
    if rank == 0:
        frames = world * b * 32 * args.steps
        # translator 7.054 GMAC per frame (Appendix A) + detector / image encoder once per source image
        gmac_frame = 7.054 + (1.817 + 0.681) / 32.0
        line = {'metric': conf['metric'], 'value': round(frames / dt, 1), 'unit': 'predicted frames/sec', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': 'evaluate.py rollout 128x128 K=15: %d source images per GPU -> %d frames per step, 2 x LSTM(1024), vae_dim 64, '
                                       'inference-mode batch norm (BASELINE configs[4])' % (b, b * 32), 'global_batch': world * b, 'parallelism': 'replicas x%d' % world,
                           'algorithmic_gmac_per_frame': round(gmac_frame, 3)},
                'step_tflops': round(2 * gmac_frame * 1e9 * frames / dt / 1e12, 2),
                'step_algorithmic_frac_of_f32_peak': round(2 * gmac_frame * 1e9 * frames / dt / world / 157.3e12, 4),
                'host_call_wall_ms_per_step': round(t_enq / args.steps * 1e3, 3), 'dist_backend': backend if launched else None}
        if world == 1:
            line['roofline'] = roofline_conv_c4(dev)       # the translator's 3x3 layers dominate the rollout; its own launch: a 256-frame slab
        print(json.dumps(line), flush=True)
        child_mark('done')
    if launched:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        child_mark('done')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None, help='image pairs (c4: source images) per GPU; default: the configuration\'s')
    ap.add_argument('--config', choices=sorted(CONFIGS), default='c1',
                    help='c1: BASELINE configs[1], the headline (128x128, K=15, 32 pairs per GPU).  c3: configs[3] (256x256, K=40, 16 pairs per GPU = 128 '
                         'over 8).  c4: configs[4], the evaluate.py rollout (64 source images -> 2048 predicted frames per step, inference only)')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help="f32: the headline / parity configuration (BASELINE configs[1]).  bf16: BASELINE configs[2] -- bf16 activation tensors in "
                         'HBM, fp32 accumulate / batch-norm statistics / master weights / Adam; a SEPARATE configuration, never the headline')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true', help='skip the kernel microbenchmarks after the timed steps (clean per-step rocprofv3 kernel statistics)')
    ap.add_argument('--no-also', action='store_true', help='skip the `also` legs (bf16 / c3 / c4 on fresh models after the headline; N = 1, --config c1 --dtype f32 only)')
    ap.add_argument('--roofline-only', action='store_true', help='only the two kernel microbenchmarks (used for the rocprofv3 cross-check)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))             # nothing in this process has touched the GPU yet
    if args.gpus > 1 and os.environ.get('KPX_BENCH_CHILD') != '1' and os.environ.get('KPX_BENCH_SUPERVISE', '1') != '0':
        sys.exit(supervise_own_rank())               # under a launcher: the launched process only supervises (no GPU use, no exec)
    launched = 'WORLD_SIZE' in os.environ            # under torch.distributed.run (or self_launch): always build the process group
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    backend = os.environ.get('KPX_DIST_BACKEND', 'nccl')              # nccl = RCCL over xGMI; gloo only to self-test on one GPU
    n_dev = max(1, torch.cuda.device_count())
    if backend == 'nccl' and world > n_dev:
        raise SystemExit('--gpus %d needs %d GPUs, %d visible (RCCL wants one GPU per rank; KPX_DIST_BACKEND=gloo shares GPUs for self-tests)'
                         % (world, world, n_dev))
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if launched:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        assert torch.distributed.get_world_size() == world

    import kpx_amd
    from kpx_amd import ops as kops
    from kpx_amd.synthetic import synthetic_pair
    kops.set_compute_dtype(args.dtype)
    conf = CONFIGS[args.config]
    RES, K_PTS = conf['res'], conf['k']
    if args.batch is None:
        args.batch = conf['batch']
    if args.config == 'c4' and args.roofline_only:
        print(json.dumps({'roofline': roofline_conv_c4(dev)}), flush=True)
        return
    if args.config == 'c4':
        return bench_rollout(args, conf, dev, rank, world, launched, backend)
    if args.roofline_only and args.config == 'c3':
        print(json.dumps({'roofline': roofline_conv_c3(dev, args.batch), 'roofline_hbm_render': roofline_render(dev, RES, K_PTS, args.batch)}), flush=True)
        return
    if args.roofline_only and args.dtype == 'bf16':
        print(json.dumps({'roofline': roofline_conv_bf16(dev), 'roofline_wgrad': roofline_wgrad_bf16(dev)}), flush=True)
        return
    if args.roofline_only:
        kops.set_compute_dtype('f32')
        print(json.dumps({'roofline': roofline_conv(dev), 'roofline_wino43_f32mfma': roofline_conv_f32mfma(dev), 'roofline_wgrad': roofline_wgrad(dev), 'roofline_wino_f23': roofline_conv_f23(dev), 'roofline_direct_conv': roofline_conv_direct(dev), 'roofline_bf16x3_conv': roofline_conv_bf16x3(dev),
                          'roofline_hbm_render': roofline_render(dev)}), flush=True)
        return
    cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': args.batch},
           'model': {'n_pts': K_PTS}, 'paths': {'log_dir': '/tmp/kpx_bench', 'vggnet': None}}
    vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19), device=dev)
    model = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=RES)
    model.build()
    pair = synthetic_pair(args.batch, res=RES, seed0=2 * rank, seed1=2 * rank + 1)
    feed = {k: torch.from_numpy(v).to(dev) for k, v in pair.items()}

    def sync():
        if launched:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    tw = time.perf_counter()
    for i in range(args.warmup):
        model.train_step(None, feed, i, args.batch)
    sync()
    child_mark('warm')
    if rank == 0:
        print('[bench] warmup %d steps: %.2fs' % (args.warmup, time.perf_counter() - tw), file=sys.stderr, flush=True)
    from kpx_amd import _lib as klib
    calls0 = klib.abi_calls[0]
    t0 = time.perf_counter()
    per_call = []
    for i in range(args.steps):
        tc = time.perf_counter()
        model.train_step(None, feed, args.warmup + i, args.batch)
        per_call.append(time.perf_counter() - tc)
    t_enq = time.perf_counter() - t0          # host wall time to enqueue K steps: includes waiting on a full launch queue
    calls_per_step = (klib.abi_calls[0] - calls0) / max(args.steps, 1)
    graphed = bool(getattr(model, '_graphs', None)) and not getattr(model, '_graph_failed', False)
    sync()
    dt = time.perf_counter() - t0
    enq = [t_enq / args.steps * 1e3]
    ms_by_rank = [dt / args.steps * 1e3]
    n_ranks_seen = 1
    modes_by_rank, host_work_by_rank = [model.launch_mode()], [min(per_call) * 1e3]
    if launched:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        n_ranks_seen = torch.distributed.get_world_size()
        e = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(world)]
        torch.distributed.all_gather(e, torch.tensor([enq[0], ms_by_rank[0], float(model.launch_mode()), min(per_call) * 1e3], dtype=torch.float64, device=dev))
        enq = [float(x[0].item()) for x in e]
        ms_by_rank = [float(x[1].item()) for x in e]          # each rank's own clock around its K steps; `ms_per_step` is their maximum
        modes_by_rank = [int(x[2].item()) for x in e]
        host_work_by_rank = [float(x[3].item()) for x in e]
        dt = float(t.item())
    losses = model.loss_values()
    assert np.isfinite(losses['loss_D']) and np.isfinite(losses['loss_G']), losses

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        gmac = step_gmac_per_pair(RES)
        out = {'metric': conf['metric'], 'value': round(value, 2),
               'unit': 'image pairs/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f32' if args.dtype == 'f32' else 'bf16 (activation tensors bf16 in HBM; fp32 accumulate, batch-norm statistics, master weights, gradients, Adam)',
               'data': 'synthetic',
               'config': {'workload': 'Penn %dx%d K=%d detector_translator %s, batch=%d per GPU (BASELINE configs[%d])'
                                      % (RES, RES, K_PTS, 'fp32' if args.dtype == 'f32' else 'bf16 + VGG19 perceptual loss', args.batch,
                                         conf['idx'] if args.dtype == 'f32' else 2),
                          'global_batch': world * args.batch, 'parallelism': 'dp%d' % world,
                          'step': 'D update + G update on one batch (generator forward shared, SURVEY 8d restructured step), '
                                  'VGG19 perceptual loss with synthetic He-normal weights, two fused Adam updates',
                          'algorithmic_gmac_per_pair': round(gmac, 2)},
               'step_tflops': round(2 * gmac * 1e9 * value / 1e12, 2),
               'step_algorithmic_frac_of_f32_peak': round(2 * gmac * 1e9 * value / world / 157.3e12, 4),
               # the chip's fp32-EQUIVALENT ceiling is the six-product bf16x3 bound (2.5 PFLOP/s / 6 = 417 TFLOP/s), not the fp32-MFMA pipe
               'step_algorithmic_frac_of_bf16x3_bound': round(2 * gmac * 1e9 * value / world / 416.7e12, 4),
               'step_algorithmic_frac_of_bf16_peak': round(2 * gmac * 1e9 * value / world / 2.5e15, 4),
               'n_ranks_seen': n_ranks_seen, 'dist_backend': backend if launched else None,
               'ms_per_step_by_rank': [round(x, 3) for x in ms_by_rank],
               'host_call_wall_ms_per_step': round(max(enq), 3), 'host_call_wall_ms_per_step_by_rank': [round(x, 3) for x in enq],
               'host_note': 'host_call_wall = wall time of the K train_step() calls / K: mostly launch-queue BACK-PRESSURE (a graph replay call blocks once '
                            'a few replays are in flight), not host work; host_work_ms_per_step_min = the shortest single call = the host work of one step',
               'host_work_ms_per_step_min': round(min(per_call) * 1e3, 3), 'host_work_ms_per_step_min_by_rank': [round(x, 3) for x in host_work_by_rank],
               'launch_mode': model.LAUNCH_MODES[model.launch_mode()] + (' (captured after 1 eager warm-up step; %d C-ABI kernel launches inside the graph)'
                                                                         % getattr(model, '_graph_launches', 0) if model.launch_mode() == 1 else ''),
               'launch_mode_by_rank': [model.LAUNCH_MODES[m].split(':')[0] for m in modes_by_rank],
               'host_abi_calls_per_step': round(calls_per_step, 1),
               'loss_D': round(losses['loss_D'], 5), 'loss_G': round(losses['loss_G'], 5)}
        if launched:
            # N > 1 runs under bench.py's supervisor (DP_FORMS): the attempts that failed or hung before this one
            out['dp_graph_form'] = model.dp_graph if model.distributed else None
            out['dp_fallbacks'] = json.loads(os.environ.get('KPX_BENCH_FALLBACKS', '[]'))
        if world == 1 and not args.no_roofline:
            if args.dtype == 'bf16':
                # the bf16 configuration's own dominant kernels (its fp32 legs are the fp32 line's)
                out['fp32_kernel_fallbacks'] = dict(kops.fallback_uses)       # bf16 tensors routed through an fp32 kernel between two conversions (whole run)
                out['roofline'] = roofline_conv_bf16(dev)
                out['roofline_wgrad'] = roofline_wgrad_bf16(dev)
                kops.set_compute_dtype('f32')
                out['roofline_hbm_render'] = roofline_render(dev, RES, K_PTS, args.batch)
            else:
                out['roofline'] = roofline_conv(dev) if args.config == 'c1' else roofline_conv_c3(dev, args.batch)
                if args.config == 'c1':
                    out['roofline_wino43_f32mfma'] = roofline_conv_f32mfma(dev)
                out['roofline_wgrad'] = roofline_wgrad(dev)
                out['roofline_wino_f23'] = roofline_conv_f23(dev)
                out['roofline_direct_conv'] = roofline_conv_direct(dev)
                out['roofline_bf16x3_conv'] = roofline_conv_bf16x3(dev)
                out['roofline_hbm_render'] = roofline_render(dev, RES, K_PTS, args.batch)
            if args.config == 'c1' and args.dtype == 'f32' and not args.no_also:
                # every other configuration's number under the same (driver's) clock: after the timed region and the roofline legs, fresh models
                del model, vgg
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                out['also'] = also_legs(dev)
            if not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
        child_mark('done')
    if launched:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        child_mark('done')


if __name__ == '__main__':
    main()
