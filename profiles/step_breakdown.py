#!/usr/bin/env python3
"""Per-step kernel-time breakdown from a rocprofv3 --kernel-trace --stats CSV of `bench.py --no-roofline --no-cpu-baseline`:

    python profiles/step_breakdown.py profiles/r03_step_kernel_stats.csv <profiled steps = warmup + steps>

Groups the kernels into the families DESIGN.md section 6 quotes, prints ms per step, launches per step and the batch-norm subtotal."""
import csv
import sys

FAMILIES = [
    ('bf16-storage 3x3 conv fwd/dgrad (conv3x3_bf16s)', ('conv3x3_bf16s_kernel',)),
    ('bf16 3x3 weight gradient (conv3x3_wgrad_bf16 + slab reduce)', ('conv3x3_wgrad_bf16_kernel', 'wgrad16_reduce_kernel')),
    ('bf16 filter preparation', ('conv_bf16s_prepare',)),
    ('bf16 <-> fp32 conversions / bf16 channel copies', ('cast_channels_kernel',)),
    ('F(4x4,3x3) Winograd fwd/dgrad, bf16x3 form (conv_wino43b)', ('conv_wino43b_kernel',)),
    ('F(4x4,3x3) Winograd fwd/dgrad, fp32 MFMA (conv_wino43)', ('conv_wino43_kernel',)),
    ('F(2x2,3x3) Winograd fwd/dgrad', ('conv_wino_v2_kernel',)),
    ('Winograd weight gradient', ('conv_wino_wgrad_kernel',)),
    ('bf16x3 gather conv fwd/dgrad (conv_gemm3)', ('conv_gemm3_kernel',)),
    ('bf16x3 weight gradient (conv_wgrad3)', ('conv_wgrad3_kernel',)),
    ('fp32-MFMA direct conv fwd/dgrad', ('conv_igemm_kernel', 'conv_small_cout_kernel')),
    ('fp32-MFMA direct weight gradients', ('conv_wgrad_kernel', 'conv_wgrad_rows', 'conv_wgrad_tap_rows', 'conv_wgrad_rows_merged', 'conv_c16_wgrad', 'wgrad_small')),
    ('split-K / weight-gradient slab reductions', ('wgrad_reduce_kernel', 'splitk_reduce_kernel')),
    ('16-cout / image-input conv kernels', ('conv3x3_c16_kernel', 'conv_rgb', 'conv_few')),
    ('batch norm', ('bn_', 'chan_reduce_kernel<1', 'chan_reduce_kernel<2', 'chan_reduce_bf16_kernel<1', 'chan_reduce_bf16_kernel<2')),
    ('bias gradients (channel sums)', ('chan_reduce_kernel<0', 'chan_reduce_bf16_kernel<0', 'chan_sum_finalize')),
    ('Winograd filter transforms', ('filter_transform',)),
    ('key-point head / heat-map render', ('kp_', 'gauss_')),
    ('VGG19 pointwise (prep, pool, feature gradient, L1)', ('vgg_', 'maxpool2', 'l1_pair')),
    ('resize / concat / copy / activation backward / blend', ('resize2x', 'copy_channels', 'act_bwd', 'head_blend', 'fill_kernel', 'axpy')),
    ('Adam', ('adam_tf',)),
    ('losses', ('xent',)),
    ('torch glue (adds, mean)', ('at::native',)),
]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    fam = {name: [0.0, 0] for name, _ in FAMILIES}
    other = [0.0, 0, []]
    for r in rows:
        ns, calls = float(r['TotalDurationNs']), int(r['Calls'])
        for name, keys in FAMILIES:
            if any(k in r['Name'] for k in keys):
                fam[name][0] += ns; fam[name][1] += calls
                break
        else:
            other[0] += ns; other[1] += calls; other[2].append(r['Name'][:50])
    total = sum(v[0] for v in fam.values()) + other[0]
    launches = sum(v[1] for v in fam.values()) + other[1]
    print('%-58s %9s %10s' % ('kernel family', 'ms/step', 'launches/step'))
    for name, _ in FAMILIES:
        ns, calls = fam[name]
        if calls:
            print('%-58s %9.3f %10.1f' % (name, ns / 1e6 / steps, calls / steps))
    if other[1]:
        print('%-58s %9.3f %10.1f   %s' % ('other', other[0] / 1e6 / steps, other[1] / steps, sorted(set(other[2]))[:6]))
    print('%-58s %9.3f %10.1f   (sum over the three streams)' % ('TOTAL', total / 1e6 / steps, launches / steps))


if __name__ == '__main__':
    main()
