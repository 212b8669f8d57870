"""Per-layer error budget of the bf16 configuration: the fp32 and the bf16 forward from the same state, rel-L2 of every conv+BN+ReLU output
at the layer boundaries, key-points and frame; then what-if runs keeping chosen tensors fp32 (layers.F32_OUT_SCOPES).
    python scratch/bf16_error_budget.py [B]      -> profiles/r06_bf16_error_budget.txt (copied by hand)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops, layers
from oracle import restatement as R          # (synthetic_pair only: the same images as the parity tests)

dev = torch.device('cuda:0')
res, k, b = 128, 15, int(sys.argv[1]) if len(sys.argv) > 1 else 2


def build():
    cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b}, 'model': {'n_pts': k}, 'paths': {'log_dir': '/tmp/kpx_test', 'vggnet': None}}
    vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19, width_div=8), device=dev)
    m = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=res)
    m.build()
    return m


def rel(a, b_):
    a, b_ = a.double(), b_.double()
    return float((a - b_).norm() / b_.norm())


im, fut = R.synthetic_pair(b, res=res, seed0=0, seed1=1)
im, fut = torch.from_numpy(im).to(dev), torch.from_numpy(fut).to(dev)


def run(dtype, keep=()):
    ops.set_compute_dtype(dtype)
    layers.F32_OUT_SCOPES = set(keep)
    layers.TRACE = []
    try:
        m = build()
        out = m.forward(im, fut)
        tr = [(n, t.float().clone()) for n, t in layers.TRACE]
        return {kk: v.float().clone() for kk, v in out.items() if torch.is_tensor(v)}, tr
    finally:
        layers.TRACE = None
        layers.F32_OUT_SCOPES = set()
        ops.set_compute_dtype('f32')


o32, t32 = run('f32')
o16, t16 = run('bf16')
print('bf16 configuration vs fp32 configuration, same weights, B=%d, %dx%d, K=%d (forward only, train-mode batch norm)' % (b, res, res, k))
print('%-44s %12s %10s' % ('conv+BN+ReLU output', 'shape', 'rel-L2'))
for (n, a), (n2, c) in zip(t32, t16):
    assert n == n2
    print('%-44s %12s %10.2e' % (n, 'x'.join(str(s) for s in a.shape[1:]), rel(c, a)))


def summary(o):
    return (float((o['current_points'] - o32['current_points']).abs().max()), float((o['future_points'] - o32['future_points']).abs().max()),
            rel(o['crude_output'], o32['crude_output']), rel(o['mask'], o32['mask']), rel(o['final_output'], o32['final_output']))


print('\n%-64s %9s %9s %9s %9s %9s' % ('tensors kept fp32 (beyond the declared ones)', 'kp cur', 'kp fut', 'crude', 'mask', 'frame'))
print('%-64s %9.2e %9.2e %9.2e %9.2e %9.2e' % (('(none)',) + summary(o16)))
names = [n for n, _ in t32]
pose = [n for n in names if n.startswith('pose_encoder')]
trans = [n for n in names if n.startswith('translator')]
imenc = [n for n in names if n.startswith('image_encoder')]
cands = [('pose_encoder: all', pose), ('translator: all', trans), ('image_encoder: all', imenc),
         ('pose decoder (conv_1_0 .. conv_7_0)', [n for n in pose if '/encoder/' not in n]), ('pose encoder blocks', [n for n in pose if '/encoder/' in n]),
         ('pose conv_5_0 .. conv_7_0', [n for n in pose if any(s in n for s in ('conv_5_', 'conv_6_', 'conv_7_'))]),
         ('translator conv_1_0 .. conv_2_1', [n for n in trans if any(s in n for s in ('conv_1_', 'conv_2_'))]),
         ('translator conv_3_0 .. conv_4_1', [n for n in trans if any(s in n for s in ('conv_3_', 'conv_4_'))]),
         ('translator conv_5_0, conv_5_1', [n for n in trans if 'conv_5_' in n]), ('translator conv_5_1', [n for n in trans if 'conv_5_1' in n]),
         ('translator conv_1_0', [n for n in trans if 'conv_1_0' in n]), ('pose + image encoders: all', pose + imenc)]
for label, keep in cands:
    o, _ = run('bf16', keep)
    print('%-64s %9.2e %9.2e %9.2e %9.2e %9.2e' % ((label + ' [%d]' % len(keep),) + summary(o)))
