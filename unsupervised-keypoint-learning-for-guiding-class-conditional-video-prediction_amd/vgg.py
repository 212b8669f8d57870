"""VGG19 feature extractor + perceptual L1 loss (reference: models/networks/vgg.py, detector_translator_model.py:274-289).

The weights are constants (``tf.constant`` in the reference, vgg.py:57-61), so the whole perceptual term is ONE autograd
node with a hand-written backward: forward runs the 16 conv+bias+ReLU layers on the batch-concatenated [gt ‖ pred]
images exactly as the reference does (detector_translator_model.py:278), backward runs dgrad only, and only on the
``pred`` half of the batch (the gt half has no trainable ancestor; TF computes and discards that gradient).
"""
import math
import os
import weakref
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import lib, check

VGG_LAYERS = [('conv1_1', 3, 64), ('conv1_2', 64, 64), ('conv2_1', 64, 128), ('conv2_2', 128, 128),
              ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256), ('conv3_4', 256, 256),
              ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512), ('conv4_4', 512, 512),
              ('conv5_1', 512, 512), ('conv5_2', 512, 512), ('conv5_3', 512, 512), ('conv5_4', 512, 512)]
# layer sequence of Vgg19.build (vgg.py:20-40); 'F' marks the returned features (vgg.py:43); pool5 is never consumed
VGG_SEQ = ['conv1_1', 'conv1_2', 'F', 'P', 'conv2_1', 'conv2_2', 'F', 'P',
           'conv3_1', 'conv3_2', 'conv3_3', 'conv3_4', 'F', 'P',
           'conv4_1', 'conv4_2', 'conv4_3', 'conv4_4', 'F', 'P',
           'conv5_1', 'conv5_2', 'conv5_3', 'conv5_4', 'F']


def synthetic_vgg19_weights(seed=19, width_div=1):
    """The real vgg19.npy is not shipped with the reference (README.md:33-34): He-normal filters, zero biases from
    RandomState(seed).  ``width_div`` shrinks the channel counts for small tests."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, ci, co in VGG_LAYERS:
        ci = ci if ci == 3 else max(ci // width_div, 1)
        co = max(co // width_div, 1)
        w = (rs.randn(3, 3, ci, co) * math.sqrt(2.0 / (9 * ci))).astype(np.float32)
        out[name] = (w, np.zeros((co,), np.float32))
    return out


class Vgg19:
    """reference Vgg19 (vgg.py:7-61).  ``vgg19_path`` is the reference's ``paths.vggnet`` .npy dict
    {'conv1_1': [HWIO filter, bias], ...}; pass ``weights=`` to inject arrays (synthetic benchmark weights)."""

    def __init__(self, vgg19_path=None, weights=None, device='cuda'):
        if weights is None:
            if vgg19_path is None or not os.path.exists(vgg19_path):
                raise Exception('file of pretrained vgg19 does not exist at: ' + str(vgg19_path))   # vgg.py:9-10
            weights = np.load(vgg19_path, encoding='latin1', allow_pickle=True).item()              # vgg.py:11
        self.device = torch.device(device)
        self.params = OrderedDict()
        keys = []
        for name, _, _ in VGG_LAYERS:
            w, b = weights[name][0], weights[name][1]
            self.params[name] = (torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(self.device),
                                 torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(self.device))
            if self.device.type == 'cuda':
                keys += ops.register_constant_filter(self.params[name][0], 'vgg/' + name)     # tf.constant filters (vgg.py:57-61): Winograd form once
        weakref.finalize(self, ops.release_filters, keys)

    def build(self, rgb01):
        """Forward only: images in [-1,1] (the (x+1)/2*255, BGR, mean shift of :262-263 / vgg.py:16-19 is fused into the
        first kernel) -> [conv1_2, conv2_2, conv3_4, conv4_4, conv5_4]."""
        feats, _ = _vgg_forward(self, rgb01.contiguous())
        return feats

    def perceptual_loss(self, gt_image, pred_image):
        """_compute_perceptual_loss (detector_translator_model.py:274-289) on images in [-1,1]."""
        return _PerceptualLossFn.apply(pred_image, gt_image, self)


def _vgg_forward(vgg, images):
    """images [N,H,W,3] in [-1,1] -> (features, tape) where tape records (kind, name, input, output)."""
    n, h, w, _ = images.shape
    dev = images.device
    x = torch.empty_like(images)
    check(lib.kpx_vgg_prep_fwd_f32(images.data_ptr(), n * h * w, x.data_ptr(), ops._stream()), 'kpx_vgg_prep_fwd_f32')
    feats, tape = [], []
    pooled = None                                        # max-pool of the last conv's output, when its epilogue already wrote it
    for pos, item in enumerate(VGG_SEQ):
        if item == 'F':
            feats.append(x)
        elif item == 'P':
            nn_, hh, ww, cc = x.shape
            if pooled is not None:
                y = pooled
            elif x.dtype == ops.BF16 and hh % 2 == 0 and ww % 2 == 0:
                y = torch.empty((nn_, hh // 2, ww // 2, cc), dtype=ops.BF16, device=dev)
                check(lib.kpx_maxpool2_fwd_bf16(x.data_ptr(), nn_, hh, ww, cc, y.data_ptr(), ops._stream()), 'kpx_maxpool2_fwd_bf16')
            elif x.dtype == ops.BF16:
                # odd feature maps (tf 'SAME' pooling: ceil): the bf16 kernel takes even sizes only -- through the fp32 kernel between two
                # conversions, counted like every other fp32 detour of the bf16 configuration
                ops.fallback_uses['other'] += 1
                xf = ops.cast(x, torch.float32)
                yf = torch.empty((nn_, (hh + 1) // 2, (ww + 1) // 2, cc), dtype=torch.float32, device=dev)
                check(lib.kpx_maxpool2_fwd_f32(xf.data_ptr(), nn_, hh, ww, cc, yf.data_ptr(), ops._stream()), 'kpx_maxpool2_fwd_f32')
                y = ops.cast(yf, ops.BF16)
            else:
                y = torch.empty((nn_, (hh + 1) // 2, (ww + 1) // 2, cc), dtype=torch.float32, device=dev)
                check(lib.kpx_maxpool2_fwd_f32(x.data_ptr(), nn_, hh, ww, cc, y.data_ptr(), ops._stream()), 'kpx_maxpool2_fwd_f32')
            tape.append(('pool', None, x, y))
            x = y
            pooled = None
        else:
            wgt, b = vgg.params[item]
            nn_, hh, ww, cc = x.shape
            cout = wgt.shape[3]
            y = torch.empty((nn_, hh, ww, cout), dtype=ops.act_dtype(), device=dev)
            pooled = None
            if FUSE_POOL_FWD and y.dtype == torch.float32 and 'P' in VGG_SEQ[pos + 1:pos + 3] and hh % 2 == 0 and ww % 2 == 0:      # conv -> ('F' ->) pool: pool in the epilogue
                pooled = torch.empty((nn_, hh // 2, ww // 2, cout), dtype=torch.float32, device=dev)
                if not ops.conv3x3_wino43_ex(x, cc, cc, wgt, b, y, cout, cout, ops.ACT_RELU, False, pool_out=pooled):
                    pooled = None
            if pooled is None:
                ops.conv_fwd_raw(x, cc, cc, wgt, b, y, cout, 1, 1, 1, ops.ACT_RELU)
            tape.append(('conv', item, x, y))
            x = y
    return feats, tape


import os as _os
FUSE_FEAT_BWD = True      # feature gradient (pool bwd + L1 bwd + ReLU bwd) in one pass
FUSE_POOL_FWD = True      # 2x2 max-pool written by the producing conv's epilogue
FUSE_RELU_BWD = True      # ReLU backward applied in the epilogue of the data gradient above it


class _PerceptualLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, vgg):
        ops._require_gpu(pred)
        b = pred.shape[0]
        both = ops.concat_batch(gt.detach(), pred.detach())          # tf.concat([gt, pred], 0) (:278)
        feats, tape = _vgg_forward(vgg, both)
        dev = pred.device
        losses = torch.empty(len(feats), dtype=torch.float32, device=dev)
        sc = ops.scratch.get('l1', 8192, dev)
        for k, f in enumerate(feats):
            half = f.numel() // 2
            if f.dtype == ops.BF16:
                check(lib.kpx_l1_pair_fwd_bf16(f.data_ptr(), half, losses[k:].data_ptr(), sc.data_ptr(), ops._stream()), 'kpx_l1_pair_fwd_bf16')
            else:
                check(lib.kpx_l1_pair_fwd_f32(f.data_ptr(), half, losses[k:].data_ptr(), sc.data_ptr(), ops._stream()), 'kpx_l1_pair_fwd_f32')
        ctx.vgg, ctx.tape, ctx.feats, ctx.b = vgg, tape, feats, b
        ctx.per_feature = losses
        return losses.mean().reshape(1)                               # tf.reduce_mean(losses) (:287): 5-element glue

    @staticmethod
    def backward(ctx, g):
        vgg, tape, feats, b = ctx.vgg, ctx.tape, ctx.feats, ctx.b
        g = g.contiguous()
        nfeat = len(feats)
        feat_ids = {id(f): k for k, f in enumerate(feats)}
        d = None                                                      # gradient wrt the current tensor, pred half only
        done = set()                                                  # tensors whose complete gradient (ReLU mask included) is already in d
        conv_outputs = {id(yy) for kk, _, _, yy in tape if kk == 'conv'}

        def feat_grad(f, dy_pooled):
            """ReLU mask of (max-pool backward of dy_pooled + L1 backward of feature f) in one pass (kpx_vgg_feat_bwd_f32)."""
            half = f.numel() // 2
            out = torch.empty((b,) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
            if f.dtype == ops.BF16:
                check(lib.kpx_vgg_feat_bwd_bf16(f.data_ptr(), half, g.data_ptr(), 1.0 / (nfeat * half), dy_pooled.data_ptr() if dy_pooled is not None else None,
                                                b, f.shape[1], f.shape[2], f.shape[3], out.data_ptr(), ops._stream()), 'kpx_vgg_feat_bwd_bf16')
                return out
            check(lib.kpx_vgg_feat_bwd_f32(f.data_ptr(), half, g.data_ptr(), 1.0 / (nfeat * half), dy_pooled.data_ptr() if dy_pooled is not None else None,
                                           b, f.shape[1], f.shape[2], f.shape[3], out.data_ptr(), ops._stream()), 'kpx_vgg_feat_bwd_f32')
            return out
        for kind, name, x, y in reversed(tape):
            if kind == 'pool' and id(x) in feat_ids and x.shape[3] % (8 if x.dtype == ops.BF16 else 4) == 0 and FUSE_FEAT_BWD:
                d = feat_grad(x, d)                                   # every pooled tensor of VGG_SEQ is a returned feature
                done.add(id(x))
                continue
            k = feat_ids.get(id(y))
            if k is not None and d is None and kind == 'conv' and y.shape[3] % (8 if y.dtype == ops.BF16 else 4) == 0 and FUSE_FEAT_BWD:
                d = feat_grad(y, None)                                # the last feature: nothing behind it
                done.add(id(y))
            elif k is not None and id(y) not in done:                 # y is a returned feature: add its L1 gradient
                if y.dtype == ops.BF16:
                    raise ops._lib.KpxError('bf16 configuration: the separate feature-L1 gradient pass is not built (vgg.FUSE_FEAT_BWD off / odd channel counts)')
                half = y.numel() // 2
                dl = torch.empty((b,) + tuple(y.shape[1:]), dtype=torch.float32, device=y.device)
                check(lib.kpx_l1_pair_bwd_f32(y.data_ptr(), half, g.data_ptr(), 1.0 / (nfeat * half), dl.data_ptr(), ops._stream()),
                      'kpx_l1_pair_bwd_f32')
                if d is None:
                    d = dl
                else:
                    ops.axpy_raw_(d, dl)
            xp, yp = x[b:], y[b:]
            if kind == 'pool':
                if xp.dtype == ops.BF16:
                    raise ops._lib.KpxError('bf16 configuration: the separate max-pool backward pass is not built (every pooled tensor is a returned feature)')
                dx = torch.empty(xp.shape, dtype=torch.float32, device=xp.device)
                nn_, hh, ww, cc = xp.shape
                check(lib.kpx_maxpool2_bwd_f32(d.data_ptr(), xp.data_ptr(), nn_, hh, ww, cc, dx.data_ptr(), ops._stream()), 'kpx_maxpool2_bwd_f32')
            else:
                wgt, _ = vgg.params[name]
                if id(y) not in done:
                    ops.act_bwd_raw_(d, yp, ops.ACT_RELU)             # d is ours: in place
                dx = torch.empty(xp.shape, dtype=xp.dtype, device=xp.device)
                # x is the ReLU output of the conv below (not a pooled tensor, not the image): its ReLU backward rides in this epilogue
                if (FUSE_RELU_BWD and d.dtype == ops.BF16 and id(x) in conv_outputs and id(x) not in feat_ids
                        and ops._bf16s_conv(d, wgt.shape[3], wgt.shape[3], wgt, None, dx, xp.shape[3], xp.shape[3], ops.ACT_NONE, True, mask=xp)):
                    done.add(id(x))
                elif (FUSE_RELU_BWD and d.dtype == torch.float32 and id(x) in conv_outputs and id(x) not in feat_ids
                        and ops.conv3x3_wino43_ex(d, wgt.shape[3], wgt.shape[3], wgt, None, dx, xp.shape[3], xp.shape[3], ops.ACT_NONE, True, mask=xp)):
                    done.add(id(x))
                else:
                    ops.conv_dgrad_raw(d, wgt.shape[3], wgt, dx, xp.shape[3], xp.shape[3], 1, 1, 1)
            d = dx
        dpred = torch.empty_like(d)
        check(lib.kpx_vgg_prep_bwd_f32(d.data_ptr(), d.shape[0] * d.shape[1] * d.shape[2], dpred.data_ptr(), ops._stream()), 'kpx_vgg_prep_bwd_f32')
        ctx.tape = ctx.feats = None
        return dpred, None, None
