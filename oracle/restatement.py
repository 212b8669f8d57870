"""CPU restatement of the reference detector_translator hot path (TEST INFRASTRUCTURE ONLY).

torch-CPU fp32, NHWC tensors, HWIO conv kernels, TF variable names -- one function per
reference function, each citing the reference file:line it follows.  Gradients come from
torch autograd over this forward restatement, which makes them an independent check of the
hand-written HIP backward kernels.

Wiring pinned / op numerics UNPINNED (see oracle/__init__.py): the graph structure is checked against the reference's
own files (tests/test_reference_graph.py); the [TF-sem] rules of SURVEY.md Appendix C are encoded here as executable
assumptions about tensorflow-gpu==1.12.0.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # models/networks/layers.py:14
BN_DECAY = 0.999       # tf.contrib.layers.batch_norm default [TF-sem]
INV_STD = 14.3         # utils/model.py:49
VGG_MEAN = [103.939, 116.779, 123.68]  # models/networks/vgg.py:16


# --------------------------------------------------------------------------- TF-sem helpers
def same_pad(in_size, k, s):
    """[TF-sem 1] SAME padding: out=ceil(in/s); extra pixel goes bottom/right."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2, out


def tf_linspace(a, b, n):
    """[TF-sem 4] tf.linspace: step=(b-a)/(n-1) in fp32; element i = a + step*i in fp32."""
    a32, b32 = np.float32(a), np.float32(b)
    step = np.float32((b32 - a32) / np.float32(n - 1))
    return (a32 + step * np.arange(n, dtype=np.float32)).astype(np.float32)


# --------------------------------------------------------------------------- layers.py
def conv(x, kernel, bias, stride, pad=0):
    """layers.conv (models/networks/layers.py:4-10): tf.pad(pad) then conv2d(padding='same').

    x NHWC, kernel HWIO, bias [Cout] or None.
    """
    n, h, w, c = x.shape
    kh, kw, ci, co = kernel.shape
    assert ci == c
    pt, pb, _ = same_pad(h + 2 * pad, kh, stride)
    pl, pr, _ = same_pad(w + 2 * pad, kw, stride)
    xp = F.pad(x.permute(0, 3, 1, 2), (pad + pl, pad + pr, pad + pt, pad + pb))
    wk = kernel.permute(3, 2, 0, 1)
    if wk.dtype == torch.float64:
        wk = wk.contiguous()                 # the float64 CPU path (slow_conv2d) wants a contiguous weight gradient
    y = F.conv2d(xp, wk, bias, stride)
    return y.permute(0, 2, 3, 1)


def batch_norm_train(x, gamma, beta):
    """layers.batch_norm train mode (layers.py:13-14) [TF-sem 3]: fused BN, biased variance."""
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    y = (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta
    return y, mean, var


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var):
    """layers.batch_norm with is_training=False (keypoint_model.py:48-50)."""
    return (x - moving_mean) * torch.rsqrt(moving_var + BN_EPS) * gamma + beta


def moving_update(moving_mean, moving_var, mean, var, count):
    """[TF-sem 3] moving -= (moving-batch)*(1-decay); variance Bessel-corrected."""
    one_minus = np.float32(1.0) - np.float32(BN_DECAY)
    unbiased = var * (float(count) / float(max(count - 1, 1)))
    return (moving_mean - (moving_mean - mean) * one_minus,
            moving_var - (moving_var - unbiased) * one_minus)


def resize2x(x):
    """tf.image.resize_images(x, 2*size) (networks/__init__.py:63,98) [TF-sem 2]:
    legacy bilinear, align_corners=False, no half-pixel centres; x first, then y."""
    n, h, w, c = x.shape

    def idx(n_in):
        dst = np.arange(2 * n_in)
        src = dst.astype(np.float32) * np.float32(0.5)
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        t = (src - lo).astype(np.float32)
        return torch.from_numpy(lo), torch.from_numpy(hi), torch.from_numpy(t)

    ylo, yhi, ty = idx(h)
    xlo, xhi, tx = idx(w)
    tx = tx.view(1, 1, -1, 1)
    ty = ty.view(1, -1, 1, 1)
    top_rows, bot_rows = x[:, ylo], x[:, yhi]
    top = top_rows[:, :, xlo] + (top_rows[:, :, xhi] - top_rows[:, :, xlo]) * tx
    bot = bot_rows[:, :, xlo] + (bot_rows[:, :, xhi] - bot_rows[:, :, xlo]) * tx
    return top + (bot - top) * ty


# --------------------------------------------------------------------------- utils/model.py
def get_coord(x, other_axis, axis_size):
    """model_utils.get_coord (utils/model.py:63-70). x [B,H,W,K]."""
    g = x.mean(dim=other_axis)                                   # :65
    m = g.max(dim=1, keepdim=True).values                        # [TF-sem 5] softmax
    e = torch.exp(g - m)
    prob = e * (1.0 / e.sum(dim=1, keepdim=True))                # :66
    coord = torch.from_numpy(tf_linspace(-1.0, 1.0, axis_size)).view(1, axis_size, 1)  # :67-68
    return (prob * coord).sum(dim=1), prob                       # :69


def get_gaussian_maps(mu, shape_hw, inv_std=INV_STD):
    """model_utils.get_gaussian_maps (utils/model.py:49-60). mu [B,K,2] (x,y) -> [B,H,W,K]."""
    mu_x, mu_y = mu[:, :, 0:1], mu[:, :, 1:2]                    # :50
    y = torch.from_numpy(tf_linspace(-1.0, 1.0, shape_hw[0])).view(1, 1, shape_hw[0], 1)
    x = torch.from_numpy(tf_linspace(-1.0, 1.0, shape_hw[1])).view(1, 1, 1, shape_hw[1])
    g_y = (y - mu_y.unsqueeze(-1)) ** 2                          # :56
    g_x = (x - mu_x.unsqueeze(-1)) ** 2                          # :57
    dist = (g_y + g_x) * float(np.float32(inv_std ** 2))         # :58 python float -> fp32 const
    return torch.exp(-dist).permute(0, 2, 3, 1)                  # :59


# --------------------------------------------------------------------------- networks/__init__.py
class Net:
    """Holds the variable dict (TF names -> torch tensors) and collects BN batch statistics.

    ``train_mode`` mirrors the reference's Python-bool ``is_training``.
    ``bn_log`` records (scope, mean, var, count) per BN call in graph order, i.e. the
    UPDATE_OPS the reference would run (detector_translator_model.py:199-202).
    """

    def __init__(self, params, train_mode=True):
        self.p = params
        self.train_mode = train_mode
        self.bn_log = []

    def conv(self, x, scope, stride=1, pad=0, use_bias=True):
        k = self.p[scope + '/conv2d/kernel']
        b = self.p[scope + '/conv2d/bias'] if use_bias else None
        return conv(x, k, b, stride, pad)

    def bn(self, x, scope):
        g, b = self.p[scope + '/gamma'], self.p[scope + '/beta']
        if self.train_mode:
            y, mean, var = batch_norm_train(x, g, b)
            self.bn_log.append((scope, mean.detach(), var.detach(), x.shape[0] * x.shape[1] * x.shape[2]))
            return y
        return batch_norm_infer(x, g, b, self.p[scope + '/moving_mean'], self.p[scope + '/moving_variance'])


def encoder(net, x, scope):
    """networks.encoder (models/networks/__init__.py:7-26)."""
    s = scope + '/encoder'
    feats = []
    x = F.relu(net.bn(net.conv(x, s + '/conv_1'), s + '/b_norm_1'))          # 7x7 s1 :10-12
    x = F.relu(net.bn(net.conv(x, s + '/conv_2'), s + '/b_norm_2'))          # :13-15
    feats.append(x)
    for i in range(3):                                                       # :17-25
        a, b = i * 2 + 3, i * 2 + 4
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d' % a, stride=2), s + '/b_norm_%d' % a))
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d' % b), s + '/b_norm_%d' % b))
        feats.append(x)
    return feats


def image_encoder(net, x):
    """networks.image_encoder (:29-33)."""
    return [x] + encoder(net, x, 'image_encoder')


def pose_encoder_logits(net, x, final_res=128):
    """networks.pose_encoder (:36-66) up to the 1x1 head logits [B,H,W,K]."""
    s = 'pose_encoder'
    feats = encoder(net, x, s)
    x = feats[-1]
    size = x.shape[1]
    conv_id = 1
    for i in range(4):
        if i > 0:
            x = torch.cat([x, feats[-1 * (i + 1)]], dim=-1)                  # :44
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_0' % conv_id), s + '/b_norm_%d_0' % conv_id))
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_1' % conv_id), s + '/b_norm_%d_1' % conv_id))
        if size == final_res:
            x = net.conv(x, s + '/conv_0')                                   # 1x1 head :54
            break
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_0' % (conv_id + 1)), s + '/b_norm_%d_0' % (conv_id + 1)))
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_1' % (conv_id + 1)), s + '/b_norm_%d_1' % (conv_id + 1)))
        x = resize2x(x)                                                      # :63
        size = x.shape[1]
        conv_id += 2
    return x


def pose_encoder(net, x, final_res=128):
    """networks.pose_encoder (:36-72): logits -> get_coord twice -> stack([x,y], axis=2)."""
    logits = pose_encoder_logits(net, x, final_res)
    gauss_y, _ = get_coord(logits, 2, logits.shape[1])                       # :69
    gauss_x, _ = get_coord(logits, 1, logits.shape[2])                       # :70
    return torch.stack([gauss_x, gauss_y], dim=2)                            # :71


def translator(net, x, final_res=128):
    """networks.translator (:75-102) -> crude [B,H,W,3], mask [B,H,W,1] (sigmoid)."""
    s = 'translator'
    size = x.shape[1]
    conv_id = 1
    while size <= final_res:
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_0' % conv_id), s + '/b_norm_%d_0' % conv_id))
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_1' % conv_id), s + '/b_norm_%d_1' % conv_id))
        if size == final_res:
            crude = net.conv(x, s + '/conv_%d_0' % (conv_id + 1))            # :87
            mask = torch.sigmoid(net.conv(x, s + '/conv_%d_1' % (conv_id + 1)))  # :88-89
            return crude, mask
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_0' % (conv_id + 1)), s + '/b_norm_%d_0' % (conv_id + 1)))
        x = F.relu(net.bn(net.conv(x, s + '/conv_%d_1' % (conv_id + 1)), s + '/b_norm_%d_1' % (conv_id + 1)))
        x = resize2x(x)                                                      # :98
        size = x.shape[1]
        conv_id += 2
    raise AssertionError('translator input larger than final_res')


def img_discr(net, x):
    """networks.img_discr (:141-151): 6x [pad1 + conv4x4 s2 SAME + bias + lrelu .01], D_logit 3x3."""
    s = 'img_discr'
    for i in range(6):
        x = F.leaky_relu(net.conv(x, s + '/conv_%d' % i, stride=2, pad=1), 0.01)
    return net.conv(x, s + '/D_logit', stride=1, pad=1, use_bias=False)


def vgg19(vgg, rgb):
    """Vgg19.build (models/networks/vgg.py:13-43); vgg = {'conv1_1': (HWIO filter, bias), ...}."""
    r, g, b = rgb[..., 0:1], rgb[..., 1:2], rgb[..., 2:3]
    x = torch.cat([b - VGG_MEAN[0], g - VGG_MEAN[1], r - VGG_MEAN[2]], dim=3)  # :17-19

    def cl(x, name):                                                          # :48-55
        return F.relu(conv(x, vgg[name][0], vgg[name][1], 1))

    def pool(x):                                                              # :45-46 (even sizes: no pad)
        return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)

    feats = []
    x = cl(cl(x, 'conv1_1'), 'conv1_2'); feats.append(x); x = pool(x)
    x = cl(cl(x, 'conv2_1'), 'conv2_2'); feats.append(x); x = pool(x)
    x = cl(cl(cl(cl(x, 'conv3_1'), 'conv3_2'), 'conv3_3'), 'conv3_4'); feats.append(x); x = pool(x)
    x = cl(cl(cl(cl(x, 'conv4_1'), 'conv4_2'), 'conv4_3'), 'conv4_4'); feats.append(x); x = pool(x)
    x = cl(cl(cl(cl(x, 'conv5_1'), 'conv5_2'), 'conv5_3'), 'conv5_4'); feats.append(x)
    return feats


# --------------------------------------------------------------------------- detector_translator_model.py
def sigmoid_xent(logits, label):
    """[TF-sem 8] sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))."""
    return torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-torch.abs(logits)))


def forward_pass(net, im, future_im, heat_hw=None, with_vis_maps=True):
    """DetectorTranslatorModel._define_forward_pass (detector_translator_model.py:160-184)."""
    res = im.shape[1]
    heat = heat_hw or res // 4                                               # literal [32,32] at 128
    embeddings = image_encoder(net, im)                                      # :165
    cur_pt = pose_encoder(net, im, final_res=res)                            # :166
    fut_pt = pose_encoder(net, future_im, final_res=res)                     # :167
    cur_map = get_gaussian_maps(cur_pt, [heat, heat])                        # :168
    fut_map = get_gaussian_maps(fut_pt, [heat, heat])                        # :169
    joint = torch.cat([embeddings[-2], cur_map, fut_map], dim=-1)            # :170
    crude, mask = translator(net, joint, final_res=res)                      # :173
    final = im * mask + crude * (1 - mask)                                   # :174
    out = dict(final_output=final, crude_output=crude, mask=mask,
               current_points=cur_pt, future_points=fut_pt,
               current_map_lo=cur_map, future_map_lo=fut_map)
    if with_vis_maps:
        out['current_keypoints_map'] = get_gaussian_maps(cur_pt, [res, res])  # :176
        out['future_keypoints_map'] = get_gaussian_maps(fut_pt, [res, res])   # :177
    return out


def loss_D(net, future_im_pred, future_im):
    """_compute_loss_D (:246-259)."""
    real_ = img_discr(net, future_im)
    fake_ = img_discr(net, future_im_pred)
    real_loss = sigmoid_xent(real_, 1.0).mean()
    fake_loss = sigmoid_xent(fake_, 0.0).mean()
    return real_loss + fake_loss, real_loss, fake_loss


def perceptual_loss(vgg, gt_image, pred_image):
    """_compute_perceptual_loss (:274-289)."""
    feats = vgg19(vgg, torch.cat([gt_image, pred_image], dim=0))             # :278-279
    losses = []
    for f in feats:
        f_gt, f_pred = torch.split(f, f.shape[0] // 2, dim=0)                # :280
        losses.append(torch.abs(f_gt - f_pred).mean())                       # :283-284
    return torch.stack(losses).mean()                                        # :287


def loss_G(net, vgg, future_im_pred, future_im):
    """_compute_loss_G (:261-272)."""
    recon = perceptual_loss(vgg, (future_im + 1) / 2.0 * 255.0, (future_im_pred + 1) / 2.0 * 255.0)
    fake_ = img_discr(net, future_im_pred)
    adv = sigmoid_xent(fake_, 1.0).mean()
    return recon + adv, recon, adv


def exponential_decay(lr0, step, decay_steps, decay):
    """tf.train.exponential_decay non-staircase (:193-195) [TF-sem 7], fp32."""
    p = np.float32(step) / np.float32(decay_steps)
    return np.float32(np.float32(lr0) * np.power(np.float32(decay), p, dtype=np.float32))


class AdamTF:
    """tf.train.AdamOptimizer(lr, beta1=0.5, beta2=0.999, eps=1e-8) [TF-sem 7] in fp32.

    ApplyAdam: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
    var -= m*alpha/(sqrt(v)+eps); the beta powers are fp32 variables multiplied after the update.
    """

    def __init__(self, names, params, beta1=0.5, beta2=0.999, eps=1e-8):
        self.names = list(names)
        self.b1, self.b2, self.eps = np.float32(beta1), np.float32(beta2), np.float32(eps)
        self.m = {n: torch.zeros_like(params[n]) for n in self.names}
        self.v = {n: torch.zeros_like(params[n]) for n in self.names}
        self.b1p, self.b2p = np.float32(beta1), np.float32(beta2)

    def alpha(self, lr):
        return np.float32(np.float32(lr) * np.sqrt(np.float32(1) - self.b2p) / (np.float32(1) - self.b1p))

    def step(self, params, grads, lr):
        a = float(self.alpha(lr))
        with torch.no_grad():
            for n in self.names:
                g = grads[n]
                self.m[n] += (g - self.m[n]) * float(np.float32(1) - self.b1)
                self.v[n] += (g * g - self.v[n]) * float(np.float32(1) - self.b2)
                params[n] -= (self.m[n] * a) / (torch.sqrt(self.v[n]) + float(self.eps))
        self.b1p = np.float32(self.b1p * self.b1)
        self.b2p = np.float32(self.b2p * self.b2)


# --------------------------------------------------------------------------- variable manifest / init
def _conv_vars(out, scope, k, cin, cout, bias=True):
    out[scope + '/conv2d/kernel'] = (k, k, cin, cout)
    if bias:
        out[scope + '/conv2d/bias'] = (cout,)


def _bn_vars(out, scope, c):
    for n in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
        out[scope + '/' + n] = (c,)


def _encoder_vars(out, scope):
    s = scope + '/encoder'
    chans = [(3, 32, 7), (32, 32, 3), (32, 64, 3), (64, 64, 3), (64, 128, 3), (128, 128, 3), (128, 256, 3), (256, 256, 3)]
    for i, (ci, co, k) in enumerate(chans, start=1):
        _conv_vars(out, s + '/conv_%d' % i, k, ci, co)
        _bn_vars(out, s + '/b_norm_%d' % i, co)


def variable_manifest(n_pts, res=128):
    """ordered {TF variable name: shape} of the stage-1 model variables
    (no optimiser slots). Creation order follows _define_forward_pass then _compute_loss."""
    out = OrderedDict()
    _encoder_vars(out, 'image_encoder')
    _encoder_vars(out, 'pose_encoder')
    # pose_encoder decoder (networks/__init__.py:42-66)
    filters, size, conv_id = 128, res // 8, 1
    cin = 256
    enc_c = [32, 64, 128, 256]
    for i in range(4):
        if i > 0:
            cin = cin + enc_c[-1 * (i + 1)]
        for j, (a, b) in enumerate([(cin, filters), (filters, filters)]):
            _conv_vars(out, 'pose_encoder/conv_%d_%d' % (conv_id, j), 3, a, b)
            _bn_vars(out, 'pose_encoder/b_norm_%d_%d' % (conv_id, j), b)
        if size == res:
            _conv_vars(out, 'pose_encoder/conv_0', 1, filters, n_pts)
            break
        for j in range(2):
            _conv_vars(out, 'pose_encoder/conv_%d_%d' % (conv_id + 1, j), 3, filters, filters)
            _bn_vars(out, 'pose_encoder/b_norm_%d_%d' % (conv_id + 1, j), filters)
        size *= 2
        conv_id += 2
        cin = filters
        filters //= 2
    # translator (networks/__init__.py:76-101)
    filters, size, conv_id = 256, res // 4, 1
    cin = 128 + 2 * n_pts
    while size <= res:
        for j, (a, b) in enumerate([(cin, filters), (filters, filters)]):
            _conv_vars(out, 'translator/conv_%d_%d' % (conv_id, j), 3, a, b)
            _bn_vars(out, 'translator/b_norm_%d_%d' % (conv_id, j), b)
        if size == res:
            _conv_vars(out, 'translator/conv_%d_0' % (conv_id + 1), 3, filters, 3)
            _conv_vars(out, 'translator/conv_%d_1' % (conv_id + 1), 3, filters, 1)
            break
        for j in range(2):
            _conv_vars(out, 'translator/conv_%d_%d' % (conv_id + 1, j), 3, filters, filters)
            _bn_vars(out, 'translator/b_norm_%d_%d' % (conv_id + 1, j), filters)
        size *= 2
        conv_id += 2
        cin = filters
        filters //= 2
    # img_discr (networks/__init__.py:142-150)
    c = 3
    ch = 64
    for i in range(6):
        _conv_vars(out, 'img_discr/conv_%d' % i, 4, c, ch)
        c, ch = ch, ch * 2
    _conv_vars(out, 'img_discr/D_logit', 3, c, 1, bias=False)
    return out


def init_variables(n_pts, res=128, seed=1234):
    """xavier-uniform kernels ([TF-sem 6]), zero biases, BN gamma=1 beta=0 moving (0,1),
    drawn from RandomState(seed) in manifest order. Returns {name: np.float32 array}."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in variable_manifest(n_pts, res).items():
        if name.endswith('/kernel'):
            kh, kw, ci, co = shape
            lim = math.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
            out[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        elif name.endswith('/gamma') or name.endswith('/moving_variance'):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


VGG_LAYERS = [('conv1_1', 3, 64), ('conv1_2', 64, 64), ('conv2_1', 64, 128), ('conv2_2', 128, 128),
              ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256), ('conv3_4', 256, 256),
              ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512), ('conv4_4', 512, 512),
              ('conv5_1', 512, 512), ('conv5_2', 512, 512), ('conv5_3', 512, 512), ('conv5_4', 512, 512)]


def synthetic_vgg(seed=19, width_div=1):
    """the real vgg19.npy is not shipped -> He-normal filters, zero biases, RandomState(seed).
    ``width_div`` shrinks channel counts for small test cases only."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, ci, co in VGG_LAYERS:
        ci = ci if ci == 3 else max(ci // width_div, 1)
        co = max(co // width_div, 1)
        w = (rs.randn(3, 3, ci, co) * math.sqrt(2.0 / (9 * ci))).astype(np.float32)
        out[name] = (w, np.zeros((co,), np.float32))
    return out


def synthetic_pair(batch, res=128, seed0=0, seed1=1):
    """SURVEY 8d synthetic Penn-shaped inputs: uint8~U{0..255}, x/255*2-1 (image_pair_dataloader.py:65-70)."""
    def one(seed):
        u = np.random.RandomState(seed).randint(0, 256, size=(batch, res, res, 3)).astype(np.float32)
        return (u / np.float32(255.0) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    return one(seed0), one(seed1)


# --------------------------------------------------------------------------- one train step
class TrainState:
    """Everything tf.train.Saver would hold for stage 1, as torch-CPU tensors."""

    def __init__(self, variables, vgg, lr_cfg=(1e-4, 20000, 0.95), dtype=torch.float32):
        """``dtype=torch.float64`` runs the same restatement in double precision: the arbiter the tests use to tell an fp32
        implementation's rounding noise (amplified by the discontinuous L1 / ReLU / max-pool gradient) from a wiring error."""
        self.dtype = dtype
        self.params = OrderedDict((k, torch.from_numpy(np.array(v, copy=True)).to(dtype)) for k, v in variables.items())
        self.vgg = {k: (torch.from_numpy(np.asarray(w)).to(dtype), torch.from_numpy(np.asarray(b)).to(dtype)) for k, (w, b) in vgg.items()}
        train_names = [k for k in self.params if not ('moving_' in k)]
        self.d_names = [k for k in train_names if 'img_discr' in k]          # :191-192
        self.g_names = [k for k in train_names if 'img_discr' not in k]
        self.opt_D = AdamTF(self.d_names, self.params)                        # :198
        self.opt_G = AdamTF(self.g_names, self.params)                        # :201
        self.global_step = 0
        self.lr_cfg = lr_cfg

    def lr(self):
        return exponential_decay(self.lr_cfg[0], self.global_step, self.lr_cfg[1], self.lr_cfg[2])


def _grads(loss, params, names):
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(params[n])) for n, g in zip(names, gs)}


def train_step(state, im, future_im, same_batch=True, im_G=None, future_im_G=None):
    """DetectorTranslatorModel.train_step (:79-117): D-run then G-run.

    The reference feeds a *new* batch to each sess.run; the benchmark convention
     is D-run + G-run on the same batch (``same_batch=True``).  BN moving statistics
    are updated by the G-run only (UPDATE_OPS gate, :199-202).  Returns a dict of scalars/tensors.
    """
    p = state.params
    dtype = getattr(state, 'dtype', torch.float32)
    im = torch.as_tensor(im).to(dtype)
    future_im = torch.as_tensor(future_im).to(dtype)
    lr = state.lr()
    for n in state.d_names + state.g_names:
        p[n].requires_grad_(True)

    # ---- D run (:93): generator forward, D loss, Adam on D vars
    net = Net(p, train_mode=True)
    fwd = forward_pass(net, im, future_im, with_vis_maps=False)
    l_d, l_real, l_fake = loss_D(net, fwd['final_output'], future_im)
    g_d = _grads(l_d, p, state.d_names)
    state.opt_D.step(p, g_d, lr)

    # ---- G run (:94): (re)forward with the updated D, G loss, Adam on G vars, BN moving update
    if not same_batch:
        im, future_im = torch.as_tensor(im_G).to(dtype), torch.as_tensor(future_im_G).to(dtype)
    net = Net(p, train_mode=True)
    fwd = forward_pass(net, im, future_im, with_vis_maps=False)
    l_g, l_recon, l_adv = loss_G(net, state.vgg, fwd['final_output'], future_im)
    g_g = _grads(l_g, p, state.g_names)
    state.opt_G.step(p, g_g, lr)
    with torch.no_grad():
        for scope, mean, var, count in net.bn_log:
            if 'img_discr' in scope:
                continue
            mm, mv = moving_update(p[scope + '/moving_mean'], p[scope + '/moving_variance'], mean, var, count)
            p[scope + '/moving_mean'].copy_(mm)
            p[scope + '/moving_variance'].copy_(mv)
    state.global_step += 1
    for n in state.d_names + state.g_names:
        p[n].requires_grad_(False)
    f = lambda t: float(t.detach())
    return dict(loss_D=f(l_d), loss_D_real=f(l_real), loss_D_fake=f(l_fake),
                loss_G=f(l_g), loss_G_recon=f(l_recon), loss_G_adv=f(l_adv),
                lr=float(lr), grads_D=g_d, grads_G=g_g,
                final_output=fwd['final_output'].detach(), crude_output=fwd['crude_output'].detach(), mask=fwd['mask'].detach(),
                current_points=fwd['current_points'].detach(),
                future_points=fwd['future_points'].detach())


def train_step_dt_gradients(state, im, future_im):
    """The discriminator and TRANSLATOR gradients of ``train_step`` (same-batch convention) without the autograd tape of the key-point detector
    and the image encoder: their forward runs under no_grad (the joint embedding is a constant of the translator's gradient), so a float64
    run at the bench batch (B=32) fits a test host -- the arbiter for the large-launch code paths of the discriminator / translator / VGG19
    kernels (tests/test_model_gpu.py).  Applies the D update like train_step; returns {'grads_D', 'grads_T'} (translator variables only)."""
    p = state.params
    dtype = getattr(state, 'dtype', torch.float32)
    im = torch.as_tensor(im).to(dtype)
    future_im = torch.as_tensor(future_im).to(dtype)
    lr = state.lr()
    t_names = [n for n in state.g_names if n.startswith('translator/')]
    for n in state.d_names + t_names:
        p[n].requires_grad_(True)
    res = im.shape[1]
    net = Net(p, train_mode=True)
    with torch.no_grad():
        embeddings = image_encoder(net, im)
        cur_pt, fut_pt = pose_encoder(net, im, final_res=res), pose_encoder(net, future_im, final_res=res)
        joint = torch.cat([embeddings[-2], get_gaussian_maps(cur_pt, [res // 4, res // 4]), get_gaussian_maps(fut_pt, [res // 4, res // 4])], dim=-1)
        del embeddings
        crude, mask = translator(net, joint, final_res=res)
        final_d = im * mask + crude * (1 - mask)
    l_d, _, _ = loss_D(net, final_d, future_im)
    g_d = _grads(l_d, p, state.d_names)
    del l_d
    state.opt_D.step(p, g_d, lr)
    net = Net(p, train_mode=True)
    crude, mask = translator(net, joint, final_res=res)
    final = im * mask + crude * (1 - mask)
    l_g, _, _ = loss_G(net, state.vgg, final, future_im)
    g_t = _grads(l_g, p, t_names)
    for n in state.d_names + t_names:
        p[n].requires_grad_(False)
    return dict(grads_D=g_d, grads_T=g_t)


def train_step_data_parallel(state, local_batches):
    """One data-parallel train step over ``len(local_batches)`` replicas: every replica runs the reference's D-run and G-run
    (:79-117) on ITS local batch from identical weights, batch-norm statistics are per replica (the reference has no cross-device batch
    norm), the gradient of each optimiser update is the MEAN over the replicas (sum all-reduce, 1/world in the optimiser), and both Adam
    updates are applied once.  ``local_batches`` = [(im, future_im), ...].  Returns per-replica losses / frames and the mean gradients;
    the moving statistics kept are replica 0's (rank 0 writes the checkpoint)."""
    p = state.params
    dtype = getattr(state, 'dtype', torch.float32)
    lr = state.lr()
    world = len(local_batches)
    for n in state.d_names + state.g_names:
        p[n].requires_grad_(True)
    batches = [(torch.as_tensor(a).to(dtype), torch.as_tensor(b).to(dtype)) for a, b in local_batches]
    per = [dict() for _ in range(world)]
    f = lambda t: float(t.detach())
    # ---- D run on every replica, mean gradient, ONE Adam update
    g_d = None
    for r, (im, fut) in enumerate(batches):
        net = Net(p, train_mode=True)
        fwd = forward_pass(net, im, fut, with_vis_maps=False)
        l_d, l_real, l_fake = loss_D(net, fwd['final_output'], fut)
        g = _grads(l_d, p, state.d_names)
        g_d = g if g_d is None else {n: g_d[n] + g[n] for n in g}
        per[r].update(loss_D=f(l_d), loss_D_real=f(l_real), loss_D_fake=f(l_fake))
    g_d = {n: t / world for n, t in g_d.items()}
    state.opt_D.step(p, g_d, lr)
    # ---- G run on every replica against the UPDATED discriminator, mean gradient, ONE Adam update
    g_g, bn_log0 = None, None
    for r, (im, fut) in enumerate(batches):
        net = Net(p, train_mode=True)
        fwd = forward_pass(net, im, fut, with_vis_maps=False)
        l_g, l_recon, l_adv = loss_G(net, state.vgg, fwd['final_output'], fut)
        g = _grads(l_g, p, state.g_names)
        g_g = g if g_g is None else {n: g_g[n] + g[n] for n in g}
        per[r].update(loss_G=f(l_g), loss_G_recon=f(l_recon), loss_G_adv=f(l_adv), final_output=fwd['final_output'].detach(),
                      current_points=fwd['current_points'].detach(), future_points=fwd['future_points'].detach())
        if r == 0:
            bn_log0 = net.bn_log
    g_g = {n: t / world for n, t in g_g.items()}
    state.opt_G.step(p, g_g, lr)
    with torch.no_grad():
        for scope, mean, var, count in bn_log0:
            if 'img_discr' in scope:
                continue
            mm, mv = moving_update(p[scope + '/moving_mean'], p[scope + '/moving_variance'], mean, var, count)
            p[scope + '/moving_mean'].copy_(mm)
            p[scope + '/moving_variance'].copy_(mv)
    state.global_step += 1
    for n in state.d_names + state.g_names:
        p[n].requires_grad_(False)
    return dict(replicas=per, grads_D=g_d, grads_G=g_g, lr=float(lr))


# --------------------------------------------------------------------------- stage-2 decoder + evaluate.py rollout
N_FUTURE_FRAMES = 32      # models/final_model.py:11


def stage2_decoder_manifest(n_pts, n_action=9, cell_info=(1024, 1024), vae_dim=64):
    """Variables of networks.vae_decoder (models/networks/__init__.py:116-129) [TF-sem naming]."""
    out = OrderedDict()
    fc_in = vae_dim + 2 * n_pts + n_action
    out['vae_decoder/fully_connected/weights'] = (fc_in, 32)                 # tf.contrib.layers.fully_connected(.., 32) :120
    out['vae_decoder/fully_connected/biases'] = (32,)
    prev = 32
    for i, units in enumerate(cell_info):                                      # layers.lstm_model (layers.py:17-21)
        out['vae_decoder/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel' % i] = (prev + units, 4 * units)
        out['vae_decoder/multi_rnn_cell/cell_%d/basic_lstm_cell/bias' % i] = (4 * units,)
        prev = units
    out['vae_decoder/fully_connected/W'] = (cell_info[-1], 2 * n_pts)         # layers.to_coord (layers.py:24-28)
    out['vae_decoder/fully_connected/b'] = (2 * n_pts,)
    return out


def init_stage2_decoder(n_pts, n_action=9, cell_info=(1024, 1024), vae_dim=64, seed=4321):
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in stage2_decoder_manifest(n_pts, n_action, cell_info, vae_dim).items():
        if name.endswith('/W'):
            out[name] = (rs.randn(*shape) * 0.02).astype(np.float32)           # random_normal_initializer(stddev=0.02)
        elif len(shape) == 2:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))                       # glorot_uniform / xavier
            out[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def vae_decoder(p, x, f_pt, act_code, cell_info, n_pts):
    """networks.vae_decoder (models/networks/__init__.py:116-129): fc(relu) -> 2-layer LSTMCell x 32 steps (zero inputs after
    step 0) -> to_coord tanh.  LSTMCell [TF-sem]: gates i,j,f,o; forget_bias 1.0; state (c,h) zero-initialised."""
    s = 'vae_decoder'
    inp = F.relu(torch.cat([x, f_pt, act_code], dim=-1) @ p[s + '/fully_connected/weights'] + p[s + '/fully_connected/biases'])
    b = x.shape[0]
    c = [torch.zeros(b, u) for u in cell_info]
    h = [torch.zeros(b, u) for u in cell_info]
    outs = []
    for step in range(N_FUTURE_FRAMES):
        xin = inp if step == 0 else torch.zeros_like(inp)
        for l, u in enumerate(cell_info):
            g = torch.cat([xin, h[l]], dim=-1) @ p[s + '/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel' % l] + \
                p[s + '/multi_rnn_cell/cell_%d/basic_lstm_cell/bias' % l]
            i, j, f, o = torch.split(g, u, dim=-1)
            c[l] = c[l] * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            h[l] = torch.tanh(c[l]) * torch.sigmoid(o)
            xin = h[l]
        outs.append(torch.tanh(h[-1] @ p[s + '/fully_connected/W'] + p[s + '/fully_connected/b']))
    return torch.stack(outs, dim=1)                                            # [B, 32, 2K]


def final_model_forward(params, im, action_code, z, n_pts, cell_info=(1024, 1024)):
    """FinalModel.build (models/final_model.py:49-122), inference-mode batch norm; z is injected (the reference draws it
    unseeded, :71).  Returns pred_im_seq / mask / pred_im_crude / fut_pt_raw."""
    net = Net(params, train_mode=False)
    res = im.shape[1]
    b = im.shape[0]
    t = N_FUTURE_FRAMES
    tiled_im = im.unsqueeze(1).expand(b, t, res, res, 3).reshape(b * t, res, res, 3)              # :57-59
    emb = image_encoder(net, im)[-2]                                                              # :61-62
    emb_t = emb.unsqueeze(1).expand(b, t, *emb.shape[1:]).reshape(b * t, *emb.shape[1:])          # :63-66
    first_pt = pose_encoder(net, im, final_res=res)                                               # :68
    pred_seq = vae_decoder(params, z, first_pt.reshape(b, n_pts * 2), action_code, cell_info, n_pts)   # :71-77
    pred_seq = pred_seq.reshape(b, t, n_pts, 2)
    heat = res // 4
    cur_map = get_gaussian_maps(first_pt, [heat, heat])                                           # :79-84
    cur_map_t = cur_map.unsqueeze(1).expand(b, t, *cur_map.shape[1:]).reshape(b * t, *cur_map.shape[1:])
    pred_map = get_gaussian_maps(pred_seq.reshape(b * t, n_pts, 2), [heat, heat])                 # :88-92
    joint = torch.cat([emb_t, cur_map_t, pred_map], dim=-1)                                       # :94
    crude, mask = translator(net, joint, final_res=res)                                           # :95
    final = tiled_im * mask + crude * (1 - mask)                                                  # :96
    return dict(pred_im_seq=torch.clamp(final, -1, 1).reshape(b, t, res, res, 3),                  # :98-99
                pred_im_crude=torch.clamp(crude, -1, 1).reshape(b, t, res, res, 3),
                mask=mask.reshape(b, t, res, res, 1), fut_pt_raw=pred_seq, first_pt=first_pt)


# --------------------------------------------------------------------------- stage-2 training: MotionGeneratorModel (last item)
def _lstm_vars(out, scope, in_size, cell_info):
    prev = in_size
    for i, units in enumerate(cell_info):                                      # layers.lstm_model (layers.py:17-21)
        out[scope + '/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel' % i] = (prev + units, 4 * units)
        out[scope + '/multi_rnn_cell/cell_%d/basic_lstm_cell/bias' % i] = (4 * units,)
        prev = units


def motion_generator_manifest(n_pts, n_action=9, cell_info=(1024, 1024), vae_dim=64, discr_cells=(1024, 1024)):
    """Trainable variables of models/motion_generator_model.py: vae_encoder (:105-114 of networks/__init__.py; dynamic_rnn names
    its cell variables under 'rnn/'), vae_decoder (:116-129) and seq_discr (:132-138)."""
    out = OrderedDict()
    _lstm_vars(out, 'vae_encoder/rnn', 2 * n_pts, cell_info)
    out['vae_encoder/fully_connected/weights'] = (cell_info[-1] + 2 * n_pts + n_action, 2 * vae_dim)
    out['vae_encoder/fully_connected/biases'] = (2 * vae_dim,)
    out.update(stage2_decoder_manifest(n_pts, n_action, cell_info, vae_dim))
    _lstm_vars(out, 'seq_discr/rnn', 2 * n_pts, discr_cells)
    out['seq_discr/fully_connected/weights'] = (discr_cells[-1], 1)
    out['seq_discr/fully_connected/biases'] = (1,)
    return out


def init_motion_generator(n_pts, n_action=9, cell_info=(1024, 1024), vae_dim=64, discr_cells=(1024, 1024), seed=777):
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in motion_generator_manifest(n_pts, n_action, cell_info, vae_dim, discr_cells).items():
        if name.endswith('/W'):
            out[name] = (rs.randn(*shape) * 0.02).astype(np.float32)
        elif len(shape) == 2:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def _dynamic_rnn(p, scope, x, cell_info):
    """tf.nn.dynamic_rnn over a MultiRNNCell of LSTMCells with zero initial state: x [B,T,In] -> top-layer outputs [B,T,U]."""
    b, t, _ = x.shape
    c = [torch.zeros(b, u) for u in cell_info]
    h = [torch.zeros(b, u) for u in cell_info]
    outs = []
    for step in range(t):
        xin = x[:, step]
        for l, u in enumerate(cell_info):
            g = torch.cat([xin, h[l]], dim=-1) @ p[scope + '/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel' % l] + \
                p[scope + '/multi_rnn_cell/cell_%d/basic_lstm_cell/bias' % l]
            i, j, f, o = torch.split(g, u, dim=-1)
            c[l] = c[l] * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            h[l] = torch.tanh(c[l]) * torch.sigmoid(o)
            xin = h[l]
        outs.append(h[-1])
    return torch.stack(outs, dim=1)


def vae_encoder(p, x, f_pt, act_code, cell_info, vae_dim):
    """networks.vae_encoder (:105-114): LSTM over the real sequence, last output ++ first point ++ action -> fully_connected
    (tf.contrib default activation: ReLU, so mu and stddev are non-negative) -> (mu, stddev)."""
    out = _dynamic_rnn(p, 'vae_encoder/rnn', x, cell_info)
    logit = F.relu(torch.cat([out[:, -1, :], f_pt, act_code], dim=-1) @ p['vae_encoder/fully_connected/weights'] +
                   p['vae_encoder/fully_connected/biases'])
    return logit[:, :vae_dim], logit[:, vae_dim:]


def seq_discr(p, x, discr_cells=(1024, 1024)):
    """networks.seq_discr (:132-138): 2-layer LSTM, fully_connected(outputs, 1) (ReLU again), logit of the last step [B,1]."""
    out = _dynamic_rnn(p, 'seq_discr/rnn', x, discr_cells)
    logit = F.relu(out @ p['seq_discr/fully_connected/weights'] + p['seq_discr/fully_connected/biases'])
    return logit[:, -1, :]


def motion_generator_losses(p, keypoints, real_seq, action_code, eps, n_pts, cell_info=(1024, 1024), vae_dim=64, discr_cells=(1024, 1024)):
    """MotionGeneratorModel._define_forward_pass (training branch, :137-150) + _compute_loss_D / _compute_loss_G (:257-308);
    ``eps`` is the N(0,1) draw of tf.random_normal (:146), injected."""
    b = keypoints.shape[0]
    first_pt = keypoints.reshape(b, n_pts * 2)
    real = real_seq.reshape(b, N_FUTURE_FRAMES, n_pts * 2)
    mu, stddev = vae_encoder(p, real, first_pt, action_code, cell_info, vae_dim)
    z = mu + stddev * eps
    pred = vae_decoder(p, z, first_pt, action_code, cell_info, n_pts)
    real_, fake_ = seq_discr(p, real, discr_cells), seq_discr(p, pred, discr_cells)
    loss_d_real = sigmoid_xent(real_, 1.0).mean()
    loss_d_fake = sigmoid_xent(fake_, 0.0).mean()
    recon = (1000 * torch.abs(pred - real)).mean()
    kl = (0.5 * torch.sum(mu * mu + stddev * stddev - torch.log(1e-8 + stddev * stddev) - 1, dim=1)).mean()
    adv = sigmoid_xent(fake_, 1.0).mean()
    return dict(loss_D=loss_d_real + loss_d_fake, loss_D_real=loss_d_real, loss_D_fake=loss_d_fake,
                loss_G=kl + recon + adv, loss_G_recon=recon, loss_G_kl=kl, loss_G_adv=adv, pred_seq=pred, mu=mu, stddev=stddev)


class MotionTrainState:
    def __init__(self, params, n_pts, n_action=9, cell_info=(1024, 1024), vae_dim=64, discr_cells=(1024, 1024), lr_cfg=None):
        self.params = OrderedDict((k, torch.as_tensor(np.asarray(v)).clone()) for k, v in params.items())
        self.cfg = dict(n_pts=n_pts, cell_info=tuple(cell_info), vae_dim=vae_dim, discr_cells=tuple(discr_cells))
        self.d_names = [n for n in self.params if 'discr' in n]                # :176-178
        self.g_names = [n for n in self.params if 'discr' not in n]
        self.opt_D = AdamTF(self.d_names, self.params)
        self.opt_G = AdamTF(self.g_names, self.params)
        self.global_step = 0
        self.lr_cfg = lr_cfg or {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}


def motion_train_step(state, keypoints, real_seq, action_code, eps_D, eps_G):
    """MotionGeneratorModel.train_step (:80-104): D-run then G-run, each a full forward (with its own random_normal draw)."""
    p = state.params
    kp, rs, ac = (torch.as_tensor(a) for a in (keypoints, real_seq, action_code))
    lr = exponential_decay(state.lr_cfg['start_val'], state.global_step, state.lr_cfg['step'], state.lr_cfg['decay'])
    for n in state.d_names + state.g_names:
        p[n].requires_grad_(True)
    out_d = motion_generator_losses(p, kp, rs, ac, torch.as_tensor(eps_D), **state.cfg)
    g_d = _grads(out_d['loss_D'], p, state.d_names)
    state.opt_D.step(p, g_d, lr)
    out_g = motion_generator_losses(p, kp, rs, ac, torch.as_tensor(eps_G), **state.cfg)
    g_g = _grads(out_g['loss_G'], p, state.g_names)
    state.opt_G.step(p, g_g, lr)
    state.global_step += 1
    for n in state.d_names + state.g_names:
        p[n].requires_grad_(False)
    f = lambda t: float(t.detach())
    return dict(loss_D=f(out_d['loss_D']), loss_G=f(out_g['loss_G']), loss_G_recon=f(out_g['loss_G_recon']), loss_G_kl=f(out_g['loss_G_kl']),
                loss_G_adv=f(out_g['loss_G_adv']), lr=float(lr), grads_D=g_d, grads_G=g_g, pred_seq=out_g['pred_seq'].detach())
