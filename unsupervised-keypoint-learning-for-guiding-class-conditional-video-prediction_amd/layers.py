"""Layer wrappers with the reference's signatures (reference: models/networks/layers.py).

``conv`` and ``batch_norm`` keep the reference's argument names / defaults and variable naming
(``<scope>/conv2d/{kernel,bias}``, ``<scope>/{beta,gamma,moving_mean,moving_variance}``); the arithmetic is the HIP
kernels of libkpx_hip.so.  Extra keyword arguments (``act``, ``cin``, ``groups``...) expose fusions the TF graph did as
separate ops (bias+activation epilogue, BN+ReLU, batched weight-sharing calls).
"""
import torch

from . import ops
from .variables import Sym, default_store, is_sym

ACT_NONE, ACT_RELU, ACT_LRELU = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LRELU
EXACT_ZERO_BIAS_GRAD = True      # see conv_bn_relu
import os as _os
FOLD_BN_INFERENCE = True      # see conv_bn_relu
# Diagnostics of the bf16 configuration (tests/test_model_gpu.py::test_bf16_error_budget, scratch/bf16_error_budget.py; nothing in the product
# sets them): TRACE, a list that receives (scope of the layer's filter, output tensor) for every conv_bn_relu unit; F32_OUT_SCOPES, scopes whose
# output stays fp32 in the bf16 configuration (what networks.py declares with out_f32=True, extended for what-if runs).
TRACE = None
F32_OUT_SCOPES = set()


def conv(x, channels, kernel=4, stride=2, pad=0, use_bias=True, scope='conv_0', act=ACT_NONE, cin=None, head31=False, bias_grad=True, bn_stats=False,
         f43_fwd=True, input_act=ACT_NONE, act_bwd_by_consumer=False, out_f32=False):
    """reference layers.conv (layers.py:4-10): tf.pad(pad) + tf.layers.conv2d(padding='same', xavier, bias).
    f43_fwd (a kernel attribute of the LAYER, recorded with its filter variable when the layer is declared): whether the forward of a 3x3
    stride-1 layer may run the F(4x4,3x3) Winograd kernel (ops.WINO43: an accuracy policy taken from a whole-step measurement).
    act_bwd_by_consumer / input_act: a contract between two chained layers -- the output of a layer with ``act`` and act_bwd_by_consumer=True
    is read by exactly ONE layer, declared with input_act=act, whose data gradient applies the activation backward in its epilogue."""
    st = default_store()
    channels = int(channels)          # the reference passes float filter counts after `filters /= 2` ([TF-sem 9])
    cin_ = int(cin) if cin is not None else int(x.shape[-1])
    with st.variable_scope(scope), st.variable_scope('conv2d'):
        kname = st.get_variable('kernel', (kernel, kernel, cin_, channels), 'kernel_head31' if head31 else 'kernel')
        bname = st.get_variable('bias', (channels,), 'zeros') if use_bias else None
    if not st.materialised:
        st.layer_attrs.setdefault(kname, {})['f43_fwd'] = bool(f43_fwd)
    if is_sym(x):
        n, h, w, _ = x.shape
        _, _, ho = ops.same_pad(h + 2 * pad, kernel, stride)
        _, _, wo = ops.same_pad(w + 2 * pad, kernel, stride)
        return Sym(n, ho, wo, channels)
    w_, wg = st.param(kname)
    b_, bg = st.param(bname) if use_bias else (None, None)
    # out_f32: the output stays fp32 in the bf16 configuration too (tensors read by fp32-only consumers: head blend, the losses)
    return ops.conv2d(x, w_, b_, stride=stride, pad=pad, act=act, cin=cin, w_grad_out=wg, b_grad_out=bg, bias_grad=bias_grad, bn_stats=bn_stats,
                      input_act=input_act, act_bwd_by_consumer=act_bwd_by_consumer, out_dtype=torch.float32 if out_f32 else None)


def conv1x1_keypoints(x, channels, scope='conv_0'):
    """layers.conv(x, channels, kernel=1, stride=1) followed by the separable-softmax key-point head (reference networks/__init__.py:54,
    68-72) as one op: same variables as ``conv`` (<scope>/conv2d/{kernel,bias}), the logits are never materialised (ops.KeypointHeadProjFn).
    Returns (mu [B,K,2] as (x,y), prob_y [B,H,K], prob_x [B,W,K])."""
    st = default_store()
    channels = int(channels)
    cin_ = int(x.shape[-1])
    with st.variable_scope(scope), st.variable_scope('conv2d'):
        kname = st.get_variable('kernel', (1, 1, cin_, channels), 'kernel')
        bname = st.get_variable('bias', (channels,), 'zeros')
    if is_sym(x):
        n, h, w, _ = x.shape
        return Sym(n, channels, 2), Sym(n, h, channels), Sym(n, w, channels)
    w_, wg = st.param(kname)
    b_, bg = st.param(bname)
    return ops.keypoint_head_proj(x, w_, b_, w_grad_out=wg, b_grad_out=bg)


def batch_norm(x, train_mode, scope='batch_norm', act=ACT_NONE, groups=1, update_moving=True, out_f32=False):
    """reference layers.batch_norm (layers.py:13-14): contrib batch_norm(eps=1e-5, center, scale, is_training)."""
    st = default_store()
    c = int(x.shape[-1])
    with st.variable_scope(scope):
        beta = st.get_variable('beta', (c,), 'zeros')
        gamma = st.get_variable('gamma', (c,), 'ones')
        mm = st.get_variable('moving_mean', (c,), 'moving_zeros')
        mv = st.get_variable('moving_variance', (c,), 'moving_ones')
    if is_sym(x):
        return Sym(*x.shape)
    g_, gg = st.param(gamma)
    b_, bg = st.param(beta)
    return ops.batch_norm(x, g_, b_, st[mm], st[mv], train=bool(train_mode), act=act, groups=groups,
                          update_moving=update_moving and bool(train_mode), g_grad_out=gg, b_grad_out=bg, out_f32=out_f32)


def conv_bn_relu(x, channels, kernel, stride, train_mode, conv_scope, bn_scope, groups=1, update_moving=True, cin=None, f43_fwd=True, out_f32=False):
    """conv -> batch_norm -> relu, the repeating unit of every generator network (reference networks/__init__.py:10-12).

    The conv keeps its bias variable (reference layers.py:4 default use_bias=True), but d(loss)/d(bias) is exactly
    zero when a train-mode batch norm follows (the batch mean removes any per-channel constant), so it is not computed: the
    flat gradient bucket holds 0 there and Adam leaves those biases at their initial value.  The reference computes fp32
    rounding noise for them, which its Adam turns into a +-lr random walk that batch norm again cancels in the forward."""
    if not train_mode and not is_sym(x) and FOLD_BN_INFERENCE and x.is_cuda and not torch.is_grad_enabled():
        # inference (KeypointModel / FinalModel): the moving-statistics batch norm is a constant per-channel affine map -- folded into the
        # filter and the bias once per checkpoint, so the layer is one conv with a ReLU epilogue and no pass over its activation
        st = default_store()
        with st.variable_scope(conv_scope), st.variable_scope('conv2d'):
            kname, bname = st.scoped('kernel'), st.scoped('bias')
        with st.variable_scope(bn_scope):
            names = [st.scoped(n) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
        if all(n in st.vars for n in [kname, bname] + names):
            wf, bf = st.folded_conv_bn(kname, bname, names[0], names[1], names[2], names[3], ops.BN_EPS)
            return ops.conv2d(x, wf, bf, stride=stride, pad=0, act=ACT_RELU, cin=cin)
    x = conv(x, channels, kernel=kernel, stride=stride, scope=conv_scope, cin=cin, bias_grad=not (EXACT_ZERO_BIAS_GRAD and train_mode),
             bn_stats=bool(train_mode), f43_fwd=f43_fwd)       # the conv epilogue delivers the batch statistics when it can
    if (TRACE is not None or F32_OUT_SCOPES) and not is_sym(x):
        st = default_store()
        with st.variable_scope(conv_scope):
            full = st.scoped('')
        y = batch_norm(x, train_mode, scope=bn_scope, act=ACT_RELU, groups=groups, update_moving=update_moving, out_f32=out_f32 or full.rstrip('/') in F32_OUT_SCOPES)
        if TRACE is not None:
            TRACE.append((full.rstrip('/'), y.detach()))
        return y
    return batch_norm(x, train_mode, scope=bn_scope, act=ACT_RELU, groups=groups, update_moving=update_moving, out_f32=out_f32)


def fully_connected(x, num_outputs, scope='fully_connected', act=ACT_RELU, trainable=False):
    """tf.contrib.layers.fully_connected (default activation ReLU; variables <scope>/{weights,biases})
    (reference networks/__init__.py:112,120,137).  ``trainable``: differentiable path writing gradients into the flat bucket."""
    st = default_store()
    with st.variable_scope(scope):
        wn = st.get_variable('weights', (int(x.shape[-1]), int(num_outputs)), 'glorot2d')
        bn = st.get_variable('biases', (int(num_outputs),), 'zeros')
    if is_sym(x):
        return Sym(x.shape[0], num_outputs)
    if trainable:
        (w_, wg), (b_, bg) = st.param(wn), st.param(bn)
        return ops.dense_train(x, w_, b_, act=act, w_grad_out=wg, b_grad_out=bg)
    return ops.dense(x, st[wn].detach(), st[bn].detach(), act=act)


class _LstmStack:
    """reference layers.lstm_model (layers.py:17-21): MultiRNNCell of LSTMCell(units, name='basic_lstm_cell') wrapped in
    DropoutWrapper(keep_prob=1.0) = identity.  Variables multi_rnn_cell/cell_<i>/basic_lstm_cell/{kernel,bias}."""

    def __init__(self, units):
        self.units = [int(u) for u in units]

    def zero_state(self, batch, device):
        return [(torch.zeros((batch, u), dtype=torch.float32, device=device), torch.zeros((batch, u), dtype=torch.float32, device=device))
                for u in self.units]

    def declare(self, input_size):
        st = default_store()
        names, prev = [], int(input_size)
        for i, u in enumerate(self.units):
            with st.variable_scope('multi_rnn_cell'), st.variable_scope('cell_%d' % i), st.variable_scope('basic_lstm_cell'):
                names.append((st.get_variable('kernel', (prev + u, 4 * u), 'glorot2d'), st.get_variable('bias', (4 * u,), 'zeros')))
            prev = u
        return names

    def sequence(self, x_seq, scope_prefix=None):
        """Differentiable run over a whole sequence x_seq [T,B,In] with zero initial state -> top layer outputs [T,B,U]
        (tf.nn.dynamic_rnn, or the unrolled cell calls of vae_decoder); gradients go to the flat bucket."""
        st = default_store()
        names = self.declare(x_seq.shape[-1])
        if is_sym(x_seq):
            return Sym(x_seq.shape[0], x_seq.shape[1], self.units[-1])
        for kn, bn in names:
            (k_, kg), (b_, bg) = st.param(kn), st.param(bn)
            x_seq = ops.lstm_layer(x_seq, k_, b_, w_grad_out=kg, b_grad_out=bg)
        return x_seq

    def __call__(self, x, state):
        st = default_store()
        names = self.declare(x.shape[-1])
        new_state = []
        for (kn, bn), (c, h) in zip(names, state):
            h2, c2 = ops.lstm_cell(x, h, c, st[kn].detach(), st[bn].detach())
            new_state.append((c2, h2))
            x = h2
        return x, new_state


def lstm_model(layers_):
    return _LstmStack(layers_)


def to_coord(input_, input_size, output_size, stddev=0.02, bias_start=0.0, trainable=False):
    """reference layers.to_coord (layers.py:24-28): tanh(x W + b), variables fully_connected/{W,b}."""
    st = default_store()
    with st.variable_scope('fully_connected'):
        wn = st.get_variable('W', (int(input_size), int(output_size)), 'normal002')
        bn = st.get_variable('b', (int(output_size),), 'zeros')
    if is_sym(input_):
        return Sym(input_.shape[0], output_size)
    if trainable:
        (w_, wg), (b_, bg) = st.param(wn), st.param(bn)
        return ops.dense_train(input_, w_, b_, act=ops.ACT_TANH, w_grad_out=wg, b_grad_out=bg)
    return ops.dense(input_, st[wn].detach(), st[bn].detach(), act=ops.ACT_TANH)
