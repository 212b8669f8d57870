"""A lazy graph-mode stand-in for the slice of the TensorFlow-1.12 Python API that the reference's stage-1 graph code calls.

TEST INFRASTRUCTURE, used only by ``make_networks_golden.py`` in the build container (it needs /root/reference).

TensorFlow 1.12 cannot be installed here (no network, no cp310 wheel), so the reference's own graph-building files

    models/networks/__init__.py   models/networks/layers.py   models/networks/vgg.py
    utils/model.py                models/base_model.py        models/detector_translator_model.py

are executed UNMODIFIED against this module registered as ``sys.modules['tensorflow']``.  What that pins is everything
the reference's Python decides: layer order, scope / variable names and their creation order, filter schedules, skip
indices, concat order, the tf.pad + padding='same' composition, which tensors feed which loss, the D / G variable split,
the optimiser wiring (learning-rate schedule arguments, betas, global_step, UPDATE_OPS control dependency) and the
two-``sess.run`` structure of ``train_step`` with a fresh batch per run.

What it does NOT pin: the arithmetic inside each op.  Every ``tf.*`` op below is an executable restatement of the
TF-1.12 semantics listed in SURVEY.md Appendix C ([TF-sem] rules: SAME padding split, legacy bilinear resize, fused
batch norm with biased / Bessel-corrected variance, linspace, softmax by reciprocal, ApplyAdam) on torch-CPU fp32.

Mechanics: every op returns a ``Tensor`` node (function + inputs) that is evaluated once at construction on zero
inputs (the "probe" run, which provides the static shapes the reference's Python control flow reads through
``x.shape.as_list()``) and re-evaluated, memoised per run, by ``Session.run``.  Variables are torch leaves, so
``Optimizer.minimize`` gets its gradients from torch autograd over the recorded forward of the same run.  Variable
scopes follow TF: ``variable_scope(None, default_name=..)`` uniquifies against the scopes opened so far inside the
enclosing scope, the counters of sub-scopes are reset when a scope closes (which is why ``tf.layers.conv2d`` inside a
re-entered ``reuse=tf.AUTO_REUSE`` scope finds its ``conv2d/kernel`` again), ``get_variable`` on an existing name
without reuse raises.
"""
import contextlib
import math
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

AUTO_REUSE = 'AUTO_REUSE'
_G = None


class _Graph:
    def __init__(self, seed):
        self.run_id = 0
        self.vars = OrderedDict()                  # full name (no ':0') -> Variable, creation order = tf.global_variables()
        self.update_ops = []
        self.scope, self.reuse = '', False
        self.scope_counts = {}
        self.control = []
        self.nodes = []
        self.rng = np.random.RandomState(seed)
        self.log = []
        self.grad_records = []                     # (optimizer index, run id, OrderedDict name -> grad tensor)
        self.optimizers = []


def reset(seed=1234):
    global _G
    _G = _Graph(seed)
    return _G


def graph():
    return _G


class _Shape(tuple):
    def as_list(self):
        return list(self)


def _const(v):
    if isinstance(v, Tensor):
        return v
    if isinstance(v, (list, tuple)) and v and all(isinstance(e, Tensor) for e in v):
        return stack(list(v), axis=0)
    if isinstance(v, (int, float)) and not isinstance(v, bool):
        return Tensor('const', lambda: v, [])      # python scalars stay weakly typed: fp32 arithmetic like TF's converted constants
    t = torch.as_tensor(np.asarray(v))
    if t.dtype == torch.float64:
        t = t.float()
    return Tensor('const', lambda: t, [])


class Tensor:
    def __init__(self, op, fn, inputs, stateful=False):
        g = _G
        self.op, self.fn, self.stateful = op, fn, stateful
        self.inputs = [_const(i) for i in inputs]
        self.control = list(g.control)
        self.scope = g.scope
        self._val, self._run = None, None
        self.index = len(g.nodes)
        g.nodes.append(self)
        if not stateful:
            self._eval(0)

    def _eval(self, run):
        if self._run == run:
            return self._val
        if run == 0:
            if self.stateful:
                return None
            with torch.no_grad():
                val = self.fn(*[i._eval(0) for i in self.inputs])
        else:
            for c in self.control:
                c._eval(run)
            val = self.fn(*[i._eval(run) for i in self.inputs])
        self._val, self._run = val, run
        return val

    @property
    def shape(self):
        v = self._val
        return _Shape(tuple(v.shape)) if torch.is_tensor(v) else _Shape(())

    def get_shape(self):
        return self.shape

    def __getitem__(self, key):
        return Tensor('getitem', lambda a: a[key], [self])

    def __add__(self, o): return Tensor('add', lambda a, b: a + b, [self, o])
    def __radd__(self, o): return Tensor('add', lambda a, b: a + b, [o, self])
    def __sub__(self, o): return Tensor('sub', lambda a, b: a - b, [self, o])
    def __rsub__(self, o): return Tensor('sub', lambda a, b: a - b, [o, self])
    def __mul__(self, o): return Tensor('mul', lambda a, b: a * b, [self, o])
    def __rmul__(self, o): return Tensor('mul', lambda a, b: a * b, [o, self])
    def __truediv__(self, o): return Tensor('div', lambda a, b: a / b, [self, o])
    def __rtruediv__(self, o): return Tensor('div', lambda a, b: a / b, [o, self])
    def __neg__(self): return Tensor('neg', lambda a: -a, [self])
    __array_priority__ = 1000        # ndarray * Tensor -> Tensor.__rmul__


class Variable(Tensor):
    def __init__(self, name, value, trainable):
        self.tensor = torch.as_tensor(value).clone()
        if trainable:
            self.tensor.requires_grad_(True)
        self.trainable = trainable
        self.var_name = name
        Tensor.__init__(self, 'variable', lambda: self.tensor, [])
        self.control = []

    @property
    def name(self):
        return self.var_name + ':0'

    @property
    def op_name(self):
        return self.var_name


# ----------------------------------------------------------------------------------------------- scopes / variables
def _unique_scope(prefix):
    g = _G
    base = g.scope + '/' + prefix if g.scope else prefix
    if g.scope_counts.get(base, 0) == 0:
        return prefix
    idx = 1
    while g.scope_counts.get('%s_%d' % (base, idx), 0) > 0:
        idx += 1
    return '%s_%d' % (prefix, idx)


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None):
    g = _G
    if name_or_scope is None:
        if reuse:
            raise ValueError('reuse=True cannot be used without a name_or_scope')
        name_or_scope = _unique_scope(default_name)
    name_or_scope = str(name_or_scope)
    old_scope, old_reuse = g.scope, g.reuse
    new = old_scope + '/' + name_or_scope if old_scope else name_or_scope
    g.scope_counts[new] = g.scope_counts.get(new, 0) + 1
    g.scope = new
    g.reuse = reuse if reuse else old_reuse          # reuse is inherited by sub-scopes
    try:
        yield new
    finally:
        for k in list(g.scope_counts):
            if k.startswith(new + '/'):
                g.scope_counts[k] = 0
        g.scope, g.reuse = old_scope, old_reuse


def _new_variable(full, value, trainable):
    g = _G
    v = Variable(full, value, trainable)
    g.vars[full] = v
    return v


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    g = _G
    full = g.scope + '/' + name if g.scope else name
    if full in g.vars:
        if not g.reuse:
            raise ValueError('Variable %s already exists, disallowed. Did you mean to set reuse=True or reuse=tf.AUTO_REUSE in VarScope?' % full)
        v = g.vars[full]
        if shape is not None and tuple(int(s) for s in shape) != tuple(v.tensor.shape):
            raise ValueError('Trying to share variable %s, but specified shape %s and found shape %s' % (full, shape, tuple(v.tensor.shape)))
        return v
    if g.reuse is True:
        raise ValueError('Variable %s does not exist, or was not created with tf.get_variable()' % full)
    shape = tuple(int(s) for s in shape)
    return _new_variable(full, initializer(shape), trainable)


def _tf_Variable(initial_value, trainable=True, name=None):
    g = _G
    base = name or 'Variable'
    full, idx = base, 0
    while full in g.vars:
        idx += 1
        full = '%s_%d' % (base, idx)
    val = np.asarray(initial_value)
    if val.dtype == np.int64:
        val = val.astype(np.int32)                   # tf.Variable(0) is int32
    if val.dtype == np.float64:
        val = val.astype(np.float32)
    return _new_variable(full, val, trainable and val.dtype == np.float32)


def _zeros(shape):
    return np.zeros(shape, np.float32)


def _ones(shape):
    return np.ones(shape, np.float32)


def _xavier_initializer():
    def init(shape):                                 # [TF-sem 6] uniform +-sqrt(6/(fan_in+fan_out)), fan = receptive field * channels
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        lim = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
        return _G.rng.uniform(-lim, lim, size=shape).astype(np.float32)
    return init


# ----------------------------------------------------------------------------------------------- op arithmetic ([TF-sem])
def _same_pad(n, k, s):                              # [TF-sem 1]
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _conv_same(x, w, stride):
    pt, pb = _same_pad(x.shape[1], w.shape[0], stride)
    pl, pr = _same_pad(x.shape[2], w.shape[1], stride)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xp, w.permute(3, 2, 0, 1), None, stride).permute(0, 2, 3, 1)


def _resize_bilinear_legacy(x, out_h, out_w):        # [TF-sem 2] align_corners=False, no half-pixel centres
    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        src = np.arange(n_out, dtype=np.float32) * scale
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        return torch.from_numpy(lo), torch.from_numpy(hi), torch.from_numpy((src - lo).astype(np.float32))
    ylo, yhi, ty = axis(x.shape[1], out_h)
    xlo, xhi, tx = axis(x.shape[2], out_w)
    tx, ty = tx.view(1, 1, -1, 1), ty.view(1, -1, 1, 1)
    top_l, top_r = x[:, ylo][:, :, xlo], x[:, ylo][:, :, xhi]
    bot_l, bot_r = x[:, yhi][:, :, xlo], x[:, yhi][:, :, xhi]
    top = top_l + (top_r - top_l) * tx
    bot = bot_l + (bot_r - bot_l) * tx
    return top + (bot - top) * ty


def _linspace(a, b, n):                              # [TF-sem 4]
    a32, b32 = np.float32(a), np.float32(b)
    step = np.float32((b32 - a32) / np.float32(n - 1))
    return torch.from_numpy((a32 + step * np.arange(n, dtype=np.float32)).astype(np.float32))


def _softmax(x, axis):                               # [TF-sem 5]
    e = torch.exp(x - x.max(dim=axis, keepdim=True).values)
    return e * (1.0 / e.sum(dim=axis, keepdim=True))


# ----------------------------------------------------------------------------------------------- tf.* functions
def pad(x, paddings):
    flat = []
    for lo, hi in reversed(list(paddings)):
        flat += [int(lo), int(hi)]
    return Tensor('pad', lambda a: F.pad(a, tuple(flat)), [x])


def layers_conv2d(inputs, filters, kernel_size, strides=1, padding='valid', use_bias=True, kernel_initializer=None, name=None):
    assert padding == 'same'
    cin = inputs.shape[-1]
    with variable_scope(name, default_name='conv2d'):
        kernel = get_variable('kernel', [kernel_size, kernel_size, cin, int(filters)], initializer=kernel_initializer or _xavier_initializer())
        bias = get_variable('bias', [int(filters)], initializer=_zeros) if use_bias else None
    y = Tensor('conv2d', lambda a, w: _conv_same(a, w, int(strides)), [inputs, kernel])
    if bias is not None:
        y = Tensor('bias_add', lambda a, b: a + b, [y, bias])
    return y


def nn_conv2d(x, filt, strides, padding='SAME', name=None):
    assert padding == 'SAME' and list(strides) == [1, 1, 1, 1]
    return Tensor('conv2d', lambda a, w: _conv_same(a, w, 1), [x, filt])


def nn_max_pool(x, ksize, strides, padding='SAME', name=None):
    assert list(ksize) == [1, 2, 2, 1] and list(strides) == [1, 2, 2, 1] and padding == 'SAME'
    # SAME on odd sizes pads at the bottom / right with -inf = ceil_mode [TF-sem 8]
    return Tensor('max_pool', lambda a: F.max_pool2d(a.permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1), [x])


def contrib_batch_norm(x, decay=0.999, center=False, scale=False, epsilon=0.001, is_training=True, scope=None, **unused):
    """[TF-sem 3] fused batch norm: batch mean / biased variance for the output, Bessel-corrected variance for the moving average,
    moving -= (moving - batch) * (1 - decay); variables beta, gamma, moving_mean, moving_variance in this order."""
    assert center and scale
    g = _G
    c = x.shape[-1]
    with variable_scope(scope, default_name='BatchNorm'):
        beta = get_variable('beta', [c], initializer=_zeros)
        gamma = get_variable('gamma', [c], initializer=_ones)
        mm = get_variable('moving_mean', [c], initializer=_zeros, trainable=False)
        mv = get_variable('moving_variance', [c], initializer=_ones, trainable=False)
    eps = float(epsilon)
    if not is_training:
        return Tensor('batch_norm_infer', lambda a, ga, be, m, v: (a - m) * torch.rsqrt(v + eps) * ga + be, [x, gamma, beta, mm, mv])
    mean = Tensor('bn_mean', lambda a: a.mean(dim=(0, 1, 2)), [x])
    var = Tensor('bn_var', lambda a, m: ((a - m) ** 2).mean(dim=(0, 1, 2)), [x, mean])
    y = Tensor('batch_norm', lambda a, m, v, ga, be: (a - m) * torch.rsqrt(v + eps) * ga + be, [x, mean, var, gamma, beta])
    count = int(np.prod(x.shape[:-1]))
    one_minus = float(np.float32(1.0) - np.float32(decay))

    def upd_mean(m):
        with torch.no_grad():
            mm.tensor -= (mm.tensor - m.detach()) * one_minus

    def upd_var(v):
        with torch.no_grad():
            mv.tensor -= (mv.tensor - v.detach() * (float(count) / float(max(count - 1, 1)))) * one_minus
    um = Tensor('assign_moving_avg', upd_mean, [mean], stateful=True)
    uv = Tensor('assign_moving_avg', upd_var, [var], stateful=True)
    um.target, uv.target = mm, mv
    g.update_ops += [um, uv]
    return y


def concat(values=None, axis=None, name=None):
    if isinstance(values, int):                     # tf.concat(axis=3, values=[...]) both spellings exist in the reference
        values, axis = axis, values
    values = list(values)
    return Tensor('concat', lambda *a: torch.cat(a, dim=axis), values)


def split(value=None, num_or_size_splits=None, axis=0, name=None):
    n = int(num_or_size_splits)
    size = value.shape[axis] // n
    assert size * n == value.shape[axis]
    return [Tensor('split', (lambda a, i=i: a.narrow(axis, i * size, size)), [value]) for i in range(n)]


def stack(values, axis=0, name=None):
    return Tensor('stack', lambda *a: torch.stack([torch.as_tensor(t) for t in a], dim=axis), list(values))


def _reduce(op, f):
    def fn(x, axis=None, keepdims=False, name=None):
        x = _const(x)
        if axis is None:
            return Tensor(op, lambda a: f(a), [x])
        return Tensor(op, lambda a: f(a, axis, keepdims), [x])
    return fn


reduce_mean = _reduce('reduce_mean', lambda a, axis=None, k=False: a.mean() if axis is None else a.mean(dim=axis, keepdim=k))
reduce_sum = _reduce('reduce_sum', lambda a, axis=None, k=False: a.sum() if axis is None else a.sum(dim=axis, keepdim=k))
reduce_max = _reduce('reduce_max', lambda a, axis=None, k=False: a.max() if axis is None else a.max(dim=axis, keepdim=k).values)


def sigmoid_xent(labels=None, logits=None, name=None):   # [TF-sem 8]
    return Tensor('sigmoid_xent', lambda z, x: torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x))), [labels, logits])


def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
    assert not staircase
    _G.lr_args = dict(learning_rate=learning_rate, global_step=global_step.var_name, decay_steps=decay_steps, decay_rate=decay_rate)

    def fn(gs):                                      # [TF-sem 7] fp32
        p = np.float32(int(gs)) / np.float32(decay_steps)
        return torch.tensor(np.float32(np.float32(learning_rate) * np.power(np.float32(decay_rate), p, dtype=np.float32)))
    return Tensor('exponential_decay', fn, [global_step])


class AdamOptimizer:
    """[TF-sem 7] ApplyAdam in fp32: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2); var -= m*alpha/(sqrt(v)+eps);
    the beta powers are variables multiplied by beta after the update.  Slot / power variable names are TF's ([TF-sem naming])."""

    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, np.float32(beta1), np.float32(beta2), np.float32(epsilon)
        self.index = len(_G.optimizers)
        _G.optimizers.append(self)

    def minimize(self, loss, var_list=None, global_step=None):
        g = _G
        self.var_list = list(var_list)
        self.loss, self.global_step = loss, global_step
        self.m, self.v = {}, {}
        for v in self.var_list:                     # _create_slots: beta powers first, then the m / v slot of each variable
            pass
        self.b1p = _tf_Variable(np.float32(self.b1), trainable=False, name='beta1_power')
        self.b2p = _tf_Variable(np.float32(self.b2), trainable=False, name='beta2_power')
        for v in self.var_list:
            self.m[v.var_name] = _new_variable(v.var_name + '/Adam', np.zeros(tuple(v.tensor.shape), np.float32), False)
            self.v[v.var_name] = _new_variable(v.var_name + '/Adam_1', np.zeros(tuple(v.tensor.shape), np.float32), False)
        lr = _const(self.lr)

        def apply(loss_val, lr_val):
            params = [v.tensor for v in self.var_list]
            grads = torch.autograd.grad(loss_val, params, allow_unused=True)
            rec = OrderedDict()
            b1p, b2p = np.float32(self.b1p.tensor.item()), np.float32(self.b2p.tensor.item())
            alpha = float(np.float32(np.float32(float(lr_val)) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p)))
            with torch.no_grad():
                for v, gr in zip(self.var_list, grads):
                    gr = torch.zeros_like(v.tensor) if gr is None else gr
                    rec[v.var_name] = gr
                    m, s = self.m[v.var_name].tensor, self.v[v.var_name].tensor
                    m += (gr - m) * float(np.float32(1) - self.b1)
                    s += (gr * gr - s) * float(np.float32(1) - self.b2)
                    v.tensor -= (m * alpha) / (torch.sqrt(s) + float(self.eps))
                self.b1p.tensor.fill_(float(np.float32(b1p * self.b1)))
                self.b2p.tensor.fill_(float(np.float32(b2p * self.b2)))
                if self.global_step is not None:
                    self.global_step.tensor += 1
            g.grad_records.append((self.index, g.run_id, rec))
            return None
        return Tensor('adam_minimize', apply, [loss, lr], stateful=True)


@contextlib.contextmanager
def control_dependencies(ops):
    g = _G
    old = g.control
    g.control = old + [o for o in ops if isinstance(o, Tensor)]
    try:
        yield
    finally:
        g.control = old


class _Summary:
    pass


class Session:
    def __init__(self, config=None):
        self.history = []
        self.graph = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        g = _G
        g.run_id += 1
        run = g.run_id

        def ev(f):
            if isinstance(f, (list, tuple)):
                return [ev(e) for e in f]
            if isinstance(f, dict):
                return {k: ev(e) for k, e in f.items()}
            if isinstance(f, Tensor):
                v = f._eval(run)
                return v.detach().numpy().copy() if torch.is_tensor(v) else v
            return None
        with torch.enable_grad():
            out = ev(fetches)
        self.history.append((run, out))
        return out


class InputSource:
    """Stands in for ``base_iterator.get_next()`` (reference train.py:46-50): ONE node that yields the next batch the first time
    it is evaluated in a run, so every ``sess.run`` that touches the inputs consumes a new batch."""

    def __init__(self, batches):
        self.batches, self.cursor, self.served = list(batches), 0, []
        probe = {k: torch.zeros_like(torch.as_tensor(v)) for k, v in self.batches[0].items()}

        def nxt():
            if _G.run_id == 0:
                return probe
            b = self.batches[self.cursor]
            self.served.append((_G.run_id, self.cursor))
            self.cursor += 1
            return {k: torch.as_tensor(v) for k, v in b.items()}
        self.node = Tensor('iterator_get_next', nxt, [])

    def get_next(self):
        return {k: Tensor('iterator_output', (lambda d, k=k: d[k]), [self.node]) for k in self.batches[0]}


def install():
    """Build the module object and register it as ``tensorflow`` (and the sub-modules the reference reaches by attribute)."""
    tf = types.ModuleType('tensorflow')
    tf.AUTO_REUSE = AUTO_REUSE
    tf.float32, tf.int32, tf.bool, tf.string = 'float32', 'int32', 'bool', 'string'
    tf.variable_scope, tf.get_variable, tf.Variable = variable_scope, get_variable, _tf_Variable
    tf.pad, tf.concat, tf.split, tf.stack = pad, concat, split, stack
    tf.reduce_mean, tf.reduce_sum, tf.reduce_max = reduce_mean, reduce_sum, reduce_max
    tf.expand_dims = lambda x, axis=None, name=None: Tensor('expand_dims', lambda a: a.unsqueeze(axis), [x])
    tf.reshape = lambda x, shape, name=None: Tensor('reshape', lambda a: a.reshape(tuple(int(s) for s in shape)), [x])
    tf.square = lambda x: Tensor('square', lambda a: a * a, [x])
    tf.exp = lambda x: Tensor('exp', torch.exp, [x])
    tf.abs = lambda x: Tensor('abs', torch.abs, [x])
    tf.transpose = lambda x, perm=None: Tensor('transpose', lambda a: a.permute(*perm), [x])
    tf.linspace = lambda a, b, n: Tensor('linspace', lambda: _linspace(a, b, n), [])
    tf.to_float = lambda x: Tensor('to_float', lambda a: a.float(), [x])
    tf.ones_like = lambda x: Tensor('ones_like', torch.ones_like, [x])
    tf.zeros_like = lambda x: Tensor('zeros_like', torch.zeros_like, [x])
    tf.clip_by_value = lambda x, lo, hi: Tensor('clip_by_value', lambda a: torch.clamp(a, lo, hi), [x])
    tf.constant = lambda value, dtype=None, shape=None, name=None: _const(np.asarray(value))
    tf.trainable_variables = lambda: [v for v in _G.vars.values() if v.trainable]
    tf.global_variables = lambda: list(_G.vars.values())
    tf.GraphKeys = types.SimpleNamespace(UPDATE_OPS='update_ops')
    tf.get_collection = lambda key: list(_G.update_ops) if key == 'update_ops' else []
    tf.control_dependencies = control_dependencies
    tf.Session = Session
    tf.ConfigProto = lambda **kw: types.SimpleNamespace(gpu_options=types.SimpleNamespace(allow_growth=False))
    tf.nn = types.SimpleNamespace(
        relu=lambda x, name=None: Tensor('relu', torch.relu, [x]),
        leaky_relu=lambda x, alpha=0.2, name=None: Tensor('leaky_relu', lambda a: torch.maximum(a, a * alpha), [x]),   # [TF-sem 8]
        sigmoid=lambda x, name=None: Tensor('sigmoid', torch.sigmoid, [x]),
        softmax=lambda x, axis=-1, name=None: Tensor('softmax', lambda a: _softmax(a, axis), [x]),
        conv2d=nn_conv2d, max_pool=nn_max_pool,
        bias_add=lambda x, b, name=None: Tensor('bias_add', lambda a, c: a + c, [x, b]),
        sigmoid_cross_entropy_with_logits=sigmoid_xent)
    tf.layers = types.SimpleNamespace(conv2d=layers_conv2d)
    tf.contrib = types.SimpleNamespace(layers=types.SimpleNamespace(batch_norm=contrib_batch_norm, xavier_initializer=_xavier_initializer))
    tf.image = types.SimpleNamespace(
        resize_images=lambda x, size, **kw: Tensor('resize_images', lambda a: _resize_bilinear_legacy(a, int(size[0]), int(size[1])), [x]))
    tf.train = types.SimpleNamespace(exponential_decay=exponential_decay, AdamOptimizer=AdamOptimizer,
                                     Saver=lambda *a, **k: None, NewCheckpointReader=None)
    tf.summary = types.SimpleNamespace(image=lambda *a, **k: _Summary(), scalar=lambda *a, **k: _Summary(), merge=lambda *a, **k: _Summary(),
                                       FileWriter=lambda *a, **k: None)
    tf.logging = types.SimpleNamespace(info=lambda msg, *a: _G.log.append(msg % a if a else msg), INFO=20, set_verbosity=lambda *a: None)
    sys.modules['tensorflow'] = tf
    return tf
