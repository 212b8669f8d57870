"""Variable store: the stand-in for TF1's graph-level variable scopes of the reference.

The reference identifies every weight by its ``tf.variable_scope`` path (e.g.
``pose_encoder/encoder/conv_3/conv2d/kernel``) and trains two variable sets split by the substring ``img_discr``
(models/detector_translator_model.py:191-192).  Here the same names index views into two flat fp32 buckets in HBM
(generator / discriminator): parameters, gradients and the two Adam slots are each ONE contiguous buffer per bucket, so
the data-parallel exchange is a single RCCL all-reduce and the optimiser a single fused kernel launch per bucket.
"""
import contextlib
import math
import weakref
from collections import OrderedDict

import numpy as np
import torch


class Bucket:
    """One flat parameter bucket with matching flat gradient and Adam-slot buffers."""

    def __init__(self, name):
        self.name = name
        self.entries = OrderedDict()     # var name -> (offset, shape)
        self.size = 0
        self.params = self.grads = self.m = self.v = None

    def add(self, name, shape):
        n = int(np.prod(shape))
        self.entries[name] = (self.size, tuple(shape))
        self.size += (n + 3) // 4 * 4      # keep every view 16-B aligned for the vectorised kernels

    def allocate(self, device):
        z = lambda: torch.zeros(max(self.size, 4), dtype=torch.float32, device=device)
        self.params, self.grads, self.m, self.v = z(), z(), z(), z()

    def view(self, flat, name):
        off, shape = self.entries[name]
        return flat[off:off + int(np.prod(shape))].view(shape)


class VariableStore:
    """name -> tensor, created on first use under the current scope (tf.get_variable semantics with AUTO_REUSE)."""

    def __init__(self, device='cpu', seed=1234):
        self.device = torch.device(device)
        self.rng = np.random.RandomState(seed)       # RandomState(1234), creation order
        self.specs = OrderedDict()                   # name -> (shape, kind)
        self.init_values = OrderedDict()             # name -> np array (until materialised)
        self.vars = OrderedDict()                    # name -> torch tensor (views into buckets once materialised)
        self.grad_views = {}
        self.buckets = OrderedDict((k, Bucket(k)) for k in ('G', 'D'))
        self._scope = []
        self.materialised = False
        self.frozen = ()                             # name substrings whose variables are used without gradients
        self.layer_attrs = {}                        # filter variable name -> per-layer kernel attributes declared by layers.conv (e.g. f43_fwd)
        self.version = 0                             # bumped by touch(): anything derived from the parameters (captured graphs included) is stale

    @contextlib.contextmanager
    def freeze(self, *substrings):
        """Use the matching variables as constants (the discriminator inside the generator update)."""
        old, self.frozen = self.frozen, tuple(substrings)
        try:
            yield
        finally:
            self.frozen = old

    def param(self, name):
        """(tensor, gradient destination) for a variable, honouring freeze()."""
        v = self.vars[name]
        if any(s in name for s in self.frozen):
            return v.detach(), None
        return v, self.grad_views.get(name)

    # ---- scopes ------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def scoped(self, name):
        return '/'.join(self._scope + [name])

    # ---- creation ----------------------------------------------------------------------------------------------
    def get_variable(self, name, shape, kind):
        """kind: 'kernel' (xavier-uniform), 'zeros', 'ones'; non-trainable kinds: 'moving_zeros', 'moving_ones'."""
        full = self.scoped(name)
        if full in self.specs:
            assert self.specs[full][0] == tuple(shape), (full, self.specs[full][0], shape)
            return full
        assert not self.materialised, 'variable %s requested after materialise()' % full
        self.specs[full] = (tuple(shape), kind)
        if kind == 'kernel':                          # tf.contrib.layers.xavier_initializer (layers.py:8)
            val = self._xavier(shape)
        elif kind == 'kernel_head31':                 # fused crude(3) + mask(1) head: two reference variables
            kh, kw, ci, co = shape
            assert co == 4
            val = np.concatenate([self._xavier((kh, kw, ci, 3)), self._xavier((kh, kw, ci, 1))], axis=-1)
        elif kind == 'glorot2d':                      # tf.get_variable default / contrib xavier for [in, out] matrices
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            val = self.rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif kind == 'normal002':                     # tf.random_normal_initializer(stddev=0.02) (layers.py:26)
            val = (self.rng.randn(*shape) * 0.02).astype(np.float32)
        elif kind in ('ones', 'moving_ones'):
            val = np.ones(shape, np.float32)
        else:
            val = np.zeros(shape, np.float32)
        self.init_values[full] = val
        return full

    def _xavier(self, shape):
        kh, kw, ci, co = shape
        lim = math.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
        return self.rng.uniform(-lim, lim, size=shape).astype(np.float32)

    @staticmethod
    def is_trainable(kind):
        return not kind.startswith('moving')

    def materialise(self):
        """Allocate the buckets on the device and turn every variable into a view of them."""
        for name, (shape, kind) in self.specs.items():
            if self.is_trainable(kind):
                self.buckets['D' if 'discr' in name else 'G'].add(name, shape)   # reference :191-192 ('discr' in var.name: img_discr, seq_discr)
        for b in self.buckets.values():
            b.allocate(self.device)
        for name, (shape, kind) in self.specs.items():
            val = torch.from_numpy(self.init_values[name])
            if self.is_trainable(kind):
                b = self.buckets['D' if 'discr' in name else 'G']
                v = b.view(b.params, name)
                v.copy_(val)
                v.requires_grad_(True)
                self.vars[name] = v
                self.grad_views[name] = b.view(b.grads, name)
            else:
                self.vars[name] = val.to(self.device)
        self.init_values.clear()
        self.materialised = True
        if self.device.type == 'cuda':
            from . import ops                            # pre-transformed Winograd filters of every trainable 3x3 kernel
            # (only filters a Winograd kernel can ever take: >= 4 produced channels in at least one direction; the discriminator's single
            # 3x3 layer, D_logit 2048 -> 1, is not one -- so an update of the D bucket never invalidates anything here, see touch())
            named = [(n, v.detach()) for n, v in self.vars.items()
                     if n.endswith('/kernel') and v.dim() == 4 and v.shape[0] == 3 and min(v.shape[2], v.shape[3]) >= 4]
            self.filter_bank = ops.FilterBank(named, self.device, attrs=self.layer_attrs)
            self._bank_buckets = {'D' if 'discr' in n else 'G' for n, _ in named}        # same split as the buckets above
            weakref.finalize(self, ops.release_filters, self.filter_bank.keys())     # the bucket's addresses may be reused later

    def touch(self, bucket=None):
        """Call after writing parameters behind torch's back (fused Adam kernel, restore): derived filter forms are refreshed lazily.
        ``bucket``: the flat bucket that was written ('G' / 'D' / ...); an update of a bucket none of whose filters has a derived form
        (the discriminator's) leaves the forms valid."""
        self.version += 1
        self.drop_folded()                            # inference-folded filters (layers.conv_bn_relu) are functions of the parameters
        bank = getattr(self, 'filter_bank', None)
        if bank is not None:
            owners = getattr(self, '_bank_buckets', None)
            if bucket is None or owners is None or bucket in owners:
                bank.touch()

    def drop_folded(self):
        """Forget the inference-mode (conv filter x batch-norm scale) products: a parameter or a moving statistic changed."""
        folded = getattr(self, '_folded', None)
        if folded:
            from . import ops
            for _, _, keys in folded.values():
                ops.release_filters(keys)
            folded.clear()

    def folded_conv_bn(self, kname, bname, gamma, beta, mm, mv, eps):
        """(w', b') with relu(conv(x, w') + b') = relu(batch_norm_inference(conv(x, w) + b)); computed once per parameter state
        (kpx_bn_fold_conv_f32) and registered like a constant filter so that the Winograd kernels take it pre-transformed."""
        folded = self.__dict__.setdefault('_folded', {})
        ent = folded.get(kname)
        if ent is None:
            from . import ops
            from ._lib import lib, check
            w = self.vars[kname].detach()
            wf, bf = torch.empty_like(w), torch.empty(w.shape[3], dtype=torch.float32, device=w.device)
            b = self.vars[bname].detach() if bname is not None else None
            check(lib.kpx_bn_fold_conv_f32(w.data_ptr(), b.data_ptr() if b is not None else None, w.numel() // w.shape[3], int(w.shape[3]),
                                           self.vars[gamma].data_ptr(), self.vars[beta].data_ptr(), self.vars[mm].data_ptr(), self.vars[mv].data_ptr(),
                                           float(eps), wf.data_ptr(), bf.data_ptr(), ops._stream()), 'kpx_bn_fold_conv_f32')
            # (inference: the F(4x4,3x3) forward policy of training -- batch statistics over few pixels behind 256-deep sums, gradients through
            #  ten batch norms -- does not apply: the batch norm is a constant folded into this filter, and a layer's own F(4x4,3x3) error is
            #  2.5e-6 with the double-precision filter transform.  KPX_INFER_F43=0 keeps the training attribute.)
            import os
            f43 = True                                    # (inference: the training-accuracy policy does not apply to a folded batch norm)
            ent = folded[kname] = (wf, bf, ops.register_constant_filter(wf, kname, f43_fwd=f43))
        return ent[0], ent[1]

    # ---- access ------------------------------------------------------------------------------------------------
    def __getitem__(self, name):
        """READ access to a variable.  The tensor is a view of the flat bucket; derived forms of the filters (Winograd U, inference-folded
        filters) are cached per store version, so a parameter must not be written through this view behind the store's back: write with
        ``assign`` (or call ``touch`` after a bulk write such as the fused Adam kernel / ``load_numpy`` do)."""
        return self.vars[name]

    def assign(self, name, value):
        """Write one variable (tf.assign): copies ``value`` into the bucket view and invalidates every derived filter form."""
        v = self.vars[name]
        with torch.no_grad():
            v.copy_(torch.as_tensor(value, dtype=torch.float32).to(v.device).reshape(v.shape))
        self.touch()

    def grad(self, name):
        return self.grad_views.get(name)

    def load_numpy(self, arrays, strict=True):
        """Copy {TF name: array} into the store; the fused translator head 'translator/conv_N_0+1' is assembled from the
        reference's separate conv_N_0 (crude, 3 ch) and conv_N_1 (mask, 1 ch) variables."""
        arrays = dict(arrays)
        with torch.no_grad():
            for name, v in self.vars.items():
                if '_0+1/' in name:
                    a = arrays.get(name.replace('_0+1/', '_0/'))
                    b = arrays.get(name.replace('_0+1/', '_1/'))
                    if a is None or b is None:
                        if strict:
                            raise KeyError(name)
                        continue
                    src = np.concatenate([np.asarray(a), np.asarray(b)], axis=-1)
                elif name in arrays:
                    src = np.asarray(arrays[name])
                else:
                    if strict:
                        raise KeyError(name)
                    continue
                v.copy_(torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).to(v.device))
        self.touch()

    def export_numpy(self, include_slots=False, beta_powers=None):
        """{TF name: array} in the reference's checkpoint naming, splitting the fused head."""
        out = OrderedDict()

        def put(name, t, suffix=''):
            a = t.detach().cpu().numpy().copy()
            if '_0+1/' in name:
                out[name.replace('_0+1/', '_0/') + suffix] = a[..., :3].copy()
                out[name.replace('_0+1/', '_1/') + suffix] = a[..., 3:].copy()
            else:
                out[name + suffix] = a

        for name, v in self.vars.items():
            put(name, v)
        if include_slots:
            for b in self.buckets.values():
                for name in b.entries:
                    put(name, b.view(b.m, name), '/Adam')
                    put(name, b.view(b.v, name), '/Adam_1')
        if beta_powers:
            out.update(beta_powers)
        return out


_default_store = []


@contextlib.contextmanager
def as_default(store):
    """The analogue of TF1's default graph: layers.* create / look up variables in the innermost default store."""
    _default_store.append(store)
    try:
        yield store
    finally:
        _default_store.pop()


def default_store():
    if not _default_store:
        raise RuntimeError('no default VariableStore: wrap the network call in `with variables.as_default(store):`')
    return _default_store[-1]


class Sym:
    """Shape-only stand-in for a tensor, used for the build pass that declares variables (the analogue of TF1 graph
    construction: models/base_model.py build() creates variables without running anything)."""

    def __init__(self, *shape):
        self.shape = tuple(int(s) for s in shape)

    def __repr__(self):
        return 'Sym%s' % (self.shape,)


def is_sym(x):
    return isinstance(x, Sym)
