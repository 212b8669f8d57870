#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Run in the build container only (needs /root/reference):

    python -B tests/golden/make_golden.py

Two kinds of vectors:

1. ``model_utils_ref.npz`` -- outputs of the REFERENCE's own ``utils/model.py`` (get_coord,
   get_gaussian_maps; /root/reference/utils/model.py:49-70) executed unmodified.  TensorFlow 1.12
   is not installable here, so the file is executed against a ~30-line numpy-backed stand-in for
   the dozen ``tf.*`` calls it makes.  That pins the reference's formula, broadcasting and axis
   conventions (x from axis 1 / y from axis 2, (x,y) order, BKHW->BHWK transpose, inv_std**2 placement);
   the float arithmetic is numpy's, not Eigen's, so TF-1.12 numerics stay UNPINNED.
   Also ``utils/training.py:get_n_iterations`` (pure python) is run directly.

2. ``tiny_e2e_oracle.npz`` -- one end-to-end tiny detector_translator train step from the CPU
   restatement (oracle/restatement.py), H=32, K=3, B=2, VGG width/8: losses, key-points, frame and a
   digest of every gradient.  These pin the *restatement* against accidental edits (they are
   self-generated, not reference outputs).

Nothing from /root/reference is copied: only inputs (seeds) and output arrays are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True


def _numpy_tf_standin():
    """Just enough of the tf.* surface for utils/model.py:49-70, float32 throughout."""
    tf = types.ModuleType('tensorflow')
    f32 = np.float32

    def linspace(a, b, n):  # [TF-sem 4]
        step = f32((f32(b) - f32(a)) / f32(n - 1))
        return (f32(a) + step * np.arange(n, dtype=f32)).astype(f32)

    def softmax(x, axis=-1):  # [TF-sem 5]
        e = np.exp(x - np.max(x, axis=axis, keepdims=True)).astype(f32)
        return (e * (f32(1.0) / np.sum(e, axis=axis, keepdims=True, dtype=f32))).astype(f32)

    tf.linspace = linspace
    tf.to_float = lambda x: np.asarray(x, dtype=f32)
    tf.expand_dims = lambda x, axis: np.expand_dims(x, axis)
    tf.reshape = lambda x, shape: np.reshape(x, shape)
    tf.square = lambda x: np.square(x).astype(f32)
    tf.exp = lambda x: np.exp(x).astype(f32)
    tf.transpose = lambda x, perm: np.transpose(x, perm)
    tf.reduce_mean = lambda x, axis=None: np.mean(x, axis=axis, dtype=f32).astype(f32)
    tf.reduce_sum = lambda x, axis=None: np.sum(x, axis=axis, dtype=f32).astype(f32)
    tf.reduce_max = lambda x, axis=None: np.max(x, axis=axis)
    tf.stack = lambda xs, axis=0: np.stack(xs, axis=axis)
    tf.nn = types.SimpleNamespace(softmax=softmax)
    return tf


def _load(path, name, extra_modules=None):
    saved = {}
    for k, v in (extra_modules or {}).items():
        saved[k] = sys.modules.get(k)
        sys.modules[k] = v
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def make_model_utils_ref():
    mu = _load(os.path.join(REF, 'utils', 'model.py'), '_ref_model_utils', {'tensorflow': _numpy_tf_standin()})
    tr = _load(os.path.join(REF, 'utils', 'training.py'), '_ref_training')
    out = {}
    # get_coord on seeded logits [2,128,128,15] and a ragged [1,24,40,5]; inputs are regenerated from the seed
    for tag, seed, shape, scale in (('a', 7, (2, 128, 128, 15), 3.0), ('b', 8, (1, 24, 40, 5), 10.0)):
        x = (np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32)
        gy, gy_prob = mu.get_coord(x, 2, shape[1])      # networks/__init__.py:69
        gx, gx_prob = mu.get_coord(x, 1, shape[2])      # networks/__init__.py:70
        out['coord_%s_seed' % tag] = np.int64(seed)
        out['coord_%s_shape' % tag] = np.array(shape, np.int64)
        out['coord_%s_scale' % tag] = np.float32(scale)
        out['coord_%s_mu' % tag] = np.stack([gx, gy], axis=2).astype(np.float32)   # :71 (x,y)
        out['coord_%s_yprob' % tag] = gy_prob.astype(np.float32)
        out['coord_%s_xprob' % tag] = gx_prob.astype(np.float32)
    # get_gaussian_maps at [32,32] (K=15,B=2), [128,128] (K=4,B=1), and a non-square [8,12]
    for tag, seed, b, k, hw in (('lo', 11, 2, 15, (32, 32)), ('hi', 12, 1, 4, (128, 128)), ('rect', 13, 2, 3, (8, 12))):
        pts = np.random.RandomState(seed).uniform(-1, 1, size=(b, k, 2)).astype(np.float32)
        out['gauss_%s_mu' % tag] = pts
        out['gauss_%s_hw' % tag] = np.array(hw, np.int64)
        out['gauss_%s_map' % tag] = mu.get_gaussian_maps(pts, list(hw)).astype(np.float32)
    out['n_iterations_cases'] = np.array([[89, 16, tr.get_n_iterations(89, 16)],
                                          [1171, 16, tr.get_n_iterations(1171, 16)],
                                          [32, 16, tr.get_n_iterations(32, 16)]], np.int64)
    np.savez_compressed(os.path.join(HERE, 'model_utils_ref.npz'), **out)
    print('wrote model_utils_ref.npz', {k: getattr(v, 'shape', None) for k, v in out.items()})


def make_tiny_e2e():
    sys.path.insert(0, REPO)
    import torch
    from oracle import restatement as R
    torch.manual_seed(0)
    torch.set_num_threads(1)        # fixed reduction order
    res, k, b = 32, 3, 2
    variables = R.init_variables(k, res=res, seed=1234)
    vgg = R.synthetic_vgg(seed=19, width_div=8)
    st = R.TrainState(variables, vgg)
    im, fut = R.synthetic_pair(b, res=res)
    r = R.train_step(st, im, fut)
    out = dict(res=np.int64(res), n_pts=np.int64(k), batch=np.int64(b),
               loss_D=np.float32(r['loss_D']), loss_G=np.float32(r['loss_G']),
               loss_G_recon=np.float32(r['loss_G_recon']), loss_G_adv=np.float32(r['loss_G_adv']),
               lr=np.float32(r['lr']),
               final_output=r['final_output'].numpy(), current_points=r['current_points'].numpy(),
               future_points=r['future_points'].numpy())
    names = sorted(list(r['grads_D']) + list(r['grads_G']))
    g = {**r['grads_D'], **r['grads_G']}
    out['grad_names'] = np.array(names)
    out['grad_l2'] = np.array([float(g[n].double().norm()) for n in names], np.float64)
    out['grad_sum'] = np.array([float(g[n].double().sum()) for n in names], np.float64)
    # parameters after the step: digest
    out['param_l2_after'] = np.array([float(st.params[n].double().norm()) for n in names], np.float64)
    bn_names = sorted(n for n in st.params if 'moving_' in n)
    out['moving_names'] = np.array(bn_names)
    out['moving_l2_after'] = np.array([float(st.params[n].double().norm()) for n in bn_names], np.float64)
    np.savez_compressed(os.path.join(HERE, 'tiny_e2e_oracle.npz'), **out)
    print('wrote tiny_e2e_oracle.npz loss_D=%.6f loss_G=%.6f' % (r['loss_D'], r['loss_G']))


if __name__ == '__main__':
    make_model_utils_ref()
    make_tiny_e2e()
