#!/usr/bin/env python3
"""Generate tests/golden/networks_ref.npz.  Run in the build container only (needs /root/reference):

    python -B tests/golden/make_networks_golden.py

Executes the REFERENCE's own stage-1 graph code, unmodified and imported from where it lies --

    /root/reference/models/networks/__init__.py  (encoder, image_encoder, pose_encoder, translator, img_discr)
    /root/reference/models/networks/layers.py    (conv, batch_norm)
    /root/reference/models/networks/vgg.py       (Vgg19.build)
    /root/reference/utils/model.py               (get_coord, get_gaussian_maps)
    /root/reference/models/base_model.py, models/detector_translator_model.py
        (DetectorTranslatorModel.__init__ / build / _define_forward_pass / _compute_loss / _compute_loss_D / _compute_loss_G /
         _compute_perceptual_loss / _define_summary / train_step / test_step)

-- against ``tf_standin`` (a lazy graph-mode stand-in for the TF-1.12 calls those files make; see its docstring for what
that does and does not pin) and stores what the reference's graph produced:

* the variable registry in ``tf.global_variables()`` order (names, shapes), the trainable subset, the D / G ``var_list`` split, the
  UPDATE_OPS targets in order, the learning-rate schedule arguments and the Adam hyper-parameters the reference passed;
* a digest of every initial value (xavier draws from RandomState(1234) in creation order);
* one forward evaluation at the initial weights (batch 0): key-points, heat-maps, crude / mask / final frames, the three
  discriminator logit maps, the five VGG19 feature maps, all loss terms and the learning rate;
* two executions of the reference's ``train_step`` (each = sess.run(D ops) then sess.run(G ops), a NEW batch per sess.run,
  reference train.py:46-50) and one of ``test_step``: fetched losses, per-variable gradient digests, parameter / moving-statistic /
  Adam-slot digests after each step, global_step, which batch every run consumed, and that test_step changed nothing.

The reference hard-codes 128x128 inputs (heat-map sizes [32,32] / [128,128] are literals, detector_translator_model.py:168-177; final_res
defaults to 128), so the case is 128x128, K=3, batch 2, with a width/8 synthetic ``vgg19.npy`` (the real file is not shipped) written to
a temporary directory in the dict format vgg.py:11 loads.  Nothing from /root/reference is copied: only seeds, names and output arrays
are stored.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))          # tests/gradproj.py

RES, K, B, N_BATCHES = 128, 3, 2, 6
VGG_SEED, VGG_WIDTH_DIV, INIT_SEED = 19, 8, 1234
VGG_LAYERS = [('conv1_1', 3, 64), ('conv1_2', 64, 64), ('conv2_1', 64, 128), ('conv2_2', 128, 128),
              ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256), ('conv3_4', 256, 256),
              ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512), ('conv4_4', 512, 512),
              ('conv5_1', 512, 512), ('conv5_2', 512, 512), ('conv5_3', 512, 512), ('conv5_4', 512, 512)]


def synthetic_batch(i):
    """SURVEY 8d synthetic Penn-shaped pair: uint8 ~ U{0..255} from RandomState(seed), x/255*2-1 (image_pair_dataloader.py:65-70)."""
    def one(seed):
        u = np.random.RandomState(seed).randint(0, 256, size=(B, RES, RES, 3)).astype(np.float32)
        return (u / np.float32(255.0) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    return {'image': one(100 + 2 * i), 'future_image': one(101 + 2 * i)}


def synthetic_vgg_file(path):
    """He-normal filters, zero biases, channel counts / VGG_WIDTH_DIV, in the layout vgg.py:11,57-61 reads."""
    rs = np.random.RandomState(VGG_SEED)
    d = {}
    for name, ci, co in VGG_LAYERS:
        ci = ci if ci == 3 else max(ci // VGG_WIDTH_DIV, 1)
        co = max(co // VGG_WIDTH_DIV, 1)
        d[name] = [(rs.randn(3, 3, ci, co) * np.sqrt(2.0 / (9 * ci))).astype(np.float32), np.zeros((co,), np.float32)]
    np.save(path, np.array(d, dtype=object), allow_pickle=True)


def digest(t):
    a = np.asarray(t.detach().numpy() if torch.is_tensor(t) else t, dtype=np.float64).ravel()
    return np.array([np.sqrt((a * a).sum()), a.sum(), np.abs(a).max() if a.size else 0.0], np.float64)


def projections(named):
    """<gradient, fixed random direction> per variable (tests/gradproj.py): changes under a sign flip / transposition / permutation."""
    from gradproj import projection
    return np.array([projection(str(n), t.detach().numpy() if torch.is_tensor(t) else t) for n, t in named], np.float64)


def sample(a, n=512):
    a = np.asarray(a).ravel()
    idx = np.linspace(0, a.size - 1, min(n, a.size)).astype(np.int64)
    return a[idx].astype(np.float32)


def main():
    import tf_standin as S
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = S.reset(seed=INIT_SEED)
    tf = S.install()
    # numpy==1.15.4 (reference requirements.txt:9) loads pickled object arrays by default; vgg.py:11 relies on that
    np_load = np.load
    np.load = lambda *a, **k: np_load(*a, **{**k, 'allow_pickle': True})
    sys.path.insert(0, REF)
    try:
        from models.detector_translator_model import DetectorTranslatorModel
        tmp = tempfile.mkdtemp(prefix='kpx_netgold_')
        vgg_path = os.path.join(tmp, 'vgg19.npy')
        synthetic_vgg_file(vgg_path)
        config = {'paths': {'data_dir': tmp, 'vggnet': vgg_path, 'log_dir': tmp},                      # keys of configs/penn.yaml
                  'training': {'n_steps': 2, 'summary_interval': 500, 'test_interval': 500, 'checkpoint_interval': 20000, 'log_interval': 1,
                               'batch_size': B, 'lr': {'start_val': 0.0001, 'step': 20000, 'decay': 0.95}},
                  'model': {'n_pts': K, 'n_action': 9, 'cell_info': [1024, 1024], 'vae_dim': 64}}
        out = {}
        with tf.Session() as sess:
            global_step = tf.Variable(0, trainable=False, name='global_step')                       # reference train.py:30
            source = S.InputSource([synthetic_batch(i) for i in range(N_BATCHES)])
            inputs = source.get_next()                                                                # train.py:50
            model = DetectorTranslatorModel(config, global_step, is_training=True)                    # train.py:119
            model.build(inputs)                                                                       # train.py:55

            # ---------------------------------------------------------------- A. registry and wiring
            names = list(g.vars)
            out['var_names'] = np.array(names)
            out['var_shapes'] = np.array([','.join(str(int(s)) for s in g.vars[n].tensor.shape) for n in names])
            out['trainable_names'] = np.array([v.var_name for v in tf.trainable_variables()])
            opt_d, opt_g = g.optimizers                                                              # D optimiser is created first (:198, :201)
            out['D_var_list'] = np.array([v.var_name for v in opt_d.var_list])
            out['G_var_list'] = np.array([v.var_name for v in opt_g.var_list])
            out['update_op_targets'] = np.array([u.target.var_name for u in g.update_ops])
            out['train_op_G_control_inputs'] = np.array([c.target.var_name for c in model.train_op_G.control])
            out['train_op_D_control_inputs'] = np.array([getattr(c, 'op', '?') for c in model.train_op_D.control])
            out['lr_args'] = np.array([g.lr_args['learning_rate'], g.lr_args['decay_steps'], g.lr_args['decay_rate']], np.float64)
            out['lr_global_step_var'] = np.array(g.lr_args['global_step'])
            out['adam_args'] = np.array([[float(o.b1), float(o.b2), float(o.eps)] for o in (opt_d, opt_g)], np.float64)
            out['adam_increments_global_step'] = np.array([o.global_step is not None for o in (opt_d, opt_g)])
            model_vars = [n for n in names if not (n.endswith('/Adam') or n.endswith('/Adam_1') or n.startswith('beta') or n == 'global_step')]
            out['init_digest'] = np.stack([digest(g.vars[n].tensor) for n in model_vars])
            out['init_head'] = np.stack([np.resize(g.vars[n].tensor.detach().numpy().ravel()[:4], 4) for n in model_vars])
            out['model_var_names'] = np.array(model_vars)

            # ---------------------------------------------------------------- C. one forward at the initial weights (batch 0)
            def nodes(op, scope=None):
                return [n for n in g.nodes if n.op == op and (scope is None or n.scope == scope)]
            stacks = nodes('stack', 'pose_encoder')                                                   # networks/__init__.py:71, two calls
            lo_maps = [n for n in nodes('transpose') if n.shape[1] == 32]                             # get_gaussian_maps(.., [32, 32]) :168-169
            d_logits = nodes('conv2d', 'img_discr/D_logit')                                           # real, fake (D loss), fake (G loss)
            feats = [[n for n in nodes('relu', 'content_vgg/' + s)][0] for s in ('conv1_2', 'conv2_2', 'conv3_4', 'conv4_4', 'conv5_4')]
            assert len(stacks) == 2 and len(lo_maps) == 2 and len(d_logits) == 3
            fetch = dict(final_output=model.final_output, crude_output=model.crude_output, mask=model.mask,
                         current_keypoints_map=model.current_keypoints_map, future_keypoints_map=model.future_keypoints_map,
                         current_points=stacks[0], future_points=stacks[1], current_map_lo=lo_maps[0], future_map_lo=lo_maps[1],
                         D_logit_real=d_logits[0], D_logit_fake=d_logits[1], D_logit_fake_G=d_logits[2],
                         loss_D_real=model.loss_D_real, loss_D_fake=model.loss_D_fake, loss_D=model.loss_D,
                         loss_G_recon=model.loss_G_recon, loss_G_adv=model.loss_G_adv, loss_G=model.loss_G, lr=model.current_lr,
                         **{'vgg_feat_%d' % i: f for i, f in enumerate(feats)})
            vals = sess.run(fetch)
            for k_, v in vals.items():
                v = np.asarray(v)
                if v.size <= 8192:
                    out['fwd_' + k_] = v.astype(np.float32)
                else:
                    out['fwd_' + k_ + '_digest'] = digest(v)
                    out['fwd_' + k_ + '_shape'] = np.array(v.shape, np.int64)
                    if v.ndim == 4 and v.shape[1] == RES and v.shape[-1] <= 4:
                        out['fwd_' + k_ + '_sub2'] = v[:, ::2, ::2, :].astype(np.float32)           # every second pixel
                    else:
                        out['fwd_' + k_ + '_sample'] = sample(v)

            # ---------------------------------------------------------------- D. the reference's train_step, twice
            def state_digests():
                return np.stack([digest(g.vars[n].tensor) for n in names if n != 'global_step'])
            out['state_names'] = np.array([n for n in names if n != 'global_step'])
            for step in range(2):
                h0 = len(sess.history)
                model.train_step(sess, {}, step, B, should_write_log=True, should_write_summary=False)   # reference :79-117
                (_, d_vals), (_, g_vals) = sess.history[h0:h0 + 2]
                out['step%d_loss_D' % step] = np.float32(d_vals[0])
                out['step%d_loss_G' % step] = np.float32(g_vals[0])
                rec_d, rec_g = g.grad_records[-2], g.grad_records[-1]
                assert rec_d[0] == opt_d.index and rec_g[0] == opt_g.index
                out['step%d_grad_D_digest' % step] = np.stack([digest(t) for t in rec_d[2].values()])
                out['step%d_grad_G_digest' % step] = np.stack([digest(t) for t in rec_g[2].values()])
                # <gradient, fixed random direction> per variable, in var_list order (keys of the records are the variable names)
                assert [str(k_) for k_ in rec_d[2]] == [v.var_name for v in opt_d.var_list] and [str(k_) for k_ in rec_g[2]] == [v.var_name for v in opt_g.var_list]
                out['step%d_grad_D_proj' % step] = projections(rec_d[2].items())
                out['step%d_grad_G_proj' % step] = projections(rec_g[2].items())
                out['step%d_state_digest' % step] = state_digests()
                out['step%d_global_step' % step] = np.int64(int(global_step.tensor))
                out['step%d_log' % step] = np.array(g.log[-1].split(': ', 1)[1].split(' (')[0])     # 'step N, loss_D = .., loss_G = ..'
            # ---------------------------------------------------------------- E. the reference's test_step
            before = state_digests()
            loss_d, loss_g, _, n_ex = model.test_step(sess, {}, 2, 1, B)                              # reference :119-141
            out['test_loss_D'], out['test_loss_G'] = np.float32(loss_d), np.float32(loss_g)
            out['test_step_changed_state'] = np.array(not np.array_equal(before, state_digests()))
            out['test_global_step'] = np.int64(int(global_step.tensor))
            out['served'] = np.array(source.served, np.int64)                                         # (run id, batch index)
            out['case'] = np.array([RES, K, B, N_BATCHES, VGG_SEED, VGG_WIDTH_DIV, INIT_SEED], np.int64)
    finally:
        np.load = np_load
        sys.path.remove(REF)
    path = os.path.join(HERE, 'networks_ref.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KB): %d variables, served=%s' % (path, os.path.getsize(path) / 1024, len(out['var_names']), out['served'].tolist()))
    for k_ in ('fwd_loss_D', 'fwd_loss_G', 'step0_loss_D', 'step0_loss_G', 'step1_loss_D', 'step1_loss_G', 'test_loss_D', 'test_loss_G', 'test_step_changed_state'):
        print('  ', k_, out[k_])


if __name__ == '__main__':
    main()
