"""The CPU restatement (oracle/restatement.py) against what the REFERENCE's own graph code produced.

tests/golden/networks_ref.npz was written by tests/golden/make_networks_golden.py, which executes the reference's
models/networks/{__init__,layers,vgg}.py, utils/model.py and models/detector_translator_model.py unmodified under the lazy
TensorFlow stand-in tests/golden/tf_standin.py (128x128, K=3, batch 2, width/8 synthetic VGG19).  This pins the restatement's
WIRING -- variable names / shapes / creation order, layer order, skip and concat order, pad + SAME composition, loss wiring,
D / G variable split, optimiser arguments, BN update-op set, two sess.run per step with a fresh batch each -- to the
reference's files.  The arithmetic inside each op is [TF-sem] on both sides and stays unpinned.
"""
import os

import numpy as np
import pytest
import torch

from gradproj import projection
from oracle import restatement as R

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def ref():
    return np.load(os.path.join(HERE, 'golden', 'networks_ref.npz'))


def batch(i, b=2, res=128):
    """Same generator as make_networks_golden.synthetic_batch (seeds 100+2i / 101+2i)."""
    return R.synthetic_pair(b, res=res, seed0=100 + 2 * i, seed1=101 + 2 * i)


def digest(t):
    a = np.asarray(t.detach().numpy() if torch.is_tensor(t) else t, dtype=np.float64).ravel()
    return np.array([np.sqrt((a * a).sum()), a.sum(), np.abs(a).max() if a.size else 0.0], np.float64)


def sample(a, n=512):
    a = np.asarray(a).ravel()
    idx = np.linspace(0, a.size - 1, min(n, a.size)).astype(np.int64)
    return a[idx].astype(np.float32)


def test_variable_registry_is_the_references(ref):
    res, k = int(ref['case'][0]), int(ref['case'][1])
    manifest = R.variable_manifest(k, res)
    names = [str(n) for n in ref['var_names']]
    shapes = {n: tuple(int(s) for s in str(sh).split(',') if s) for n, sh in zip(names, ref['var_shapes'])}
    model_vars = [str(n) for n in ref['model_var_names']]
    # the model variables, in tf.global_variables() creation order, are exactly the restatement's manifest
    assert model_vars == list(manifest)
    for n, s in manifest.items():
        assert shapes[n] == tuple(s), n
    # trainable = everything but the moving statistics; split by the substring 'img_discr' (reference :191-192)
    trainable = [str(n) for n in ref['trainable_names']]
    assert trainable == [n for n in manifest if 'moving_' not in n]
    assert [str(n) for n in ref['D_var_list']] == [n for n in trainable if 'img_discr' in n]
    assert [str(n) for n in ref['G_var_list']] == [n for n in trainable if 'img_discr' not in n]
    # checkpoint extras (Appendix B; TF slot naming is [TF-sem]): global_step first (train.py:30), D optimiser's powers unsuffixed
    assert names[0] == 'global_step' and shapes['global_step'] == ()
    extras = [n for n in names if n not in manifest and n != 'global_step']
    want = {'beta1_power', 'beta2_power', 'beta1_power_1', 'beta2_power_1'}
    want |= {n + s for n in trainable for s in ('/Adam', '/Adam_1')}
    assert set(extras) == want
    assert names.index('beta1_power') < names.index('beta1_power_1')
    assert names.index('img_discr/conv_0/conv2d/kernel/Adam') < names.index('beta1_power_1')       # D slots are created first (:198)
    # UPDATE_OPS: one moving_mean + one moving_variance update per BN *call*: image_encoder 8, pose_encoder 22 x 2 calls, translator 10
    upd = [str(n) for n in ref['update_op_targets']]
    assert len(upd) == 2 * (8 + 22 * 2 + 10)
    assert upd.count('pose_encoder/b_norm_1_0/moving_mean') == 2 and upd.count('image_encoder/encoder/b_norm_8/moving_variance') == 1
    assert not any('img_discr' in n for n in upd)
    assert [str(n) for n in ref['train_op_G_control_inputs']] == upd and ref['train_op_D_control_inputs'].size == 0   # (:199-202)
    np.testing.assert_allclose(ref['lr_args'], [1e-4, 20000, 0.95])
    np.testing.assert_allclose(ref['adam_args'], [[0.5, 0.999, 1e-8]] * 2, rtol=1e-6)
    assert ref['adam_increments_global_step'].tolist() == [False, True] and str(ref['lr_global_step_var']) == 'global_step'


def test_initial_values_follow_creation_order(ref):
    """xavier-uniform draws from RandomState(1234) in creation order: equal digests <=> equal names, shapes AND order."""
    res, k, seed = int(ref['case'][0]), int(ref['case'][1]), int(ref['case'][6])
    init = R.init_variables(k, res=res, seed=seed)
    for i, n in enumerate(str(s) for s in ref['model_var_names']):
        np.testing.assert_allclose(digest(init[n]), ref['init_digest'][i], rtol=1e-12, atol=0, err_msg=n)
        np.testing.assert_array_equal(np.resize(init[n].ravel()[:4], 4), ref['init_head'][i])


@pytest.fixture(scope='module')
def restated(ref):
    """One forward at the initial weights, two train steps and one evaluation pass of the restatement, fed like the reference's
    session was (batch 0: forward probe; 1,2 and 3,4: D-run / G-run of the two train steps; 5: test_step)."""
    res, k, b = (int(v) for v in ref['case'][:3])
    assert ref['served'][:, 1].tolist() == [0, 1, 2, 3, 4, 5]
    torch.manual_seed(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    vgg = R.synthetic_vgg(seed=int(ref['case'][4]), width_div=int(ref['case'][5]))
    st = R.TrainState(R.init_variables(k, res=res, seed=int(ref['case'][6])), vgg)
    out = {}
    with torch.no_grad():
        im, fut = (torch.from_numpy(a) for a in batch(0, b, res))
        net = R.Net(st.params, train_mode=True)
        f = R.forward_pass(net, im, fut)
        f['D_logit_real'] = R.img_discr(net, fut)
        f['D_logit_fake'] = R.img_discr(net, f['final_output'])
        f['loss_D'], f['loss_D_real'], f['loss_D_fake'] = R.loss_D(net, f['final_output'], fut)
        f['loss_G'], f['loss_G_recon'], f['loss_G_adv'] = R.loss_G(net, st.vgg, f['final_output'], fut)
        feats = R.vgg19(st.vgg, torch.cat([(fut + 1) / 2.0 * 255.0, (f['final_output'] + 1) / 2.0 * 255.0], dim=0))
        for i, t in enumerate(feats):
            f['vgg_feat_%d' % i] = t
    out['fwd'] = f
    out['steps'] = []
    for step in range(2):
        (im_d, fut_d), (im_g, fut_g) = batch(1 + 2 * step, b, res), batch(2 + 2 * step, b, res)
        r = R.train_step(st, im_d, fut_d, same_batch=False, im_G=im_g, future_im_G=fut_g)
        r['state'] = {n: t.detach().clone() for n, t in st.params.items()}
        for tag, opt in (('D', st.opt_D), ('G', st.opt_G)):
            for n in opt.names:
                r['state'][n + '/Adam'], r['state'][n + '/Adam_1'] = opt.m[n].clone(), opt.v[n].clone()
        r['state'].update(beta1_power=st.opt_D.b1p, beta2_power=st.opt_D.b2p, beta1_power_1=st.opt_G.b1p, beta2_power_1=st.opt_G.b2p)
        r['global_step'] = st.global_step
        out['steps'].append(r)
    with torch.no_grad():
        im, fut = (torch.from_numpy(a) for a in batch(5, b, res))
        net = R.Net(st.params, train_mode=True)                    # is_training is a Python bool: test_step uses batch statistics
        f = R.forward_pass(net, im, fut, with_vis_maps=False)
        out['test'] = (float(R.loss_D(net, f['final_output'], fut)[0]), float(R.loss_G(net, st.vgg, f['final_output'], fut)[0]))
    return out


def test_forward_matches_reference_graph(ref, restated):
    f = restated['fwd']
    for key in ('current_points', 'future_points'):
        np.testing.assert_allclose(f[key].numpy(), ref['fwd_' + key], atol=2e-6)
    for key in ('current_map_lo', 'future_map_lo', 'D_logit_real', 'D_logit_fake'):
        np.testing.assert_allclose(f[key].numpy(), ref['fwd_' + key], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(f['D_logit_fake'].numpy(), ref['fwd_D_logit_fake_G'], rtol=1e-4, atol=1e-5)   # the G loss re-applies img_discr (:264)
    for key in ('final_output', 'crude_output', 'mask', 'current_keypoints_map', 'future_keypoints_map'):
        a = f[key].numpy()
        assert tuple(ref['fwd_%s_shape' % key]) == a.shape
        np.testing.assert_allclose(a[:, ::2, ::2, :], ref['fwd_%s_sub2' % key], rtol=1e-4, atol=2e-5, err_msg=key)
        np.testing.assert_allclose(digest(a)[:2], ref['fwd_%s_digest' % key][:2], rtol=1e-5, err_msg=key)
    for i in range(5):
        a = f['vgg_feat_%d' % i].numpy()
        assert tuple(ref['fwd_vgg_feat_%d_shape' % i]) == a.shape
        np.testing.assert_allclose(sample(a), ref['fwd_vgg_feat_%d_sample' % i], rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(digest(a)[:2], ref['fwd_vgg_feat_%d_digest' % i][:2], rtol=1e-5)
    for key in ('loss_D_real', 'loss_D_fake', 'loss_D', 'loss_G_recon', 'loss_G_adv', 'loss_G'):
        assert abs(float(f[key]) - float(ref['fwd_' + key])) <= 1e-5 * max(1.0, abs(float(ref['fwd_' + key]))), key
    assert float(ref['fwd_lr']) == float(R.exponential_decay(1e-4, 0, 20000, 0.95))


def test_train_and_test_steps_match_reference_graph(ref, restated):
    state_names = [str(n) for n in ref['state_names']]
    d_names, g_names = [str(n) for n in ref['D_var_list']], [str(n) for n in ref['G_var_list']]
    for step, r in enumerate(restated['steps']):
        for key in ('loss_D', 'loss_G'):
            want = float(ref['step%d_%s' % (step, key)])
            assert abs(r[key] - want) <= (2e-5 if step == 0 else 1e-3) * max(1.0, abs(want)), (step, key, r[key], want)
        assert r['global_step'] == int(ref['step%d_global_step' % step]) == step + 1
        # gradient digests (l2 norm per variable).  Biases in front of a batch norm / the key-point softmax carry rounding noise only.
        for names, grads, tag in ((d_names, r['grads_D'], 'D'), (g_names, r['grads_G'], 'G')):
            want = ref['step%d_grad_%s_digest' % (step, tag)]
            num = den = 0.0
            for i, n in enumerate(names):
                got = digest(grads[n])
                if n.endswith('/bias') and not n.startswith('img_discr') and 'translator/conv_6' not in n:
                    assert got[0] < 1e-3 and want[i][0] < 1e-3, (step, n)
                    continue
                num += (got[0] - want[i][0]) ** 2
                den += want[i][0] ** 2
                # per variable loose (the perceptual-L1 / max-pool / ReLU gradient is discontinuous: the pose_encoder gradients of two
                # runs of the SAME code with different thread counts already differ by ~1 %), norm-weighted aggregate tight
                assert abs(got[0] - want[i][0]) <= (5e-2 if step == 0 else 0.15) * want[i][0] + 1e-9, (step, tag, n, got[0], want[i][0])
            assert (num / den) ** 0.5 < 1e-2, (step, tag, (num / den) ** 0.5)
            # ... and the DIRECTION of every gradient: <g, fixed random direction> (tests/gradproj.py).  The norm above is blind to a sign
            # flip or a permuted / transposed filter gradient; the projection moves by ~|g| under any of them.
            wantp = ref['step%d_grad_%s_proj' % (step, tag)]
            pnum = 0.0
            for i, n in enumerate(names):
                if n.endswith('/bias') and not n.startswith('img_discr') and 'translator/conv_6' not in n:
                    continue
                dp = projection(n, grads[n].detach().numpy()) - wantp[i]
                pnum += dp ** 2
                assert abs(dp) <= (5e-2 if step == 0 else 0.15) * want[i][0] + 1e-9, (step, tag, n, dp, want[i][0])
            print('step %d %s: gradient projections off by %.2e of the gradient norm (aggregate)' % (step, tag, (pnum / den) ** 0.5))
            # (step 1 starts from weights that two fp32 implementations have already separated by +-lr on noise-level elements)
            assert (pnum / den) ** 0.5 < (1e-3 if step == 0 else 5e-2), (step, tag, 'projection', (pnum / den) ** 0.5)
        want = ref['step%d_state_digest' % step]
        for i, n in enumerate(state_names):
            got = digest(r['state'][n])
            # after an Adam step every element has moved by ~lr: the l2 norm of a tensor is stable to ~lr * sqrt(n) * (fraction of flips)
            base = n.replace('/Adam_1', '').replace('/Adam', '')
            if base.endswith('/bias') and not base.startswith('img_discr') and 'translator/conv_6' not in base:
                # exact gradient 0 (conv feeding a batch norm / the key-point softmax): rounding noise that Adam turns into
                # a +-lr random walk on both sides -- only the excursion bound is comparable
                if '/Adam' not in n:
                    assert got[2] <= (step + 1) * 1.01e-4 and want[i][2] <= (step + 1) * 1.01e-4, (step, n, got[2], want[i][2])
                continue
            tol = (5e-3 if step == 0 else 3e-2) if '/Adam' in n else 2e-5 * (step + 1)   # slots inherit the gradient tolerance;      # elements with a noise-level gradient may step the other way (+-lr)
            assert abs(got[0] - want[i][0]) <= tol * max(want[i][0], 1.0) + 1e-7, (step, n, got[0], want[i][0])
    assert not bool(ref['test_step_changed_state']) and int(ref['test_global_step']) == 2
    for got, key in zip(restated['test'], ('test_loss_D', 'test_loss_G')):
        # two Adam steps in, elements with noise-level gradients have stepped +-lr differently on the two sides: 1e-3, not 2e-5
        assert abs(got - float(ref[key])) <= 1e-3 * max(1.0, abs(float(ref[key]))), (key, got, float(ref[key]))
