"""Stage-2 training (MotionGeneratorModel) against the CPU restatement (oracle.motion_train_step) on a small configuration."""
import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
K, A, CELLS, VAE, DISCR = 5, 9, (32, 32), 8, (32, 32)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _make(dev):
    import kpx_amd
    cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': 3},
           'model': {'n_pts': K, 'n_action': A, 'cell_info': list(CELLS), 'vae_dim': VAE}, 'paths': {'log_dir': '/tmp/kpx_motion'}}
    model = kpx_amd.MotionGeneratorModel(cfg, device=dev, discr_cells=DISCR)
    model.build()
    return model


def test_manifest_and_train_steps_match_oracle():
    dev = torch.device('cuda:0')
    model = _make(dev)
    manifest = R.motion_generator_manifest(K, A, CELLS, VAE, DISCR)
    assert {n: tuple(model.store.vars[n].shape) for n in model.store.vars} == {n: tuple(s) for n, s in manifest.items()}
    params = R.init_motion_generator(K, A, CELLS, VAE, DISCR, seed=5)
    model.store.load_numpy(params, strict=True)
    st = R.MotionTrainState(params, K, A, CELLS, VAE, DISCR)
    rs = np.random.RandomState(1)
    b = 3
    for step in range(2):
        kp = (rs.rand(b, K, 2) * 1.6 - 0.8).astype(np.float32)
        seq = (rs.rand(b, 32, K, 2) * 1.6 - 0.8).astype(np.float32)
        ac = np.eye(A, dtype=np.float32)[rs.randint(0, A, size=b)]
        e_d, e_g = rs.randn(b, VAE).astype(np.float32), rs.randn(b, VAE).astype(np.float32)
        want = R.motion_train_step(st, kp, seq, ac, e_d, e_g)
        feed = {k: torch.from_numpy(v).to(dev) for k, v in dict(keypoints=kp, real_seq=seq, action_code=ac, eps_D=e_d, eps_G=e_g).items()}
        model.train_step(None, feed, step, b)
        got = model.loss_values()
        for key in ('loss_D', 'loss_G', 'loss_G_recon', 'loss_G_kl', 'loss_G_adv'):
            assert abs(got[key] - want[key]) <= 2e-5 * max(1.0, abs(want[key])), (step, key, got[key], want[key])
        assert rel_l2(model.last['pred_seq'].cpu().numpy(), want['pred_seq'].numpy()) < 1e-5
        # gradients of the G bucket still hold this step's values
        for n, g in want['grads_G'].items():
            if float(g.abs().max()) > 1e-7:
                assert rel_l2(model.store.grad(n).cpu().numpy(), g.numpy()) < 2e-4, (step, n)
        # parameters after Adam: elements whose gradient is rounding noise may flip sign (+-lr), like in stage 1
        exp = model.store.export_numpy(include_slots=False)
        for n, p in st.params.items():
            diff = np.abs(exp[n] - p.numpy())
            assert diff.max() <= (step + 1) * 2.05e-4 and np.mean(diff > 1e-5) < 0.02, (step, n, float(diff.max()), float(np.mean(diff > 1e-5)))


def test_checkpoint_names_and_sampling():
    dev = torch.device('cuda:0')
    model = _make(dev)
    arrays = model.checkpoint_arrays()
    for name in ('vae_encoder/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel', 'vae_encoder/fully_connected/weights',
                 'vae_decoder/multi_rnn_cell/cell_1/basic_lstm_cell/bias', 'vae_decoder/fully_connected/W', 'seq_discr/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel',
                 'seq_discr/fully_connected/biases', 'seq_discr/fully_connected/weights/Adam', 'vae_decoder/fully_connected/b/Adam_1', 'global_step', 'beta2_power_1'):
        assert name in arrays, name
    assert np.asarray(arrays['global_step']).dtype == np.int32      # tf.Variable(0) (reference train.py:30) is DT_INT32
    kp = torch.rand(4, K, 2, device=dev) * 1.6 - 0.8
    ac = torch.eye(A, device=dev)[:4].contiguous()
    z = torch.randn(4, VAE, device=dev)
    seq = model.sample(kp, ac, z)
    want = R.vae_decoder({k: torch.from_numpy(np.asarray(v)) for k, v in model.store.export_numpy(include_slots=False).items()},
                         z.cpu(), kp.reshape(4, -1).cpu(), ac.cpu(), CELLS, K)
    assert tuple(seq.shape) == (4, 32, 2 * K) and rel_l2(seq.cpu().numpy(), want.numpy()) < 1e-5
