"""Device side of the input pipeline: uint8 -> [-1,1] kernel and the pinned / async batcher (kpx_amd.data)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def kpx():
    import kpx_amd
    return kpx_amd


def test_u8_to_unit_kernel_is_exact_for_every_byte(kpx):
    from kpx_amd._lib import lib, check
    dev = torch.device('cuda', 0)
    src = torch.arange(256, dtype=torch.uint8).repeat(37).to(dev)
    dst = torch.empty(src.numel(), dtype=torch.float32, device=dev)
    check(lib.kpx_u8_to_unit_f32(src.data_ptr(), src.numel(), dst.data_ptr(), torch.cuda.current_stream().cuda_stream), 'u8')
    want = (src.cpu().numpy().astype(np.float64) / 255.0).astype(np.float32) * np.float32(2) - np.float32(1)
    assert np.array_equal(dst.cpu().numpy(), want)


def test_device_batches_equal_host_batches(kpx, tmp_path):
    golden = np.load(os.path.join(HERE, 'golden', 'image_pair_ref.npz'))
    for k, rel in enumerate(str(n) for n in golden['file_names']):
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(golden['file_%03d' % k].tobytes())
    (tmp_path / 'train_set.txt').write_bytes(golden['listing'].tobytes())
    import random
    from kpx_amd.data import ImagePairDataLoader
    mk = lambda: ImagePairDataLoader(str(tmp_path), 'train', random_order=True, randomness=True, rng=random.Random(5), np_rng=np.random.RandomState(5))
    dev = torch.device('cuda', 0)
    it_d, it_h = iter(mk().batches(4, dev, repeat=True)), iter(mk().batches(4, 'cpu', repeat=True))
    for _ in range(5):                       # more batches than pinned slots: the ring is reused
        bd, bh = next(it_d), next(it_h)
        for key in ('image', 'future_image'):
            assert bd[key].device.type == 'cuda' and bd[key].dtype == torch.float32 and tuple(bd[key].shape) == (4, 128, 128, 3)
            assert torch.equal(bd[key].cpu(), bh[key])


def test_train_py_runs_on_the_jpeg_pipeline(kpx, tmp_path):
    """train.py without --synthetic: ImagePairDataLoader batches feed real train / test steps (tiny dataset, 2 steps)."""
    import yaml
    golden = np.load(os.path.join(HERE, 'golden', 'image_pair_ref.npz'))
    data = tmp_path / 'penn'
    for k, rel in enumerate(str(n) for n in golden['file_names']):
        p = data / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(golden['file_%03d' % k].tobytes())
    (data / 'train_set.txt').write_bytes(golden['listing'].tobytes())
    (data / 'test_set.txt').write_bytes(golden['listing'].tobytes())
    cfg = {'paths': {'data_dir': str(data), 'vggnet': str(tmp_path / 'none.npy'), 'log_dir': str(tmp_path / 'results')},
           'training': {'n_steps': 2, 'summary_interval': 500, 'test_interval': 1, 'checkpoint_interval': 1000, 'log_interval': 1,
                        'batch_size': 2, 'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}},
           'model': {'n_pts': 5, 'n_action': 9, 'cell_info': [32, 32], 'vae_dim': 8}}
    cfg_path = tmp_path / 'cfg.yaml'
    cfg_path.write_text(yaml.safe_dump(cfg))
    import train
    train.main(['--mode', 'detector_translator', '--config', str(cfg_path), '--synthetic-vgg', '--steps', '2'])
    assert (tmp_path / 'results' / 'detector_translator' / 'model.ckpt-0.npz').exists()


def _fixture_dataset(root):
    golden = np.load(os.path.join(HERE, 'golden', 'image_pair_ref.npz'))
    for k, rel in enumerate(str(n) for n in golden['file_names']):
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(golden['file_%03d' % k].tobytes())
    for subset in ('train', 'test'):
        (root / (subset + '_set.txt')).write_bytes(golden['listing'].tobytes())
    return golden


def test_pseudo_label_and_evaluate_scripts_run_on_jpeg_data(kpx, tmp_path):
    """make_pseudo_labels.py (KeypointDataLoader) then evaluate.py (SequenceDataLoader with the future frames) on the fixture videos."""
    import yaml
    data = tmp_path / 'penn'
    golden = _fixture_dataset(data)
    k = int(golden['n_points'])
    cfg = {'paths': {'data_dir': str(data), 'vggnet': str(tmp_path / 'none.npy'), 'log_dir': str(tmp_path / 'results')},
           'training': {'n_steps': 1, 'summary_interval': 500, 'test_interval': 500, 'checkpoint_interval': 1000, 'log_interval': 1,
                        'batch_size': 2, 'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}},
           'model': {'n_pts': k, 'n_action': int(golden['n_action']), 'cell_info': [32, 32], 'vae_dim': 8}}
    cfg_path = tmp_path / 'cfg.yaml'
    cfg_path.write_text(yaml.safe_dump(cfg))
    dev = torch.device('cuda', 0)
    fm = kpx.FinalModel(cfg, device=dev); fm.build(None); fm.initialize_loggers(str(tmp_path / 'ckpt'))
    ckpt = fm.save_checkpoint(None, 0)
    import make_pseudo_labels, evaluate
    make_pseudo_labels.main(['--config', str(cfg_path), '--checkpoint', ckpt])
    for vid, n in zip(golden['kp_idx'], golden['kp_len']):
        pts = np.load(str(data / 'pseudo_labels' / ('%04d.npy' % int(vid))))
        assert pts.shape == (int(n), k, 2) and np.isfinite(pts).all() and np.abs(pts).max() <= 1.0
    evaluate.main(['--config', str(cfg_path), '--checkpoint_stage1', ckpt, '--checkpoint_stage2', ckpt, '--save_dir', str(tmp_path / 'eval'), '--batch', '2'])
    assert sorted(os.listdir(str(tmp_path / 'eval'))) == ['0000', '0001', '0002']
    assert len(os.listdir(str(tmp_path / 'eval' / '0000' / 'pred_seq'))) == 32 and len(os.listdir(str(tmp_path / 'eval' / '0000' / 'real_seq'))) == 32


def test_train_py_motion_generator_on_sequence_loader(kpx, tmp_path):
    """train.py --mode motion_generator on the fixture videos + pseudo labels (SequenceDataLoader batches), tiny cells."""
    import yaml
    data = tmp_path / 'penn'
    golden = _fixture_dataset(data)
    cfg = {'paths': {'data_dir': str(data), 'vggnet': None, 'log_dir': str(tmp_path / 'results')},
           'training': {'n_steps': 3, 'summary_interval': 500, 'test_interval': 2, 'checkpoint_interval': 2, 'log_interval': 1,
                        'batch_size': 2, 'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}},
           'model': {'n_pts': int(golden['n_points']), 'n_action': int(golden['n_action']), 'cell_info': [32, 32], 'vae_dim': 8}}
    cfg_path = tmp_path / 'cfg.yaml'
    cfg_path.write_text(yaml.safe_dump(cfg))
    import train
    train.main(['--mode', 'motion_generator', '--config', str(cfg_path), '--steps', '3'])
    ck = tmp_path / 'results' / 'motion_generator' / 'model.ckpt-2.npz'
    assert ck.exists()
    names = set(k.replace('|', '/') for k in np.load(str(ck)).files)
    assert 'seq_discr/rnn/multi_rnn_cell/cell_1/basic_lstm_cell/kernel/Adam' in names and 'vae_encoder/fully_connected/weights' in names
