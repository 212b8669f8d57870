#!/usr/bin/env python3
"""Training entry point with the reference's CLI (reference: train.py:12-114):

    python train.py --mode detector_translator --config configs/penn.yaml [--synthetic] [--steps N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --mode ... (data parallel)

Both modes are built: ``detector_translator`` (stage 1, the BASELINE.json hot path) and ``motion_generator`` (stage 2, the
sequence VAE-GAN over the pseudo labels written by make_pseudo_labels.py).  Without ``--synthetic`` the frames come from ``paths.data_dir`` through kpx_amd.data.ImagePairDataLoader (the
reference's data/image_pair_dataloader.py + tf.data pipeline: threaded PIL augmentation, uint8 batches in pinned memory, async
copy + on-device conversion); ``--synthetic`` feeds Penn-shaped random pairs with the same output contract (float32 NHWC in
[-1,1], keys image / future_image).
"""
import logging
import os
import sys
from argparse import ArgumentParser

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_config(config_path):
    """reference utils/__init__.py:8-10"""
    with open(config_path, 'r') as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def _get_model_by_mode(mode, config, global_step, **kw):
    """reference train.py:117-123"""
    if mode == 'detector_translator':
        from kpx_amd import DetectorTranslatorModel
        return DetectorTranslatorModel(config, global_step, is_training=True, **kw)
    if mode == 'motion_generator':
        from kpx_amd import MotionGeneratorModel
        kw.pop('vgg', None)
        return MotionGeneratorModel(config, global_step, is_training=True, **kw)
    raise Exception('unknown model %s' % mode)


def _train_motion_generator(args, config, dev, rank, world):
    """reference train.py:136-147 (SequenceDataLoader, is_train -> random_order + randomness) + the common loop (:84-113)."""
    import numpy as np
    import kpx_amd
    paths_config, train_config, model_config = config['paths'], config['training'], config['model']
    model = _get_model_by_mode('motion_generator', config, 0, device=dev)
    print('model initialized')
    model.build(None)
    batch_size = train_config['batch_size']
    n_steps = args.steps if args.steps is not None else train_config['n_steps']
    k, a = model_config['n_pts'], model_config['n_action']
    train_it = test_batches = None
    if not args.synthetic:
        import random
        random.seed(1000 + rank); np.random.seed(1000 + rank)
        mk = lambda subset, train: kpx_amd.SequenceDataLoader(paths_config['data_dir'], subset, n_points=k, n_action=a,
                                                              random_order=train, randomness=train)
        train_it = mk('train', True).batches(batch_size, dev, repeat=True)
        if os.path.exists(os.path.join(str(paths_config['data_dir']), 'test_set.txt')):
            test_loader = mk('test', False)
            test_batches = lambda: test_loader.batches(batch_size, dev, repeat=False)

    def synthetic(seed):
        rs = np.random.RandomState(seed)
        d = {'keypoints': rs.rand(batch_size, k, 2) * 1.6 - 0.8, 'real_seq': rs.rand(batch_size, 32, k, 2) * 1.6 - 0.8,
             'action_code': np.eye(a)[rs.randint(0, a, size=batch_size)]}
        return {n: torch.from_numpy(v.astype(np.float32)).to(dev) for n, v in d.items()}
    model.initialize_loggers(paths_config['log_dir'], None)
    print('training start')
    for step in range(model.global_step, n_steps):
        feed = next(train_it) if train_it is not None else synthetic(step * world + rank)
        model.train_step(None, feed, step, batch_size, should_write_log=step % train_config['log_interval'] == 0 and rank == 0)
        if step % train_config['checkpoint_interval'] == 0 and rank == 0:
            model.save_checkpoint(None, step, fmt=args.ckpt_format)
        if step % train_config['test_interval'] == 0 and rank == 0:
            feeds = test_batches() if test_batches is not None else [synthetic(10 ** 6)]
            model.collect_test_results([model.test_step(None, f, step, i, f['keypoints'].shape[0]) for i, f in enumerate(feeds)], step)
    if world > 1:
        torch.distributed.destroy_process_group()


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--mode', type=str, choices=['detector_translator', 'motion_generator'], help='which mode to train')
    parser.add_argument('--config', type=str, help='path of the configuration file')
    parser.add_argument('--synthetic', action='store_true', help='synthetic Penn-shaped pairs instead of the JPEG pipeline')
    parser.add_argument('--synthetic-vgg', action='store_true', help='He-normal VGG19 weights when paths.vggnet is absent')
    parser.add_argument('--steps', type=int, default=None, help='override training.n_steps')
    parser.add_argument('--ckpt-format', choices=['npz', 'tf', 'bundle'], default=os.environ.get('KPX_CKPT_FORMAT', 'npz'),
                        help="checkpoint container: 'tf' = TensorFlow V2 bundle (model.ckpt-N.index / .data-00000-of-00001 + `checkpoint`, the "
                             "layout tf.train.Saver writes and the reference restores, models/base_model.py:74-91); 'npz' (default) = one "
                             "numpy archive with the same variable names -- readable by this repo only")
    parser.add_argument('--batch-per-run', action='store_true',
                        help="feed a NEW batch to the G update like the reference's two sess.run (train.py:46-50); default: one batch per "
                             'step with a shared generator forward')
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='%(message)s')

    config = load_config(args.config)
    paths_config, train_config = config['paths'], config['training']
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        torch.distributed.init_process_group('nccl', device_id=dev)

    import kpx_amd
    from kpx_amd.synthetic import synthetic_pair
    vgg = None
    if args.mode == 'motion_generator':
        return _train_motion_generator(args, config, dev, rank, world)
    if args.synthetic_vgg or not os.path.exists(str(paths_config.get('vggnet'))):
        if not args.synthetic_vgg:
            raise Exception('file of pretrained vgg19 does not exist at: %s (pass --synthetic-vgg for benchmarking)' % paths_config.get('vggnet'))
        vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(), device=dev)
    model = _get_model_by_mode(args.mode, config, 0, device=dev, vgg=vgg)
    print('model initialized')
    model.build(None)
    batch_size = train_config['batch_size']          # per process, like the reference's single-device batch
    train_it = test_it = None
    if not args.synthetic:                           # reference train.py:125-135, 60-70: is_train -> random_order + randomness
        data_dir = paths_config['data_dir']
        if not os.path.exists(os.path.join(str(data_dir), 'train_set.txt')):
            raise Exception('no train_set.txt under paths.data_dir = %s (pass --synthetic for benchmarking)' % data_dir)
        import random
        import numpy as np
        random.seed(1000 + rank); np.random.seed(1000 + rank)          # different sample streams per data-parallel rank
        train_it = iter(kpx_amd.ImagePairDataLoader(data_dir, 'train', random_order=True, randomness=True)
                        .batches(batch_size, dev, repeat=True, shuffle=True))
        if os.path.exists(os.path.join(str(data_dir), 'test_set.txt')):
            test_loader = kpx_amd.ImagePairDataLoader(data_dir, 'test', random_order=False, randomness=False)
            test_it = lambda: iter(test_loader.batches(batch_size, dev, repeat=False, shuffle=False))
    n_steps = args.steps if args.steps is not None else train_config['n_steps']
    model.initialize_loggers(paths_config['log_dir'], None)
    print('training start')
    for step in range(model.global_step, n_steps):   # reference train.py:84-113
        should_write_log = step % train_config['log_interval'] == 0
        if train_it is not None:
            feed_dict = next(train_it)
            if args.batch_per_run:                   # the reference's input node serves a new batch to each of the two sess.run (:46-50)
                feed_dict = dict(feed_dict, **{k + '_G': v for k, v in next(train_it).items() if k in ('image', 'future_image')})
        else:
            pair = synthetic_pair(batch_size, res=model.image_size, seed0=2 * (step * world + rank), seed1=2 * (step * world + rank) + 1)
            feed_dict = {k: torch.from_numpy(v).to(dev) for k, v in pair.items()}
            if args.batch_per_run:
                s2 = 10 ** 7 + 2 * (step * world + rank)
                pair = synthetic_pair(batch_size, res=model.image_size, seed0=s2, seed1=s2 + 1)
                feed_dict.update({k + '_G': torch.from_numpy(v).to(dev) for k, v in pair.items()})
        model.train_step(None, feed_dict, step, batch_size, should_write_log=should_write_log and rank == 0,
                         should_write_summary=False)
        if step % train_config['checkpoint_interval'] == 0 and rank == 0:
            model.save_checkpoint(None, step, fmt=args.ckpt_format)
        if step % train_config['test_interval'] == 0 and rank == 0:
            if test_it is not None:                  # reference train.py:97-110: one pass over the test subset
                results = [model.test_step(None, fd, step, i, fd['image'].shape[0]) for i, fd in enumerate(test_it())]
            else:
                tp = synthetic_pair(batch_size, res=model.image_size, seed0=10 ** 6, seed1=10 ** 6 + 1)
                results = [model.test_step(None, {k: torch.from_numpy(v).to(dev) for k, v in tp.items()}, step, 0, batch_size)]
            model.collect_test_results(results, step)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
