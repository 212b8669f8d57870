#!/usr/bin/env python3
"""Evaluation entry point with the reference's CLI (reference: evaluate.py:13-163):

    python evaluate.py --config configs/penn.yaml --checkpoint_stage1 S1.npz --checkpoint_stage2 S2.npz [--save_dir results/eval]
                       [--synthetic N] [--no-save]

One source image -> 32 predicted frames per sample (FinalModel).  PNG writer semantics as the reference (:137-156).  The
test subset of ``paths.data_dir`` is read through kpx_amd.data.SequenceDataLoader (the reference's data/sequence_dataloader.py);
``--synthetic N`` feeds N random images with random one-hot action codes instead.  Checkpoints: the ``.npz`` containers written
by this repo or TensorFlow V2 bundle prefixes (TF variable names either way).
"""
import os
import sys
import time
from argparse import ArgumentParser
from os import path as osp

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _save_img(file_path, img, rescale=False):
    """reference :137-149"""
    from PIL import Image
    mode = None
    if img.shape[2] <= 2:
        img = np.squeeze(img, axis=2)
        mode = 'L'
    if rescale:
        img = 0.5 * (img + 1.0)
    img = (img * 255).astype(np.uint8)
    Image.fromarray(img, mode=mode).save(file_path)


def _save_img_sequence(output_dir, img_seq, rescale=False):
    """reference :151-156"""
    os.makedirs(output_dir, exist_ok=True)
    for i in range(img_seq.shape[0]):
        _save_img(osp.join(output_dir, '%06d.png' % i), img_seq[i], rescale=rescale)


def main(argv=None):
    from train import load_config
    parser = ArgumentParser()
    parser.add_argument('--config', type=str, required=True, help='path of the configuration file')
    parser.add_argument('--checkpoint_stage1', type=str, required=True, help='path of the stage1 checkpoint')
    parser.add_argument('--checkpoint_stage2', type=str, required=True, help='path of the stage2 checkpoint')
    parser.add_argument('--save_dir', type=str, required=False, help='root dir to save results', default='results/eval')
    parser.add_argument('--synthetic', type=int, default=0)
    parser.add_argument('--batch', type=int, default=8)             # reference :27
    parser.add_argument('--no-save', action='store_true')
    args = parser.parse_args(argv)
    config = load_config(args.config)
    for p in (args.checkpoint_stage1, args.checkpoint_stage2):       # reference :34-38
        if not (osp.exists(p) or osp.exists(p + '.index')):                 # .npz file or TensorFlow bundle prefix
            raise Exception('checkpoint not found at %s' % p)
    import kpx_amd
    dev = torch.device('cuda', 0)
    model = kpx_amd.FinalModel(config, device=dev)
    print('model initialized')
    model.build(None)
    model.restore(None, args.checkpoint_stage1)                      # reference :76-77: two partial restores by name
    model.restore(None, args.checkpoint_stage2)
    n_action = config['model']['n_action']
    sample_idx, frames, t0 = 0, 0, time.time()
    if not args.synthetic:                       # reference :43-51, :84-130: the test subset through the sequence loader
        loader = kpx_amd.data.SequenceDataLoader(config['paths']['data_dir'], 'test', n_points=config['model']['n_pts'], n_action=n_action,
                                                 random_order=False, randomness=False, with_image_seq=True)
        for batch in loader.batches(args.batch, dev, repeat=False, num_preprocess_threads=12):
            outputs = model.run(None, {'image': batch['image'], 'action_code': batch['action_code']})
            bsz = batch['image'].shape[0]
            frames += bsz * 32
            if not args.no_save:
                o = {k: v.cpu().numpy() for k, v in outputs.items() if isinstance(v, torch.Tensor)}
                real = batch['real_im_seq'].cpu().numpy()
                for batch_idx in range(bsz):
                    d = osp.join(args.save_dir, '%04d' % sample_idx)
                    os.makedirs(d, exist_ok=True)
                    _save_img(osp.join(d, 'input_im.png'), o['im'][batch_idx], rescale=True)
                    _save_img_sequence(osp.join(d, 'real_seq'), real[batch_idx], rescale=True)
                    _save_img_sequence(osp.join(d, 'pred_seq'), o['pred_im_seq'][batch_idx], rescale=True)
                    _save_img_sequence(osp.join(d, 'mask'), o['mask'][batch_idx], rescale=False)
                    _save_img_sequence(osp.join(d, 'crude'), o['pred_im_crude'][batch_idx], rescale=True)
                    sample_idx += 1
        torch.cuda.synchronize()
        print('iteration through test set finished: %d predicted frames, %.1f frames/sec' % (frames, frames / (time.time() - t0)))
        return
    for start in range(0, args.synthetic, args.batch):
        bsz = min(args.batch, args.synthetic - start)
        rs = np.random.RandomState(start)
        im = (rs.randint(0, 256, size=(bsz, 128, 128, 3)).astype(np.float32) / 255.0 * 2.0 - 1.0)
        act = np.eye(n_action, dtype=np.float32)[rs.randint(0, n_action, size=bsz)]
        outputs = model.run(None, {'image': torch.from_numpy(im).to(dev), 'action_code': torch.from_numpy(act).to(dev)})
        frames += bsz * 32
        if not args.no_save:
            o = {k: v.cpu().numpy() for k, v in outputs.items() if isinstance(v, torch.Tensor)}
            for batch_idx in range(bsz):
                d = osp.join(args.save_dir, '%04d' % sample_idx)
                os.makedirs(d, exist_ok=True)
                _save_img(osp.join(d, 'input_im.png'), o['im'][batch_idx], rescale=True)
                _save_img_sequence(osp.join(d, 'pred_seq'), o['pred_im_seq'][batch_idx], rescale=True)
                _save_img_sequence(osp.join(d, 'mask'), o['mask'][batch_idx], rescale=False)
                _save_img_sequence(osp.join(d, 'crude'), o['pred_im_crude'][batch_idx], rescale=True)
                sample_idx += 1
    torch.cuda.synchronize()
    print('iteration through synthetic set finished: %d predicted frames, %.1f frames/sec' % (frames, frames / (time.time() - t0)))


if __name__ == '__main__':
    main()
