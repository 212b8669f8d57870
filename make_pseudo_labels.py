#!/usr/bin/env python3
"""Pseudo-label writer with the reference's CLI (reference: make_pseudo_labels.py:12-105):

    python make_pseudo_labels.py --config configs/penn.yaml --checkpoint results/detector_translator/model.ckpt-N.npz [--synthetic V]

For every video writes ``<data_dir>/pseudo_labels/NNNN.npy`` = float32 [len, K, 2] key-points (reference :98-101).  Videos come
from the train and test subsets of ``paths.data_dir`` through kpx_amd.data.KeypointDataLoader (the reference's
data/keypoint_dataloader.py); ``--synthetic V`` feeds V random videos padded to 663 frames with the loader's output contract
({'image': [1,663,128,128,3] in [-1,1], 'idx', 'len'}) instead.
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
MAX_FRAMES = 663                       # reference data/keypoint_dataloader.py:13


def _save_output(keypoints_root_dir, outputs):
    """reference :98-101"""
    idx, n = int(outputs['idx'][0]), int(outputs['len'][0])
    np.save(osp.join(keypoints_root_dir, '{:04d}.npy'.format(idx)), outputs['pts'][0, :n].cpu().numpy())


def main(argv=None):
    from train import load_config
    parser = ArgumentParser()
    parser.add_argument('--config', type=str, required=True, help='path of the configuration file')
    parser.add_argument('--checkpoint', type=str, required=True, help='path of the pretrained keypoints detector')
    parser.add_argument('--synthetic', type=int, default=0, help='number of synthetic videos instead of paths.data_dir')
    parser.add_argument('--frames', type=int, default=MAX_FRAMES)
    args = parser.parse_args(argv)
    config = load_config(args.config)
    keypoints_root_dir = osp.join(config['paths']['data_dir'], 'pseudo_labels')
    os.makedirs(keypoints_root_dir, exist_ok=True)
    if not (osp.exists(args.checkpoint) or osp.exists(args.checkpoint + '.index')):      # .npz file or TensorFlow bundle prefix
        raise Exception('checkpoint not found at %s' % args.checkpoint)          # reference :31-32
    import kpx_amd
    dev = torch.device('cuda', 0)
    model = kpx_amd.KeypointModel(config, device=dev)
    print('model initialized')
    model.build(None)
    restored = model.restore(None, args.checkpoint)
    print('restored %d arrays' % len(restored))
    t0, frames = time.time(), 0
    if not args.synthetic:                       # reference :36-37, :79-95: every video of the train and the test subset
        data_dir = config['paths']['data_dir']
        for subset in ('train', 'test'):
            for v in kpx_amd.data.KeypointDataLoader(data_dir, subset).videos(dev):
                outputs = model.run(None, {'image': v['image'][None], 'idx': np.array([v['idx']]), 'len': np.array([v['len']])})
                _save_output(keypoints_root_dir, outputs)
                frames += v['len']
            print('iteration through %s set finished' % subset)
        torch.cuda.synchronize()
        print('%d frames, %.1f frames/sec' % (frames, frames / (time.time() - t0)))
        return
    for v in range(args.synthetic):
        rs = np.random.RandomState(v)
        n = int(rs.randint(args.frames // 4, args.frames + 1))
        im = np.zeros((1, args.frames, 128, 128, 3), np.float32)
        im[0, :n] = (rs.randint(0, 256, size=(n, 128, 128, 3)).astype(np.float32) / 255.0 * 2.0 - 1.0)
        outputs = model.run(None, {'image': torch.from_numpy(im).to(dev), 'idx': np.array([v]), 'len': np.array([n])})
        _save_output(keypoints_root_dir, outputs)
        frames += args.frames
    torch.cuda.synchronize()
    print('iteration through synthetic set finished: %d frames, %.1f frames/sec' % (frames, frames / (time.time() - t0)))


if __name__ == '__main__':
    main()
