#!/usr/bin/env python3
"""Generate tests/golden/image_pair_ref.npz.  Run in the build container only (needs /root/reference):

    python -B tests/golden/make_input_golden.py

Pins the input pipelines against the REFERENCE's own loaders, executed unmodified
(/root/reference/data/{image_pair,keypoint,sequence}_dataloader.py + utils/data.py) on a small synthetic
Penn-Action-shaped dataset:

* the dataset: three "videos" frames/0001..3 (landscape 144x108 x 40 frames, portrait 96x136 x 70, square 120x120 x 20 -- i.e.
  frame gaps 1, 2 and the "short video" branch of the sequence loader) drawn by this script (a smooth random background with a
  moving disc) plus random float32 pseudo_labels/*.npy key points; the encoded bytes are stored in the fixture so the test can
  rebuild the directory tree;
* KeypointDataLoader: SHA-256 of every video's centre-cropped frames, length and id; SequenceDataLoader: 12 augmented samples
  with the future-frame sequence and 3 sequential ones (key-point arrays in full, images as SHA-256);
* the expected samples: ``ImagePairDataLoader.sample_generator()`` of the reference after ``random.seed(S)`` /
  ``np.random.seed(S)`` -- 30 samples with random_order=True, randomness=True (rotation, random crop, flip, the ten
  filters) and the 3 sequential samples with both switched off.  Stored as the uint8 image (the reference yields
  uint8 / 255.0 in float64, checked here to be exactly that) -- full arrays for four samples, a SHA-256 of the bytes and
  an 8x8 block-mean signature for all of them.

TensorFlow is not installable here; the reference module only touches ``tf.float32`` (dtype table) and ``tf.name_scope``
(map_fn) outside of ``get_dataset``, so an empty stand-in module is enough: the pixel path is PIL / numpy / random only.
Nothing from /root/reference is copied: only inputs (JPEG bytes made here, seeds) and output arrays are stored.
"""
import hashlib
import io
import os
import random
import sys
import tempfile
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
SEED = 20190611
sys.dont_write_bytecode = True

VIDEOS = (('frames/0001', 144, 108, 40, 3), ('frames/0002', 96, 136, 70, 7), ('frames/0003', 120, 120, 20, 0))     # name, w, h, frames, action id
N_POINTS, N_ACTION = 5, 9


def make_dataset(root):
    """-> {relative path: bytes}; also written under root"""
    files = {}
    rs = np.random.RandomState(7)
    for name, w, h, n, act in VIDEOS:
        os.makedirs(os.path.join(root, name), exist_ok=True)
        os.makedirs(os.path.join(root, 'pseudo_labels'), exist_ok=True)
        kp = (rs.rand(n, N_POINTS, 2) * 1.6 - 0.8).astype(np.float32)         # what make_pseudo_labels.py writes: [frames, K, 2] float32
        buf = io.BytesIO(); np.save(buf, kp)
        rel = name.replace('frames', 'pseudo_labels') + '.npy'
        files[rel] = buf.getvalue()
        with open(os.path.join(root, rel), 'wb') as f:
            f.write(files[rel])
        low = rs.rand(h // 12 + 2, w // 12 + 2, 3)
        bg = np.asarray(Image.fromarray((low * 255).astype(np.uint8)).resize((w, h), Image.BILINEAR)).astype(np.float32)
        yy, xx = np.mgrid[0:h, 0:w]
        for i in range(n):
            cx, cy = w * (0.2 + 0.6 * i / n), h * (0.5 + 0.25 * np.sin(i * 0.7))
            disc = ((xx - cx) ** 2 + (yy - cy) ** 2) < (min(w, h) * 0.12) ** 2
            img = bg.copy()
            img[disc] = (250, 40 + 10 * i, 30)
            buf = io.BytesIO()
            Image.fromarray(img.astype(np.uint8)).save(buf, format='JPEG', quality=85)
            rel = '%s/%06d.jpg' % (name, i + 1)
            files[rel] = buf.getvalue()
            with open(os.path.join(root, rel), 'wb') as f:
                f.write(files[rel])
    listing = ''.join('%s %d\n' % (name, act) for name, _, _, _, act in VIDEOS)
    with open(os.path.join(root, 'train_set.txt'), 'w') as f:
        f.write(listing)
    return files, listing


def reference_loader_classes():
    tf = types.ModuleType('tensorflow')
    tf.float32, tf.int16 = 'float32', 'int16'
    sys.modules['tensorflow'] = tf
    sys.path.insert(0, REF)
    try:
        from data import ImagePairDataLoader, KeypointDataLoader, SequenceDataLoader
    finally:
        sys.path.remove(REF)
    return ImagePairDataLoader, KeypointDataLoader, SequenceDataLoader


def signature(u8):
    return u8.reshape(8, 16, 8, 16, 3).astype(np.float64).mean(axis=(1, 3))


def collect(loader, n_samples):
    out = []
    while len(out) < n_samples:
        for s in loader.sample_generator():
            pair = []
            for key in ('image', 'future_image'):
                x = s[key]
                u8 = np.rint(x * 255.0).astype(np.uint8)
                assert x.dtype == np.float64 and x.shape == (128, 128, 3) and np.array_equal(u8 / 255.0, x)
                pair.append(u8)
            out.append(np.stack(pair))
            if len(out) == n_samples:
                break
    return np.stack(out)


def main():
    ref_cls, ref_kp_cls, ref_seq_cls = reference_loader_classes()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    extra = {}
    with tempfile.TemporaryDirectory() as root:
        files, listing = make_dataset(root)
        random.seed(SEED); np.random.seed(SEED)
        rand = collect(ref_cls(root, 'train', random_order=True, randomness=True), 30)
        random.seed(SEED); np.random.seed(SEED)
        seq = collect(ref_cls(root, 'train', random_order=False, randomness=False), 3)
        # whole-video loader (data/keypoint_dataloader.py): SHA-256 of the real frames as uint8, the padding is checked to be zeros
        kp_sha, kp_len, kp_idx = [], [], []
        for s in ref_kp_cls(root, 'train').sample_generator():
            x, n = s['image'], s['len']
            assert x.shape == (663, 128, 128, 3) and x.dtype == np.float64 and not x[n:].any()
            u8 = np.rint(x[:n] * 255.0).astype(np.uint8)
            assert np.array_equal(u8 / 255.0, x[:n])
            kp_sha.append(sha(u8)); kp_len.append(n); kp_idx.append(s['idx'])
        extra.update(kp_sha256=np.array(kp_sha), kp_len=np.array(kp_len), kp_idx=np.array(kp_idx))
        # sequence loader (data/sequence_dataloader.py): 12 random augmented samples with the future frames, 3 plain sequential ones
        for tag, kw, n_s in (('sq_rand', dict(with_image_seq=True, random_order=True, randomness=True), 12),
                             ('sq_seq', dict(with_image_seq=False, random_order=False, randomness=False), 3)):
            random.seed(SEED + 1); np.random.seed(SEED + 1)
            loader = ref_seq_cls(root, 'train', N_POINTS, N_ACTION, **kw)
            got = []
            while len(got) < n_s:
                for s in loader.sample_generator():
                    got.append(s)
                    if len(got) == n_s:
                        break
            for key in ('keypoints', 'real_seq', 'action_code'):
                extra['%s_%s' % (tag, key)] = np.stack([g[key] for g in got])
            ims = [np.rint(g['image'] * 255.0).astype(np.uint8) for g in got]
            assert all(np.array_equal(u / 255.0, g['image']) for u, g in zip(ims, got))
            extra['%s_image_sha256' % tag] = np.array([sha(u) for u in ims])
            if kw['with_image_seq']:
                sq = [np.rint(g['real_im_seq'] * 255.0).astype(np.uint8) for g in got]
                assert all(np.array_equal(u / 255.0, g['real_im_seq']) and u.shape == (32, 128, 128, 3) for u, g in zip(sq, got))
                extra['%s_im_seq_sha256' % tag] = np.array([sha(u) for u in sq])
    out = {'seed': np.int64(SEED), 'listing': np.frombuffer(listing.encode(), dtype=np.uint8),
           'file_names': np.array(sorted(files)),
           'rand_sha256': np.array([hashlib.sha256(a.tobytes()).hexdigest() for a in rand]),
           'rand_sig': np.stack([np.stack([signature(f) for f in a]) for a in rand]),
           'rand_full_idx': np.array([0, 7, 18, 29]), 'rand_full': rand[[0, 7, 18, 29]],
           'seq_sha256': np.array([hashlib.sha256(a.tobytes()).hexdigest() for a in seq]), 'seq_full': seq}
    out.update(extra)
    out['n_points'], out['n_action'] = np.int64(N_POINTS), np.int64(N_ACTION)
    for k, name in enumerate(sorted(files)):
        out['file_%03d' % k] = np.frombuffer(files[name], dtype=np.uint8)
    path = os.path.join(HERE, 'image_pair_ref.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB;', len(files), 'files,', len(rand), '+', len(seq), 'pair samples')


if __name__ == '__main__':
    main()
