#!/usr/bin/env python3
"""Generate tests/golden/image_pair_ref.npz.  Run in the build container only (needs /root/reference):

    python -B tests/golden/make_input_golden.py

Pins the image-pair input pipeline against the REFERENCE's own loader, executed unmodified
(/root/reference/data/image_pair_dataloader.py + utils/data.py) on a small synthetic Penn-Action-shaped dataset:

* the dataset: three "videos" (landscape 176x132, portrait 120x168, square 150x150) of 11-14 JPEG frames drawn by this
  script (a smooth random background with a moving disc); the encoded JPEG bytes are stored in the fixture so the test
  can rebuild the directory tree;
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

VIDEOS = (('0001', 176, 132, 12, 3), ('0002', 120, 168, 14, 7), ('0003', 150, 150, 11, 0))     # name, w, h, frames, action id


def make_dataset(root):
    """-> {relative path: bytes}; also written under root"""
    files = {}
    rs = np.random.RandomState(7)
    for name, w, h, n, act in VIDEOS:
        os.makedirs(os.path.join(root, name), exist_ok=True)
        low = rs.rand(h // 12 + 2, w // 12 + 2, 3)
        bg = np.asarray(Image.fromarray((low * 255).astype(np.uint8)).resize((w, h), Image.BILINEAR)).astype(np.float32)
        yy, xx = np.mgrid[0:h, 0:w]
        for i in range(n):
            cx, cy = w * (0.2 + 0.6 * i / n), h * (0.5 + 0.25 * np.sin(i * 0.7))
            disc = ((xx - cx) ** 2 + (yy - cy) ** 2) < (min(w, h) * 0.12) ** 2
            img = bg.copy()
            img[disc] = (250, 40 + 10 * i, 30)
            buf = io.BytesIO()
            Image.fromarray(img.astype(np.uint8)).save(buf, format='JPEG', quality=90)
            rel = '%s/%06d.jpg' % (name, i + 1)
            files[rel] = buf.getvalue()
            with open(os.path.join(root, rel), 'wb') as f:
                f.write(files[rel])
    listing = ''.join('%s %d\n' % (name, act) for name, _, _, _, act in VIDEOS)
    with open(os.path.join(root, 'train_set.txt'), 'w') as f:
        f.write(listing)
    return files, listing


def reference_loader_class():
    tf = types.ModuleType('tensorflow')
    tf.float32 = 'float32'
    sys.modules['tensorflow'] = tf
    sys.path.insert(0, REF)
    try:
        from data.image_pair_dataloader import ImagePairDataLoader
    finally:
        sys.path.remove(REF)
    return ImagePairDataLoader


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
    ref_cls = reference_loader_class()
    with tempfile.TemporaryDirectory() as root:
        files, listing = make_dataset(root)
        random.seed(SEED); np.random.seed(SEED)
        rand = collect(ref_cls(root, 'train', random_order=True, randomness=True), 30)
        random.seed(SEED); np.random.seed(SEED)
        seq = collect(ref_cls(root, 'train', random_order=False, randomness=False), 3)
    out = {'seed': np.int64(SEED), 'listing': np.frombuffer(listing.encode(), dtype=np.uint8),
           'file_names': np.array(sorted(files)),
           'rand_sha256': np.array([hashlib.sha256(a.tobytes()).hexdigest() for a in rand]),
           'rand_sig': np.stack([np.stack([signature(f) for f in a]) for a in rand]),
           'rand_full_idx': np.array([0, 7, 18, 29]), 'rand_full': rand[[0, 7, 18, 29]],
           'seq_sha256': np.array([hashlib.sha256(a.tobytes()).hexdigest() for a in seq]), 'seq_full': seq}
    for k, name in enumerate(sorted(files)):
        out['file_%03d' % k] = np.frombuffer(files[name], dtype=np.uint8)
    path = os.path.join(HERE, 'image_pair_ref.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB;', len(files), 'jpeg frames,', len(rand), '+', len(seq), 'samples')


if __name__ == '__main__':
    main()
