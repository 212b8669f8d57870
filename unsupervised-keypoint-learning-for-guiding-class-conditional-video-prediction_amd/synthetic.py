"""Synthetic Penn-shaped inputs: the output contract of the reference's ImagePairDataLoader
(data/image_pair_dataloader.py:38-48,65-70): float32 NHWC in [-1,1], keys 'image' and 'future_image'."""
import numpy as np


def synthetic_pair(batch, res=128, seed0=0, seed1=1):
    def one(seed):
        u = np.random.RandomState(seed).randint(0, 256, size=(batch, res, res, 3)).astype(np.float32)
        return (u / np.float32(255.0) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)   # map_fn: x/255*2-1
    return {'image': one(seed0), 'future_image': one(seed1)}
