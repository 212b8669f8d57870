"""Inference-only key-point extractor (reference: models/keypoint_model.py) -- SURVEY 8f row 3, the first "next" row.

Runs ``pose_encoder`` with ``is_training=False`` (batch norm on the moving statistics, reference keypoint_model.py:48-50) over
whole videos and returns ``pts`` [B, T, K, 2].  The reference hard-codes T=663 and K=40 in a reshape (:52); here both follow
the input / config.  Frames are processed in slabs so a 663-frame video (9.7 GB of activations at once) stays bounded.
"""
import torch

from . import networks, ops, variables
from .base_model import BaseModel
from .variables import Sym


class KeypointModel(BaseModel):
    name = 'stage1'                      # reference :18

    def __init__(self, config, device='cuda', image_size=128, frames_per_launch=128, seed=1234):
        super(KeypointModel, self).__init__(False)
        self.n_points = config['model']['n_pts']
        self.log_dir = config['paths']['log_dir']
        self.image_size = image_size
        self.frames_per_launch = frames_per_launch
        self.device = ops.normalize_device(device)
        self.store = variables.VariableStore(device=self.device, seed=seed)

    def build(self, inputs=None):
        """Declares the pose_encoder variables (same scopes as stage 1, so DetectorTranslatorModel checkpoints restore by
        name intersection, reference base_model.py:83-91)."""
        r = self.image_size
        with variables.as_default(self.store):
            networks.pose_encoder(Sym(1, r, r, 3), self.n_points, False, final_res=r)
        self.store.materialise()

    def run(self, sess, feed_dict):
        """reference run (:60-61).  feed_dict: {'image': [B,T,H,W,3] in [-1,1], 'idx': [B], 'len': [B]}."""
        im = feed_dict['image']
        b, t = im.shape[0], im.shape[1]
        frames = im.reshape(b * t, self.image_size, self.image_size, 3)
        outs = []
        with variables.as_default(self.store), torch.no_grad():
            for s in range(0, b * t, self.frames_per_launch):
                outs.append(networks.pose_encoder(frames[s:s + self.frames_per_launch].contiguous(), self.n_points, False,
                                                  final_res=self.image_size))
        pts = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
        return {'pts': pts.reshape(b, t, self.n_points, 2), 'idx': feed_dict.get('idx'), 'len': feed_dict.get('len'), 'im': im}

    def train_step(self, sess, feed_dict, step, batch_size, should_write_log=False, should_write_summary=False):
        raise NotImplementedError          # reference :63-72: this model is not trainable

    def test_step(self, sess, feed_dict, step, test_idx, batch_size):
        raise NotImplementedError

    def collect_test_results(self, results, step):
        raise NotImplementedError
