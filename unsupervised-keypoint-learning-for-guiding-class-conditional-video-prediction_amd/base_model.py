"""Abstract model API with the reference's method names (reference: models/base_model.py:7-92).

``sess`` arguments are kept for signature parity and ignored (there is no session: kernels are launched eagerly on the
current HIP stream).  Checkpoints keep the reference's location ``<log_dir>/<name>/model.ckpt-<step>`` and variable
names, in an ``.npz`` container (default) or as a TensorFlow V2 checkpoint bundle (``tf_bundle.py``, SURVEY 8f-2;
``KPX_CKPT_FORMAT=tf``).
"""
import os
from abc import ABC, abstractmethod
from os import path as osp

import numpy as np


class BaseModel(ABC):
    name = 'base_model'
    trainable = True

    def __init__(self, is_training=True):
        super(BaseModel, self).__init__()
        self.is_training = is_training
        self.log_dir = None
        self.store = None

    @abstractmethod
    def build(self, inputs):
        raise NotImplementedError

    @abstractmethod
    def train_step(self, sess, feed_dict, step, batch_size, should_write_log=False, should_write_summary=False):
        raise NotImplementedError

    @abstractmethod
    def test_step(self, sess, feed_dict, step, test_idx, batch_size):
        raise NotImplementedError

    @abstractmethod
    def collect_test_results(self, results, step):
        raise NotImplementedError

    def initialize_loggers(self, log_dir, sess=None):
        """reference :62-75 (the TensorBoard FileWriters are out of scope; the checkpoint directory is created)."""
        self.log_dir = log_dir
        os.makedirs(osp.join(log_dir, self.__class__.name), exist_ok=True)

    def checkpoint_arrays(self):
        """Everything tf.train.Saver(tf.global_variables()) would hold (reference :74)."""
        return self.store.export_numpy(include_slots=self.is_training)

    def save_checkpoint(self, sess, step, fmt=None):
        """reference :77-81 -> <log_dir>/<name>/model.ckpt-<step>.  ``fmt`` (or $KPX_CKPT_FORMAT): 'npz' (default, one file) or
        'tf' = a TensorFlow V2 bundle (.index + .data-00000-of-00001 + ``checkpoint``) that tf.train.Saver can restore."""
        fmt = fmt or os.environ.get('KPX_CKPT_FORMAT', 'npz')
        if fmt not in ('npz', 'tf', 'bundle'):           # ('bundle' = 'tf'; anything else is a typo that must not silently write the other container)
            raise ValueError("checkpoint format %r: expected 'npz', 'tf' or 'bundle' (KPX_CKPT_FORMAT / --ckpt-format)" % (fmt,))
        prefix = osp.join(self.log_dir, self.__class__.name, 'model.ckpt-%d' % step)
        if fmt in ('tf', 'bundle'):
            from . import tf_bundle
            tf_bundle.write_bundle(prefix, self.checkpoint_arrays())
            return prefix
        np.savez(prefix + '.npz', **{k.replace('/', '|'): v for k, v in self.checkpoint_arrays().items()})
        return prefix + '.npz'

    def restore(self, sess, checkpoint_path):
        """reference :83-91: restore the intersection of checkpoint variables and model variables, by name.  Accepts the .npz
        container or the prefix of a TensorFlow V2 bundle (e.g. the published stage-1 / stage-2 checkpoints)."""
        from . import tf_bundle
        if tf_bundle.is_bundle(checkpoint_path):
            arrays = tf_bundle.read_bundle(checkpoint_path)
        else:
            data = np.load(checkpoint_path)
            arrays = {k.replace('|', '/'): data[k] for k in data.files}
        self.store.load_numpy(arrays, strict=False)
        self._restore_extra(arrays)
        return sorted(arrays)

    def _restore_extra(self, arrays):
        pass
