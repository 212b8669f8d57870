"""Stage-2 model: class-conditional sequence VAE-GAN over key-point sequences (reference: models/motion_generator_model.py).

vae_encoder (2-layer LSTM over the 32 future key-point sets) -> z = mu + stddev * eps -> vae_decoder (2-layer LSTM, 32 steps)
-> seq_discr (2-layer LSTM(1024)).  Losses (reference :257-308): D = xent(D(real), 1) + xent(D(fake), 0);
G = KL + 1000 * mean|pred - real| + xent(D(pred), 1).  A step is the reference's D-run followed by its G-run (:80-104); each
run evaluates the whole graph with its own tf.random_normal draw, so two noise tensors are used per step (injectable for tests).
The matmuls run on the implicit-GEMM conv kernels (1x1 "convolutions" over [B,1,1,In]), the LSTM gate math, KL / sampling, L1,
cross entropy and Adam are HIP kernels; the weight gradient of an LSTM layer is one GEMM over all T*B rows.
"""
import logging
import time
from datetime import datetime

import numpy as np
import torch

from . import networks, ops, variables
from .base_model import BaseModel
from .variables import Sym

log = logging.getLogger('kpx')
N_FUTURE_FRAMES = 32       # reference :9


class MotionGeneratorModel(BaseModel):
    name = 'motion_generator'          # reference :16 (checkpoint sub-directory)

    def __init__(self, config, global_step=None, is_training=True, device='cuda', process_group=None, seed=777,
                 discr_cells=(1024, 1024)):
        super(MotionGeneratorModel, self).__init__(is_training)
        train_config, model_config, paths_config = config['training'], config['model'], config['paths']
        self.lr = train_config['lr'] if self.is_training else None
        self.batch_size = train_config['batch_size']
        self.log_dir = paths_config['log_dir']
        self.n_points = model_config['n_pts']
        self.n_action = model_config.get('n_action', 9)
        self.cell_info = list(model_config['cell_info'])
        self.vae_dim = model_config['vae_dim']
        self.discr_cells = tuple(discr_cells)          # literal [1024, 1024] in the reference (networks/__init__.py:134)
        self.device = ops.normalize_device(device)
        self.global_step = int(global_step or 0)
        self.process_group = process_group
        self.world_size = torch.distributed.get_world_size(process_group) if (
            torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        self.store = variables.VariableStore(device=self.device, seed=seed)
        self.beta1, self.beta2, self.adam_eps = np.float32(0.5), np.float32(0.999), np.float32(1e-8)
        self.beta_power = {'D': [np.float32(0.5), np.float32(0.999)], 'G': [np.float32(0.5), np.float32(0.999)]}
        self.last = {}

    def build(self, inputs=None):
        b, k2 = 2, self.n_points * 2
        with variables.as_default(self.store):
            networks.vae_encoder(Sym(b, N_FUTURE_FRAMES, k2), Sym(b, k2), Sym(b, self.n_action), self.cell_info, self.vae_dim)
            networks.vae_decoder(Sym(b, self.vae_dim), Sym(b, k2), Sym(b, self.n_action), self.cell_info, self.vae_dim, self.n_points)
            networks.seq_discr(Sym(b, N_FUTURE_FRAMES, k2), self.discr_cells)
        self.store.materialise()
        if self.device.type == 'cuda':
            self._e0 = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float32, device=self.device)
            self._one = torch.ones(1, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------------------------------------ graph pieces
    def _generate(self, keypoints, real_seq, action_code, eps):
        """reference _define_forward_pass, training branch (:137-150) -> (pred_seq [B,32,2K], kl [1])"""
        b = keypoints.shape[0]
        first_pt = keypoints.reshape(b, self.n_points * 2)
        real = real_seq.reshape(b, N_FUTURE_FRAMES, self.n_points * 2)
        logit = networks.vae_encoder(real, first_pt, action_code, self.cell_info, self.vae_dim)
        z, kl = ops.vae_sample_kl(logit, eps)
        pred = networks.vae_decoder_train(z, first_pt, action_code, self.cell_info, self.vae_dim, self.n_points)
        return pred, kl, real

    def sample(self, keypoints, action_code, z=None):
        """Inference branch (:151-159): z ~ N(0,1) -> pred_seq."""
        b = keypoints.shape[0]
        if z is None:
            z = torch.randn(b, self.vae_dim, device=self.device)
        with variables.as_default(self.store), torch.no_grad():
            return networks.vae_decoder(z, keypoints.reshape(b, self.n_points * 2).contiguous(), action_code, self.cell_info, self.vae_dim,
                                        self.n_points)

    def _loss_D(self, pred, real):
        """reference _compute_loss_D (:257-270); real and fake through seq_discr as one batch -> [loss_D, D_real, D_fake]"""
        n = real.shape[0]
        logits = networks.seq_discr(torch.cat([real, pred], dim=0), self.discr_cells)
        return ops.sigmoid_xent(logits, n, 1.0, n, 0.0)

    def _loss_G(self, pred, real, kl):
        """reference _compute_loss_G (:272-308) -> (recon [1], kl [1], adv [3]) with the discriminator weights as constants"""
        recon = ops.l1_mean(pred, real, 1000.0)
        with self.store.freeze('seq_discr'):
            logits = networks.seq_discr(pred, self.discr_cells)
        return recon, kl, ops.sigmoid_xent(logits, logits.numel(), 1.0)

    def current_lr(self):
        p = np.float32(self.global_step) / np.float32(self.lr['step'])
        return np.float32(np.float32(self.lr['start_val']) * np.power(np.float32(self.lr['decay']), p, dtype=np.float32))

    def _apply_adam(self, which, lr):
        bucket = self.store.buckets[which]
        ops.join_side_stream(self.device)
        if self.world_size > 1:                      # data parallel: one all-reduce per flat bucket, 1/world folded into Adam
            torch.distributed.all_reduce(bucket.grads, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
        b1p, b2p = self.beta_power[which]
        alpha = np.float32(np.float32(lr) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p))
        ops.adam_tf_flat_(bucket.params, bucket.grads, bucket.m, bucket.v, alpha, self.beta1, self.beta2, self.adam_eps,
                          gscale=1.0 / self.world_size)
        self.store.touch()
        self.beta_power[which] = [np.float32(b1p * self.beta1), np.float32(b2p * self.beta2)]

    def _noise(self, feed_dict, key, b):
        eps = feed_dict.get(key)
        return eps if eps is not None else torch.randn(b, self.vae_dim, device=self.device)

    # ------------------------------------------------------------------------------------------------ reference surface
    def train_step(self, sess, feed_dict, step, batch_size, should_write_log=False, should_write_summary=False):
        """reference train_step (:68-104).  feed_dict: 'keypoints' [B,K,2], 'real_seq' [B,32,K,2], 'action_code' [B,A]
        (the SequenceDataLoader batch); optional 'eps_D' / 'eps_G' [B,vae_dim] pin the two random_normal draws."""
        kp, rs, ac = feed_dict['keypoints'], feed_dict['real_seq'], feed_dict['action_code']
        b = kp.shape[0]
        start_time = time.time()
        lr = self.current_lr()
        with variables.as_default(self.store):
            # ---- D run (:80)
            with self.store.freeze('vae_encoder', 'vae_decoder'):
                pred, _, real = self._generate(kp, rs, ac, self._noise(feed_dict, 'eps_D', b))
            d_losses = self._loss_D(pred.detach(), real)
            ops.begin_backward()
            torch.autograd.backward([d_losses], [self._e0])
            self._apply_adam('D', lr)
            # ---- G run (:81): new forward, updated discriminator
            pred, kl, real = self._generate(kp, rs, ac, self._noise(feed_dict, 'eps_G', b))
            recon, kl, adv = self._loss_G(pred, real, kl)
            ops.begin_backward()
            torch.autograd.backward([recon, kl, adv], [self._one, self._one, self._e0])
            self._apply_adam('G', lr)
        self.global_step += 1
        self.last = dict(d_losses=d_losses.detach(), recon=recon.detach(), kl=kl.detach(), adv=adv.detach(), lr=float(lr), pred_seq=pred.detach())
        if should_write_log:
            v = self.loss_values()
            duration = time.time() - start_time
            log.info('%s: step %d, loss_D = %.4f, loss_G = %.4f (%.1f examples/sec) %.3f sec/batch',
                     datetime.now(), step, v['loss_D'], v['loss_G'], batch_size / float(duration), duration)

    def loss_values(self):
        d = self.last['d_losses'].cpu().numpy()
        recon, kl, adv = float(self.last['recon'].cpu()[0]), float(self.last['kl'].cpu()[0]), float(self.last['adv'].cpu()[0])
        return dict(loss_D=float(d[0]), loss_D_real=float(d[1]), loss_D_fake=float(d[2]), loss_G_recon=recon, loss_G_kl=kl,
                    loss_G_adv=adv, loss_G=kl + recon + adv, lr=self.last['lr'])

    def test_step(self, sess, feed_dict, step, test_idx, batch_size):
        """reference test_step (:106-117): both losses of one forward, no updates."""
        kp, rs, ac = feed_dict['keypoints'], feed_dict['real_seq'], feed_dict['action_code']
        start_time = time.time()
        with variables.as_default(self.store), torch.no_grad():
            pred, kl, real = self._generate(kp, rs, ac, self._noise(feed_dict, 'eps_G', kp.shape[0]))
            d = self._loss_D(pred, real)
            recon, kl, adv = self._loss_G(pred, real, kl)
        loss_g = float(recon.cpu()[0]) + float(kl.cpu()[0]) + float(adv.cpu()[0])
        return float(d.cpu()[0]), loss_g, time.time() - start_time, batch_size

    def collect_test_results(self, results, step):
        """reference collect_test_results (:119-135)."""
        average_loss_D = sum(x[0] for x in results) / len(results)
        average_loss_G = sum(x[1] for x in results) / len(results)
        total_duration = sum(x[2] for x in results)
        log.info('test: %s: step %d, loss_D = %.4f, loss_G = %.4f (%.1f examples/sec) %.3f sec/batch', datetime.now(), step,
                 average_loss_D, average_loss_G, sum(x[3] for x in results) / total_duration, total_duration / len(results))
        return average_loss_D, average_loss_G

    def checkpoint_arrays(self):
        arrays = self.store.export_numpy(include_slots=self.is_training)
        arrays['global_step'] = np.int32(self.global_step)   # tf.Variable(0) in train.py:30 is int32
        if self.is_training:
            arrays.update({'beta1_power': self.beta_power['D'][0], 'beta2_power': self.beta_power['D'][1],
                           'beta1_power_1': self.beta_power['G'][0], 'beta2_power_1': self.beta_power['G'][1]})
        return arrays

    def _restore_extra(self, arrays):
        if 'global_step' in arrays:
            self.global_step = int(arrays['global_step'])
        for which, suffix in (('D', ''), ('G', '_1')):
            if 'beta1_power' + suffix in arrays:
                self.beta_power[which] = [np.float32(arrays['beta1_power' + suffix]), np.float32(arrays['beta2_power' + suffix])]
