"""Stage-1 model: key-point detector + translator + GAN / perceptual losses
(reference: models/detector_translator_model.py).

The step is the reference's D-run followed by its G-run (train_step :79-117) on ONE batch, restructured so that the
generator forward -- identical in both runs because the D update does not touch generator variables -- is computed once
(SURVEY 8d "restructured step"); the discriminator is re-run with its UPDATED weights for the generator's adversarial
term exactly as the second sess.run would.  Data parallelism: one process per GPU, per-replica BN statistics, one
all-reduce (RCCL) per flat gradient bucket per optimiser update, 1/world scaling inside the fused Adam kernel.
"""
import contextlib
import logging
import os
import time
from datetime import datetime

import numpy as np
import torch

from . import model_utils, networks, ops, variables
from .base_model import BaseModel
from .variables import Sym
from .vgg import Vgg19

# the discriminator update on an auxiliary HIP stream beside the VGG19 forward of the G run; the image encoder (forward and backward) beside the
# key-point detector; the G run's adversarial branch on that stream as well.  Module constants: the three-stream == one-stream bit-identity
# test sets them to False.  (Measured alternative, deleted: the perceptual VGG19 chain on the auxiliary stream instead, +0.9 ms.)
AUX_STREAM = True
AUX_STREAM_FWD = True
AUX_STREAM_ADV = True

# The whole step -- ~700 launches on three streams -- as ONE HIP graph, captured from the second call on a given input shape and replayed
# afterwards (KPX_GRAPH=0: every step is enqueued from Python).  Single-process steps on the shared batch only; see _train_step_graphed.
GRAPH = os.environ.get('KPX_GRAPH', '1') != '0'
# Data parallel, optional: exchange the named flat gradient buckets as bf16 (half the bytes on xGMI; the 178.9 MB discriminator bucket is the
# candidate).  OFF by default -- it rounds every gradient element to 8 mantissa bits before the sum, which the fp32 configuration must not do;
# e.g. KPX_DP_BF16=D.  Parameters, Adam moments and the local gradients stay fp32.
DP_BF16_BUCKETS = tuple(b for b in os.environ.get('KPX_DP_BF16', '').split(',') if b)


class _Bf16Exchange:
    """Handle of a bf16 gradient exchange: ``wait()`` orders the collective before the caller's stream and widens the sums back into the
    fp32 gradient buffer (the same interface as the work handle of an asynchronous all-reduce)."""

    def __init__(self, work, buf, grads):
        self.work, self.buf, self.grads = work, buf, grads

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.grads.copy_(self.buf)
# Data parallel, how the step reaches the GPU (KPX_DP_GRAPH):
#   'segments' (default on every backend with more than one rank) four captured segments replayed around the two collectives, which are
#              enqueued from Python between them (_train_step_dp): +1.2 ms per step, host work 1.3 ms.  The form two-rank runs have proven
#              (gloo tests: replicas bit-identical to the eager and inline forms);
#   'one'      ONE captured graph: the single-GPU step unchanged, with the two all-reduces issued synchronously (async_op=False) on the stream
#              they belong to -- ProcessGroupNCCL (torch 2.10) runs such a collective on the caller's CURRENT stream, so nothing forks from the
#              auxiliary stream (ops.py: stream discipline) and the collectives are two more nodes of the graph.  Measured at one rank with both
#              collectives kept in (KPX_DP_FORCE_EXCHANGE=1): 22.76 ms against 22.74-22.78 ms for the non-distributed graph.  It has only ever
#              run with a ONE-rank RCCL communicator, so it is opt-in (KPX_DP_GRAPH=one; a one-rank group takes it by default) until a run
#              with two or more ranks has shown the replicas bit-identical under it; a failed capture falls back to 'segments' in process,
#              a capture that HANGS is bench.py's supervisor's business (fresh processes with the next form);
#   'inline'   round 3's form: one eager pass with the collectives inline (also taken by separate-batch steps and with KPX_GRAPH=0 + this value).
DP_GRAPH = os.environ.get('KPX_DP_GRAPH', '')
GRAPH_WARMUP_STEPS = 1          # eager steps before the capture: they create every lazily allocated scratch buffer and kernel attribute

log = logging.getLogger('kpx')


class DetectorTranslatorModel(BaseModel):
    name = 'detector_translator'       # reference :15 (checkpoint sub-directory)

    def __init__(self, config, global_step=None, is_training=True, device='cuda', vgg=None, process_group=None,
                 image_size=128, seed=1234):
        super(DetectorTranslatorModel, self).__init__(is_training)
        train_config, model_config, paths_config = config['training'], config['model'], config['paths']
        self.lr = train_config['lr'] if self.is_training else None      # reference :24-27
        self.batch_size = train_config['batch_size']
        self.n_points = model_config['n_pts']
        self.log_dir = paths_config['log_dir']
        self.vgg19_path = paths_config.get('vggnet')
        self.image_size = image_size            # literal 128 in the reference loaders (image_pair_dataloader.py:13)
        self.heat_size = image_size // 4        # literal [32, 32] (:168-169)
        self.device = ops.normalize_device(device)
        self.global_step = int(global_step or 0)
        self.process_group = process_group
        # a process group of more than one rank means: exchange through it.  A group of ONE rank (`torch.distributed.run --nproc-per-node 1`)
        # has nothing to exchange -- a sum over one rank is the identity -- and runs the plain single-GPU step (graph replay included);
        # KPX_DP_FORCE_EXCHANGE=1 keeps the two collectives in, which is how the cost of the RCCL path alone is measured on one GPU
        # (+1.6 ms per step)
        in_group = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world_size = torch.distributed.get_world_size(process_group) if in_group else 1
        self.distributed = in_group and (self.world_size > 1 or os.environ.get('KPX_DP_FORCE_EXCHANGE', '0') != '0')
        backend = torch.distributed.get_backend(process_group) if in_group else ''
        if DP_GRAPH not in ('', 'one', 'segments', 'inline'):
            raise ValueError("KPX_DP_GRAPH must be 'one', 'segments' or 'inline' (got %r)" % DP_GRAPH)
        self.dp_graph = DP_GRAPH or ('one' if (backend == 'nccl' and self.world_size == 1) else 'segments')
        self.store = variables.VariableStore(device=self.device, seed=seed)
        self.vgg = vgg
        # Adam state (two optimisers, reference :198 and :201): fp32 beta powers like TF's beta{1,2}_power variables
        self.beta1, self.beta2, self.adam_eps = np.float32(0.5), np.float32(0.999), np.float32(1e-8)
        self.beta_power = {'D': [np.float32(0.5), np.float32(0.999)], 'G': [np.float32(0.5), np.float32(0.999)]}
        self.last = {}
        self._graphs = {}               # input shape -> captured step (graph, static inputs, outputs)
        self._eager_steps = {}          # input shape -> eager steps taken so far
        self._capturing = False
        self._graph_failed = False
        self._alpha_dev = None          # {'D','G'} -> [1] device tensors holding Adam's step size for the captured launches

    # ------------------------------------------------------------------------------------------------ build
    def build(self, inputs=None):
        """Declare every variable with a shape-only pass over the same network code (the analogue of TF graph
        construction, reference :68-77), then allocate the flat buckets on the device."""
        b, r = 2, self.image_size
        with variables.as_default(self.store):
            self._define_forward_pass(Sym(b, r, r, 3), Sym(b, r, r, 3))
            networks.img_discr(Sym(b, r, r, 3))
        self.store.materialise()
        dev = self.device
        if dev.type == 'cuda':
            self._e0 = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float32, device=dev)
            self._one = torch.ones(1, dtype=torch.float32, device=dev)
        if self.is_training and self.vgg is None and dev.type == 'cuda':
            self.vgg = Vgg19(self.vgg19_path, device=dev)             # raises like vgg.py:9-10 when the file is missing
        if self.distributed and self.world_size > 1:
            self.broadcast_parameters(0)

    # ------------------------------------------------------------------------------------------------ forward
    def _define_forward_pass(self, im, future_im, with_vis_maps=False, update_moving=True):
        """reference _define_forward_pass (:160-184)."""
        train = self.is_training
        sym = variables.is_sym(im)
        b = im.shape[0]
        # :165 -- the image encoder (half-batch, small launches) is independent of the key-point detector: on the auxiliary stream beside it.
        # The Winograd filter forms are re-derived (one batched launch) BEFORE the fork: both branches read them.
        aux = self._aux_stream() if (AUX_STREAM_FWD and not sym and im.is_cuda) else None
        if aux is not None:
            bank = getattr(self.store, 'filter_bank', None)
            if bank is not None:
                bank.ensure_fresh()
            aux.wait_stream(torch.cuda.current_stream(im.device))
        with (torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext()):
            embeddings = networks.image_encoder(im, train, update_moving=update_moving)
        # :166-167 -- the two weight-sharing pose_encoder calls as one batched launch, BN statistics per call
        both = Sym(2 * b, *im.shape[1:]) if sym else ops.concat_batch(im, future_im)
        # (the reference also fetches the head's logits here and never uses them: not requesting them lets the 1x1 head fold into
        # the key-point head, networks.FUSE_KEYPOINT_HEAD)
        pts = networks.pose_encoder(both, self.n_points, train, final_res=self.image_size, bn_groups=2,
                                    update_moving=update_moving)
        if sym:
            joint = Sym(b, self.heat_size, self.heat_size, (embeddings[-2].shape[-1] + 2 * self.n_points + 3) // 4 * 4)
            cur_pt = fut_pt = None
        else:
            cur_pt, fut_pt = pts[:b], pts[b:]
            if aux is not None:
                torch.cuda.current_stream(im.device).wait_stream(aux)
            joint = ops.joint_embedding(embeddings[-2], cur_pt, fut_pt)                    # :168-170
        raw4 = networks.translator(joint, train, final_res=self.image_size,
                                   cin=embeddings[-2].shape[-1] + 2 * self.n_points,
                                   update_moving=update_moving)                            # :173
        if sym:
            return None
        final_output, crude_output, mask = ops.head_blend(im, raw4)                        # :174
        out = dict(final_output=final_output, crude_output=crude_output, mask=mask,
                   current_points=cur_pt, future_points=fut_pt)
        if with_vis_maps:                                                                  # :176-177
            hw = [self.image_size, self.image_size]
            out['current_keypoints_map'] = model_utils.get_gaussian_maps(cur_pt.detach(), hw)
            out['future_keypoints_map'] = model_utils.get_gaussian_maps(fut_pt.detach(), hw)
        return out

    def forward(self, im, future_im, with_vis_maps=True):
        """Fetch the model outputs only: like a sess.run of the output tensors, this runs no UPDATE_OPS (they ride on train_op_G,
        reference :199-202), so the BN moving statistics stay untouched."""
        with variables.as_default(self.store), torch.no_grad():
            return self._define_forward_pass(im, future_im, with_vis_maps=with_vis_maps, update_moving=False)

    # ------------------------------------------------------------------------------------------------ losses
    def _loss_D(self, future_im_pred, future_im):
        """reference _compute_loss_D (:246-259); real and fake go through img_discr as one batch."""
        n = future_im.shape[0]
        logits = networks.img_discr(ops.concat_batch(future_im, future_im_pred))
        per = logits.numel() // (2 * n)
        return ops.sigmoid_xent(logits, n * per, 1.0, n * per, 0.0)       # [loss_D, D_real, D_fake]

    def _loss_G_recon(self, future_im_pred, future_im):
        """reference _compute_loss_G (:261-263): VGG19 perceptual term (independent of the discriminator)."""
        return self.vgg.perceptual_loss(future_im, future_im_pred)         # [1]

    def _loss_G_adv(self, future_im_pred):
        """reference _compute_loss_G (:264-267): adversarial term vs ones, discriminator weights as constants."""
        with self.store.freeze('img_discr'):
            logits = networks.img_discr(future_im_pred)
        return ops.sigmoid_xent(logits, logits.numel(), 1.0)               # [adv, adv, 0]

    def _loss_G(self, future_im_pred, future_im):
        return self._loss_G_recon(future_im_pred, future_im), self._loss_G_adv(future_im_pred)

    def current_lr(self):
        """tf.train.exponential_decay, non-staircase, fp32 (reference :193-195)."""
        p = np.float32(self.global_step) / np.float32(self.lr['step'])
        return np.float32(np.float32(self.lr['start_val']) * np.power(np.float32(self.lr['decay']), p, dtype=np.float32))

    def exchange_gradients(self, which, async_op=False):
        """Data-parallel exchange: ONE all-reduce(sum) of the bucket's flat fp32 gradient buffer (RCCL over xGMI on the
        GPUs; gloo in the CPU tests).  The 1/world scaling happens inside the fused Adam kernel.  With ``async_op`` the
        collective runs on RCCL's own stream and the returned handle is waited for just before the Adam update, so the
        178.9 MB discriminator exchange overlaps the VGG19 forward of the generator's perceptual loss."""
        bucket = self.store.buckets[which]
        ops.join_side_stream(self.device)       # weight gradients are written on the side stream
        if self.distributed:
            if which in DP_BF16_BUCKETS:
                bufs = self.__dict__.setdefault('_bf16_exchange_bufs', {})
                buf = bufs.get(which)
                if buf is None:
                    buf = bufs[which] = torch.empty(bucket.grads.numel(), dtype=torch.bfloat16, device=bucket.grads.device)
                buf.copy_(bucket.grads)              # round to nearest even, like every fp32 -> bf16 conversion on this path
                work = torch.distributed.all_reduce(buf, op=torch.distributed.ReduceOp.SUM, group=self.process_group, async_op=async_op)
                handle = _Bf16Exchange(work if async_op else None, buf, bucket.grads)
                if async_op:
                    return handle
                handle.wait()
                return None
            work = torch.distributed.all_reduce(bucket.grads, op=torch.distributed.ReduceOp.SUM, group=self.process_group,
                                                async_op=async_op)
            return work if async_op else None
        return None

    def _aux_stream(self):
        st = getattr(self, '_aux', None)
        if st is None:
            st = self._aux = torch.cuda.Stream(device=self.device)
        return st

    def _apply_adam(self, which, lr, pending=None, exchanged=False):
        bucket = self.store.buckets[which]
        if pending is not None:
            pending.wait()                      # stream-level wait for the asynchronous all-reduce
        elif not exchanged:
            self.exchange_gradients(which)
        if self._capturing:
            # a captured launch replays with frozen arguments: the step size lives in device memory and is refreshed before each replay
            ops.adam_tf_flat_dev_alpha_(bucket.params, bucket.grads, bucket.m, bucket.v, self._alpha_dev[which], self.beta1, self.beta2,
                                        self.adam_eps, gscale=1.0 / self.world_size)
        else:
            ops.adam_tf_flat_(bucket.params, bucket.grads, bucket.m, bucket.v, self._adam_alpha(which, lr), self.beta1, self.beta2, self.adam_eps,
                              gscale=1.0 / self.world_size)
            self._advance_beta_powers(which)
        self.store.touch(which)                 # the filters changed: their Winograd forms are re-derived before the next use

    def _adam_alpha(self, which, lr):
        """lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) in fp32, like tf.train.AdamOptimizer._prepare / _apply_dense."""
        b1p, b2p = self.beta_power[which]
        return np.float32(np.float32(lr) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p))

    def _advance_beta_powers(self, which):
        b1p, b2p = self.beta_power[which]
        self.beta_power[which] = [np.float32(b1p * self.beta1), np.float32(b2p * self.beta2)]

    # ------------------------------------------------------------------------------------------------ steps
    def train_step(self, sess, feed_dict, step, batch_size, should_write_log=False, should_write_summary=False):
        """reference train_step (:79-117).  feed_dict = {'image': [B,H,W,3], 'future_image': [B,H,W,3]} in [-1,1].

        With only those two keys the D-run and the G-run see the same batch and share one generator forward (the benchmark
        convention).  The reference's input node hands a NEW batch to each sess.run (train.py:46-50):
        pass that second batch as 'image_G' / 'future_image_G' and the G-run recomputes the forward on it, exactly like the
        reference's second sess.run (BN moving statistics then come from the G-run's batch only, :199-202)."""
        start_time = time.time()
        if self.distributed and self.dp_graph == 'segments' and self._phased_eligible(feed_dict):
            self._train_step_dp(feed_dict)
        elif self._graph_eligible(feed_dict):
            self._train_step_graphed(feed_dict)
        else:
            self._train_step_eager(feed_dict)
        if should_write_log:
            vals = self.loss_values()
            duration = time.time() - start_time
            log.info('%s: step %d, loss_D = %.4f, loss_G = %.4f (%.1f examples/sec) %.3f sec/batch',
                     datetime.now(), step, vals['loss_D'], vals['loss_G'], batch_size / float(duration), duration)

    # ---- the step as a HIP graph ------------------------------------------------------------------------------------------
    def _graph_eligible(self, feed_dict):
        im = feed_dict['image']
        return (GRAPH and not self._graph_failed and (not self.distributed or self.dp_graph == 'one') and 'image_G' not in feed_dict and self.device.type == 'cuda'
                and im.is_cuda and im.dtype == torch.float32 and feed_dict['future_image'].shape == im.shape)

    def _train_step_graphed(self, feed_dict):
        """One train step through a captured HIP graph.

        The step is ~700 kernel launches on three streams whose host side (Python, ctypes, the autograd tape) costs 11-18 ms -- as much as
        the GPU needs for it.  Nothing in it depends on host values except the two Adam step sizes, so from the second call on an input
        shape the whole step (forward, both backward passes on their streams, both fused Adam updates, the Winograd filter re-derivations)
        is captured ONCE into a HIP graph and replayed: inputs are copied into the capture's static buffers, the two step sizes into two
        device floats.  Every kernel, its arguments and the stream order are those of the eager step, so the result is bit-identical to it
        (tests/test_model_gpu.py::test_graph_replay_is_bit_identical_to_the_eager_step).  Host state the step advances -- beta powers,
        global_step -- is advanced here per replay.  A capture that fails (a runtime without the needed support) falls back to eager."""
        im, fut = feed_dict['image'], feed_dict['future_image']
        key = (tuple(im.shape), im.device.index, ops.graph_knobs())
        ent = self._graphs.get(key)
        if ent is None:
            if self._eager_steps.get(key, 0) < GRAPH_WARMUP_STEPS:
                self._eager_steps[key] = self._eager_steps.get(key, 0) + 1
                return self._train_step_eager(feed_dict)
            ent = self._capture_step(key, im, fut)
            if ent is None:
                return self._train_step_eager(feed_dict)
        graph, static, outputs = ent
        lr = self.current_lr()
        ops.flat_copy_raw(im.contiguous().data_ptr(), static['image'].data_ptr(), im.numel())
        ops.flat_copy_raw(fut.contiguous().data_ptr(), static['future_image'].data_ptr(), fut.numel())
        for which in ('D', 'G'):
            ops.fill_raw_(self._alpha_dev[which], float(self._adam_alpha(which, lr)))
        graph.replay()
        for which in ('D', 'G'):
            self._advance_beta_powers(which)
            self.store.touch(which)         # host-side bookkeeping of the replayed Adam updates: derived filter forms are stale for eager code
        self.global_step += 1
        # NOTE: these tensors (losses, the forward outputs) are the capture's static buffers -- the next replayed step of this shape
        # overwrites them in place; read (or clone) what is needed before calling train_step again.  Eager steps return fresh tensors.
        self.last = dict(outputs, lr=float(lr))

    def _capture_mode(self):
        """Stream-capture error mode of the step captures.  In a process with a process group, ProcessGroupNCCL's watchdog THREAD polls the
        completion events of the eager collectives issued so far, and an event query from any thread is an error while a capture in the
        default 'global' mode is open (it aborted a run: 'operation not permitted when stream is capturing' from WorkNCCL::isCompleted).
        'thread_local' restricts the check to the capturing thread -- which only enqueues kernels -- so the watchdog's queries are legal
        whenever they come: no sleep, no race.  Single-process runs keep the stricter default."""
        return 'thread_local' if self.distributed else 'global'

    def _quiesce_collectives(self):
        """Before a capture: finish the GPU work of the eager steps (the collectives among it), so that nothing issued before the capture is
        still running beside it."""
        if self.distributed:
            torch.cuda.synchronize(self.device)

    def _capture_fault(self):
        """Test seam: tests replace this to raise inside an open capture (the in-process fall-back is what they check)."""

    def _capture_step(self, key, im, fut):
        self._quiesce_collectives()
        dev = self.device
        if self._alpha_dev is None:
            self._alpha_dev = {w: torch.zeros(1, dtype=torch.float32, device=dev) for w in ('D', 'G')}
        static = {'image': torch.empty_like(im, memory_format=torch.contiguous_format),
                  'future_image': torch.empty_like(fut, memory_format=torch.contiguous_format)}
        saved = (dict(self.beta_power), self.global_step, self.last)
        bank = getattr(self.store, 'filter_bank', None)
        if bank is not None:
            bank.touch()                    # the capture must CONTAIN the re-derivation of the filter forms, whatever ran before it
        graph = torch.cuda.CUDAGraph()
        self._capturing = True
        from . import _lib
        calls0 = _lib.abi_calls[0]
        try:
            with torch.cuda.graph(graph, capture_error_mode=self._capture_mode()):
                self._train_step_eager(static)
                self._capture_fault()
            outputs = {k: v for k, v in self.last.items() if k != 'lr'}
            self._graph_launches = _lib.abi_calls[0] - calls0          # diagnostics: C-ABI launches recorded in the graph
        except Exception as e:              # noqa: BLE001 -- whatever the runtime refuses: stay on the eager path
            if self.distributed and self.dp_graph == 'one':
                log.warning('capturing the data-parallel step with its collectives failed (%s: %s); continuing with captured segments', type(e).__name__, e)
                self.dp_graph = 'segments'
            else:
                log.warning('HIP graph capture of the train step failed (%s: %s); continuing with eager launches', type(e).__name__, e)
                self._graph_failed = True
            torch.cuda.synchronize(dev)
            # the failed capture advanced host bookkeeping for launches that never ran (FilterBank.synced, pending side-stream joins, tile
            # statistics in flight): mark every derived filter form stale and forget the rest before the eager fallback
            self.store.touch()
            ops.reset_after_failed_capture()
            return None
        finally:
            self._capturing = False
            self.beta_power, self.global_step, self.last = dict(saved[0]), saved[1], saved[2]      # nothing has executed yet
        ent = self._graphs[key] = (graph, static, outputs)
        return ent

    # ---- the data-parallel step: captured segments around the two gradient exchanges ---------------------------------------
    def _phased_eligible(self, feed_dict):
        im = feed_dict['image']
        return ('image_G' not in feed_dict and self.device.type == 'cuda' and im.is_cuda and im.dtype == torch.float32
                and feed_dict['future_image'].shape == im.shape)

    def _step_phases(self, feed_dict, device_alpha):
        """The shared-batch train step cut at its two gradient exchanges (a generator: everything between two yields is one segment):

            A   generator forward, discriminator forward + backward on (real, generated)                 -> yields 'exchange_D'
            B1  VGG19 forward of the perceptual loss: independent of the exchange, runs beside it        -> yields 'wait_D'
            B2  Adam-D + adversarial branch with the UPDATED discriminator (auxiliary stream) beside the VGG19 data gradients
                (main stream), then the generator backward                                               -> yields 'exchange_G'
            C   Adam-G

        Same kernels on the same operands as _train_step_eager (bit-identical results); what differs is the stream layout: each segment forks
        and joins inside itself (it must be capturable alone; every join goes into the main stream), and the 178.9 MB discriminator exchange --
        launched by the caller between A and B1 -- runs beside the VGG19 forward.  (Measured alternatives, one rank: the VGG19 forward in A
        beside the discriminator update 24.66 ms; this 24.61 ms; all four segments in one graph with the collectives captured 24.62 ms -- the
        +1.2 ms over the single-GPU step is the stream layout the boundaries force, not the boundaries.)
        ``device_alpha``: Adam reads its step size from self._alpha_dev (captured segments) instead of a host scalar."""
        im, future_im = feed_dict['image'], feed_dict['future_image']
        lr = self.current_lr()
        dev = self.device

        def join_all():
            if getattr(self, '_aux', None) is not None:
                torch.cuda.current_stream(dev).wait_stream(self._aux)
            ops.join_side_stream(dev)

        def adam(which):
            bucket = self.store.buckets[which]
            if device_alpha:
                ops.adam_tf_flat_dev_alpha_(bucket.params, bucket.grads, bucket.m, bucket.v, self._alpha_dev[which], self.beta1, self.beta2,
                                            self.adam_eps, gscale=1.0 / self.world_size)
            else:
                ops.adam_tf_flat_(bucket.params, bucket.grads, bucket.m, bucket.v, self._adam_alpha(which, lr), self.beta1, self.beta2,
                                  self.adam_eps, gscale=1.0 / self.world_size)
                self._advance_beta_powers(which)
            self.store.touch(which)

        with variables.as_default(self.store):
            # ---- A
            fwd = self._define_forward_pass(im, future_im)
            final = fwd['final_output']
            final_d = final.detach()
            aux = self._aux_stream() if AUX_STREAM else None
            d_losses = self._loss_D(final_d, future_im)
            ops.begin_backward()
            torch.autograd.backward([d_losses], [self._e0])
            join_all()
        yield 'exchange_D'
        with variables.as_default(self.store):
            # ---- B1
            recon = self._loss_G_recon(final, future_im)              # VGG19 forward beside the exchange
            join_all()
        yield 'wait_D'
        with variables.as_default(self.store):
            # ---- B2
            if aux is None:
                adam('D')
                adv = self._loss_G_adv(final)             # the UPDATED discriminator, like the reference's second sess.run
                g_adv = torch.autograd.grad([adv], [final], [self._e0])[0]
                g_recon = torch.autograd.grad([recon], [final], [self._one])[0]
            else:
                # Adam-D and the adversarial branch (frozen discriminator: no weight gradients, nothing forks from this stream) on the auxiliary
                # stream beside the VGG19 data gradients; `final` predates the fork, g_adv / adv are read on main after the join
                aux.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(aux):
                    adam('D')
                    adv = self._loss_G_adv(final)
                    g_adv = torch.autograd.grad([adv], [final], [self._e0])[0]
                g_recon = torch.autograd.grad([recon], [final], [self._one])[0]
                torch.cuda.current_stream(dev).wait_stream(aux)
            ops.begin_backward()
            final.backward(g_recon + g_adv)
            join_all()
        yield 'exchange_G'
        with variables.as_default(self.store):
            # ---- C
            adam('G')
        self._phase_out = dict(d_losses=d_losses.detach(), recon=recon.detach(), adv=adv.detach(), fwd={k: v.detach() for k, v in fwd.items()})
        self._phase_lr = float(lr)

    def _between_phases(self, tag, state):
        """The collectives of the data-parallel step, always enqueued from Python between two segments."""
        if tag == 'exchange_D':
            state['pending'] = self.exchange_gradients('D', async_op=True)      # on RCCL's stream, beside segment B1
        elif tag == 'wait_D':
            if state.get('pending') is not None:
                state['pending'].wait()                                         # stream-level: B2 starts behind the exchange
        elif tag == 'exchange_G':
            self.exchange_gradients('G')

    def _train_step_dp(self, feed_dict):
        """One data-parallel step: the segments of _step_phases with the two all-reduces between them.  From the second call on an input
        shape the four segments are four captured HIP graphs (one shared memory pool, always replayed in capture order) and a step is four
        replays + two collectives: ~1 ms of host work per rank instead of ~20 ms of launches (eight ranks share one host).  KPX_GRAPH=0, or a
        failed capture, runs the same segments launch by launch.  Replicas stay bit-identical either way (2-rank tests)."""
        im, fut = feed_dict['image'], feed_dict['future_image']
        key = ('dp', tuple(im.shape), im.device.index, ops.graph_knobs())
        ent = self._graphs.get(key) if (GRAPH and not self._graph_failed) else None
        if ent is None and GRAPH and not self._graph_failed and self._eager_steps.get(key, 0) >= GRAPH_WARMUP_STEPS:
            ent = self._capture_phases(key, im, fut)
        if ent is None:
            self._eager_steps[key] = self._eager_steps.get(key, 0) + 1
            state = {}
            for tag in self._step_phases(feed_dict, device_alpha=False):
                self._between_phases(tag, state)
            self.global_step += 1
            self.last = dict(self._phase_out, lr=self._phase_lr)
            return
        graphs, static, outputs = ent
        lr = self.current_lr()
        ops.flat_copy_raw(im.contiguous().data_ptr(), static['image'].data_ptr(), im.numel())
        ops.flat_copy_raw(fut.contiguous().data_ptr(), static['future_image'].data_ptr(), fut.numel())
        for which in ('D', 'G'):
            ops.fill_raw_(self._alpha_dev[which], float(self._adam_alpha(which, lr)))
        state = {}
        for graph, tag in zip(graphs, ('exchange_D', 'wait_D', 'exchange_G', None)):
            graph.replay()
            if tag is not None:
                self._between_phases(tag, state)
        for which in ('D', 'G'):
            self._advance_beta_powers(which)
            self.store.touch(which)
        self.global_step += 1
        self.last = dict(outputs, lr=float(lr))          # (the capture's static buffers: valid until the next step, see _train_step_graphed)

    def _capture_phases(self, key, im, fut):
        """Capture the four segments, in order, into four graphs that share one memory pool.  Nothing executes during the capture and no
        collective is issued; the first real execution is the first replay."""
        self._quiesce_collectives()
        dev = self.device
        if self._alpha_dev is None:
            self._alpha_dev = {w: torch.zeros(1, dtype=torch.float32, device=dev) for w in ('D', 'G')}
        static = {'image': torch.empty_like(im, memory_format=torch.contiguous_format),
                  'future_image': torch.empty_like(fut, memory_format=torch.contiguous_format)}
        saved = (dict(self.beta_power), self.global_step, self.last)
        bank = getattr(self.store, 'filter_bank', None)
        if bank is not None:
            bank.touch()                    # segment A must CONTAIN the re-derivation of the filter forms, whatever ran before it
        graphs = []
        self._capturing = True
        pool = None
        try:
            gen = self._step_phases(static, device_alpha=True)
            done = False
            while not done:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode=self._capture_mode()):
                    try:
                        next(gen)
                    except StopIteration:
                        done = True
                pool = g.pool()
                graphs.append(g)
            assert len(graphs) == 4, len(graphs)
            outputs = dict(self._phase_out)
        except Exception as e:              # noqa: BLE001 -- whatever the runtime refuses: stay on the eager segments
            log.warning('HIP graph capture of the data-parallel step failed (%s: %s); continuing with eager launches', type(e).__name__, e)
            self._graph_failed = True
            torch.cuda.synchronize(dev)
            self.store.touch()
            ops.reset_after_failed_capture()
            return None
        finally:
            self._capturing = False
            self.beta_power, self.global_step, self.last = dict(saved[0]), saved[1], saved[2]      # nothing has executed yet
        ent = self._graphs[key] = (graphs, static, outputs)
        return ent

    def broadcast_parameters(self, src=0):
        """Data parallel: every replica starts from rank ``src``'s state -- parameters, Adam slots, moving statistics, step counters.
        The replicas are seeded identically anyway; this is insurance against a drifted seed or a partially restored replica (one
        broadcast per flat buffer, once)."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or self.world_size <= 1:
            return
        group_src = torch.distributed.get_global_rank(self.process_group, src) if self.process_group is not None else src
        with torch.no_grad():
            for b in self.store.buckets.values():
                for flat in (b.params, b.m, b.v):
                    if flat is not None:
                        torch.distributed.broadcast(flat, group_src, group=self.process_group)
            for name, (shape, kind) in self.store.specs.items():
                if not self.store.is_trainable(kind):
                    torch.distributed.broadcast(self.store.vars[name], group_src, group=self.process_group)
            counters = torch.tensor([float(self.global_step)] + [float(v) for w in ('D', 'G') for v in self.beta_power[w]],
                                    dtype=torch.float64, device=self.device)
            torch.distributed.broadcast(counters, group_src, group=self.process_group)
            c = counters.cpu().numpy()
        self.global_step = int(c[0])
        self.beta_power = {'D': [np.float32(c[1]), np.float32(c[2])], 'G': [np.float32(c[3]), np.float32(c[4])]}
        self.store.touch()

    def _train_step_eager(self, feed_dict):
        """The step enqueued launch by launch (also the function a capture records)."""
        im, future_im = feed_dict['image'], feed_dict['future_image']
        separate = 'image_G' in feed_dict
        lr = self.current_lr()
        with variables.as_default(self.store):
            # ---- D run (:93)
            if separate:
                with torch.no_grad():
                    final_d = self._define_forward_pass(im, future_im, update_moving=False)['final_output']
            else:
                fwd = self._define_forward_pass(im, future_im)
                final_d = fwd['final_output'].detach()
            # The whole discriminator update (forward on real + fake, backward, exchange, Adam) and the G run's adversarial branch behind it
            # (discriminator forward with the UPDATED weights + its data gradients) are independent of the perceptual half of the G run (VGG19
            # on fake + real, forward and data gradients): with one shared batch the two halves run side by side, the gradient of each G-loss
            # term with respect to the generated frame is taken separately and the generator is walked once with their sum.  (With a separate
            # G batch the generator forward below re-derives every Winograd filter form, the discriminator's included, so the D update must be
            # complete first: no overlap there.)
            aux = self._aux_stream() if (AUX_STREAM and not separate and self.device.type == 'cuda') else None
            if aux is not None:
                aux.wait_stream(torch.cuda.current_stream(self.device))
            with (torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext()):
                d_losses = self._loss_D(final_d, future_im)
                ops.begin_backward()
                # on the auxiliary stream the update continues right here (exchange, Adam), so its weight gradients run inline on this
                # stream: a side stream forked from a forked stream must never be joined back into it (ops: stream discipline)
                with (ops.inline_wgrad() if aux is not None else contextlib.nullcontext()):
                    torch.autograd.backward([d_losses], [self._e0])
                # (under capture the collective is issued synchronously: it then runs on THIS stream, see DP_GRAPH 'one')
                pending = self.exchange_gradients('D', async_op=not self._capturing)
                if aux is not None:
                    self._apply_adam('D', lr, pending=pending, exchanged=True)
            # ---- G run (:94): the perceptual forward does not involve the discriminator, so it runs first ...
            if separate:
                im, future_im = feed_dict['image_G'], feed_dict['future_image_G']
                fwd = self._define_forward_pass(im, future_im)
            final = fwd['final_output']
            recon = self._loss_G_recon(final, future_im)
            if aux is not None and AUX_STREAM_ADV:
                g_recon = torch.autograd.grad([recon], [final], [self._one])[0]
                # CROSS-STREAM LIFETIME INVARIANT (no record_stream anywhere in this step): a tensor that crosses main <-> aux is produced
                # BEFORE the fork whose wait_stream orders it and stays referenced from Python until AFTER the join that orders its last
                # reader, so the caching allocator never recycles it under a kernel of the other stream:
                #   main -> aux: `final`, `final_d`, `future_im` predate aux.wait_stream(main) above (the fork of the discriminator update) and
                #                are locals of this frame until the join below;
                #   aux -> main: `g_adv`, `d_losses`, `adv` are read on main only after main.wait_stream(aux) below.
                # No fresh aux.wait_stream(main) here on purpose: it would order the adversarial branch behind the VGG19 chain just enqueued
                # on main (g_recon) and serialise the two halves; `final` is already ordered by the earlier fork.
                with torch.cuda.stream(aux):
                    adv = self._loss_G_adv(final)
                    g_adv = torch.autograd.grad([adv], [final], [self._e0])[0]
                torch.cuda.current_stream(self.device).wait_stream(aux)
                ops.begin_backward()
                final.backward(g_recon + g_adv)
            else:
                if aux is not None:
                    torch.cuda.current_stream(self.device).wait_stream(aux)
                else:
                    self._apply_adam('D', lr, pending=pending, exchanged=True)
                # ... and the adversarial term sees the UPDATED discriminator, exactly as the reference's second sess.run
                adv = self._loss_G_adv(final)
                ops.begin_backward()
                torch.autograd.backward([recon, adv], [self._one, self._e0])
            if self.device.type == 'cuda' and getattr(self, '_aux', None) is not None:
                # backward nodes recorded on the auxiliary stream ran there (their weight gradients on the side stream forked from it):
                # the main stream joins the auxiliary stream here and every side stream in exchange_gradients('G') (ops: stream discipline)
                torch.cuda.current_stream(self.device).wait_stream(self._aux)
            self._apply_adam('G', lr)
        self.global_step += 1                                             # incremented by the G optimiser (:201-202)
        self.last = dict(d_losses=d_losses.detach(), recon=recon.detach(), adv=adv.detach(), lr=float(lr),
                         fwd={k: v.detach() for k, v in fwd.items()})

    LAUNCH_MODES = ('eager: every kernel enqueued from Python', 'one HIP graph replay per step',
                    'data parallel: four captured segments replayed around the two gradient all-reduces',
                    'data parallel: one HIP graph replay per step, the two all-reduces captured inside')

    def launch_mode(self):
        """Index into LAUNCH_MODES: how train_step currently reaches the GPU."""
        if self._graph_failed or not self._graphs:
            return 0
        if any(k[0] == 'dp' for k in self._graphs):
            return 2
        return 3 if self.distributed else 1

    def loss_values(self):
        """Host copies of the last step's scalars (synchronises)."""
        d = self.last['d_losses'].cpu().numpy()
        recon = float(self.last['recon'].cpu()[0])
        adv = float(self.last['adv'].cpu()[0])
        return dict(loss_D=float(d[0]), loss_D_real=float(d[1]), loss_D_fake=float(d[2]),
                    loss_G_recon=recon, loss_G_adv=adv, loss_G=recon + adv, lr=self.last['lr'])

    def test_step(self, sess, feed_dict, step, test_idx, batch_size):
        """reference test_step (:119-141): losses only, BN in batch-statistics mode, no updates."""
        im, future_im = feed_dict['image'], feed_dict['future_image']
        start_time = time.time()
        with variables.as_default(self.store), torch.no_grad():
            fwd = self._define_forward_pass(im, future_im, update_moving=False)
            d = self._loss_D(fwd['final_output'], future_im)
            recon, adv = self._loss_G(fwd['final_output'], future_im)
        loss_d = float(d.cpu()[0])
        loss_g = float(recon.cpu()[0]) + float(adv.cpu()[0])
        return loss_d, loss_g, time.time() - start_time, batch_size

    def collect_test_results(self, results, step):
        """reference collect_test_results (:143-158)."""
        average_loss_D = sum(x[0] for x in results) / len(results)
        average_loss_G = sum(x[1] for x in results) / len(results)
        total_duration = sum(x[2] for x in results)
        num_examples = sum(x[3] for x in results)
        log.info('test: %s: step %d, loss_D = %.4f, loss_G = %.4f (%.1f examples/sec) %.3f sec/batch', datetime.now(), step,
                 average_loss_D, average_loss_G, num_examples / total_duration, total_duration / len(results))
        return average_loss_D, average_loss_G

    # ------------------------------------------------------------------------------------------------ checkpoints
    def checkpoint_arrays(self):
        powers = {'beta1_power': self.beta_power['D'][0], 'beta2_power': self.beta_power['D'][1],     # D optimiser first (:198)
                  'beta1_power_1': self.beta_power['G'][0], 'beta2_power_1': self.beta_power['G'][1],
                  'global_step': np.int32(self.global_step)}
        return self.store.export_numpy(include_slots=self.is_training, beta_powers={k: np.asarray(v) for k, v in powers.items()})

    def _restore_extra(self, arrays):
        if 'global_step' in arrays:
            self.global_step = int(arrays['global_step'])
        for which, sfx in (('D', ''), ('G', '_1')):
            if 'beta1_power' + sfx in arrays:
                self.beta_power[which] = [np.float32(arrays['beta1_power' + sfx]), np.float32(arrays['beta2_power' + sfx])]
        if self.is_training:
            with torch.no_grad():
                for b in self.store.buckets.values():
                    for name in b.entries:
                        for slot, flat in (('/Adam', b.m), ('/Adam_1', b.v)):
                            if '_0+1/' in name:
                                k0, k1 = name.replace('_0+1/', '_0/') + slot, name.replace('_0+1/', '_1/') + slot
                                if k0 in arrays and k1 in arrays:
                                    src = np.concatenate([arrays[k0], arrays[k1]], axis=-1)
                                else:
                                    continue
                            elif name + slot in arrays:
                                src = arrays[name + slot]
                            else:
                                continue
                            b.view(flat, name).copy_(torch.from_numpy(np.ascontiguousarray(src, np.float32)).to(flat.device))
