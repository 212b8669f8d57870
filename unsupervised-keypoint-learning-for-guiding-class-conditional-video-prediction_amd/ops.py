"""torch.autograd.Function wrappers around the C ABI (include/kpx.h).

PyTorch is used as plumbing only: device buffers (caching allocator), the current HIP stream and the autograd tape.
Every tensor op on the hot path is a kernel from libkpx_hip.so; there is no eager-PyTorch fallback -- a missing
library fails at import (``_lib``) and a failed launch raises ``KpxError``.

Weight / bias / gamma / beta gradients are written by the kernels straight into caller-provided views of the
flat gradient bucket (``*_grad_out``), so the RCCL all-reduce and the fused Adam step run on one contiguous buffer
without a gather pass.
"""
import os as _os

import torch

from . import _lib
from ._lib import lib, check

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
BN_EPS = 1e-5        # reference: models/networks/layers.py:14
BN_DECAY = 0.999     # tf.contrib.layers.batch_norm default


# raw handle of torch's current HIP stream on the current device: the private accessors are ~10x cheaper than
# torch.cuda.current_stream().cuda_stream, which was 2 x 2 ms of host time per train step (one lookup per kernel launch)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(t):
    if not t.is_cuda:
        raise _lib.KpxError('kpx ops run on MI355X only (got a %s tensor); there is no CPU fallback' % t.device)
    if t.dtype != torch.float32 and t.dtype != torch.bfloat16:
        raise _lib.KpxError('kpx ops take fp32 tensors, or bf16 activation tensors in the bf16 configuration (got %s)' % t.dtype)


def same_pad(in_size, k, s):
    """TF SAME padding: out = ceil(in/s); the extra pixel goes to the bottom/right."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2, out


def _nhwc(t):
    """Return (tensor, pixel stride) for an NHWC tensor that may be a channel slice of a wider buffer."""
    _require_gpu(t)
    assert t.dim() == 4
    n, h, w, c = t.shape
    ld = t.stride(2) if w > 1 else (t.stride(1) if h > 1 else (t.stride(0) if n > 1 else c))
    ok = (c == 1 or t.stride(3) == 1) and ld >= c
    ok = ok and (w == 1 or t.stride(2) == ld) and (h == 1 or t.stride(1) == w * ld) and (n == 1 or t.stride(0) == h * w * ld)
    if not ok:
        t = t.contiguous()
        ld = c
    return t, ld


class _Scratch:
    """Per-device, per-stream scratch buffers; kernels on one stream are ordered, so consecutive ops may share them."""

    def __init__(self):
        self._bufs = {}

    def get(self, key, nbytes, device):
        k = (key, device, _stream())
        b = self._bufs.get(k)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(max(nbytes, 1 << 16)), dtype=torch.uint8, device=device)
            self._bufs[k] = b
        return b

    def reduce(self, c, device):
        return self.get('reduce', lib.kpx_chan_reduce_scratch_bytes(int(c)), device)


scratch = _Scratch()

# Weight gradients are off the backward critical path (only the optimiser consumes them), so they are launched on a second
# HIP stream where they fill the ramp-up / tail gaps of the dgrad chain on the main stream.  join_side_stream() is called
# before the gradient exchange / Adam update.
# (The switches below are module constants, not environment variables: the tests that compare a fused path with its separate passes
#  monkeypatch them; DESIGN.md lists the environment variables the repo does read.)
SIDE_WGRAD = True
FUSE_BN_STATS = True     # batch statistics from the conv epilogue
# the discriminator's leaky-ReLU backward in the epilogue of the data gradient above it (kpx_conv2d_dgrad_act_f32) instead of a pass of its own
FUSE_ACT_BWD = True
# batch-norm backward sums (and the ReLU mask of the gradient) from the epilogue of the data gradient that produces it, F(4x4,3x3) and
# F(2x2,3x3) kernels: the reduction pass over (dz, y) of kpx_bn_train_bwd_f32 is then skipped.  Round 2 measured the F(2x2,3x3)-only version a
# wash (31.4-31.7 vs 31.3 ms); with the F(4x4,3x3) epilogue and the batched finalize it is -0.15 ms (23.05 vs 23.20, A/B in one session)
FUSE_BN_BWD = True
# where a layer's weight gradient forks from its stream: 'after' its data gradient has been enqueued, 'before' it, or 'capture' (default):
# before it while the step is being captured into a HIP graph, after it in eager mode -- measured on MI355X at B=32: eager 26.57 ms (after) /
# 27.22 ms (before); graph replay 27.44 ms (after) / 26.96 ms (before)
FORK_BEFORE_DGRAD = 'capture'
_side_streams = {}       # (device, raw handle of the stream the weight gradients were forked from) -> side stream
_side_dirty = {}         # the same keys -> True while un-joined kernels are pending on that side stream
_side_keep = []          # tensors read by kernels on a side stream: kept alive until join_side_stream() (cheaper than
                         # Tensor.record_stream, whose pending events the caching allocator polls on every allocation)


# Stream discipline (it is what makes the step capturable into a HIP graph, and it is harmless in eager mode):
#   * one side stream PER ORIGIN stream: backward nodes on the main stream fork side(main), nodes of a branch that ran on the auxiliary
#     stream fork side(aux) -- the fork graph is a tree  main -> {aux -> {side(aux)}, side(main)};
#   * EVERY join goes into the main stream; a forked stream never waits for a stream forked from itself.  The HIP runtime under torch 2.10
#     re-parents any non-origin stream that waits on a captured event, so a fork from aux that is later joined back INTO aux makes the two
#     streams each other's "parallel capture stream" and hipStreamEndCapture recurses until the stack overflows (found with rocgdb; the
#     pattern is legal HIP).  Where a branch on the auxiliary stream needs its weight gradients before it continues (the discriminator
#     update: backward -> Adam on that stream), they are launched inline on that stream: ``inline_wgrad()``.
_inline_wgrad = [False]


class inline_wgrad:
    """Context: weight / bias gradients of the backward nodes inside run on the node's own stream instead of a side stream."""

    def __enter__(self):
        self.old, _inline_wgrad[0] = _inline_wgrad[0], True

    def __exit__(self, *exc):
        _inline_wgrad[0] = self.old
        return False


def _side_stream(device, origin):
    """The weight-gradient stream forked from ``origin`` (raw stream handle)."""
    key = (device, origin)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return key, st


# Gradients are written (not accumulated) into the flat bucket.  If one variable is used by several ops inside one backward
# pass (e.g. the reference's two weight-sharing pose_encoder calls made as two calls instead of one batched call), the second
# and later writers must ADD.  begin_backward() opens a new epoch; a destination already written in the current epoch is
# accumulated into through a temporary + axpy.
_grad_epoch = [0]
_grad_written = {}


def begin_backward():
    _grad_epoch[0] += 1
    if len(_grad_written) > 4096:
        _grad_written.clear()
    _pending_bwd_stats.clear()


def _claim_grad(dst):
    """True if ``dst`` has not been written in this backward epoch yet (and mark it written)."""
    key = (dst.data_ptr(), dst.numel())
    first = _grad_written.get(key) != _grad_epoch[0]
    _grad_written[key] = _grad_epoch[0]
    return first


def reset_after_failed_capture():
    """A stream capture that failed half way leaves host bookkeeping advanced for kernels that were only recorded: pending side-stream
    joins, tensors kept alive for them, tile statistics handed between ops.  Forget all of it (the caller also touches its store so that
    the derived filter forms are re-derived) before falling back to eager launches."""
    _side_dirty.clear()
    _side_keep.clear()
    _pending_stats.clear()
    _pending_bwd_stats.clear()


def graph_knobs():
    """Everything besides the input shape that a captured step freezes: the compute dtype and the module-level kernel-selection
    switches.  Part of the graph cache keys, so that flipping one re-captures instead of replaying the old arithmetic."""
    return (_compute_dtype[0], SIDE_WGRAD, FUSE_BN_STATS, FUSE_BN_BWD, FUSE_ACT_BWD, FORK_BEFORE_DGRAD, WINO43, WINO43_MIN_WORKGROUPS, WINO43_NMIN)


def normalize_device(device):
    """torch.device with an explicit index: 'cuda' (index None) becomes the current device, so that it compares equal to the
    ``tensor.device`` values the ops record (torch.device('cuda') != torch.device('cuda:0'))."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


def join_side_stream(device=None):
    """Make the current stream wait for the weight-gradient kernels launched on the side streams of ``device`` (all devices if None).
    Call it on the MAIN stream (see the stream discipline above)."""
    if device is not None:
        device = normalize_device(device)
    for key in list(_side_dirty):
        dev, origin = key
        if device is not None and dev != device:
            continue
        torch.cuda.current_stream(dev).wait_stream(_side_streams[key])
        del _side_dirty[key]
    if not _side_dirty:
        _side_keep.clear()       # every side kernel is now ordered before something the consumers wait for: the blocks may be recycled


# ----------------------------------------------------------------------------------------------- compute dtype
# 'f32' (default, the parity configuration) or 'bf16' (BASELINE configs[2], "bf16 storage + fp32 accumulate"): ACTIVATION TENSORS ARE bf16
# IN HBM -- every producer writes bf16, every consumer reads bf16 -- with fp32 accumulation, fp32 batch-norm statistics, fp32 master
# weights / gradient buckets / Adam.  Images, the generated frame, the translator's 4-channel head, the tensor the key-point head reads,
# key-points, heat-map arithmetic and all losses stay fp32.  A layer whose shape the bf16 kernels do not take runs through the fp32 kernel
# of the fp32 configuration between two conversion passes (`_fallback_uses` counts them: the bench configuration should not need any).
_compute_dtype = [_os.environ.get('KPX_DTYPE', 'f32')]
BF16 = torch.bfloat16
fallback_uses = {'conv_fwd': 0, 'conv_dgrad': 0, 'conv_wgrad': 0, 'other': 0}        # diagnostics: bf16 tensors routed through fp32 kernels


def set_compute_dtype(name):
    if name not in ('f32', 'bf16'):
        raise ValueError("compute dtype must be 'f32' or 'bf16'")
    _compute_dtype[0] = name


def compute_dtype():
    return _compute_dtype[0]


def act_dtype():
    """Storage type of activation tensors in the current configuration."""
    return BF16 if _compute_dtype[0] == 'bf16' else torch.float32


def _arith():
    """The `arith` argument of kpx_conv2d_{fwd,dgrad,wgrad}_f32 (include/kpx.h): the implicit-GEMM kernels of the strided / 4x4 / 1x1
    layers and the direct weight gradients take three bf16 terms per fp32 operand (fp32-equivalent) or one (the bf16 configuration)."""
    return 1 if _compute_dtype[0] == 'bf16' else 0


def cast_channels_raw(src_ptr, ldsrc, dst_ptr, lddst, pixels, c, kind):
    """dst[p][0:c] = src[p][0:c]; kind 0: f32 -> bf16, 1: bf16 -> f32, 2: bf16 -> bf16, 3: f32 -> bf16 into 8 channels, c .. 7 zeroed."""
    check(lib.kpx_cast_channels(src_ptr, ldsrc, dst_ptr, lddst, pixels, c, kind, _stream()), 'kpx_cast_channels')


def cast(t, dtype):
    """A contiguous copy of ``t`` in ``dtype`` (fp32 <-> bf16); ``t`` itself when it already has it."""
    if t.dtype == dtype:
        return t
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=dtype, device=t.device)
    n = t.numel()
    c = next(k for k in (4096, 512, 64, 8, 1) if n % k == 0)
    if n:
        cast_channels_raw(t.data_ptr(), c, out.data_ptr(), c, n // c, c, 0 if dtype == BF16 else 1)
    return out


def _esz(t):
    return 2 if t.dtype == BF16 else 4


def _slice_f32(x, ldx, cin):
    """fp32 contiguous [N,H,W,cin] copy of the first ``cin`` channels of the NHWC tensor ``x`` (pixel stride ldx)."""
    n, h, w = x.shape[0], x.shape[1], x.shape[2]
    out = torch.empty((n, h, w, cin), dtype=torch.float32, device=x.device)
    if x.dtype == BF16:
        cast_channels_raw(x.data_ptr(), ldx, out.data_ptr(), cin, n * h * w, cin, 1)
    else:
        copy_channels_raw(x.data_ptr(), ldx, out.data_ptr(), cin, n * h * w, cin)
    return out


def _bf16s_bnbwd(dy, lddy, k, w, dx, lddx, nn, bn_y, beta):
    """(slab, tiles per image) or False: data gradient + the backward sums of the batch norm whose ReLU'd output ``bn_y`` the convolution read."""
    n, h, wd = dy.shape[0], dy.shape[1], dy.shape[2]
    if not (k % 32 == 0 or k in (8, 16)):
        k = 8 if k < 8 else 16 if k < 16 else (k + 31) // 32 * 32            # (as _bf16s_conv: the prepared filters are zero there)
        if lddy < k:
            return False
    if (bn_y.dtype != BF16 or not bn_y.is_contiguous() or bn_y.shape[3] != nn or nn % 8 or lddx % 8
            or not lib.kpx_conv3x3_bf16s_eligible(n, h, wd, k, nn, lddy, dy.data_ptr())):
        return False
    wf = _bf16s_prepared(w, True)
    tiles = lib.kpx_conv3x3_bf16s_stats_tiles(n, h, wd, k, nn)
    slab = torch.empty(tiles * 2 * nn, dtype=torch.float32, device=dy.device)
    rc = lib.kpx_conv3x3_bf16s_bnbwd(dy.data_ptr(), n, h, wd, k, lddy, wf.data_ptr(), dx.data_ptr(), nn, lddx, bn_y.data_ptr(), nn, beta.data_ptr(), slab.data_ptr(), _stream())
    if rc == -1:
        return False
    check(rc, 'kpx_conv3x3_bf16s_bnbwd')
    conv_kernel_uses_bf16s[0] += 1
    from fractions import Fraction
    return slab, Fraction(tiles, n)


# prepared (fragment-ordered bf16) filters of the bf16-storage 3x3 kernel: (filter data_ptr, dgrad) -> (Wf, owning FilterBank or None, weak ref, (Cin, Cout))
_bf16s_w = {}
conv_kernel_uses_bf16s = [0]


def _bf16s_prepared(w, dgrad):
    ent = _cached_u(_bf16s_w, w, dgrad)
    if ent is None:
        # a filter nobody registered (tests, one-off layers): prepare per call
        cin, cout = int(w.shape[2]), int(w.shape[3])
        k, nn = (cout, cin) if dgrad else (cin, cout)
        wf = scratch.get('bf16s_w%d' % (1 if dgrad else 0), lib.kpx_conv3x3_bf16s_weights_bytes(k, nn), w.device)
        check(lib.kpx_conv3x3_bf16s_prepare_f32(w.data_ptr(), cin, cout, 1 if dgrad else 0, wf.data_ptr(), _stream()), 'kpx_conv3x3_bf16s_prepare_f32')
        return wf
    if ent[1] is not None:
        ent[1].ensure_fresh('bf16s')
    return ent[0]


def _bf16s_conv(inp, ld_in, k, w, bias, out, ld_out, nn, act, dgrad, want_stats=False, mask=None):
    """3x3 stride-1 SAME convolution on bf16 tensors (csrc/conv_bf16s.hip); False when the shape is not one the kernel takes.
    With want_stats returns (slab, tiles per image as a Fraction)."""
    n, h, wd = inp.shape[0], inp.shape[1], inp.shape[2]
    out_f32 = out.dtype == torch.float32
    if inp.dtype != BF16:
        return False
    # ragged channel counts (the 158-channel joint embedding in its 160-wide buffer): the kernel runs on the rounded-up counts -- the
    # prepared filters are zero there, so the extra gathered channels (zero-filled / finite by the producer's contract) contribute nothing
    # and the extra produced channels come out as zeros
    if not (k % 32 == 0 or k in (8, 16)):
        k0 = k
        k = 8 if k < 8 else 16 if k < 16 else (k + 31) // 32 * 32
        if ld_in < k:
            return False
        # CONTRACT of the rounded-up gather: channels k0..k of the caller's buffer are multiplied by zero filter rows, so they must be FINITE
        # (0 * inf = nan).  Every producer on the path zero-fills its pad channels (JointEmbeddingFn, the concat buffers); KPX_DEBUG_FINITE=1
        # checks it on every call (a device synchronisation: diagnostics only).
        if DEBUG_FINITE and not bool(torch.isfinite(inp.reshape(-1, inp.shape[-1])[:, k0:k].float()).all()):
            raise _lib.KpxError('bf16 3x3 conv: non-finite values in the pad channels %d..%d of a %d-wide buffer' % (k0, k, inp.shape[-1]))
    g = 4 if out_f32 else 8
    if nn % g:
        nn = (nn + g - 1) // g * g
        if ld_out < nn:
            return False
    if ld_out % g or not lib.kpx_conv3x3_bf16s_eligible(n, h, wd, k, nn, ld_in, inp.data_ptr()):
        return False
    if out.data_ptr() % 16 or (mask is not None and (out_f32 or mask.dtype != BF16 or mask.data_ptr() % 16)):
        return False
    wf = _bf16s_prepared(w, dgrad)
    slab = None
    if want_stats:
        tiles = lib.kpx_conv3x3_bf16s_stats_tiles(n, h, wd, k, nn)
        slab = torch.empty(tiles * 2 * nn, dtype=torch.float32, device=inp.device)
    check(lib.kpx_conv3x3_bf16s(inp.data_ptr(), n, h, wd, k, ld_in, wf.data_ptr(), bias.data_ptr() if bias is not None else None,
                                out.data_ptr(), nn, ld_out, 1 if out_f32 else 0, act, mask.data_ptr() if mask is not None else None,
                                mask.stride(2) if mask is not None else 0, slab.data_ptr() if slab is not None else None, _stream()), 'kpx_conv3x3_bf16s')
    conv_kernel_uses_bf16s[0] += 1
    if want_stats:
        from fractions import Fraction
        return slab, Fraction(tiles, n)
    return True


# ----------------------------------------------------------------------------------------------- pre-transformed Winograd filters
# U = G g G^T is a function of the filter only.  Constant filters (VGG19) are transformed once; the trainable 3x3 filters of a
# VariableStore are transformed together in ONE launch (kpx_wino_filter_transform_batch_f32) the first time a convolution needs
# them after the store changed (optimiser update / restore), instead of once per convolution call.
import ctypes as _ctypes
import struct as _struct

_wino_u = {}          # (filter data_ptr, dgrad) -> (U tensor, owning FilterBank or None, weak reference to the filter tensor, (Cin, Cout))
_wino43_u = {}        # the same for the F(4x4,3x3) form of the filters whose layers may use it
_wino43b_u = {}       # ... and its bf16x3 form (three exact bf16 planes in fragment order) for csrc/conv_wino43b.hip
import weakref as _weakref


def _cached_u(table, w, dgrad):
    """The cached transform of filter ``w``, or None.  The key is the filter's ADDRESS, so an entry is only trusted while the tensor it
    was made from is allocated (a bank keeps its filters alive itself; a constant / folded filter is tracked by a weak reference to its storage) and has this
    shape -- the allocator hands a freed filter's address to unrelated tensors."""
    key = (w.data_ptr(), 1 if dgrad else 0)
    ent = table.get(key)
    if ent is None:
        return None
    if (ent[2] is not None and ent[2]() is None) or ent[3] != (int(w.shape[2]), int(w.shape[3])):
        table.pop(key, None)
        return None
    return ent
conv_kernel_uses = {'wino43': 0, 'wino43b': 0}      # diagnostics / tests: launches of the F(4x4,3x3) kernels (fp32-MFMA form / bf16x3 form; 'wino43' counts both)

# F(4x4,3x3) (csrc/conv_wino43.hip) does 2.25x fewer multiplies than F(2x2,3x3) at ~6x its rounding error (2-3e-6 rel-L2 per layer, still
# inside the 1e-5 bar of one layer).  Where it runs is a PER-LAYER ATTRIBUTE and a direction, from the measured effect on a whole train
# step against the float64 arbiter (tests/test_model_gpu.py::test_configs0..., profiles/wino43_policy.py):
#   * data gradients: everywhere (the detector's included: its eight eligible layers change no digit of the gradient-error figures);
#   * forward: only layers declared with ``f43_fwd=True`` (the default of layers.conv; networks.py passes False where the measurement
#     said so): VGG19 and the translator's second and third stage (conv_3_* .. conv_5_*).  The detector, the image encoder and the
#     translator's first stage (256-deep sums feeding batch norms over few pixels) stay on F(2x2,3x3): with them on F(4x4,3x3) the
#     generated frame moves 3e-5 instead of 1.7e-5 from the oracle's and the discriminator gradient 4.6x instead of 1.2x as far from the
#     float64 gradient as the fp32 oracle's own.
# The attribute lives with the layer's filter variable (VariableStore.layer_attrs), not in its name: renaming a scope changes no kernel.
# (Round 4 measured the alternative -- F(4x4,3x3) on every forward layer: -0.13..-0.2 ms per step for a generated frame 3.6e-5 instead of
#  1.7e-5 from the oracle's; the margin was kept and the experiment switches are gone.)  KPX_WINO43=0: F(2x2,3x3) everywhere (tests).
WINO43 = _os.environ.get('KPX_WINO43', '1') != '0'
DEBUG_FINITE = _os.environ.get('KPX_DEBUG_FINITE', '0') == '1'      # assert the finite-pad-channel contract of the rounded-up bf16 gathers (_bf16s_conv)
# launches of at most this many F(4x4,3x3) workgroups stay on F(2x2,3x3).  Round 2 set 128 from a kernel measured ALONE (128 workgroups: 0.187 vs
# 0.158 ms -- one F(4x4) workgroup owns its CU); inside the step, where the other streams fill the idle CUs, F(4x4,3x3) wins on those layers too:
# 23.73-23.83 ms per step at 0 / 32 / 64 against 24.01-24.16 at 128 (three repetitions each, B=32) -- so no threshold
WINO43_MIN_WORKGROUPS = 0          # (a module constant the model-level parity tests set to 128 and back)
WINO43_NMIN = 33                   # produced channels from which a layer takes the 64-cout F(4x4,3x3) workgroups (32 / 16 measured: no change)
# F(4x4,3x3) with the transform-domain GEMMs fp32-EQUIVALENT on the bf16 matrix pipe (csrc/conv_wino43b.hip: exact three-term operands, six
# products): the same layers and directions as WINO43, same per-layer policy; the shapes it takes (no packed 16x16 images, K % 4 == 0) and
# where it is preferred over the fp32-MFMA kernel: _wino43b_preferred.  KPX_WINO43B=0: the fp32-MFMA F(4x4,3x3) kernel everywhere (A/B, tests).
WINO43B = _os.environ.get('KPX_WINO43B', '1') != '0'
WINO43B_MAX_CHANNELS = 512         # gathered x produced channels up to which the bf16x3 form is prepared (its U is 1.5x the fp32 U: L2 traffic)


def _wino43b_form_wanted(k, nn):
    return WINO43B and k >= 16 and k % 4 == 0 and nn >= WINO43_NMIN and max(k, nn) <= WINO43B_MAX_CHANNELS


def _wino43b_preferred(n, h, wd, k, nn):
    """Launch shapes on which the bf16x3 kernel is the faster F(4x4,3x3) form (profiles/r06_wino43b_layers.txt)."""
    return True


def _wino43_wanted(name, cin, cout, dgrad, f43_fwd=True):
    """f43_fwd: the layer's attribute (layers.conv(..., f43_fwd=)); name: unused (kept for the callers' signatures)."""
    k, nn = (cout, cin) if dgrad else (cin, cout)
    if not (WINO43 and k >= 16 and nn >= WINO43_NMIN):
        return False
    return bool(dgrad or f43_fwd)


class FilterBank:
    wino_needed_in_bf16 = False      # set by the first lazy Winograd refresh inside the bf16 configuration (ensure_fresh)

    def __init__(self, named_filters, device, attrs=None):
        """named_filters: iterable of (name, [3,3,Cin,Cout] tensor views whose storage never moves); attrs: {name: {'f43_fwd': bool}}."""
        attrs = attrs or {}
        self.version = 0            # bumped by whoever writes the filters (Adam step, restore)
        self.synced = -1
        self.filters = [(n, w) for n, w in named_filters if w.dim() == 4 and w.shape[0] == 3 and w.shape[1] == 3]
        self.device = device
        if not self.filters or torch.device(device).type != 'cuda':
            self.filters = []
            return
        sizes = [lib.kpx_wino_u_bytes(int(w.shape[2]), int(w.shape[3])) // 4 for _, w in self.filters]
        sizes43 = [lib.kpx_wino43_u_bytes(int(w.shape[2]), int(w.shape[3])) // 4 for _, w in self.filters]
        want43 = [[_wino43_wanted(n, int(w.shape[2]), int(w.shape[3]), d, attrs.get(n, {}).get('f43_fwd', True)) for d in (0, 1)] for n, w in self.filters]
        self.arena = torch.empty(2 * sum(sizes) + sum(n43 * sum(wt) for n43, wt in zip(sizes43, want43)), dtype=torch.float32, device=device)
        table, table43, off = b'', b'', 0
        for (_, w), n, n43, wt in zip(self.filters, sizes, sizes43, want43):
            for dgrad in (0, 1):
                u = self.arena[off:off + n]
                off += n
                _wino_u[(w.data_ptr(), dgrad)] = (u, self, None, (int(w.shape[2]), int(w.shape[3])))
                table += _struct.pack('<QQiiii', w.data_ptr(), u.data_ptr(), int(w.shape[2]), int(w.shape[3]), dgrad, 0)
                if wt[dgrad]:
                    u = self.arena[off:off + n43]
                    off += n43
                    _wino43_u[(w.data_ptr(), dgrad)] = (u, self, None, (int(w.shape[2]), int(w.shape[3])))
                    table43 += _struct.pack('<QQiiii', w.data_ptr(), u.data_ptr(), int(w.shape[2]), int(w.shape[3]), dgrad, 0)
        self.n_desc = 2 * len(self.filters)
        self.table = torch.frombuffer(bytearray(table), dtype=torch.uint8).to(device)
        self.n_desc43 = len(table43) // 32
        self.table43 = torch.frombuffer(bytearray(table43), dtype=torch.uint8).to(device) if table43 else None
        # the bf16x3 form of the same (filter, direction) pairs where the kernel can take the layer
        want43b = [[wt[d] and _wino43b_form_wanted(*((int(w.shape[3]), int(w.shape[2])) if d else (int(w.shape[2]), int(w.shape[3])))) for d in (0, 1)]
                   for (_, w), wt in zip(self.filters, want43)]
        sizes43b = [lib.kpx_wino43b_u_bytes(int(w.shape[2]), int(w.shape[3])) for _, w in self.filters]
        self.arena43b = torch.empty(sum(nb * sum(wb) for nb, wb in zip(sizes43b, want43b)) or 1, dtype=torch.uint8, device=device)
        table43b, off = b'', 0
        for (_, w), nb, wb in zip(self.filters, sizes43b, want43b):
            for dgrad in (0, 1):
                if wb[dgrad]:
                    u = self.arena43b[off:off + nb]
                    off += nb
                    _wino43b_u[(w.data_ptr(), dgrad)] = (u, self, None, (int(w.shape[2]), int(w.shape[3])))
                    table43b += _struct.pack('<QQiiii', w.data_ptr(), u.data_ptr(), int(w.shape[2]), int(w.shape[3]), dgrad, 0)
        self.n_desc43b = len(table43b) // 32
        self.table43b = torch.frombuffer(bytearray(table43b), dtype=torch.uint8).to(device) if table43b else None

    def touch(self):
        self.version += 1

    def _build_bf16s(self):
        """The fragment-ordered bf16 copies of every filter for the bf16-storage kernel (both directions), one arena, one descriptor table."""
        sizes = [[lib.kpx_conv3x3_bf16s_weights_bytes(*((int(w.shape[3]), int(w.shape[2])) if d else (int(w.shape[2]), int(w.shape[3])))) for d in (0, 1)]
                 for _, w in self.filters]
        self.arena16 = torch.empty(sum(sum(sz) for sz in sizes), dtype=torch.uint8, device=self.device)
        table, off = b'', 0
        for (_, w), sz in zip(self.filters, sizes):
            cin, cout = int(w.shape[2]), int(w.shape[3])
            for dgrad in (0, 1):
                wf = self.arena16[off:off + sz[dgrad]]
                off += sz[dgrad]
                _bf16s_w[(w.data_ptr(), dgrad)] = (wf, self, None, (cin, cout))
                nn = cin if dgrad else cout
                table += _struct.pack('<QQiiii', w.data_ptr(), wf.data_ptr(), cin, cout, dgrad, (nn + 127) // 128 * 4)
        self.table16 = torch.frombuffer(bytearray(table), dtype=torch.uint8).to(self.device)
        self.n_desc16 = len(table) // 32

    def ensure_fresh(self, form=None):
        """Refresh the derived filter forms a convolution is about to read, if the filters changed since they were made.  form: 'wino' (the
        Winograd forms of the fp32 kernels), 'bf16s' (the prepared bf16 fragments), None = the form the current configuration's kernels use
        (the model calls this before it forks streams).  The other form stays stale until a kernel asks for it (an fp32 fall-back inside the bf16
        configuration): a bf16 step does not pay for the Winograd transforms."""
        if not self.filters:
            return
        if form is None:
            form = 'bf16s' if _compute_dtype[0] == 'bf16' else 'wino'
            if form == 'bf16s' and FilterBank.wino_needed_in_bf16:
                self.ensure_fresh('wino')                # an earlier step fell back to an fp32 Winograd kernel: refresh that form before the fork too
        if form == 'bf16s':
            if getattr(self, 'synced16', -1) != self.version:
                if getattr(self, 'table16', None) is None:
                    self._build_bf16s()
                check(lib.kpx_conv3x3_bf16s_prepare_batch_f32(self.table16.data_ptr(), self.n_desc16, _stream()), 'kpx_conv3x3_bf16s_prepare_batch_f32')
                self.synced16 = self.version
            return
        if self.synced != self.version:
            check(lib.kpx_wino_filter_transform_batch_f32(self.table.data_ptr(), self.n_desc, _stream()), 'kpx_wino_filter_transform_batch_f32')
            if self.n_desc43:
                check(lib.kpx_wino43_filter_transform_batch_f32(self.table43.data_ptr(), self.n_desc43, _stream()), 'kpx_wino43_filter_transform_batch_f32')
            if self.n_desc43b:
                check(lib.kpx_wino43b_filter_transform_batch_f32(self.table43b.data_ptr(), self.n_desc43b, _stream()), 'kpx_wino43b_filter_transform_batch_f32')
            self.synced = self.version
            # A LAZY refresh (an fp32 3x3 fall-back inside the bf16 configuration, whose pre-fork ensure_fresh() only refreshes the bf16
            # fragments) runs on whichever stream reaches it first: later readers on OTHER streams must wait for these launches
            self._wino_fresh_ev = None
            if _compute_dtype[0] == 'bf16' and not torch.cuda.is_current_stream_capturing():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._wino_fresh_ev = (ev, torch.cuda.current_stream().cuda_stream)
                FilterBank.wino_needed_in_bf16 = True      # from now on the model's pre-fork refresh covers the Winograd forms too
        elif getattr(self, '_wino_fresh_ev', None) is not None and torch.cuda.current_stream().cuda_stream != self._wino_fresh_ev[1]:
            torch.cuda.current_stream().wait_event(self._wino_fresh_ev[0])

    def keys(self):
        return [(w.data_ptr(), dgrad) for _, w in self.filters for dgrad in (0, 1)]

    def mode_key(self):
        return _compute_dtype[0]


def register_constant_filter(w, name='', f43_fwd=True):
    """Transform a filter that never changes (VGG19, reference vgg.py:57-61 tf.constant) once, for both directions.
    f43_fwd: whether the layer's forward may run F(4x4,3x3) (see WINO43 above).
    Returns the cache keys; the owner must pass them to release_filters() when the filter memory is given up."""
    if w.dim() != 4 or w.shape[0] != 3 or w.shape[1] != 3 or not w.is_cuda:
        return []
    n = lib.kpx_wino_u_bytes(int(w.shape[2]), int(w.shape[3])) // 4
    keys = []
    for dgrad in (0, 1):
        u = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib.kpx_wino_filter_transform_f32(w.data_ptr(), int(w.shape[2]), int(w.shape[3]), dgrad, u.data_ptr(), _stream()), 'kpx_wino_filter_transform_f32')
        _wino_u[(w.data_ptr(), dgrad)] = (u, None, _weakref.ref(w.untyped_storage()), (int(w.shape[2]), int(w.shape[3])))
        keys.append((w.data_ptr(), dgrad))
        if _wino43_wanted(name, int(w.shape[2]), int(w.shape[3]), dgrad, f43_fwd):
            u = torch.empty(lib.kpx_wino43_u_bytes(int(w.shape[2]), int(w.shape[3])) // 4, dtype=torch.float32, device=w.device)
            check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), int(w.shape[2]), int(w.shape[3]), dgrad, u.data_ptr(), _stream()), 'kpx_wino43_filter_transform_f32')
            _wino43_u[(w.data_ptr(), dgrad)] = (u, None, _weakref.ref(w.untyped_storage()), (int(w.shape[2]), int(w.shape[3])))
            kk, nn_ = (int(w.shape[3]), int(w.shape[2])) if dgrad else (int(w.shape[2]), int(w.shape[3]))
            if _wino43b_form_wanted(kk, nn_):
                ub = torch.empty(lib.kpx_wino43b_u_bytes(int(w.shape[2]), int(w.shape[3])), dtype=torch.uint8, device=w.device)
                check(lib.kpx_wino43b_filter_transform_f32(w.data_ptr(), int(w.shape[2]), int(w.shape[3]), dgrad, ub.data_ptr(), _stream()), 'kpx_wino43b_filter_transform_f32')
                _wino43b_u[(w.data_ptr(), dgrad)] = (ub, None, _weakref.ref(w.untyped_storage()), (int(w.shape[2]), int(w.shape[3])))
        if _compute_dtype[0] == 'bf16' and int(w.shape[2]) % 8 == 0:
            cin, cout = int(w.shape[2]), int(w.shape[3])
            k, nn = (cout, cin) if dgrad else (cin, cout)
            wf = torch.empty(lib.kpx_conv3x3_bf16s_weights_bytes(k, nn), dtype=torch.uint8, device=w.device)
            check(lib.kpx_conv3x3_bf16s_prepare_f32(w.data_ptr(), cin, cout, dgrad, wf.data_ptr(), _stream()), 'kpx_conv3x3_bf16s_prepare_f32')
            _bf16s_w[(w.data_ptr(), dgrad)] = (wf, None, _weakref.ref(w.untyped_storage()), (cin, cout))
    return keys


def release_filters(keys):
    """Forget cached Winograd forms (the address of a freed filter may be handed to an unrelated tensor later)."""
    for k in keys:
        _wino_u.pop(k, None)
        _wino43_u.pop(k, None)
        _wino43b_u.pop(k, None)
        _bf16s_w.pop(k, None)


def _wino_pretransformed(inp, ld_in, k, w, bias, out, ld_out, nn, act, dgrad, want_stats=False, bn_src=None):
    """Run the fused Winograd kernel on a cached U; False when there is none for this filter or the shape is not eligible.
    want_stats: also have the epilogue write the per-tile batch-norm sums of the output; returns (slab, tiles per image) then."""
    ent = _cached_u(_wino_u, w, dgrad)
    if ent is None:
        return False
    n, h, wd = inp.shape[0], inp.shape[1], inp.shape[2]
    if not lib.kpx_conv3x3_wino_eligible(n, h, wd, k, nn, ld_in, inp.data_ptr()):
        return False
    u, bank = ent[0], ent[1]
    if bank is not None:
        bank.ensure_fresh('wino')
    bptr = bias.data_ptr() if bias is not None else None
    ent43 = _cached_u(_wino43_u, w, dgrad) if WINO43 else None
    tiles = lib.kpx_conv3x3_wino43_stats_tiles(n, h, wd) if want_stats else 1      # 0: no statistics from this kernel for the shape (16 x 16)
    # (WINO43_MIN_WORKGROUPS: see its definition -- 0 by default)
    wgs43 = (n // 2 if wd == 16 else n * (h // 16) * (wd // 32)) * ((nn + 63) // 64)
    # the bf16x3 form of F(4x4,3x3) (same policy, same epilogue options) where it is prepared, eligible and the faster one
    ent43b = _cached_u(_wino43b_u, w, dgrad) if (WINO43B and ent43 is not None) else None
    if (ent43b is not None and tiles and wgs43 > WINO43_MIN_WORKGROUPS and _wino43b_preferred(n, h, wd, k, nn)
            and lib.kpx_conv3x3_wino43b_eligible(n, h, wd, k, nn, ld_in, inp.data_ptr())):
        bnbwd = bn_src is not None and dgrad and nn % 64 == 0 and wd != 16        # (packed 16 x 16 images: no statistics epilogue, as on the fp32-MFMA form)
        slab = None
        if bnbwd or want_stats:
            slab = torch.empty(lib.kpx_conv3x3_wino43_stats_tiles(n, h, wd) * 2 * nn, dtype=torch.float32, device=inp.device)
        rc = lib.kpx_conv3x3_wino43b_f32(inp.data_ptr(), n, h, wd, k, ld_in, ent43b[0].data_ptr(), None if bnbwd else bptr, out.data_ptr(), nn, ld_out,
                                         ACT_NONE if bnbwd else act, None, 0, None, 0, slab.data_ptr() if slab is not None else None,
                                         bn_src[0].data_ptr() if bnbwd else None, bn_src[0].shape[3] if bnbwd else 0, bn_src[1].data_ptr() if bnbwd else None, _stream())
        if rc != -1:
            check(rc, 'kpx_conv3x3_wino43b_f32')
            conv_kernel_uses['wino43'] += 1
            conv_kernel_uses['wino43b'] += 1
            return (slab, lib.kpx_conv3x3_wino43_stats_tiles(n, h, wd) // n) if slab is not None else True
    if (ent43 is not None and bn_src is not None and dgrad and wd != 16 and nn % 64 == 0 and wgs43 > WINO43_MIN_WORKGROUPS
            and lib.kpx_conv3x3_wino43_eligible(n, h, wd, k, nn, ld_in, inp.data_ptr())):
        # data gradient towards a ReLU'd batch norm's output on F(4x4,3x3): its epilogue masks the gradient and reduces the batch norm's backward sums
        bn_y, beta = bn_src
        tiles43 = lib.kpx_conv3x3_wino43_stats_tiles(n, h, wd)
        slab = torch.empty(tiles43 * 2 * nn, dtype=torch.float32, device=inp.device)
        rc = lib.kpx_conv3x3_wino43_bnbwd_stats_f32(inp.data_ptr(), n, h, wd, k, ld_in, ent43[0].data_ptr(), out.data_ptr(), nn, ld_out,
                                                    bn_y.data_ptr(), bn_y.shape[3], beta.data_ptr(), slab.data_ptr(), _stream())
        if rc != -1:
            check(rc, 'kpx_conv3x3_wino43_bnbwd_stats_f32')
            conv_kernel_uses['wino43'] += 1
            return slab, tiles43 // n
    # (a layer F(4x4,3x3) takes but whose shape its statistics epilogue does not -- 16x16 images, a produced-channel count that is not a
    #  multiple of 64 -- stays on F(4x4,3x3) without the sums: the batch norm then makes its own reduction pass)
    if (ent43 is not None and tiles and wgs43 > WINO43_MIN_WORKGROUPS
            and lib.kpx_conv3x3_wino43_eligible(n, h, wd, k, nn, ld_in, inp.data_ptr())):
        conv_kernel_uses['wino43'] += 1
        if want_stats:
            slab = torch.empty(tiles * 2 * nn, dtype=torch.float32, device=inp.device)
            check(lib.kpx_conv3x3_wino43_stats_f32(inp.data_ptr(), n, h, wd, k, ld_in, ent43[0].data_ptr(), bptr, out.data_ptr(), nn, ld_out, act,
                                                   slab.data_ptr(), _stream()), 'kpx_conv3x3_wino43_stats_f32')
            return slab, tiles // n
        check(lib.kpx_conv3x3_wino43_f32(inp.data_ptr(), n, h, wd, k, ld_in, ent43[0].data_ptr(), bptr, out.data_ptr(), nn, ld_out, act, _stream()),
              'kpx_conv3x3_wino43_f32')
        return True
    if bn_src is not None and dgrad and h % 16 == 0 and wd % 16 == 0:
        bn_y, beta = bn_src                              # the gathered tensor's consumer-side twin: y = relu(BN(.)) that this conv read
        tiles = lib.kpx_conv3x3_wino_stats_tiles(n, h, wd)
        slab = torch.empty(tiles * 2 * nn, dtype=torch.float32, device=inp.device)
        check(lib.kpx_conv3x3_wino_bnbwd_stats_f32(inp.data_ptr(), n, h, wd, k, ld_in, u.data_ptr(), out.data_ptr(), nn, ld_out,
                                                   bn_y.data_ptr(), bn_y.shape[3], beta.data_ptr(), slab.data_ptr(), _stream()),
              'kpx_conv3x3_wino_bnbwd_stats_f32')
        return slab, tiles // n
    if want_stats and h % 16 == 0 and wd % 16 == 0:
        tiles = lib.kpx_conv3x3_wino_stats_tiles(n, h, wd)
        slab = torch.empty(tiles * 2 * nn, dtype=torch.float32, device=inp.device)
        check(lib.kpx_conv3x3_wino_stats_f32(inp.data_ptr(), n, h, wd, k, ld_in, u.data_ptr(), bptr, out.data_ptr(), nn, ld_out, act,
                                             slab.data_ptr(), _stream()), 'kpx_conv3x3_wino_stats_f32')
        return slab, tiles // n
    check(lib.kpx_conv3x3_wino_f32(inp.data_ptr(), n, h, wd, k, ld_in, u.data_ptr(), bptr, out.data_ptr(), nn, ld_out, act, _stream()),
          'kpx_conv3x3_wino_f32')
    return True


def conv3x3_wino43_ex(inp, ld_in, k, w, bias, out, ld_out, nn, act, dgrad, mask=None, pool_out=None):
    """The F(4x4,3x3) kernel with VGG19's epilogue options: ``mask`` -- zero the output where mask <= 0 (ReLU backward of the tensor the
    data gradient belongs to), ``pool_out`` -- also write the 2x2 max-pool of the activated output.  True if it ran; False when the
    layer / launch is not one the kernel takes (the caller then uses the plain path and the separate kernels)."""
    ent43 = _cached_u(_wino43_u, w, dgrad) if WINO43 else None
    if ent43 is None or nn % 64:
        return False
    n, h, wd = inp.shape[0], inp.shape[1], inp.shape[2]
    wgs43 = (n // 2 if wd == 16 else n * (h // 16) * (wd // 32)) * ((nn + 63) // 64)
    if wgs43 <= WINO43_MIN_WORKGROUPS or not lib.kpx_conv3x3_wino43_eligible(n, h, wd, k, nn, ld_in, inp.data_ptr()):
        return False
    if ent43[1] is not None:
        ent43[1].ensure_fresh('wino')
    ent43b = _cached_u(_wino43b_u, w, dgrad) if WINO43B else None
    if (ent43b is not None and _wino43b_preferred(n, h, wd, k, nn) and lib.kpx_conv3x3_wino43b_eligible(n, h, wd, k, nn, ld_in, inp.data_ptr())):
        rc = lib.kpx_conv3x3_wino43b_f32(inp.data_ptr(), n, h, wd, k, ld_in, ent43b[0].data_ptr(), bias.data_ptr() if bias is not None else None,
                                         out.data_ptr(), nn, ld_out, act, mask.data_ptr() if mask is not None else None,
                                         mask.shape[3] if mask is not None else 0, pool_out.data_ptr() if pool_out is not None else None,
                                         pool_out.shape[3] if pool_out is not None else 0, None, None, 0, None, _stream())
        if rc != -1:
            check(rc, 'kpx_conv3x3_wino43b_f32')
            conv_kernel_uses['wino43'] += 1
            conv_kernel_uses['wino43b'] += 1
            return True
    rc = lib.kpx_conv3x3_wino43_ex_f32(inp.data_ptr(), n, h, wd, k, ld_in, ent43[0].data_ptr(), bias.data_ptr() if bias is not None else None,
                                       out.data_ptr(), nn, ld_out, act, mask.data_ptr() if mask is not None else None,
                                       mask.shape[3] if mask is not None else 0, pool_out.data_ptr() if pool_out is not None else None,
                                       pool_out.shape[3] if pool_out is not None else 0, _stream())
    if rc == -1:                                         # KPX_EINVAL: alignment / channel-count preconditions of the options
        return False
    check(rc, 'kpx_conv3x3_wino43_ex_f32')
    conv_kernel_uses['wino43'] += 1
    return True


# ----------------------------------------------------------------------------------------------- raw launchers
def conv_fwd_raw(x, ldx, cin, w, bias, y, ldy, stride, pad_t, pad_l, act, want_stats=False):
    """Returns None, or (tile-statistics slab, tiles per image) when ``want_stats`` and the layer ran on the kernel that provides them."""
    n, hi, wi = x.shape[0], x.shape[1], x.shape[2]
    kh, kw, _, cout = w.shape
    if x.dtype == BF16 or y.dtype == BF16:
        # bf16 configuration: the bf16-storage kernel, or (shapes it does not take, image-input layers) the fp32 kernel between conversions
        if (x.dtype == BF16 and kh == 3 and kw == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and act != ACT_TANH and y.shape[1] == hi and y.shape[2] == wi):
            r = _bf16s_conv(x, ldx, cin, w, bias, y, ldy, cout, act, False, want_stats=want_stats)
            if r:
                return r if isinstance(r, tuple) else None
        if x.dtype == torch.float32 and y.dtype == BF16 and cin <= 4 and ldx == cin:
            # image-input layers: the LDS-resident image kernel writes bf16
            rc = lib.kpx_conv_image_fwd_bf16(x.data_ptr(), n, hi, wi, cin, w.data_ptr(), kh, kw, bias.data_ptr() if bias is not None else None,
                                             y.data_ptr(), y.shape[1], y.shape[2], cout, ldy, stride, pad_t, pad_l, act, _stream())
            if rc != -1:
                check(rc, 'kpx_conv_image_fwd_bf16')
                return None
        if x.dtype == BF16 and act != ACT_TANH:
            # strided / 4x4 / 1x1 layers: the gather kernel of the bf16 matrix pipe on bf16 tensors
            nbytes = lib.kpx_conv2d_fwd_workspace_bytes(n, y.shape[1], y.shape[2], cin, cout, kh, kw)
            ws = scratch.get('splitk', nbytes, x.device) if nbytes else None
            rc = lib.kpx_conv2d_fwd_bf16(x.data_ptr(), n, hi, wi, cin, ldx, w.data_ptr(), kh, kw, bias.data_ptr() if bias is not None else None,
                                         y.data_ptr(), 1 if y.dtype == torch.float32 else 0, y.shape[1], y.shape[2], cout, ldy, stride, pad_t, pad_l, act,
                                         ws.data_ptr() if ws is not None else None, nbytes, _stream())
            if rc != -1:
                check(rc, 'kpx_conv2d_fwd_bf16')
                return None
        if x.dtype == BF16:
            fallback_uses['conv_fwd'] += 1
            x, ldx = _slice_f32(x, ldx, cin), cin
        yf = y if y.dtype == torch.float32 else torch.empty((n, y.shape[1], y.shape[2], cout), dtype=torch.float32, device=x.device)
        conv_fwd_raw(x, ldx, cin, w, bias, yf, ldy if yf is y else cout, stride, pad_t, pad_l, act)
        if yf is not y:
            cast_channels_raw(yf.data_ptr(), cout, y.data_ptr(), ldy, n * y.shape[1] * y.shape[2], cout, 0)
        return None
    if (kh == 3 and kw == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and act != ACT_TANH and cout == 16 and y.shape[1] == hi and y.shape[2] == wi
            and lib.kpx_conv3x3_c16_eligible(n, hi, wi, cin, cout, ldx, ldy, x.data_ptr())):
        # exactly 16 produced channels: 16x16x4 MFMA blocks, no cout padding (csrc/conv_c16.hip)
        slab = torch.empty(n * (hi // 16) * (wi // 16) * 2 * 16, dtype=torch.float32, device=x.device) if want_stats else None
        check(lib.kpx_conv3x3_c16_f32(x.data_ptr(), n, hi, wi, cin, ldx, w.data_ptr(), 0, bias.data_ptr() if bias is not None else None,
                                      y.data_ptr(), ldy, act, slab.data_ptr() if slab is not None else None, _stream()), 'kpx_conv3x3_c16_f32')
        return (slab, (hi // 16) * (wi // 16)) if want_stats else None
    if (kh == 3 and kw == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and act != ACT_TANH and y.shape[1] == hi and y.shape[2] == wi
            and not (cout <= 4 and cin <= 128 and n * hi * wi >= 65536)):        # (few produced channels over a large image: VALU kernel inside the library)
        r = _wino_pretransformed(x, ldx, cin, w, bias, y, ldy, cout, act, False, want_stats=want_stats)
        if r:
            return r if isinstance(r, tuple) else None
    nbytes = lib.kpx_conv2d_fwd_workspace_bytes(n, y.shape[1], y.shape[2], cin, cout, kh, kw)
    ws = scratch.get('splitk', nbytes, x.device) if nbytes else None
    check(lib.kpx_conv2d_fwd_f32(x.data_ptr(), n, hi, wi, cin, ldx, w.data_ptr(), kh, kw,
                                 bias.data_ptr() if bias is not None else None,
                                 y.data_ptr(), y.shape[1], y.shape[2], cout, ldy, stride, pad_t, pad_l, act, _arith(),
                                 ws.data_ptr() if ws is not None else None, nbytes, _stream()),
          'kpx_conv2d_fwd_f32')


def conv_dgrad_raw(dy, lddy, w, dx, lddx, cin, stride, pad_t, pad_l, bn_src=None, mul=None):
    """bn_src = (y, beta) of the ReLU'd batch norm whose output this convolution read: when the layer runs on the fused Winograd kernel
    its epilogue also reduces that batch norm's backward sums; returns (slab, tiles per image) then, else None.
    mul = (y_in, act): dx is the gradient of the ACTIVATED tensor y_in (the convolution's input); its activation backward is applied in the
    data-gradient epilogue (kpx_conv2d_dgrad_act_f32) -- the producer of y_in then skips its own pass."""
    n, ho, wo = dy.shape[0], dy.shape[1], dy.shape[2]
    kh, kw, _, cout = w.shape
    if dy.dtype == BF16 or dx.dtype == BF16:
        if dy.dtype == BF16 and mul is None and kh == 3 and kw == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and dx.shape[1] == ho and dx.shape[2] == wo:
            if bn_src is not None and dx.dtype == BF16:
                # towards a ReLU'd batch norm's output: the epilogue gates the gradient and reduces that batch norm's backward sums
                r = _bf16s_bnbwd(dy, lddy, cout, w, dx, lddx, cin, bn_src[0], bn_src[1])
                if r:
                    return r
            if _bf16s_conv(dy, lddy, cout, w, None, dx, lddx, cin, ACT_NONE, True):
                return None
        if dy.dtype == BF16 and (mul is None or mul[0].dtype == BF16):
            nbytes = lib.kpx_conv2d_dgrad_workspace_bytes(n, dx.shape[1], dx.shape[2], cin, cout, kh, kw, stride)
            ws = scratch.get('splitk', nbytes, dy.device) if nbytes else None
            rc = lib.kpx_conv2d_dgrad_bf16(dy.data_ptr(), n, ho, wo, cout, lddy, w.data_ptr(), kh, kw, dx.data_ptr(), 1 if dx.dtype == torch.float32 else 0,
                                           dx.shape[1], dx.shape[2], cin, lddx, stride, pad_t, pad_l,
                                           mul[0].data_ptr() if mul is not None else None, mul[0].shape[3] if mul is not None else 0, mul[1] if mul is not None else 0,
                                           ws.data_ptr() if ws is not None else None, nbytes, _stream())
            if rc != -1:
                check(rc, 'kpx_conv2d_dgrad_bf16')
                return None
        if dy.dtype == BF16 and dx.dtype == torch.float32 and cin <= 4 and mul is None:
            # gradient towards an image: the image kernel reads the bf16 gradient
            rc = lib.kpx_conv_image_dgrad_bf16(dy.data_ptr(), n, ho, wo, cout, lddy, w.data_ptr(), kh, kw, dx.data_ptr(), dx.shape[1], dx.shape[2], cin, lddx,
                                               stride, pad_t, pad_l, _stream())
            if rc != -1:
                check(rc, 'kpx_conv_image_dgrad_bf16')
                return None
        if dy.dtype == BF16:
            fallback_uses['conv_dgrad'] += 1
            dy, lddy = _slice_f32(dy, lddy, cout), cout
        dxf = dx if dx.dtype == torch.float32 else torch.empty((n, dx.shape[1], dx.shape[2], cin), dtype=torch.float32, device=dy.device)
        if mul is not None and mul[0].dtype == BF16:
            mul = (cast(mul[0], torch.float32), mul[1])
        conv_dgrad_raw(dy, lddy, w, dxf, lddx if dxf is dx else cin, cin, stride, pad_t, pad_l, mul=mul)
        if dxf is not dx:
            cast_channels_raw(dxf.data_ptr(), cin, dx.data_ptr(), lddx, n * dx.shape[1] * dx.shape[2], cin, 0)
        return None
    if mul is not None:
        y_in, act_in = mul
        nbytes = lib.kpx_conv2d_dgrad_workspace_bytes(n, dx.shape[1], dx.shape[2], cin, cout, kh, kw, stride)
        ws = scratch.get('splitk', nbytes, dy.device) if nbytes else None
        check(lib.kpx_conv2d_dgrad_act_f32(dy.data_ptr(), n, ho, wo, cout, lddy, w.data_ptr(), kh, kw,
                                           dx.data_ptr(), dx.shape[1], dx.shape[2], cin, lddx, stride, pad_t, pad_l, _arith(),
                                           y_in.data_ptr(), y_in.shape[3], act_in,
                                           ws.data_ptr() if ws is not None else None, nbytes, _stream()), 'kpx_conv2d_dgrad_act_f32')
        return
    if (kh == 3 and kw == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and cin == 16 and bn_src is None and dx.shape[1] == ho and dx.shape[2] == wo
            and lib.kpx_conv3x3_c16_eligible(n, ho, wo, cout, cin, lddy, lddx, dy.data_ptr())):
        check(lib.kpx_conv3x3_c16_f32(dy.data_ptr(), n, ho, wo, cout, lddy, w.data_ptr(), 1, None, dx.data_ptr(), lddx, ACT_NONE, None, _stream()),
              'kpx_conv3x3_c16_f32')
        return
    if kh == 3 and kw == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and dx.shape[1] == ho and dx.shape[2] == wo:
        r = _wino_pretransformed(dy, lddy, cout, w, None, dx, lddx, cin, ACT_NONE, True, bn_src=bn_src)
        if r:
            return r if isinstance(r, tuple) else None
    nbytes = lib.kpx_conv2d_dgrad_workspace_bytes(n, dx.shape[1], dx.shape[2], cin, cout, kh, kw, stride)
    ws = scratch.get('splitk', nbytes, dy.device) if nbytes else None
    check(lib.kpx_conv2d_dgrad_f32(dy.data_ptr(), n, ho, wo, cout, lddy, w.data_ptr(), kh, kw,
                                   dx.data_ptr(), dx.shape[1], dx.shape[2], cin, lddx, stride, pad_t, pad_l, _arith(),
                                   ws.data_ptr() if ws is not None else None, nbytes, _stream()),
          'kpx_conv2d_dgrad_f32')


def conv_wgrad_raw(x, ldx, cin, dy, lddy, dw, stride, pad_t, pad_l):
    n, hi, wi = x.shape[0], x.shape[1], x.shape[2]
    ho, wo = dy.shape[1], dy.shape[2]
    kh, kw, _, cout = dw.shape
    if (x.dtype == BF16 and dy.dtype == BF16 and kh == 3 and kw == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and ho == hi and wo == wi
            and lib.kpx_conv3x3_wgrad_bf16_eligible(n, hi, wi, cin, ldx, cout, lddy, x.data_ptr(), dy.data_ptr())):
        nbytes = lib.kpx_conv3x3_wgrad_bf16_workspace_bytes(n, hi, wi, cin, cout)
        ws = scratch.get('wgrad', nbytes, x.device)
        check(lib.kpx_conv3x3_wgrad_bf16(x.data_ptr(), n, hi, wi, cin, ldx, dy.data_ptr(), cout, lddy, dw.data_ptr(), ws.data_ptr(), nbytes, _stream()),
              'kpx_conv3x3_wgrad_bf16')
        return
    if x.dtype == BF16 and dy.dtype == BF16:
        nbytes = lib.kpx_conv2d_wgrad_workspace_bytes(n, ho, wo, cin, cout, kh, kw)
        ws = scratch.get('wgrad', nbytes, x.device) if nbytes else None
        rc = lib.kpx_conv2d_wgrad_bf16(x.data_ptr(), n, hi, wi, cin, ldx, dy.data_ptr(), ho, wo, cout, lddy, dw.data_ptr(), kh, kw, stride, pad_t, pad_l,
                                       ws.data_ptr() if ws is not None else None, nbytes, _stream())
        if rc != -1:
            check(rc, 'kpx_conv2d_wgrad_bf16')
            return
    if x.dtype == torch.float32 and dy.dtype == BF16 and cin <= 4:
        # image-input layers: fp32 image, bf16 gradient
        nbytes = lib.kpx_conv2d_wgrad_workspace_bytes(n, ho, wo, cin, cout, kh, kw)
        ws = scratch.get('wgrad', nbytes, x.device) if nbytes else None
        rc = lib.kpx_conv_image_wgrad_bf16(x.data_ptr(), n, hi, wi, cin, ldx, dy.data_ptr(), ho, wo, cout, lddy, dw.data_ptr(), kh, kw, stride, pad_t, pad_l,
                                           ws.data_ptr() if ws is not None else None, nbytes, _stream())
        if rc != -1:
            check(rc, 'kpx_conv_image_wgrad_bf16')
            return
    if x.dtype == BF16 or dy.dtype == BF16:
        fallback_uses['conv_wgrad'] += 1
        if x.dtype == BF16:
            x, ldx = _slice_f32(x, ldx, cin), cin
        if dy.dtype == BF16:
            dy, lddy = _slice_f32(dy, lddy, cout), cout
    nbytes = lib.kpx_conv2d_wgrad_workspace_bytes(n, ho, wo, cin, cout, kh, kw)
    ws = scratch.get('wgrad', nbytes, x.device) if nbytes else None
    check(lib.kpx_conv2d_wgrad_f32(x.data_ptr(), n, hi, wi, cin, ldx, dy.data_ptr(), ho, wo, cout, lddy,
                                   dw.data_ptr(), kh, kw, stride, pad_t, pad_l, _arith(),
                                   ws.data_ptr() if ws is not None else None, nbytes, _stream()),
          'kpx_conv2d_wgrad_f32')


def chan_sum_raw(x, ldx, pixels, c, out):
    if x.dtype == BF16:
        if c % 8 == 0 and ldx % 8 == 0 and x.data_ptr() % 16 == 0:
            check(lib.kpx_chan_sum_bf16(x.data_ptr(), pixels, c, ldx, out.data_ptr(), scratch.reduce(c, x.device).data_ptr(), _stream()), 'kpx_chan_sum_bf16')
            return
        fallback_uses['other'] += 1
        xf = torch.empty((pixels, c), dtype=torch.float32, device=x.device)
        cast_channels_raw(x.data_ptr(), ldx, xf.data_ptr(), c, pixels, c, 1)
        x, ldx = xf, c
    check(lib.kpx_chan_sum_f32(x.data_ptr(), pixels, c, ldx, out.data_ptr(), scratch.reduce(c, x.device).data_ptr(), _stream()),
          'kpx_chan_sum_f32')


def act_bwd_raw(dy, y, dz, act):
    """dz = dy * act'(y); dz may be dy itself (in place)."""
    if dy.dtype == BF16:
        assert y.dtype == BF16 and dz.dtype == BF16 and dy.numel() % 8 == 0
        check(lib.kpx_act_bwd_bf16(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), act, _stream()), 'kpx_act_bwd_bf16')
        return
    check(lib.kpx_act_bwd_f32(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), act, _stream()), 'kpx_act_bwd_f32')


def act_bwd_raw_(dy, y, act):
    act_bwd_raw(dy, y, dy, act)


def axpy_raw_(y, x, a=1.0):
    check(lib.kpx_axpy_f32(y.data_ptr(), x.data_ptr(), y.numel(), a, _stream()), 'kpx_axpy_f32')


def fill_raw_(t, v=0.0):
    if t.dtype == BF16:
        assert v == 0.0 and t.numel() % 2 == 0, 'bf16 tensors are only ever zero-filled'
        check(lib.kpx_fill_f32(t.data_ptr(), t.numel() // 2, 0.0, _stream()), 'kpx_fill_f32')
        return
    check(lib.kpx_fill_f32(t.data_ptr(), t.numel(), v, _stream()), 'kpx_fill_f32')


def copy_channels_raw(src_ptr, ldsrc, dst_ptr, lddst, pixels, c):
    check(lib.kpx_copy_channels_f32(src_ptr, ldsrc, dst_ptr, lddst, pixels, c, _stream()), 'kpx_copy_channels_f32')


def flat_copy_tensor(src, dst_ptr):
    """dst[0:numel] = src (contiguous, fp32 or bf16) as a raw copy."""
    nbytes = src.numel() * _esz(src)
    assert nbytes % 4 == 0
    flat_copy_raw(src.data_ptr(), dst_ptr, nbytes // 4)


def flat_copy_raw(src_ptr, dst_ptr, n):
    """dst[0:n] = src[0:n] (floats): a degenerate channel copy, 16 B per lane when alignment allows."""
    if n % 4 == 0 and src_ptr % 16 == 0 and dst_ptr % 16 == 0:
        copy_channels_raw(src_ptr, 4, dst_ptr, 4, n // 4, 4)
    else:
        copy_channels_raw(src_ptr, 1, dst_ptr, 1, n, 1)


# ----------------------------------------------------------------------------------------------- conv
class Conv2dFn(torch.autograd.Function):
    """layers.conv: tf.pad(pad) + conv2d(SAME) + bias [+ activation] (reference models/networks/layers.py:4-10)."""

    @staticmethod
    def forward(ctx, x, w, b, w_grad_out, b_grad_out, stride, pad, act, cin, bias_grad, bn_stats=False, bn_src=None, input_act=ACT_NONE,
                act_bwd_by_consumer=False, out_dtype=None):
        # input_act: x is the activated output of a layer declared with act_bwd_by_consumer=True -- this layer's data gradient applies that
        # activation's backward in its epilogue (it saves x anyway) and the producer skips its own pass.  A static contract between the two
        # layers, declared where the network is built (networks.img_discr): x must have NO other consumer.
        x, ldx = _nhwc(x)
        ctx.input_act = input_act if (input_act != ACT_NONE and x.is_contiguous() and (cin is None or cin == x.shape[3])) else ACT_NONE
        if input_act != ACT_NONE and ctx.input_act == ACT_NONE:
            raise _lib.KpxError('input_act needs the whole contiguous activated tensor as the convolution input')
        ctx.act_bwd_by_consumer = bool(act_bwd_by_consumer) and act != ACT_NONE
        ctx.bn_src = bn_src if (bn_src is not None and x.is_contiguous() and (cin is None or cin == x.shape[3])) else None
        _require_gpu(w)
        w = w.contiguous()
        n, h, wd, cx = x.shape
        kh, kw, wcin, cout = w.shape
        cin = cx if cin is None else cin
        assert wcin == cin and cin <= cx
        pt, _, ho = same_pad(h + 2 * pad, kh, stride)
        pl, _, wo = same_pad(wd + 2 * pad, kw, stride)
        pad_t, pad_l = pad + pt, pad + pl
        y = torch.empty((n, ho, wo, cout), dtype=out_dtype if out_dtype is not None else act_dtype(), device=x.device)
        st = conv_fwd_raw(x, ldx, cin, w, b, y, cout, stride, pad_t, pad_l, act, want_stats=bn_stats and act == ACT_NONE)
        if st is not None:
            _pending_stats[y.data_ptr()] = st
        ctx.geom = (stride, pad_t, pad_l, act, cin, ldx)
        ctx.has_bias = b is not None and bias_grad
        ctx.w_grad_out, ctx.b_grad_out = w_grad_out, b_grad_out
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad_t, pad_l, act, cin, ldx = ctx.geom
        dy = dy.contiguous()
        cout = w.shape[3]
        if act != ACT_NONE and not ctx.act_bwd_by_consumer:
            dz = torch.empty_like(dy)          # the incoming gradient tensor is not ours to overwrite
            act_bwd_raw(dy, y, dz, act)
            dy = dz
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        dy_sum = dy
        if (dy.dtype == torch.float32 and x.dtype == BF16 and cout < 8 and w.shape[0] == 3 and w.shape[1] == 3 and stride == 1
                and ctx.input_act == ACT_NONE):
            # the translator's 4-channel head: its fp32 gradient as a zero-padded 8-channel bf16 operand of the bf16 3x3 kernels
            dy16 = torch.empty((dy.shape[0], dy.shape[1], dy.shape[2], 8), dtype=BF16, device=dy.device)
            cast_channels_raw(dy.data_ptr(), cout, dy16.data_ptr(), 8, dy.shape[0] * dy.shape[1] * dy.shape[2], cout, 3)
            dy = dy16
        lddy = dy.shape[3]
        # side stream only when the gradient goes straight into the flat bucket (nobody on the main stream reads it
        # before join_side_stream())
        side = (SIDE_WGRAD and not _inline_wgrad[0] and (want_w or want_b) and (not want_w or ctx.w_grad_out is not None)
                and (not want_b or ctx.b_grad_out is not None))
        fork = None
        if side and (FORK_BEFORE_DGRAD == 'before' or (FORK_BEFORE_DGRAD == 'capture' and torch.cuda.is_current_stream_capturing())):
            # The weight gradient needs dy and x, not dx: the fork point is recorded BEFORE the data gradient is launched, so the two run
            # side by side -- and, in a captured graph, the data-gradient chain stays the first successor of its predecessor (the graph
            # runtime continues a queue along the first successor; forking after the data gradient made the chain hop queues per layer).
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(x.device))
        if ctx.needs_input_grad[0]:
            cx = x.shape[3]
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            if cin < cx:
                fill_raw_(dx, 0.0)
            st = conv_dgrad_raw(dy, lddy, w, dx, cx, cin, stride, pad_t, pad_l,
                                bn_src=(x, ctx.bn_src[0]) if ctx.bn_src is not None else None,
                                mul=(x, ctx.input_act) if ctx.input_act != ACT_NONE else None)
            if st is not None:
                # (dx itself is kept: a second reference stops the autograd engine from accumulating another consumer's gradient INTO this
                #  buffer in place -- the sums would then belong to a part of the gradient only; a sum allocates a new tensor, whose address
                #  does not match, and the batch norm falls back to its own reduction pass)
                _pending_bwd_stats[dx.data_ptr()] = (st[0], st[1], ctx.bn_src[1], dx)
        if side:
            main = torch.cuda.current_stream(x.device)   # the stream this backward node runs on (= its forward's stream)
            skey, st = _side_stream(x.device, main.cuda_stream)
            if fork is not None:
                st.wait_event(fork)                      # dy (and x) were ready when the fork point was recorded
            else:
                st.wait_stream(main)                     # ... or wait for everything enqueued so far, the data gradient included
            _side_keep.append((dy, dy_sum, x))                   # keep the allocator from recycling them under the side kernels
            _side_dirty[skey] = True
            stream_ctx = torch.cuda.stream(st)
        else:
            import contextlib
            stream_ctx = contextlib.nullcontext()
        with stream_ctx:
            if want_w:
                direct = ctx.w_grad_out is not None and _claim_grad(ctx.w_grad_out)
                dw_buf = ctx.w_grad_out if direct else torch.empty_like(w)
                conv_wgrad_raw(x, ldx, cin, dy, lddy, dw_buf, stride, pad_t, pad_l)
                if ctx.w_grad_out is not None and not direct:
                    axpy_raw_(ctx.w_grad_out, dw_buf)                 # second use of this variable in one backward
                dw = None if ctx.w_grad_out is not None else dw_buf
            if want_b:
                direct = ctx.b_grad_out is not None and _claim_grad(ctx.b_grad_out)
                db_buf = ctx.b_grad_out if direct else torch.empty(cout, dtype=torch.float32, device=x.device)
                chan_sum_raw(dy_sum, cout, dy.shape[0] * dy.shape[1] * dy.shape[2], cout, db_buf)
                if ctx.b_grad_out is not None and not direct:
                    axpy_raw_(ctx.b_grad_out, db_buf)
                db = None if ctx.b_grad_out is not None else db_buf
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None


_pending_stats = {}      # output data_ptr -> (tile-statistics slab, tiles per image), handed from Conv2dFn.forward to conv2d()
_pending_bwd_stats = {}  # dx data_ptr -> (slab, tiles per image, id of the batch norm it belongs to), from Conv2dFn.backward to BatchNormFn.backward
_bn_counter = [0]
fused_bn_uses = {'stats_from_conv_epilogue': 0, 'backward_sums_from_dgrad_epilogue': 0}    # diagnostics: how often the fused paths ran


def conv2d(x, w, b=None, stride=1, pad=0, act=ACT_NONE, cin=None, w_grad_out=None, b_grad_out=None, bias_grad=True, bn_stats=False,
           input_act=ACT_NONE, act_bwd_by_consumer=False, out_dtype=None):
    """out_dtype: storage type of the output (default: the configuration's activation type; torch.float32 keeps a tensor fp32 in the bf16
    configuration -- the translator's 4-channel head, the discriminator's logits).
    bias_grad=False: the bias gradient is known to be exactly zero (conv feeding a batch norm) and is not computed.
    bn_stats=True: a train-mode batch norm consumes the output next; when the layer runs on the fused Winograd kernel its epilogue
    also writes the per-tile channel sums, which ``batch_norm`` then uses instead of a statistics pass over the activation."""
    y = Conv2dFn.apply(x, w, b, w_grad_out, b_grad_out, stride, pad, act, cin, bias_grad, bn_stats and FUSE_BN_STATS,
                       getattr(x, '_kpx_bn', None) if FUSE_BN_BWD else None, input_act if FUSE_ACT_BWD else ACT_NONE,
                       act_bwd_by_consumer and FUSE_ACT_BWD, out_dtype)
    if bn_stats:
        st = _pending_stats.pop(y.data_ptr(), None)
        if st is not None:
            y._kpx_tile_stats = st
    return y


# ----------------------------------------------------------------------------------------------- batch norm
def _whole_tiles_per_group(n, groups, tiles_per_image):
    """True when every statistics tile of a producing convolution lies inside ONE batch-norm group.  The bf16 3x3 kernel packs G images into a
    tile on 8x8 / narrow 16x16 layers (tiles per image = 1/G, a Fraction): with n / groups not a multiple of G a tile would straddle two
    groups, and int(ng * tpi) would silently drop or misattribute images."""
    from fractions import Fraction
    return n % groups == 0 and (Fraction(tiles_per_image) * (n // groups)).denominator == 1


class BatchNormFn(torch.autograd.Function):
    """layers.batch_norm (+ fused ReLU) -- tf.contrib.layers.batch_norm(eps=1e-5) (reference layers.py:13-14).

    ``groups`` > 1 computes separate batch statistics for consecutive batch slices: this is how the two weight-sharing
    pose_encoder calls of the reference (detector_translator_model.py:166-167, separate BN statistics per call) run as
    ONE launch over the concatenated batch.  Moving statistics are updated once per group, in order.
    """

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, g_grad_out, b_grad_out, train, act, groups, update_moving, tile_stats=None, bn_id=0, out_f32=False):
        ctx.bn_id = bn_id
        _require_gpu(x)
        x = x.contiguous()
        n, h, w, c = x.shape
        assert n % groups == 0
        dev = x.device
        if x.dtype == BF16 and not (train and c % 8 == 0):
            fallback_uses['other'] += 1                  # (inference-mode / ragged batch norm on bf16 tensors: through the fp32 kernels)
            x = cast(x, torch.float32)
        ctx.x16 = x.dtype == BF16
        y = torch.empty(x.shape, dtype=torch.float32 if (out_f32 or not ctx.x16) else BF16, device=dev)
        if train and ctx.x16:
            mean = torch.empty((groups, c), dtype=torch.float32, device=dev)
            invstd = torch.empty((groups, c), dtype=torch.float32, device=dev)
            ng = n // groups
            pix = ng * h * w
            sc = scratch.get('bn%d' % groups, lib.kpx_bn_train_scratch_bytes(c, groups), dev)
            slab_ptr, tpg = None, 0
            if tile_stats is not None:
                slab, tpi = tile_stats
                fused_bn_uses['stats_from_conv_epilogue'] += groups
                slab_ptr, tpg = slab.data_ptr(), int(ng * tpi)
            check(lib.kpx_bn_train_fwd_bf16(x.data_ptr(), pix, groups, c, c, slab_ptr, tpg, BN_EPS, gamma.data_ptr(), beta.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), moving_mean.data_ptr() if update_moving else None,
                                            moving_var.data_ptr() if update_moving else None, BN_DECAY, y.data_ptr(), c, 1 if y.dtype == torch.float32 else 0, act,
                                            sc.data_ptr(), _stream()), 'kpx_bn_train_fwd_bf16')
        elif train:
            # all `groups` weight-sharing calls in one launch per phase (statistics finalize + apply; + one reduction pass when the
            # producing convolution's epilogue did not deliver the per-tile sums)
            mean = torch.empty((groups, c), dtype=torch.float32, device=dev)
            invstd = torch.empty((groups, c), dtype=torch.float32, device=dev)
            ng = n // groups
            pix = ng * h * w
            sc = scratch.get('bn%d' % groups, lib.kpx_bn_train_scratch_bytes(c, groups), dev)
            slab_ptr, tpg = None, 0
            if tile_stats is not None:                   # sums from the producing convolution's epilogue: no pass over x
                slab, tpi = tile_stats
                fused_bn_uses['stats_from_conv_epilogue'] += groups
                slab_ptr, tpg = slab.data_ptr(), int(ng * tpi)
            check(lib.kpx_bn_train_fwd_f32(x.data_ptr(), pix, groups, c, c, slab_ptr, tpg, BN_EPS, gamma.data_ptr(), beta.data_ptr(),
                                           mean.data_ptr(), invstd.data_ptr(), moving_mean.data_ptr() if update_moving else None,
                                           moving_var.data_ptr() if update_moving else None, BN_DECAY, y.data_ptr(), c, act, sc.data_ptr(), _stream()),
                  'kpx_bn_train_fwd_f32')
        else:
            mean = moving_mean.reshape(1, c)
            invstd = torch.empty((1, c), dtype=torch.float32, device=dev)
            check(lib.kpx_bn_invstd_f32(moving_var.data_ptr(), c, BN_EPS, invstd.data_ptr(), _stream()), 'kpx_bn_invstd_f32')
            check(lib.kpx_bn_apply_f32(x.data_ptr(), n * h * w, c, c, mean.data_ptr(), invstd.data_ptr(),
                                       gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), c, act, _stream()), 'kpx_bn_apply_f32')
        ctx.train, ctx.act, ctx.groups = train, act, groups
        ctx.g_grad_out, ctx.b_grad_out = g_grad_out, b_grad_out
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.train:
            raise _lib.KpxError('backward through inference-mode batch norm is not part of the hot path')
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        n, h, w, c = x.shape
        groups = ctx.groups
        ng = n // groups
        pix = ng * h * w
        dev = x.device
        dx = torch.empty_like(x)                         # (the batch norm's input type: bf16 in the bf16 configuration)
        dg = ctx.g_grad_out if ctx.g_grad_out is not None else torch.empty(c, dtype=torch.float32, device=dev)
        db = ctx.b_grad_out if ctx.b_grad_out is not None else torch.empty(c, dtype=torch.float32, device=dev)
        fresh = True
        if ctx.g_grad_out is not None:
            fresh = _claim_grad(ctx.g_grad_out)
            _claim_grad(ctx.b_grad_out)
        ent = _pending_bwd_stats.pop(dy.data_ptr(), None)        # sums reduced by the epilogue of the dgrad kernel that produced dy
        if ent is not None and (ent[2] != ctx.bn_id or ctx.act != ACT_RELU or tuple(ent[0].shape) != (n * ent[1] * 2 * c,) or not _whole_tiles_per_group(n, groups, ent[1])):
            ent = None
        # (reduction,) finalize, apply: one launch each for all groups; with `ent` the reduction was done by the epilogue of the data-gradient
        # kernel that produced dy (per-tile sums, ng * tiles-per-image tiles per group)
        sc = scratch.get('bn%d' % groups, lib.kpx_bn_train_scratch_bytes(c, groups), dev)
        if ctx.x16:
            if ent is not None:
                fused_bn_uses['backward_sums_from_dgrad_epilogue'] += groups
            check(lib.kpx_bn_train_bwd_bf16(dy.data_ptr(), c, 1 if dy.dtype == torch.float32 else 0, x.data_ptr(), c, pix, groups, c, mean.data_ptr(), invstd.data_ptr(),
                                            gamma.data_ptr(), beta.data_ptr(), ctx.act, dx.data_ptr(), c, dg.data_ptr(), db.data_ptr(), 0 if fresh else 1,
                                            ent[0].data_ptr() if ent is not None else None, int(ng * ent[1]) if ent is not None else 0,
                                            sc.data_ptr(), _stream()), 'kpx_bn_train_bwd_bf16')
            return (dx, None if ctx.g_grad_out is not None else dg, None if ctx.b_grad_out is not None else db,
                    None, None, None, None, None, None, None, None, None, None, None)
        if dy.dtype == BF16:
            dy = cast(dy, torch.float32)
        if ent is not None:
            fused_bn_uses['backward_sums_from_dgrad_epilogue'] += groups
        check(lib.kpx_bn_train_bwd_f32(dy.data_ptr(), c, x.data_ptr(), c, pix, groups, c, mean.data_ptr(), invstd.data_ptr(),
                                       gamma.data_ptr(), beta.data_ptr(), ctx.act, dx.data_ptr(), c, dg.data_ptr(), db.data_ptr(),
                                       0 if fresh else 1, ent[0].data_ptr() if ent is not None else None, ng * ent[1] if ent is not None else 0,
                                       sc.data_ptr(), _stream()), 'kpx_bn_train_bwd_f32')
        return (dx, None if ctx.g_grad_out is not None else dg, None if ctx.b_grad_out is not None else db,
                None, None, None, None, None, None, None, None, None, None, None)


def batch_norm(x, gamma, beta, moving_mean, moving_var, train=True, act=ACT_RELU, groups=1, update_moving=True,
               g_grad_out=None, b_grad_out=None, out_f32=False):
    ts = getattr(x, '_kpx_tile_stats', None) if train else None
    if ts is not None and (x.shape[0] % groups or tuple(ts[0].shape) != (x.shape[0] * ts[1] * 2 * x.shape[3],) or not _whole_tiles_per_group(x.shape[0], groups, ts[1])):
        ts = None                                        # (a tile that packs images of two groups cannot be split: the separate reduction pass runs)
    _bn_counter[0] += 1
    y = BatchNormFn.apply(x, gamma, beta, moving_mean, moving_var, g_grad_out, b_grad_out, train, act, groups, update_moving, ts, _bn_counter[0], out_f32)
    if train and act == ACT_RELU:
        y._kpx_bn = (beta.detach(), _bn_counter[0])      # a 3x3 conv reading y can reduce this batch norm's backward sums in its dgrad epilogue
    return y


# ----------------------------------------------------------------------------------------------- resize + concat
class UpsampleConcatFn(torch.autograd.Function):
    """tf.image.resize_images(x, 2x) [+ tf.concat([up, skip], -1)] (reference networks/__init__.py:63,98 and :44).
    The up-sampled tensor is written straight into its channel slice of the concat buffer."""

    @staticmethod
    def forward(ctx, x, skip):
        _require_gpu(x)
        x = x.contiguous()
        n, h, w, c1 = x.shape
        c2 = 0
        if skip is not None:
            skip = skip.contiguous()
            assert skip.shape[:3] == (n, 2 * h, 2 * w)
            c2 = skip.shape[3]
        ld = c1 + c2
        out = torch.empty((n, 2 * h, 2 * w, ld), dtype=x.dtype, device=x.device)
        if x.dtype == BF16:
            check(lib.kpx_resize2x_fwd_bf16(x.data_ptr(), n, h, w, c1, c1, out.data_ptr(), ld, _stream()), 'kpx_resize2x_fwd_bf16')
            if c2:
                skip = cast(skip, BF16)
                cast_channels_raw(skip.data_ptr(), c2, out.data_ptr() + 2 * c1, ld, n * 4 * h * w, c2, 2)
        else:
            check(lib.kpx_resize2x_fwd_f32(x.data_ptr(), n, h, w, c1, c1, out.data_ptr(), ld, _stream()), 'kpx_resize2x_fwd_f32')
            if c2:
                copy_channels_raw(cast(skip, torch.float32).data_ptr(), c2, out.data_ptr() + 4 * c1, ld, n * 4 * h * w, c2)
        ctx.dims = (n, h, w, c1, c2)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, h, w, c1, c2 = ctx.dims
        dout = dout.contiguous()
        ld = c1 + c2
        dx = dskip = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, h, w, c1), dtype=dout.dtype, device=dout.device)
            if dout.dtype == BF16:
                check(lib.kpx_resize2x_bwd_bf16(dout.data_ptr(), n, h, w, c1, ld, dx.data_ptr(), c1, _stream()), 'kpx_resize2x_bwd_bf16')
            else:
                check(lib.kpx_resize2x_bwd_f32(dout.data_ptr(), n, h, w, c1, ld, dx.data_ptr(), c1, _stream()), 'kpx_resize2x_bwd_f32')
        if c2 and ctx.needs_input_grad[1]:
            dskip = torch.empty((n, 2 * h, 2 * w, c2), dtype=dout.dtype, device=dout.device)
            if dout.dtype == BF16:
                cast_channels_raw(dout.data_ptr() + 2 * c1, ld, dskip.data_ptr(), c2, n * 4 * h * w, c2, 2)
            else:
                copy_channels_raw(dout.data_ptr() + 4 * c1, ld, dskip.data_ptr(), c2, n * 4 * h * w, c2)
        return dx, dskip


def upsample2x_concat(x, skip=None):
    return UpsampleConcatFn.apply(x, skip)


class ConcatChannelsFn(torch.autograd.Function):
    """tf.concat([a, b], axis=-1) for two NHWC tensors (batch-norm'd feature + skip) without resize."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        _require_gpu(a)
        n, h, w, c1 = a.shape
        c2 = b.shape[3]
        out = torch.empty((n, h, w, c1 + c2), dtype=a.dtype, device=a.device)
        if a.dtype == BF16:
            cast_channels_raw(a.data_ptr(), c1, out.data_ptr(), c1 + c2, n * h * w, c1, 2)
            cast_channels_raw(cast(b, BF16).data_ptr(), c2, out.data_ptr() + 2 * c1, c1 + c2, n * h * w, c2, 2)
        else:
            copy_channels_raw(a.data_ptr(), c1, out.data_ptr(), c1 + c2, n * h * w, c1)
            copy_channels_raw(cast(b, torch.float32).data_ptr(), c2, out.data_ptr() + 4 * c1, c1 + c2, n * h * w, c2)
        ctx.dims = (n, h, w, c1, c2)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, h, w, c1, c2 = ctx.dims
        dout = dout.contiguous()
        da = torch.empty((n, h, w, c1), dtype=dout.dtype, device=dout.device)
        db = torch.empty((n, h, w, c2), dtype=dout.dtype, device=dout.device)
        if dout.dtype == BF16:
            cast_channels_raw(dout.data_ptr(), c1 + c2, da.data_ptr(), c1, n * h * w, c1, 2)
            cast_channels_raw(dout.data_ptr() + 2 * c1, c1 + c2, db.data_ptr(), c2, n * h * w, c2, 2)
        else:
            copy_channels_raw(dout.data_ptr(), c1 + c2, da.data_ptr(), c1, n * h * w, c1)
            copy_channels_raw(dout.data_ptr() + 4 * c1, c1 + c2, db.data_ptr(), c2, n * h * w, c2)
        return da, db


def concat_channels(a, b):
    return ConcatChannelsFn.apply(a, b)


class ConcatBatchFn(torch.autograd.Function):
    """tf.concat([a, b], axis=0): used to run the two pose_encoder calls / the real+fake discriminator passes batched."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        _require_gpu(a)
        assert a.shape[1:] == b.shape[1:] and a.dtype == b.dtype
        out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
        flat_copy_tensor(a, out.data_ptr())
        flat_copy_tensor(b, out.data_ptr() + _esz(a) * a.numel())
        ctx.na = a.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        return dout[:ctx.na], dout[ctx.na:]


def concat_batch(a, b):
    return ConcatBatchFn.apply(a, b)


# ----------------------------------------------------------------------------------------------- key-points
class KeypointHeadFn(torch.autograd.Function):
    """get_coord x2 + stack (reference utils/model.py:63-70, networks/__init__.py:68-72): logits -> [B,K,2] (x,y)."""

    @staticmethod
    def forward(ctx, logits):
        _require_gpu(logits)
        logits = logits.contiguous()
        b, h, w, k = logits.shape
        dev = logits.device
        mu = torch.empty((b, k, 2), dtype=torch.float32, device=dev)
        py = torch.empty((b, h, k), dtype=torch.float32, device=dev)
        px = torch.empty((b, w, k), dtype=torch.float32, device=dev)
        sc = scratch.get('kphead', lib.kpx_keypoint_head_scratch_bytes(b, h, w, k), dev)
        check(lib.kpx_keypoint_head_fwd_f32(logits.data_ptr(), b, h, w, k, mu.data_ptr(), py.data_ptr(), px.data_ptr(),
                                            sc.data_ptr(), _stream()), 'kpx_keypoint_head_fwd_f32')
        ctx.dims = (b, h, w, k)
        ctx.save_for_backward(mu, py, px)
        ctx.mark_non_differentiable(py, px)
        return mu, py, px

    @staticmethod
    def backward(ctx, dmu, _dpy, _dpx):
        mu, py, px = ctx.saved_tensors
        b, h, w, k = ctx.dims
        dmu = dmu.contiguous()
        dl = torch.empty((b, h, w, k), dtype=torch.float32, device=dmu.device)
        check(lib.kpx_keypoint_head_bwd_f32(dmu.data_ptr(), mu.data_ptr(), py.data_ptr(), px.data_ptr(), b, h, w, k,
                                            dl.data_ptr(), _stream()), 'kpx_keypoint_head_bwd_f32')
        return dl


def keypoint_head(logits):
    return KeypointHeadFn.apply(logits)


class KeypointHeadProjFn(torch.autograd.Function):
    """layers.conv(x, n_pts, kernel=1) + get_coord x2 + stack with the logits never formed (reference networks/__init__.py:54,68-72;
    utils/model.py:63-70): the axis means commute with the 1x1 projection.  x [B,H,W,C], w [1,1,C,K], b [K] -> mu [B,K,2] (x,y)."""

    @staticmethod
    def forward(ctx, x, w, b, w_grad_out, b_grad_out):
        _require_gpu(x)
        x = x.contiguous()
        bsz, h, wd, c = x.shape
        k = w.shape[3]
        dev = x.device
        mu = torch.empty((bsz, k, 2), dtype=torch.float32, device=dev)
        py = torch.empty((bsz, h, k), dtype=torch.float32, device=dev)
        px = torch.empty((bsz, wd, k), dtype=torch.float32, device=dev)
        xs_y = torch.empty((bsz, h, c), dtype=torch.float32, device=dev)
        xs_x = torch.empty((bsz, wd, c), dtype=torch.float32, device=dev)
        sc = scratch.get('kpproj', lib.kpx_keypoint_head_proj_scratch_bytes(bsz, h, wd, c, k), dev)
        check(lib.kpx_keypoint_head_proj_fwd_f32(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, bsz, h, wd, c, k,
                                                 mu.data_ptr(), py.data_ptr(), px.data_ptr(), xs_y.data_ptr(), xs_x.data_ptr(),
                                                 sc.data_ptr(), _stream()), 'kpx_keypoint_head_proj_fwd_f32')
        ctx.dims = (bsz, h, wd, c, k)
        ctx.has_bias = b is not None
        ctx.w_grad_out, ctx.b_grad_out = w_grad_out, b_grad_out
        ctx.save_for_backward(mu, py, px, xs_y, xs_x, w)
        ctx.mark_non_differentiable(py, px)
        return mu, py, px

    @staticmethod
    def backward(ctx, dmu, _dpy, _dpx):
        mu, py, px, xs_y, xs_x, w = ctx.saved_tensors
        bsz, h, wd, c, k = ctx.dims
        dev = dmu.device
        dmu = dmu.contiguous()
        dx = torch.empty((bsz, h, wd, c), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        # both parameter gradients leave one launch: they go straight into the flat bucket only when both slots are fresh
        direct = (want_w and ctx.w_grad_out is not None and (not want_b or ctx.b_grad_out is not None))
        if direct:
            fresh_w = _claim_grad(ctx.w_grad_out)
            fresh_b = _claim_grad(ctx.b_grad_out) if want_b else fresh_w
            assert fresh_w == fresh_b, 'kernel and bias of one 1x1 head are always used together'
            dw_buf, db_buf, acc = ctx.w_grad_out, (ctx.b_grad_out if want_b else None), 0 if fresh_w else 1
        else:
            dw_buf = torch.empty_like(w) if want_w else None
            db_buf = torch.empty(k, dtype=torch.float32, device=dev) if want_b else None
            acc = 0
        sc = scratch.get('kpproj', lib.kpx_keypoint_head_proj_scratch_bytes(bsz, h, wd, c, k), dev)
        check(lib.kpx_keypoint_head_proj_bwd_f32(dmu.data_ptr(), mu.data_ptr(), py.data_ptr(), px.data_ptr(), xs_y.data_ptr(), xs_x.data_ptr(),
                                                 w.data_ptr(), bsz, h, wd, c, k, dx.data_ptr() if dx is not None else None,
                                                 dw_buf.data_ptr() if dw_buf is not None else None,
                                                 db_buf.data_ptr() if db_buf is not None else None, acc, sc.data_ptr(), _stream()),
              'kpx_keypoint_head_proj_bwd_f32')
        if direct:
            return dx, None, None, None, None
        return dx, dw_buf, db_buf, None, None


def keypoint_head_proj_eligible(x_shape, k):
    """True when the folded 1x1 + key-point head operator takes an activation of this [B,H,W,C] shape with K points
    (kpx_keypoint_head_proj_eligible); the caller runs the conv + keypoint_head pair otherwise."""
    b, h, w, c = (int(v) for v in x_shape)
    return bool(lib.kpx_keypoint_head_proj_eligible(b, h, w, c, int(k)))


def keypoint_head_proj(x, w, b=None, w_grad_out=None, b_grad_out=None):
    return KeypointHeadProjFn.apply(x, w, b, w_grad_out, b_grad_out)


class GaussianMapsFn(torch.autograd.Function):
    """get_gaussian_maps (reference utils/model.py:49-60): [B,K,2] -> [B,H,W,K]."""

    @staticmethod
    def forward(ctx, mu, h, w, inv_std):
        _require_gpu(mu)
        mu = mu.contiguous()
        b, k, _ = mu.shape
        out = torch.empty((b, h, w, k), dtype=torch.float32, device=mu.device)
        check(lib.kpx_gaussian_maps_fwd_f32(mu.data_ptr(), b, k, h, w, float(inv_std), out.data_ptr(), k, _stream()),
              'kpx_gaussian_maps_fwd_f32')
        ctx.dims = (b, k, h, w, float(inv_std))
        ctx.save_for_backward(mu)
        return out

    @staticmethod
    def backward(ctx, dout):
        (mu,) = ctx.saved_tensors
        b, k, h, w, inv_std = ctx.dims
        dout = dout.contiguous()
        dmu = torch.empty_like(mu)
        check(lib.kpx_gaussian_maps_bwd_f32(dout.data_ptr(), k, mu.data_ptr(), b, k, h, w, inv_std, dmu.data_ptr(), _stream()),
              'kpx_gaussian_maps_bwd_f32')
        return dmu, None, None, None


def gaussian_maps(mu, h, w, inv_std=14.3):
    return GaussianMapsFn.apply(mu, h, w, inv_std)


class JointEmbeddingFn(torch.autograd.Function):
    """tf.concat([embedding, current_map, future_map], -1) (reference detector_translator_model.py:168-170) with the
    two heat-maps rendered straight into their channel slices.  Output pixel stride is padded to a multiple of 4
    (16-B aligned pixels); the consumer conv reads only the first C+2K channels (``cin``)."""

    @staticmethod
    def forward(ctx, emb, cur_pt, fut_pt, inv_std):
        _require_gpu(emb)
        emb, cur_pt, fut_pt = emb.contiguous(), cur_pt.contiguous(), fut_pt.contiguous()
        b, h, w, c = emb.shape
        k = cur_pt.shape[1]
        if emb.dtype == BF16:
            # bf16 configuration: the joint buffer is bf16 with its pixel stride rounded up to a multiple of 32 channels (the consumer's
            # chunk size; the pad channels are zeros and meet zero filter rows).  The two heat-maps are rendered in fp32 (key-point / heat-map
            # arithmetic stays fp32) into a small [B,h,w,2K(+pad)] tensor and rounded into their channel slice.
            ld = (c + 2 * k + 31) // 32 * 32
            out = torch.empty((b, h, w, ld), dtype=BF16, device=emb.device)
            cast_channels_raw(emb.data_ptr(), c, out.data_ptr(), ld, b * h * w, c, 2)
            maps = torch.empty((b, h, w, ld - c), dtype=torch.float32, device=emb.device)
            if ld - c > 2 * k:
                fill_raw_(maps, 0.0)
            for i, pt in enumerate((cur_pt, fut_pt)):
                check(lib.kpx_gaussian_maps_fwd_f32(pt.data_ptr(), b, k, h, w, float(inv_std), maps.data_ptr() + 4 * i * k, ld - c, _stream()),
                      'kpx_gaussian_maps_fwd_f32')
            cast_channels_raw(maps.data_ptr(), ld - c, out.data_ptr() + 2 * c, ld, b * h * w, ld - c, 0)
            ctx.dims = (b, h, w, c, k, ld, float(inv_std))
            ctx.save_for_backward(cur_pt, fut_pt)
            return out
        ld = (c + 2 * k + 3) // 4 * 4
        out = torch.empty((b, h, w, ld), dtype=torch.float32, device=emb.device)
        if ld > c + 2 * k:
            fill_raw_(out, 0.0)
        copy_channels_raw(emb.data_ptr(), c, out.data_ptr(), ld, b * h * w, c)
        for i, pt in enumerate((cur_pt, fut_pt)):
            check(lib.kpx_gaussian_maps_fwd_f32(pt.data_ptr(), b, k, h, w, float(inv_std), out.data_ptr() + 4 * (c + i * k), ld, _stream()),
                  'kpx_gaussian_maps_fwd_f32')
        ctx.dims = (b, h, w, c, k, ld, float(inv_std))
        ctx.save_for_backward(cur_pt, fut_pt)
        return out

    @staticmethod
    def backward(ctx, dout):
        cur_pt, fut_pt = ctx.saved_tensors
        b, h, w, c, k, ld, inv_std = ctx.dims
        dout = dout.contiguous()
        if dout.dtype == BF16:
            demb = torch.empty((b, h, w, c), dtype=BF16, device=dout.device)
            cast_channels_raw(dout.data_ptr(), ld, demb.data_ptr(), c, b * h * w, c, 2)
            dmaps = torch.empty((b, h, w, 2 * k), dtype=torch.float32, device=dout.device)
            cast_channels_raw(dout.data_ptr() + 2 * c, ld, dmaps.data_ptr(), 2 * k, b * h * w, 2 * k, 1)
            grads = []
            for i, pt in enumerate((cur_pt, fut_pt)):
                d = torch.empty_like(pt)
                check(lib.kpx_gaussian_maps_bwd_f32(dmaps.data_ptr() + 4 * i * k, 2 * k, pt.data_ptr(), b, k, h, w, inv_std, d.data_ptr(), _stream()),
                      'kpx_gaussian_maps_bwd_f32')
                grads.append(d)
            return demb, grads[0], grads[1], None
        demb = torch.empty((b, h, w, c), dtype=torch.float32, device=dout.device)
        copy_channels_raw(dout.data_ptr(), ld, demb.data_ptr(), c, b * h * w, c)
        grads = []
        for i, pt in enumerate((cur_pt, fut_pt)):
            d = torch.empty_like(pt)
            check(lib.kpx_gaussian_maps_bwd_f32(dout.data_ptr() + 4 * (c + i * k), ld, pt.data_ptr(), b, k, h, w, inv_std,
                                                d.data_ptr(), _stream()), 'kpx_gaussian_maps_bwd_f32')
            grads.append(d)
        return demb, grads[0], grads[1], None


def joint_embedding(emb, cur_pt, fut_pt, inv_std=14.3):
    return JointEmbeddingFn.apply(emb, cur_pt, fut_pt, inv_std)


# ----------------------------------------------------------------------------------------------- heads / losses
class HeadBlendFn(torch.autograd.Function):
    """mask = sigmoid(raw[...,3]); final = im*mask + crude*(1-mask) (reference networks/__init__.py:87-89,
    detector_translator_model.py:174).  raw4 = crude(3) ‖ mask-logit(1) from ONE 4-channel conv."""

    @staticmethod
    def forward(ctx, im, raw4):
        _require_gpu(raw4)
        im, raw4 = im.contiguous(), raw4.contiguous()
        n, h, w, _ = raw4.shape
        dev = raw4.device
        final = torch.empty((n, h, w, 3), dtype=torch.float32, device=dev)
        crude = torch.empty((n, h, w, 3), dtype=torch.float32, device=dev)
        mask = torch.empty((n, h, w, 1), dtype=torch.float32, device=dev)
        check(lib.kpx_head_blend_fwd_f32(im.data_ptr(), raw4.data_ptr(), n * h * w, final.data_ptr(), crude.data_ptr(),
                                         mask.data_ptr(), _stream()), 'kpx_head_blend_fwd_f32')
        ctx.save_for_backward(im, raw4)
        ctx.mark_non_differentiable(crude, mask)
        return final, crude, mask

    @staticmethod
    def backward(ctx, dfinal, _dc, _dm):
        im, raw4 = ctx.saved_tensors
        dfinal = dfinal.contiguous()
        draw = torch.empty_like(raw4)
        n, h, w, _ = raw4.shape
        check(lib.kpx_head_blend_bwd_f32(dfinal.data_ptr(), im.data_ptr(), raw4.data_ptr(), n * h * w, draw.data_ptr(), _stream()),
              'kpx_head_blend_bwd_f32')
        return None, draw


def head_blend(im, raw4):
    return HeadBlendFn.apply(im, raw4)


class SigmoidXentFn(torch.autograd.Function):
    """reduce_mean(sigmoid_cross_entropy_with_logits) for one or two label groups
    (reference detector_translator_model.py:249-254, 265-267).  Returns [total, mean_group0, mean_group1]."""

    @staticmethod
    def forward(ctx, logits, n0, label0, n1, label1):
        _require_gpu(logits)
        logits = logits.contiguous()
        assert logits.numel() == n0 + n1
        out = torch.empty(3, dtype=torch.float32, device=logits.device)
        check(lib.kpx_sigmoid_xent_fwd_f32(logits.data_ptr(), n0, label0, n1, label1, out.data_ptr(), _stream()), 'kpx_sigmoid_xent_fwd_f32')
        ctx.args = (n0, label0, n1, label1)
        ctx.save_for_backward(logits)
        return out

    @staticmethod
    def backward(ctx, dout):
        (logits,) = ctx.saved_tensors
        n0, label0, n1, label1 = ctx.args
        dout = dout.contiguous()          # only dout[0] (the summed loss) is differentiated
        dl = torch.empty_like(logits)
        check(lib.kpx_sigmoid_xent_bwd_f32(logits.data_ptr(), n0, label0, n1, label1, dout.data_ptr(), 1.0, dl.data_ptr(), _stream()),
              'kpx_sigmoid_xent_bwd_f32')
        return dl, None, None, None, None


def sigmoid_xent(logits, n0, label0, n1=0, label1=0.0):
    return SigmoidXentFn.apply(logits, n0, float(label0), n1, float(label1))


# ----------------------------------------------------------------------------------------------- optimiser
def adam_tf_flat_(p, g, m, v, alpha, beta1, beta2, eps, gscale=1.0):
    """In-place tf.train.AdamOptimizer update of one flat bucket (reference detector_translator_model.py:198-202)."""
    for t in (p, g, m, v):
        _require_gpu(t)
    check(lib.kpx_adam_tf_flat_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                   float(alpha), float(beta1), float(beta2), float(eps), float(gscale), _stream()), 'kpx_adam_tf_flat_f32')


def adam_tf_flat_dev_alpha_(p, g, m, v, alpha_dev, beta1, beta2, eps, gscale=1.0):
    """The same update with the step size in device memory (a [1] fp32 tensor): the form a captured HIP graph replays."""
    for t in (p, g, m, v, alpha_dev):
        _require_gpu(t)
    check(lib.kpx_adam_tf_flat_dev_alpha_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), alpha_dev.data_ptr(),
                                             float(beta1), float(beta2), float(eps), float(gscale), _stream()), 'kpx_adam_tf_flat_dev_alpha_f32')


# ----------------------------------------------------------------------------------------------- rollout (forward only)
def dense(x, w, b, act=ACT_NONE):
    """act(x @ w + b) for x [B,In], w [In,Out]: a 1x1 convolution over a [B,1,1,In] tensor on the implicit-GEMM kernel
    (tf.contrib.layers.fully_connected / LSTMCell matmul / to_coord, reference networks/__init__.py:120, layers.py:17-28)."""
    _require_gpu(x)
    x = x.contiguous()
    bsz, n_in = x.shape
    n_out = w.shape[1]
    y = torch.empty((bsz, 1, 1, n_out), dtype=torch.float32, device=x.device)
    conv_fwd_raw(x.view(bsz, 1, 1, n_in), n_in, n_in, w.contiguous().view(1, 1, n_in, n_out), b, y, n_out, 1, 0, 0, act)
    return y.view(bsz, n_out)


def lstm_cell(x, h, c, kernel, bias, forget_bias=1.0):
    """One tf.nn.rnn_cell.LSTMCell step: returns (h', c')."""
    bsz, units = h.shape
    xin = torch.empty((bsz, x.shape[1] + units), dtype=torch.float32, device=x.device)
    copy_channels_raw(x.contiguous().data_ptr(), x.shape[1], xin.data_ptr(), xin.shape[1], bsz, x.shape[1])
    copy_channels_raw(h.contiguous().data_ptr(), units, xin.data_ptr() + 4 * x.shape[1], xin.shape[1], bsz, units)
    gates = dense(xin, kernel, bias)
    c2, h2 = torch.empty_like(c), torch.empty_like(h)
    check(lib.kpx_lstm_pointwise_f32(gates.data_ptr(), c.contiguous().data_ptr(), float(forget_bias), c2.data_ptr(), h2.data_ptr(),
                                     bsz, units, _stream()), 'kpx_lstm_pointwise_f32')
    return h2, c2


def tile_batch(x, t, out=None, out_ld=None, out_channel_offset=0):
    """tf.tile over a new time axis + reshape: [B, ..., C] -> [B*T, ..., C] (optionally into a channel slice of ``out``)."""
    _require_gpu(x)
    x = x.contiguous()
    bsz, c = x.shape[0], x.shape[-1]
    pix = x.numel() // (bsz * c)
    if out is None:
        out = torch.empty((bsz * t,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        out_ld = c
    check(lib.kpx_tile_batch_f32(x.data_ptr(), c, bsz, t, pix, c, out.data_ptr() + 4 * out_channel_offset, out_ld, _stream()),
          'kpx_tile_batch_f32')
    return out


def head_blend_tiled(im, raw4, t, clip=True):
    """final = tile(im,T)*mask + crude*(1-mask) with clip_by_value(-1,1) (reference final_model.py:95-99)."""
    raw4 = raw4.contiguous()
    n, h, w, _ = raw4.shape
    dev = raw4.device
    final = torch.empty((n, h, w, 3), dtype=torch.float32, device=dev)
    crude = torch.empty((n, h, w, 3), dtype=torch.float32, device=dev)
    mask = torch.empty((n, h, w, 1), dtype=torch.float32, device=dev)
    check(lib.kpx_head_blend_tiled_fwd_f32(im.contiguous().data_ptr(), raw4.data_ptr(), n * h * w, h * w, t, 1 if clip else 0,
                                           final.data_ptr(), crude.data_ptr(), mask.data_ptr(), _stream()), 'kpx_head_blend_tiled_fwd_f32')
    return final, crude, mask


# ----------------------------------------------------------------------------------------------- stage-2 training (motion generator)
def dense_train(x, w, b, act=ACT_NONE, w_grad_out=None, b_grad_out=None):
    """Differentiable ``dense``: a 1x1 convolution over [B,1,1,In] (forward / dgrad / wgrad on the conv kernels)."""
    bsz, n_in = x.shape
    n_out = w.shape[1]
    y = conv2d(x.reshape(bsz, 1, 1, n_in), w.view(1, 1, n_in, n_out), b, stride=1, pad=0, act=act,
               w_grad_out=w_grad_out.view(1, 1, n_in, n_out) if w_grad_out is not None else None, b_grad_out=b_grad_out)
    return y.reshape(bsz, n_out)


class LstmLayerFn(torch.autograd.Function):
    """One LSTMCell layer over a whole sequence with zero initial state (tf.nn.dynamic_rnn / the unrolled cell calls of
    reference networks/__init__.py:105-138): x [T,B,In] -> h [T,B,U].  The time loops run inside the library
    (kpx_lstm_layer_fwd_f32 / _bwd_f32): per step [x_t, h_{t-1}] @ kernel + bias on the conv kernel + the gate math, backwards the
    gate-math backward + the dgrad GEMM (the recurrence needs dh_{t-1}); the weight gradient is ONE GEMM over all T*B rows, the
    bias gradient one channel sum."""

    @staticmethod
    def forward(ctx, x, kernel, bias, w_grad_out, b_grad_out):
        _require_gpu(x)
        x = x.contiguous()
        t, bsz, n_in = x.shape
        units = kernel.shape[1] // 4
        dev = x.device
        xin = torch.empty((t, bsz, n_in + units), dtype=torch.float32, device=dev)
        gates = torch.empty((t, bsz, 4 * units), dtype=torch.float32, device=dev)
        cs = torch.empty((t, bsz, units), dtype=torch.float32, device=dev)
        hs = torch.empty((t, bsz, units), dtype=torch.float32, device=dev)
        zero = torch.zeros((bsz, units), dtype=torch.float32, device=dev)
        width = n_in + units
        nbytes = lib.kpx_conv2d_fwd_workspace_bytes(bsz, 1, 1, width, 4 * units, 1, 1)
        ws = scratch.get('splitk', nbytes, dev) if nbytes else None
        check(lib.kpx_lstm_layer_fwd_f32(x.data_ptr(), t, bsz, n_in, kernel.data_ptr(), bias.data_ptr(), units, xin.data_ptr(), gates.data_ptr(),
                                         cs.data_ptr(), hs.data_ptr(), zero.data_ptr(), ws.data_ptr() if ws is not None else None, nbytes, _stream()),
              'kpx_lstm_layer_fwd_f32')
        ctx.save_for_backward(xin, gates, cs, kernel)
        ctx.w_grad_out, ctx.b_grad_out, ctx.n_in = w_grad_out, b_grad_out, n_in
        return hs

    @staticmethod
    def backward(ctx, dhs):
        xin, gates, cs, kernel = ctx.saved_tensors
        t, bsz, width = xin.shape
        n_in, units = ctx.n_in, gates.shape[2] // 4
        dev = xin.device
        dhs = dhs.contiguous()
        dgates = torch.empty_like(gates)
        dx = torch.empty((t, bsz, n_in), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dxin = torch.empty((bsz, width), dtype=torch.float32, device=dev)
        dh = torch.empty((bsz, units), dtype=torch.float32, device=dev)
        dc0 = torch.zeros((bsz, units), dtype=torch.float32, device=dev)
        dc1 = torch.empty((bsz, units), dtype=torch.float32, device=dev)
        nbytes = lib.kpx_conv2d_dgrad_workspace_bytes(bsz, 1, 1, width, 4 * units, 1, 1, 1)
        ws = scratch.get('splitk', nbytes, dev) if nbytes else None
        check(lib.kpx_lstm_layer_bwd_f32(dhs.data_ptr(), t, bsz, n_in, kernel.data_ptr(), units, gates.data_ptr(), cs.data_ptr(), dgates.data_ptr(),
                                         dx.data_ptr() if dx is not None else None, dxin.data_ptr(), dh.data_ptr(), dc0.data_ptr(), dc1.data_ptr(),
                                         ws.data_ptr() if ws is not None else None, nbytes, _stream()), 'kpx_lstm_layer_bwd_f32')
        dw = db = None
        if ctx.needs_input_grad[1]:
            direct = ctx.w_grad_out is not None and _claim_grad(ctx.w_grad_out)
            dw_buf = ctx.w_grad_out if direct else torch.empty_like(kernel)
            conv_wgrad_raw(xin.view(t * bsz, 1, 1, width), width, width, dgates.view(t * bsz, 1, 1, 4 * units), 4 * units,
                           dw_buf.view(1, 1, width, 4 * units), 1, 0, 0)
            if ctx.w_grad_out is not None and not direct:
                axpy_raw_(ctx.w_grad_out, dw_buf)
            dw = None if ctx.w_grad_out is not None else dw_buf
        if ctx.needs_input_grad[2]:
            direct = ctx.b_grad_out is not None and _claim_grad(ctx.b_grad_out)
            db_buf = ctx.b_grad_out if direct else torch.empty(4 * units, dtype=torch.float32, device=dev)
            chan_sum_raw(dgates, 4 * units, t * bsz, 4 * units, db_buf)
            if ctx.b_grad_out is not None and not direct:
                axpy_raw_(ctx.b_grad_out, db_buf)
            db = None if ctx.b_grad_out is not None else db_buf
        return dx, dw, db, None, None


def lstm_layer(x_seq, kernel, bias, w_grad_out=None, b_grad_out=None):
    return LstmLayerFn.apply(x_seq, kernel, bias, w_grad_out, b_grad_out)


class VaeSampleKlFn(torch.autograd.Function):
    """logit [B,2V] = [mu | stddev], eps [B,V] -> (z [B,V], kl [1])  (reference motion_generator_model.py:146, :291-293)."""

    @staticmethod
    def forward(ctx, logit, eps):
        _require_gpu(logit)
        logit, eps = logit.contiguous(), eps.contiguous()
        bsz, v = eps.shape
        z = torch.empty((bsz, v), dtype=torch.float32, device=logit.device)
        kl = torch.empty(1, dtype=torch.float32, device=logit.device)
        check(lib.kpx_vae_sample_kl_fwd_f32(logit.data_ptr(), eps.data_ptr(), z.data_ptr(), kl.data_ptr(), bsz, v, _stream()), 'kpx_vae_sample_kl_fwd_f32')
        ctx.save_for_backward(logit, eps)
        return z, kl

    @staticmethod
    def backward(ctx, dz, dkl):
        logit, eps = ctx.saved_tensors
        bsz, v = eps.shape
        dlogit = torch.empty_like(logit)
        dz = dz.contiguous() if dz is not None else None
        dkl = dkl.contiguous() if dkl is not None else None
        check(lib.kpx_vae_sample_kl_bwd_f32(logit.data_ptr(), eps.data_ptr(), dz.data_ptr() if dz is not None else None,
                                            dkl.data_ptr() if dkl is not None else None, 1.0 if dkl is not None else 0.0,
                                            dlogit.data_ptr(), bsz, v, _stream()), 'kpx_vae_sample_kl_bwd_f32')
        return dlogit, None


def vae_sample_kl(logit, eps):
    return VaeSampleKlFn.apply(logit, eps)


class L1MeanFn(torch.autograd.Function):
    """scale * mean(|target - pred|), differentiated wrt pred (reference motion_generator_model.py:288-289, scale = 1000)."""

    @staticmethod
    def forward(ctx, pred, target, scale):
        _require_gpu(pred)
        both = torch.cat([target.reshape(-1), pred.reshape(-1)]).contiguous()          # kpx_l1_pair works on [a | b] halves
        half = pred.numel()
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        sc = scratch.get('l1', 8192, pred.device)
        check(lib.kpx_l1_pair_fwd_f32(both.data_ptr(), half, out.data_ptr(), sc.data_ptr(), _stream()), 'kpx_l1_pair_fwd_f32')
        ctx.save_for_backward(both)
        ctx.shape, ctx.scale = pred.shape, float(scale)
        return out * float(scale)

    @staticmethod
    def backward(ctx, g):
        (both,) = ctx.saved_tensors
        half = both.numel() // 2
        dpred = torch.empty(half, dtype=torch.float32, device=both.device)
        check(lib.kpx_l1_pair_bwd_f32(both.data_ptr(), half, g.contiguous().data_ptr(), ctx.scale / half, dpred.data_ptr(), _stream()), 'kpx_l1_pair_bwd_f32')
        return dpred.view(ctx.shape), None, None


def l1_mean(pred, target, scale=1.0):
    return L1MeanFn.apply(pred, target, scale)
