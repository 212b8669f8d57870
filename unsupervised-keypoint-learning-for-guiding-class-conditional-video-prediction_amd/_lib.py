"""ctypes binding of libkpx_hip.so (C ABI: include/kpx.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module raises at
import time, and every op raises ``KpxError`` on a non-zero return code.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_size_t, c_void_p

import torch  # noqa: F401  -- MUST precede the CDLL below: PyTorch-ROCm ships its own libamdhip64; if libkpx_hip.so were loaded
#                first it would pull in the system ROCm runtime and the process would end up with two HIP runtimes (tensors
#                allocated by one are unknown to the other: launches fail with hipErrorNoDevice).  Loaded after torch, the
#                library's libamdhip64 dependency resolves to the runtime that is already in the process.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KPX_LIB') or os.path.join(_HERE, 'libkpx_hip.so')      # (KPX_LIB: a diagnostic build of the same library)


class KpxError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        'libkpx_hip.so not found at %s -- build it first: python -c "import __graft_entry__ as g; g.build()" '
        '(or `make -C %s/csrc`). There is no CPU / PyTorch fallback for the hot path.' % (LIB_PATH, _HERE))

lib = ctypes.CDLL(LIB_PATH)

P = c_void_p
# name -> (restype, argtypes); mirrors include/kpx.h one to one (tests/test_abi.py checks header vs this table)
ABI_VERSION = 3        # = KPX_ABI_VERSION of include/kpx.h (tests/test_abi.py holds the two equal); bumped whenever a signature changes
SIGNATURES = {
    'kpx_abi_version': (c_int, []),
    'kpx_reload_env': (c_int, []),
    'kpx_conv2d_fwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'kpx_conv2d_fwd_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P,
                                   P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'kpx_conv2d_dgrad_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'kpx_conv2d_dgrad_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int,
                                     P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'kpx_conv2d_dgrad_act_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int,
                                         P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_size_t, P]),
    'kpx_conv2d_wgrad_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'kpx_conv2d_wgrad_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int,
                                     P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'kpx_wino_u_bytes': (c_size_t, [c_int, c_int]),
    'kpx_wino_filter_transform_f32': (c_int, [P, c_int, c_int, c_int, P, P]),
    'kpx_wino_filter_transform_batch_f32': (c_int, [P, c_int, P]),
    'kpx_conv3x3_wino_eligible': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'kpx_conv3x3_wino_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P]),
    'kpx_conv3x3_c16_eligible': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'kpx_conv3x3_c16_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, c_int, P, P]),
    'kpx_conv3x3_wino43_eligible': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'kpx_wino43_u_bytes': (c_size_t, [c_int, c_int]),
    'kpx_wino43_filter_transform_f32': (c_int, [P, c_int, c_int, c_int, P, P]),
    'kpx_wino43_filter_transform_batch_f32': (c_int, [P, c_int, P]),
    'kpx_conv3x3_wino43_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P]),
    'kpx_conv3x3_wino43_ex_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P, c_int, P, c_int, P]),
    'kpx_conv3x3_wino43b_eligible': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'kpx_wino43b_u_bytes': (c_size_t, [c_int, c_int]),
    'kpx_wino43b_filter_transform_f32': (c_int, [P, c_int, c_int, c_int, P, P]),
    'kpx_wino43b_filter_transform_batch_f32': (c_int, [P, c_int, P]),
    'kpx_conv3x3_wino43b_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P, c_int, P, c_int, P, P, c_int, P, P]),
    'kpx_conv3x3_wino43_stats_tiles': (c_size_t, [c_int, c_int, c_int]),
    'kpx_conv3x3_wino43_stats_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P, P]),
    'kpx_conv3x3_wino_stats_tiles': (c_size_t, [c_int, c_int, c_int]),
    'kpx_conv3x3_wino_stats_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P, P]),
    'kpx_conv3x3_wino_bnbwd_stats_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, c_int, P, P, P]),
    'kpx_bn_bwd_from_tiles_f32': (c_int, [P, c_int, P, c_int, c_size_t, c_int, P, P, P, P, c_int, P, c_int, P, P, c_int, P, c_size_t, c_size_t, P, P]),
    'kpx_bn_stats_from_tiles_f32': (c_int, [P, c_size_t, c_size_t, c_int, c_int, c_float, P, P, P, P, P, c_float, P]),
    'kpx_conv3x3_bf16s_weights_bytes': (c_size_t, [c_int, c_int]),
    'kpx_conv3x3_bf16s_eligible': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'kpx_conv3x3_bf16s_stats_tiles': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'kpx_conv3x3_bf16s_prepare_f32': (c_int, [P, c_int, c_int, c_int, P, P]),
    'kpx_conv3x3_bf16s_prepare_batch_f32': (c_int, [P, c_int, P]),
    'kpx_conv3x3_bf16s': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, P]),
    'kpx_conv2d_fwd_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'kpx_conv2d_dgrad_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_size_t, P]),
    'kpx_conv2d_wgrad_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'kpx_conv_image_fwd_bf16': (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'kpx_conv_image_dgrad_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'kpx_conv_image_wgrad_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'kpx_conv3x3_wgrad_bf16_eligible': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    'kpx_conv3x3_wgrad_bf16_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'kpx_conv3x3_wgrad_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, c_size_t, P]),
    'kpx_conv3x3_bf16s_bnbwd': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, c_int, P, P, P]),
    'kpx_cast_channels': (c_int, [P, c_int, P, c_int, c_size_t, c_int, c_int, P]),
    'kpx_chan_sum_bf16': (c_int, [P, c_size_t, c_int, c_int, P, P, P]),
    'kpx_bn_train_fwd_bf16': (c_int, [P, c_size_t, c_int, c_int, c_int, P, c_size_t, c_float, P, P, P, P, P, P, c_float, P, c_int, c_int, c_int, P, P]),
    'kpx_bn_train_bwd_bf16': (c_int, [P, c_int, c_int, P, c_int, c_size_t, c_int, c_int, P, P, P, P, c_int, P, c_int, P, P, c_int, P, c_size_t, P, P]),
    'kpx_act_bwd_bf16': (c_int, [P, P, P, c_size_t, c_int, P]),
    'kpx_resize2x_fwd_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    'kpx_resize2x_bwd_bf16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    'kpx_maxpool2_fwd_bf16': (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    'kpx_vgg_feat_bwd_bf16': (c_int, [P, c_size_t, P, c_float, P, c_int, c_int, c_int, c_int, P, P]),
    'kpx_l1_pair_fwd_bf16': (c_int, [P, c_size_t, P, P, P]),
    'kpx_act_bwd_f32': (c_int, [P, P, P, c_size_t, c_int, P]),
    'kpx_chan_reduce_scratch_bytes': (c_size_t, [c_int]),
    'kpx_chan_sum_f32': (c_int, [P, c_size_t, c_int, c_int, P, P, P]),
    'kpx_bn_stats_f32': (c_int, [P, c_size_t, c_int, c_int, c_float, P, P, P, P, P, c_float, P, P]),
    'kpx_bn_invstd_f32': (c_int, [P, c_int, c_float, P, P]),
    'kpx_bn_fold_conv_f32': (c_int, [P, P, c_size_t, c_int, P, P, P, P, c_float, P, P, P]),
    'kpx_bn_train_scratch_bytes': (c_size_t, [c_int, c_int]),
    'kpx_bn_train_fwd_f32': (c_int, [P, c_size_t, c_int, c_int, c_int, P, c_size_t, c_float, P, P, P, P, P, P, c_float, P, c_int, c_int, P, P]),
    'kpx_bn_train_bwd_f32': (c_int, [P, c_int, P, c_int, c_size_t, c_int, c_int, P, P, P, P, c_int, P, c_int, P, P, c_int, P, c_size_t, P, P]),
    'kpx_conv3x3_wino43_bnbwd_stats_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, c_int, P, P, P]),
    'kpx_bn_apply_f32': (c_int, [P, c_size_t, c_int, c_int, P, P, P, P, P, c_int, c_int, P]),
    'kpx_bn_bwd_f32': (c_int, [P, c_int, P, c_int, c_size_t, c_int, P, P, P, P, c_int, P, c_int, P, P, c_int, P, P]),
    'kpx_resize2x_fwd_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    'kpx_resize2x_bwd_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    'kpx_copy_channels_f32': (c_int, [P, c_int, P, c_int, c_size_t, c_int, P]),
    'kpx_keypoint_head_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'kpx_keypoint_head_fwd_f32': (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    'kpx_keypoint_head_bwd_f32': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    'kpx_keypoint_head_proj_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'kpx_keypoint_head_proj_eligible': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'kpx_keypoint_head_proj_fwd_f32': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P]),
    'kpx_keypoint_head_proj_bwd_f32': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P]),
    'kpx_gaussian_maps_fwd_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_double, P, c_int, P]),
    'kpx_gaussian_maps_bwd_f32': (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_double, P, P]),
    'kpx_head_blend_fwd_f32': (c_int, [P, P, c_size_t, P, P, P, P]),
    'kpx_head_blend_bwd_f32': (c_int, [P, P, P, c_size_t, P, P]),
    'kpx_vgg_prep_fwd_f32': (c_int, [P, c_size_t, P, P]),
    'kpx_vgg_prep_bwd_f32': (c_int, [P, c_size_t, P, P]),
    'kpx_maxpool2_fwd_f32': (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    'kpx_maxpool2_bwd_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    'kpx_l1_pair_fwd_f32': (c_int, [P, c_size_t, P, P, P]),
    'kpx_l1_pair_bwd_f32': (c_int, [P, c_size_t, P, c_float, P, P]),
    'kpx_vgg_feat_bwd_f32': (c_int, [P, c_size_t, P, c_float, P, c_int, c_int, c_int, c_int, P, P]),
    'kpx_sigmoid_xent_fwd_f32': (c_int, [P, c_size_t, c_float, c_size_t, c_float, P, P]),
    'kpx_sigmoid_xent_bwd_f32': (c_int, [P, c_size_t, c_float, c_size_t, c_float, P, c_float, P, P]),
    'kpx_adam_tf_flat_f32': (c_int, [P, P, P, P, c_size_t, c_float, c_float, c_float, c_float, c_float, P]),
    'kpx_adam_tf_flat_dev_alpha_f32': (c_int, [P, P, P, P, c_size_t, P, c_float, c_float, c_float, c_float, P]),
    'kpx_lstm_pointwise_f32': (c_int, [P, P, c_float, P, P, c_int, c_int, P]),
    'kpx_tile_batch_f32': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    'kpx_head_blend_tiled_fwd_f32': (c_int, [P, P, c_size_t, c_int, c_int, c_int, P, P, P, P]),
    'kpx_fill_f32': (c_int, [P, c_size_t, c_float, P]),
    'kpx_axpy_f32': (c_int, [P, P, c_size_t, c_float, P]),
    'kpx_lstm_pointwise_bwd_f32': (c_int, [P, P, P, P, c_float, P, P, c_int, c_int, P]),
    'kpx_lstm_layer_fwd_f32': (c_int, [P, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P, c_size_t, P]),
    'kpx_lstm_layer_bwd_f32': (c_int, [P, c_int, c_int, c_int, P, c_int, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    'kpx_vae_sample_kl_fwd_f32': (c_int, [P, P, P, P, c_int, c_int, P]),
    'kpx_vae_sample_kl_bwd_f32': (c_int, [P, P, P, P, c_float, P, c_int, c_int, P]),
    'kpx_u8_to_unit_f32': (c_int, [P, c_size_t, P, P]),
    'kpx_crc32c_host': (ctypes.c_uint, [ctypes.c_uint, P, c_size_t]),
}

for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as e:  # pragma: no cover
        raise ImportError('libkpx_hip.so does not export %s (stale build?)' % _name) from e
    _fn.restype = _res
    _fn.argtypes = _args

if lib.kpx_abi_version() != ABI_VERSION:
    raise ImportError('libkpx_hip.so ABI version mismatch')


abi_calls = [0]          # diagnostics: C-ABI calls checked so far (one kernel launch each, a few entries fan out to two or three)


def check(rc, what):
    abi_calls[0] += 1
    if rc != 0:
        raise KpxError('%s failed with code %d (%s)' % (what, rc, 'bad argument' if rc == -1 else 'hipError %d' % -rc))
