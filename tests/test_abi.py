"""CPU: the C-ABI library loads and exports every symbol include/kpx.h declares, with matching arity in the ctypes table."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _header_functions():
    src = open(os.path.join(REPO, 'include', 'kpx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(int|size_t)\s+(kpx_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = m.group(3).strip()
        out[m.group(2)] = 0 if args in ('', 'void') else args.count(',') + 1
    return out


def test_library_exports_every_declared_symbol():
    import kpx_amd  # noqa: F401  (fails loudly if the .so is missing)
    from kpx_amd import _lib
    decl = _header_functions()
    assert len(decl) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(lib, name), 'missing export %s' % name
        assert name in _lib.SIGNATURES, 'no ctypes signature for %s' % name
        assert len(_lib.SIGNATURES[name][1]) == nargs, (name, len(_lib.SIGNATURES[name][1]), nargs)
    assert set(_lib.SIGNATURES) == set(decl)
    header_version = int(re.search(r'#define\s+KPX_ABI_VERSION\s+(\d+)', open(os.path.join(REPO, 'include', 'kpx.h')).read()).group(1))
    assert lib.kpx_abi_version() == header_version == _lib.ABI_VERSION          # header, library and ctypes table agree


def test_bad_arguments_are_rejected_without_a_gpu():
    from kpx_amd._lib import lib
    assert lib.kpx_conv2d_fwd_f32(None, 1, 8, 8, 4, 4, None, 3, 3, None, None, 8, 8, 4, 4, 1, 1, 1, 0, 0, None, 0, None) == -1
    assert lib.kpx_bn_stats_f32(None, 10, 4, 4, 1e-5, None, None, None, None, None, 0.999, None, None) == -1
    assert lib.kpx_conv2d_wgrad_workspace_bytes(32, 128, 128, 64, 64, 3, 3) > 0
    assert lib.kpx_conv2d_wgrad_workspace_bytes(1, 4, 4, 1024, 2048, 4, 4) == 0
    # per-call arithmetic selector and the activation-backward factor: range-checked before anything touches a device
    assert lib.kpx_conv2d_fwd_f32(1, 1, 8, 8, 4, 4, 1, 3, 3, None, 1, 8, 8, 4, 4, 1, 1, 1, 0, 2, None, 0, None) == -1           # arith = 2
    assert lib.kpx_conv2d_dgrad_act_f32(1, 1, 8, 8, 4, 4, 1, 3, 3, 1, 8, 8, 4, 4, 1, 1, 1, 0, None, 4, 2, None, 0, None) == -1   # y_in = NULL
    assert lib.kpx_conv2d_dgrad_act_f32(1, 1, 8, 8, 4, 4, 1, 3, 3, 1, 8, 8, 4, 4, 1, 1, 1, 0, 1, 4, 3, None, 0, None) == -1      # tanh: not an epilogue factor


def test_keypoint_head_fold_eligibility_is_a_host_question():
    """networks.pose_encoder asks kpx_keypoint_head_proj_eligible before choosing the folded 1x1 + key-point head operator; a shape outside
    the operator's limits (channels, key-points, profile rows within 64 KB of LDS) takes the conv + head pair instead of failing in the launch."""
    from kpx_amd._lib import lib
    assert lib.kpx_keypoint_head_proj_eligible(64, 128, 128, 16, 15) == 1 and lib.kpx_keypoint_head_proj_eligible(32, 256, 256, 16, 40) == 1
    assert lib.kpx_keypoint_head_proj_eligible(2, 128, 128, 18, 15) == 0        # C % 4
    assert lib.kpx_keypoint_head_proj_eligible(2, 128, 128, 512, 15) == 0       # C > 256
    assert lib.kpx_keypoint_head_proj_eligible(2, 1024, 1024, 16, 15) == 0      # image side > 512
    assert lib.kpx_keypoint_head_proj_eligible(2, 128, 128, 16, 100) == 0       # K > 64


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from kpx_amd import ops
    from kpx_amd._lib import KpxError
    with pytest.raises(KpxError):
        ops.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(3, 3, 4, 4))
