"""Channel sums of an F(4x4,3x3) data gradient of a zero-mean tensor (the bias / batch-norm gradient sums): error of the sums relative to the sums, new vs old kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
n, h, w, c1, c2 = 2, 64, 64, 128, 128
rs = np.random.RandomState(64 + 128 + 128)
_ = rs.randn(n, h, w, 64); _ = rs.randn(3, 3, 64, c1)
wb = (rs.randn(3, 3, c1, c2) / np.sqrt(9 * c1)).astype(np.float32)
_ = rs.uniform(0.5, 1.5, c1); _ = rs.randn(c1)
gy = rs.randn(n, h, w, c2).astype(np.float32)
wt = torch.from_numpy(wb).to(dev); dy = torch.from_numpy(gy).to(dev)
want = torch.nn.functional.conv2d(dy.cpu().double().permute(0, 3, 1, 2), wt.cpu().double().flip(0, 1).permute(2, 3, 0, 1), padding=1).permute(0, 2, 3, 1)   # [n,h,w,c1]
ub = torch.empty(lib.kpx_wino43b_u_bytes(c1, c2), dtype=torch.uint8, device=dev); uo = torch.empty(lib.kpx_wino43_u_bytes(c1, c2), dtype=torch.uint8, device=dev)
check(lib.kpx_wino43b_filter_transform_f32(wt.data_ptr(), c1, c2, 1, ub.data_ptr(), ops._stream()), 't')
check(lib.kpx_wino43_filter_transform_f32(wt.data_ptr(), c1, c2, 1, uo.data_ptr(), ops._stream()), 't')
yb = torch.empty(n, h, w, c1, device=dev); yo = torch.empty(n, h, w, c1, device=dev)
check(lib.kpx_conv3x3_wino43b_f32(dy.data_ptr(), n, h, w, c2, c2, ub.data_ptr(), None, yb.data_ptr(), c1, c1, 0, None, 0, None, 0, None, None, 0, None, ops._stream()), 'b')
check(lib.kpx_conv3x3_wino43_f32(dy.data_ptr(), n, h, w, c2, c2, uo.data_ptr(), None, yo.data_ptr(), c1, c1, 0, ops._stream()), 'o')
for name, y in (('bf16x3', yb), ('fp32 mfma', yo)):
    e = y.cpu().double() - want
    s_true = want.sum((0, 1, 2)); s_err = e.sum((0, 1, 2))
    print('%-10s rel-L2 %.2e | channel sums: |err| / |sum| (L2 over channels) %.2e | mean err per channel / rms err: max %.3f (random: ~%.3f)'
          % (name, float(e.norm() / want.norm()), float(s_err.norm() / s_true.norm()), float((e.mean((0, 1, 2)).abs() / e.pow(2).mean((0, 1, 2)).sqrt()).max()), 1 / np.sqrt(n * h * w)))
    # is the error correlated with the output itself (a scale error)?
    print('           <err, out> / <out, out> = %.3e' % float((e * want).sum() / (want * want).sum()))
