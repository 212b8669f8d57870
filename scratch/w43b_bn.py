import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
for (n, h, w, k, nn) in [(2, 32, 64, 64, 64), (2, 64, 64, 128, 128), (2, 128, 128, 64, 64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, k, generator=g).to(dev); wt = (torch.randn(3, 3, nn, k, generator=g) / (9 * k) ** 0.5).to(dev)   # dgrad: filter [3,3,cin=nn,cout=k]
    z = torch.randn(n, h, w, nn, generator=g).to(dev); beta = torch.randn(nn, generator=g).to(dev)
    ub = torch.empty(lib.kpx_wino43b_u_bytes(nn, k), dtype=torch.uint8, device=dev); uo = torch.empty(lib.kpx_wino43_u_bytes(nn, k), dtype=torch.uint8, device=dev)
    check(lib.kpx_wino43b_filter_transform_f32(wt.data_ptr(), nn, k, 1, ub.data_ptr(), ops._stream()), 't')
    check(lib.kpx_wino43_filter_transform_f32(wt.data_ptr(), nn, k, 1, uo.data_ptr(), ops._stream()), 't')
    tiles = lib.kpx_conv3x3_wino43_stats_tiles(n, h, w)
    yb = torch.empty(n, h, w, nn, device=dev); yo = torch.empty(n, h, w, nn, device=dev)
    sb = torch.full((tiles * 2 * nn,), float('nan'), device=dev); so = torch.full((tiles * 2 * nn,), float('nan'), device=dev)
    check(lib.kpx_conv3x3_wino43b_f32(x.data_ptr(), n, h, w, k, k, ub.data_ptr(), None, yb.data_ptr(), nn, nn, 0, None, 0, None, 0, sb.data_ptr(), z.data_ptr(), nn, beta.data_ptr(), ops._stream()), 'b')
    check(lib.kpx_conv3x3_wino43_bnbwd_stats_f32(x.data_ptr(), n, h, w, k, k, uo.data_ptr(), yo.data_ptr(), nn, nn, z.data_ptr(), nn, beta.data_ptr(), so.data_ptr(), ops._stream()), 'o')
    torch.cuda.synchronize()
    sb, so = sb.view(tiles, 2, nn).double(), so.view(tiles, 2, nn).double()
    print(n, h, w, k, nn, 'dz new vs old rel %.2e' % float((yb - yo).norm() / yo.norm()), ' nan in new slab:', int(torch.isnan(sb).sum()), ' old:', int(torch.isnan(so).sum()))
    # per strip: recompute from the new kernel's own dz
    dz = yb.double(); zz = z.double() - beta.double()
    ref0 = dz.view(n, h // 16, 4, 4, w // 32, 2, 16, nn).sum((3, 6))          # [n, by, tyrow, bx, half, nn]
    ref1 = (dz * zz).view(n, h // 16, 4, 4, w // 32, 2, 16, nn).sum((3, 6))
    ref0 = ref0.permute(0, 1, 3, 2, 4, 5).reshape(tiles, nn); ref1 = ref1.permute(0, 1, 3, 2, 4, 5).reshape(tiles, nn)
    print('   per-strip max abs err: sum %.3e  sumq %.3e   (old kernel vs its own dz: %.3e)' % (float((sb[:, 0] - ref0).abs().max()), float((sb[:, 1] - ref1).abs().max()),
          float((so[:, 0] - yo.double().view(n, h // 16, 4, 4, w // 32, 2, 16, nn).sum((3, 6)).permute(0, 1, 3, 2, 4, 5).reshape(tiles, nn)).abs().max())))
    bad = ((sb[:, 0] - ref0).abs() > 1e-3).nonzero()
    print('   bad entries:', bad[:8].tolist(), 'of', bad.shape[0])
