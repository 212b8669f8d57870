"""Random eligible shapes of conv_wino43b_kernel against the fp32-MFMA kernel (rel-L2 < 1e-5) and against itself (three launches: bitwise equal).
   python scratch/w43b_stress.py [cases] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    pack = rnd.random() < 0.3
    if pack:
        n, h, w = 2 * rnd.randint(1, 24), 16, 16
    else:
        n, h, w = rnd.randint(1, 9), 16 * rnd.randint(1, 6), 32 * rnd.randint(1, 4)
    cin = 4 * rnd.randint(4, 80)
    cout = rnd.randint(33, 300)
    act = rnd.randint(0, 2)
    ldx = cin + 4 * rnd.randint(0, 3)
    g = torch.Generator().manual_seed(c)
    xb = torch.randn(n, h, w, ldx, generator=g).to(dev)
    wt = (torch.randn(3, 3, cin, cout, generator=g) / (9 * cin) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    if not lib.kpx_conv3x3_wino43b_eligible(n, h, w, cin, cout, ldx, xb.data_ptr()) or not lib.kpx_conv3x3_wino43_eligible(n, h, w, cin, cout, ldx, xb.data_ptr()):
        continue
    u = torch.empty(lib.kpx_wino43b_u_bytes(cin, cout), dtype=torch.uint8, device=dev)
    check(lib.kpx_wino43b_filter_transform_f32(wt.data_ptr(), cin, cout, 0, u.data_ptr(), ops._stream()), 'tf')
    uo = torch.empty(lib.kpx_wino43_u_bytes(cin, cout), dtype=torch.uint8, device=dev)
    check(lib.kpx_wino43_filter_transform_f32(wt.data_ptr(), cin, cout, 0, uo.data_ptr(), ops._stream()), 'tf')
    ys = []
    for rep in range(3):
        y = torch.full((n, h, w, cout), float('nan'), device=dev)
        check(lib.kpx_conv3x3_wino43b_f32(xb.data_ptr(), n, h, w, cin, ldx, u.data_ptr(), b.data_ptr(), y.data_ptr(), cout, cout, act, None, 0, None, 0, None, None, 0, None, ops._stream()), 'conv')
        ys.append(y)
    yo = torch.empty((n, h, w, cout), device=dev)
    check(lib.kpx_conv3x3_wino43_f32(xb.data_ptr(), n, h, w, cin, ldx, uo.data_ptr(), b.data_ptr(), yo.data_ptr(), cout, cout, act, ops._stream()), 'conv')
    torch.cuda.synchronize()
    rel = float((ys[0].double() - yo.double()).norm() / yo.double().norm())
    same = bool(torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]))
    ok = rel < 1e-5 and same and bool(torch.isfinite(ys[0]).all())
    bad += not ok
    print('%s n%d %dx%d %d(ld %d)->%d act%d  rel %.2e  repeat-equal %s' % ('ok ' if ok else 'BAD', n, h, w, cin, ldx, cout, act, rel, same), flush=True)
print('bad:', bad)
