"""bf16 configuration vs fp32 configuration of the HIP path on STRUCTURED inputs: smooth synthetic frames (a few moving Gaussian blobs on a
gradient background), a few train steps from the seeded initial state.  Prints losses, frame distance, gradient cosines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['KPX_GRAPH'] = '0'
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
dev = torch.device('cuda:0')
res, k, b = 128, 15, 8

def smooth_pair(bsz, res, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, res), np.linspace(-1, 1, res), indexing='ij')
    ims = []
    for shift in (0.0, 0.15):
        im = np.zeros((bsz, res, res, 3), np.float32)
        for i in range(bsz):
            rs_i = np.random.RandomState(seed * 1000 + i)
            base = 0.3 * xx * rs_i.uniform(-1, 1) + 0.3 * yy * rs_i.uniform(-1, 1)
            img = np.stack([base + 0.1 * c for c in range(3)], -1)
            for j in range(5):
                cx, cy = rs_i.uniform(-0.6, 0.6, 2)
                col = rs_i.uniform(-1, 1, 3)
                g = np.exp(-(((xx - cx - shift * (j % 2)) ** 2 + (yy - cy - shift * ((j + 1) % 2)) ** 2) / 0.03))
                img = img + g[..., None] * col
            im[i] = np.clip(img, -1, 1)
        ims.append(im)
    return ims[0], ims[1]

def run(dtype, steps=3):
    ops.set_compute_dtype(dtype)
    cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b}, 'model': {'n_pts': k}, 'paths': {'log_dir': '/tmp/x', 'vggnet': None}}
    vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19, width_div=1), device=dev)
    m = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=res); m.build()
    out = []
    for s in range(steps):
        im, fut = smooth_pair(b, res, 100 + s)
        m.train_step(None, {'image': torch.from_numpy(im).to(dev), 'future_image': torch.from_numpy(fut).to(dev)}, s, b)
        torch.cuda.synchronize()
        out.append((m.loss_values(), m.last['fwd']['final_output'].cpu().numpy().copy(),
                    {n: m.store.grad(n).cpu().numpy().copy() for n in m.store.buckets['G'].entries if n.endswith('kernel')}))
    ops.set_compute_dtype('f32')
    return out
a, c = run('f32'), run('bf16')
for s in range(len(a)):
    dot = gg = ww = 0.0
    per = {}
    for n in a[s][2]:
        g, w = c[s][2][n].astype(np.float64), a[s][2][n].astype(np.float64)
        dot += (g * w).sum(); gg += (g * g).sum(); ww += (w * w).sum()
        sc = n.split('/')[0]
        d = per.setdefault(sc, [0.0, 0.0, 0.0]); d[0] += (g * w).sum(); d[1] += (g * g).sum(); d[2] += (w * w).sum()
    fr = np.linalg.norm(c[s][1] - a[s][1]) / np.linalg.norm(a[s][1])
    print('step %d: loss_G %.5f / %.5f  loss_D %.5f / %.5f  frame rel-L2 %.2e  G-grad cosine %.4f  per net %s' % (
        s, a[s][0]['loss_G'], c[s][0]['loss_G'], a[s][0]['loss_D'], c[s][0]['loss_D'], fr, dot / (gg * ww) ** 0.5,
        {k_: round(v[0] / (v[1] * v[2]) ** 0.5, 4) for k_, v in per.items()}))
