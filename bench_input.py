"""Throughput of the image-pair input pipeline (kpx_amd.data) on Penn-Action-sized synthetic JPEG frames (480x270):
pairs/s of the threaded decode + augment + pinned H2D + on-device conversion, alone and feeding the train step."""
import argparse, io, os, sys, tempfile, time
import numpy as np
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, ROOT)


def make_dataset(root, videos=8, frames=40, w=480, h=270):
    rs = np.random.RandomState(0)
    for v in range(videos):
        os.makedirs(os.path.join(root, '%04d' % v))
        low = rs.rand(h // 16 + 2, w // 16 + 2, 3)
        bg = np.asarray(Image.fromarray((low * 255).astype(np.uint8)).resize((w, h), Image.BILINEAR)).astype(np.float32)
        for i in range(frames):
            img = np.clip(bg + rs.randn(h, w, 3) * 6 + i, 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, '%04d' % v, '%06d.jpg' % (i + 1)), format='JPEG', quality=90)
    with open(os.path.join(root, 'train_set.txt'), 'w') as f:
        f.write(''.join('%04d 0\n' % v for v in range(videos)))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--batches', type=int, default=40)
    ap.add_argument('--threads', type=int, default=12); ap.add_argument('--train', action='store_true')
    a = ap.parse_args()
    import kpx_amd
    dev = torch.device('cuda', 0)
    with tempfile.TemporaryDirectory() as root:
        make_dataset(root)
        it = iter(kpx_amd.ImagePairDataLoader(root, 'train', random_order=True, randomness=True).batches(a.batch, dev, num_preprocess_threads=a.threads))
        for _ in range(3): next(it)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.batches): b = next(it)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('input pipeline alone: %.0f pairs/s (%d threads, batch %d)' % (a.batch * a.batches / dt, a.threads, a.batch))
        if a.train:
            cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': a.batch}, 'model': {'n_pts': 15},
                   'paths': {'log_dir': '/tmp/kpx_bench', 'vggnet': None}}
            vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19), device=dev)
            model = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=128); model.build()
            for i in range(3): model.train_step(None, next(it), i, a.batch)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(a.batches): model.train_step(None, next(it), 3 + i, a.batch)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print('train step fed by the pipeline: %.0f pairs/s' % (a.batch * a.batches / dt))


if __name__ == '__main__':
    main()
