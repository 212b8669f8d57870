"""Stage-2 (motion generator) train-step time at the reference's size: K=40 key points, LSTM 2x1024, vae_dim 64, batch 16 / 64."""
import sys, time, torch
sys.path.insert(0, '.')
import kpx_amd, numpy as np
dev = torch.device('cuda', 0)
for b in (16, 64):
    cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b},
           'model': {'n_pts': 40, 'n_action': 9, 'cell_info': [1024, 1024], 'vae_dim': 64}, 'paths': {'log_dir': '/tmp/kpx_motion'}}
    m = kpx_amd.MotionGeneratorModel(cfg, device=dev); m.build()
    rs = np.random.RandomState(0)
    feed = {'keypoints': torch.from_numpy((rs.rand(b, 40, 2) * 1.6 - .8).astype(np.float32)).to(dev),
            'real_seq': torch.from_numpy((rs.rand(b, 32, 40, 2) * 1.6 - .8).astype(np.float32)).to(dev),
            'action_code': torch.from_numpy(np.eye(9, dtype=np.float32)[rs.randint(0, 9, size=b)]).to(dev)}
    for i in range(3): m.train_step(None, feed, i, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): m.train_step(None, feed, 3 + i, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print('motion generator train step, batch %d: %.1f ms = %.0f sequences/s  (loss_D %.3f loss_G %.1f)' % (b, dt * 1e3, b / dt, m.loss_values()['loss_D'], m.loss_values()['loss_G']))
