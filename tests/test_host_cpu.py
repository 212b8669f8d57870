"""CPU tests of the host logic: variable declaration pass (shape-only build), reference naming / checkpoint layout,
config surface, LR schedule, and the data-parallel gradient exchange on 2 gloo processes."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

from oracle import restatement as R

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': 4},
       'model': {'n_pts': 15}, 'paths': {'log_dir': '/tmp/kpx_cpu', 'vggnet': None}}


def build_cpu_model(n_pts=15, res=128):
    import kpx_amd
    cfg = {**CFG, 'model': {'n_pts': n_pts}}
    m = kpx_amd.DetectorTranslatorModel(cfg, device='cpu', image_size=res)
    m.build()
    return m


def test_build_declares_the_reference_variable_manifest():
    m = build_cpu_model()
    exp = m.store.export_numpy()
    man = R.variable_manifest(15)
    assert list(exp.keys()).sort() == list(man.keys()).sort() and set(exp) == set(man)
    for k, shape in man.items():
        assert tuple(exp[k].shape) == tuple(shape), k
    # same RandomState(1234) creation order as SURVEY 8d / the oracle -> bit-identical initial weights
    init = R.init_variables(15)
    for k in man:
        assert np.array_equal(exp[k], init[k]), k
    # variable split by the substring 'img_discr' (reference detector_translator_model.py:191-192)
    assert sum(int(np.prod(s)) for n, (o, s) in m.store.buckets['G'].entries.items()) == 6419107
    assert sum(int(np.prod(s)) for n, (o, s) in m.store.buckets['D'].entries.items()) == 44721088
    for b in m.store.buckets.values():
        for name, (off, shape) in b.entries.items():
            assert off % 4 == 0                      # 16-B aligned views for the vectorised kernels


def test_checkpoint_arrays_follow_tf_saver_naming():
    m = build_cpu_model(n_pts=3, res=32)
    arrays = m.checkpoint_arrays()
    for name in ('global_step', 'beta1_power', 'beta2_power', 'beta1_power_1', 'beta2_power_1',
                 'translator/conv_6_0/conv2d/kernel', 'translator/conv_6_1/conv2d/kernel', 'translator/conv_6_0/conv2d/kernel/Adam',
                 'pose_encoder/conv_0/conv2d/bias/Adam_1', 'img_discr/D_logit/conv2d/kernel', 'pose_encoder/b_norm_7_1/moving_variance'):
        assert name in arrays, name
    assert not any(k.startswith('content_vgg') or 'vgg' in k for k in arrays)        # VGG weights are constants, not saved
    assert not any('moving_mean/Adam' in k for k in arrays)
    assert arrays['translator/conv_6_1/conv2d/kernel'].shape == (3, 3, 64, 1)


def test_lr_schedule_and_config_surface():
    m = build_cpu_model(n_pts=3, res=32)
    for step in (0, 1, 20000, 50000):
        m.global_step = step
        assert abs(float(m.current_lr()) - float(R.exponential_decay(1e-4, step, 20000, 0.95))) == 0.0
    cfg = yaml.load(open(os.path.join(REPO, 'configs', 'penn.yaml')), Loader=yaml.FullLoader)
    assert set(cfg) == {'paths', 'training', 'model'}
    assert set(cfg['training']) == {'n_steps', 'summary_interval', 'test_interval', 'checkpoint_interval', 'log_interval', 'batch_size', 'lr'}
    assert cfg['model']['n_pts'] == 40 and cfg['training']['batch_size'] == 16 and cfg['training']['lr'] == {'start_val': 0.0001, 'step': 20000, 'decay': 0.95}
    import train
    with pytest.raises(Exception, match='unknown model'):            # reference train.py:117-123
        train._get_model_by_mode('no_such_mode', cfg, 0)
    mg = train._get_model_by_mode('motion_generator', cfg, 0, device='cpu')          # stage 2 (constructed only: no kernels run on CPU)
    assert mg.name == 'motion_generator' and mg.cell_info == [1024, 1024] and mg.vae_dim == 64


def test_synthetic_pairs_follow_the_loader_contract():
    from kpx_amd.synthetic import synthetic_pair
    p = synthetic_pair(3, res=16, seed0=4, seed1=5)
    assert set(p) == {'image', 'future_image'}
    for v in p.values():
        assert v.shape == (3, 16, 16, 3) and v.dtype == np.float32 and v.min() >= -1 and v.max() <= 1
    a, b = R.synthetic_pair(3, res=16, seed0=4, seed1=5)
    assert np.array_equal(p['image'], a) and np.array_equal(p['future_image'], b)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, REPO)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    m = build_cpu_model(n_pts=3, res=32)
    assert m.world_size == world
    out = {}
    for which in ('D', 'G'):
        b = m.store.buckets[which]
        b.grads.copy_(torch.arange(b.grads.numel(), dtype=torch.float32) * 1e-3 + (rank + 1))
        work = m.exchange_gradients(which, async_op=(which == 'D'))      # D: asynchronous handle, G: blocking
        if work is not None:
            work.wait()
        g = b.grads
        want = torch.arange(b.grads.numel(), dtype=torch.float32) * 1e-3 * world + sum(range(1, world + 1))
        out[which] = bool(torch.allclose(g, want, rtol=1e-6))
        # parameter views alias the flat bucket: a bucket update is visible through every named variable
        b.params.add_(1.0)
    name = 'translator/conv_1_0/conv2d/kernel'
    out['alias'] = bool(torch.allclose(m.store[name], torch.from_numpy(R.init_variables(3, res=32)[name]) + 1.0))
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_gradient_exchange_two_gloo_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in results:
        assert out == {'D': True, 'G': True, 'alias': True}, (rank, out)


def _dp_bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), KPX_DP_BF16='D')
    import torch.distributed
    sys.path.insert(0, REPO)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    m = build_cpu_model(n_pts=3, res=32)
    out = {}
    for which in ('D', 'G'):
        b = m.store.buckets[which]
        local = [torch.sin(torch.arange(b.grads.numel(), dtype=torch.float32) * 0.37 + r) * (r + 1.5) for r in range(world)]
        b.grads.copy_(local[rank])
        work = m.exchange_gradients(which, async_op=(which == 'D'))
        if work is not None:
            work.wait()
        if which == 'D':     # the flagged bucket: every rank's gradient rounded to bf16, summed in bf16, widened back
            want = local[0].to(torch.bfloat16)
            for r in range(1, world):
                want = want + local[r].to(torch.bfloat16)
            out[which] = bool(torch.equal(b.grads, want.float())) and b.grads.dtype == torch.float32
            out['D_close_to_fp32_sum'] = bool(torch.allclose(b.grads, sum(local), rtol=2e-2, atol=2e-2))
        else:                # the other bucket keeps the exact fp32 exchange
            out[which] = bool(torch.equal(b.grads, local[0] + local[1]))
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_bf16_exchange_flag_two_gloo_ranks():
    """KPX_DP_BF16=D (off by default): the discriminator's flat gradient bucket crosses the links as bf16, the generator's stays fp32."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in results:
        assert out == {'D': True, 'D_close_to_fp32_sum': True, 'G': True}, (rank, out)


def test_f43_layer_policy_follows_the_float64_arbiter_findings():
    """Which layers may run the F(4x4,3x3) kernel is host logic (a per-layer attribute declared in networks.py, ops.WINO43): data gradients everywhere, forward only on VGG19
    and the translator's 64x64 / 128x128 layers; the key-point detector, the image encoder and the translator's 32x32 layers keep
    F(2x2,3x3) in the forward direction.  Channel limits: K >= 16 gathered, more than 32 produced."""
    from kpx_amd import ops
    if not ops.WINO43:
        pytest.skip('KPX_WINO43=0')
    m = build_cpu_model()
    fwd, dgrad = set(), set()
    for name, v in m.store.vars.items():
        if name.endswith('/kernel') and v.dim() == 4 and v.shape[0] == 3:
            cin, cout = int(v.shape[2]), int(v.shape[3])
            f43 = m.store.layer_attrs[name]['f43_fwd']
            if ops._wino43_wanted(name, cin, cout, 0, f43):
                fwd.add(name.split('/conv2d')[0])
            if ops._wino43_wanted(name, cin, cout, 1, f43):
                dgrad.add(name.split('/conv2d')[0])
    assert fwd == {'translator/conv_%d_%d' % (i, j) for i in (3, 4, 5) for j in (0, 1)}
    assert not any(n.startswith(('pose_encoder', 'image_encoder', 'translator/conv_1', 'translator/conv_2', 'img_discr')) for n in fwd)
    # data gradients: every 3x3 layer gathering >= 16 and producing > 32 channels, the detector's included; never the 4-channel head's input side
    assert {'translator/conv_1_0', 'translator/conv_2_1', 'pose_encoder/encoder/conv_4', 'pose_encoder/conv_7_0', 'image_encoder/encoder/conv_6'} <= dgrad
    assert not any(n.startswith('translator/conv_6') for n in dgrad | fwd)          # the 4-channel head: K = 4 gathered in the gradient, 4 produced in the forward
    assert ops._wino43_wanted('vgg/conv3_2', 256, 256, 0) and ops._wino43_wanted('vgg/conv3_2', 256, 256, 1)
    assert not ops._wino43_wanted('vgg/conv1_1', 3, 64, 0)
    # the attribute, not the name, decides: the same filter name with the attribute cleared stays off F(4x4,3x3) in the forward direction only
    assert not ops._wino43_wanted('translator/conv_3_1/conv2d/kernel', 128, 128, 0, False) and ops._wino43_wanted('translator/conv_3_1/conv2d/kernel', 128, 128, 1, False)
    assert ops._wino43_wanted('pose_encoder/renamed_scope/conv2d/kernel', 128, 128, 0, True)


# ------------------------------------------------------------------------------------------------ bench.py supervisor (N > 1 cannot hang)
_STUB_CHILD = r'''
import json, os, sys, time
d, k, r = os.environ['KPX_BENCH_DIR'], os.environ['KPX_BENCH_ATTEMPT'], os.environ['RANK']
form = os.environ['KPX_DP_GRAPH']
mark = lambda sfx: open(os.path.join(d, 'a%s_r%s.%s' % (k, r, sfx)), 'w').close()
behaviour = os.environ.get('STUB_' + form.upper(), 'ok')
if behaviour == 'hang_before_warmup' and r == os.environ.get('STUB_BAD_RANK', '1'):
    time.sleep(3600)
if behaviour == 'crash' and r == os.environ.get('STUB_BAD_RANK', '1'):
    sys.exit(7)
mark('warm')
while not all(os.path.exists(os.path.join(d, 'a%s_r%d.warm' % (k, q))) for q in range(int(os.environ['WORLD_SIZE']))):
    time.sleep(0.02)          # (the real ranks meet in a barrier here: rank 0 cannot print while another rank hangs)
if behaviour == 'hang_in_steps':
    time.sleep(3600)
if r == '0':
    print(json.dumps({'form': form, 'graph_env': os.environ.get('KPX_GRAPH', ''), 'world': os.environ['WORLD_SIZE'],
                      'dp_fallbacks': json.loads(os.environ['KPX_BENCH_FALLBACKS'])}), flush=True)
mark('done')
'''


def _supervisor_proc(ranks, world, stub, key, q):
    sys.path.insert(0, REPO)
    import bench
    rc, attempts = bench.supervise(ranks, world, [sys.executable, stub], key, plan=[('one', {}), ('segments', {}), ('inline', {'KPX_GRAPH': '0'})],
                                   warm_deadline=3.0, run_deadline=3.0, settle=10.0, log=open(os.devnull, 'w'))
    q.put((ranks, rc, attempts))


@pytest.mark.parametrize('one,segments,want_form', [('hang_before_warmup', 'ok', 'segments'), ('hang_in_steps', 'crash', 'inline'), ('ok', 'ok', 'one')])
@pytest.mark.parametrize('layout', ['one_supervisor', 'supervisor_per_rank'])
def test_bench_supervisor_restarts_hung_or_failed_ranks_with_the_next_form(tmp_path, capfd, monkeypatch, one, segments, want_form, layout):
    """`bench.py --gpus N` cannot hang: a rank that hangs before its warm-up completes, hangs in the timed steps or crashes makes the
    supervisor(s) kill every child of the attempt and start FRESH children with the next data-parallel form; the line rank 0 prints names
    the failed attempts.  Both layouts: one supervisor for all ranks (`python bench.py --gpus 2`) and one supervisor per rank that agree
    through the shared directory (under torch.distributed.run)."""
    import multiprocessing as mp
    stub = str(tmp_path / 'stub_child.py')
    open(stub, 'w').write(_STUB_CHILD)
    monkeypatch.setenv('STUB_ONE', one)
    monkeypatch.setenv('STUB_SEGMENTS', segments)
    key = 'test_%d_%s_%s' % (os.getpid(), layout, want_form)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    groups = [[0, 1]] if layout == 'one_supervisor' else [[0], [1]]
    procs = [ctx.Process(target=_supervisor_proc, args=(g, 2, stub, key, q)) for g in groups]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for ranks, rc, attempts in results:
        assert rc == 0, (ranks, attempts)
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines                                    # exactly ONE JSON line, from the attempt that worked
    import json
    line = json.loads(lines[0])
    assert line['form'] == want_form and line['world'] == '2'
    failed = {'segments': ['one'], 'inline': ['one', 'segments'], 'one': []}[want_form]
    assert [a['form'] for a in line['dp_fallbacks']] == failed
    assert line['graph_env'] == ('0' if want_form == 'inline' else '')


def test_bench_attempt_plan_puts_an_explicit_form_first(monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.delenv('KPX_DP_GRAPH', raising=False)
    assert [f for f, _ in bench.dp_attempt_plan()] == ['segments', 'inline']
    monkeypatch.setenv('KPX_DP_GRAPH', 'one')
    assert [f for f, _ in bench.dp_attempt_plan()] == ['one', 'segments', 'inline']
    monkeypatch.setenv('KPX_DP_GRAPH', 'inline')
    assert bench.dp_attempt_plan() == [('inline', {'KPX_GRAPH': '0'}), ('segments', {})]
