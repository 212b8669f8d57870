#!/bin/bash
# The command-line surface on the GPU box: a few synthetic stage-1 steps (single GPU, and one rank under the launcher with the RCCL exchange kept in),
# a stage-2 run, pseudo labels and the rollout from the checkpoints they write.
set -e
O=/tmp/kpx_cli; rm -rf $O; mkdir -p $O
python3 - <<'PY'
import yaml
c = yaml.safe_load(open('configs/penn.yaml'))
c['paths']['log_dir'] = '/tmp/kpx_cli/logs'; c['paths']['vggnet'] = None
c['training'].update(n_steps=6, log_interval=2, checkpoint_interval=3, test_interval=1000000, batch_size=4)
yaml.safe_dump(c, open('/tmp/kpx_cli/penn.yaml', 'w'))
PY
timeout 300 python3 train.py --mode detector_translator --config $O/penn.yaml --synthetic --synthetic-vgg --steps 6 2>&1 | tail -4
KPX_DP_FORCE_EXCHANGE=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 train.py --mode detector_translator --config $O/penn.yaml --synthetic --synthetic-vgg --steps 6 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -4
ls $O/logs/detector_translator | head
