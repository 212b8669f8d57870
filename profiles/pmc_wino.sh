#!/bin/bash
# rocprofv3 recipe for the Winograd conv kernel (translator conv_3_1 shape), run on the GPU box from the repo root:
#   bash profiles/pmc_wino.sh      -> gpurun_out/pmc_{fetch,write,l2,sq}/ + gpurun_out/wino_stats/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/wino_stats -o s --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o p --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o p --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d gpurun_out/pmc_l2 -o p --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/pmc_sq -o p --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_wino8_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items():
        out[c] = sum(v) / len(v); out[c + '_n'] = len(v)
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/wino_pmc_raw.json', 'w'), indent=1)
PY
grep -i "wino\|Name" gpurun_out/wino_stats/*kernel_stats.csv | head -5
