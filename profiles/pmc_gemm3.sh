#!/bin/bash
# rocprofv3 counters for the bf16x3 gather-conv kernel on one discriminator layer (run on the GPU box from the repo root):
#   bash profiles/pmc_gemm3.sh [layer filter, default img_discr/conv_2]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
L=${1:-img_discr/conv_2}
O=gpurun_out/pmc_g3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 bench_layers.py --filter $L > /dev/null 2>&1
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" \
         "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace -d $d -o p --output-format csv -- python3 bench_layers.py --filter $L > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
O = 'gpurun_out/pmc_g3'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'conv_gemm3' in r['Kernel_Name'] or 'conv_igemm' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
json.dump(out, open(O + '/summary.json', 'w'), indent=1)
for f in glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'gemm3' in row['Name'] or 'igemm' in row['Name'] or 'reduce' in row['Name']:
            print(row['Name'][:80], row['Calls'], 'avg_us', float(row['AverageNs']) / 1e3)
PY
