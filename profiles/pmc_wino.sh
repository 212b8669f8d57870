cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace -d gpurun_out/pmcA -o a --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmcB -o b --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace -d gpurun_out/pmcC -o c --output-format csv -- python3 bench_layers.py --filter translator/conv_3_1 > /dev/null 2>&1
find gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC -name "*counter_collection.csv" | head
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        if 'wino_pp' in k or 'conv_igemm' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in acc.items():
        print(f.split('/')[1], k, {c: sum(v)/len(v) for c,v in d.items()}, 'n=', len(next(iter(d.values()))))
PY
