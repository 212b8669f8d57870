#!/bin/bash
# Round-2 profile collection, run on the GPU box from the repo root:   bash profiles/collect_r02.sh
# Writes raw rocprofv3 output under gpurun_out/r02/ and the summaries that are committed under profiles/ (names r02_*).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02; mkdir -p $O
# 1. the bench line itself + per-kernel stats of the same command
python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/step -o s --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_profiled.json 2>/dev/null
# 2. the roofline micro-benchmarks alone (kernel averages must agree with the bench line's avg_launch_ms)
rocprofv3 --kernel-trace --stats -d $O/roof -o s --output-format csv -- python3 bench.py --roofline-only > $O/roofline_only.json 2>/dev/null
# 3. PMC, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never with --stats / sys-trace)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  d=$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace -d $d -o p --output-format csv -- python3 bench.py --roofline-only > /dev/null 2>&1
done
# 4. the bf16 configuration
python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bf16.json 2>/dev/null
python3 - <<'PY'
import csv, glob, collections, json, shutil, os
O = 'gpurun_out/r02'
shutil.copy(glob.glob(O + '/step/*kernel_stats.csv')[0], 'profiles/r02_bench_b32_kernel_stats.csv')
shutil.copy(glob.glob(O + '/roof/*kernel_stats.csv')[0], 'profiles/r02_roofline_only_kernel_stats.csv')
for f, dst in (('bench.json', 'r02_bench.json'), ('bench_bf16.json', 'r02_bench_bf16.json'), ('roofline_only.json', 'r02_roofline_only.json')):
    shutil.copy(os.path.join(O, f), 'profiles/' + dst)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        for key, tag in (('conv_wino43_kernel<0, false>', 'wino43'), ('conv_wino_v2_kernel<2, 0>', 'wino'), ('gauss_fwd', 'render'), ('conv3x3_bf16_kernel', 'bf16'), ('conv_igemm_kernel<128, 128', 'direct')):
            if key in r['Kernel_Name']:
                pmc[tag][r['Counter_Name']].append(float(r['Counter_Value']))
def mean(v): return sum(v) / len(v) if v else None
out = {}
for tag, d in pmc.items():
    out[tag] = {c: mean(v) for c, v in d.items()}
    out[tag]['dispatches'] = {c: len(v) for c, v in d.items()}
json.dump(out, open(O + '/pmc_raw.json', 'w'), indent=1)
w, r, w43 = out.get('wino', {}), out.get('render', {}), out.get('wino43', {})
if w43.get('FETCH_SIZE') is not None and w43.get('WRITE_SIZE') is not None:
    fetch, write = w43['FETCH_SIZE'] * 1024 * 2, w43['WRITE_SIZE'] * 1024
    json.dump({'kernel': 'conv_wino43_kernel<0, false> F(4x4,3x3) fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1)',
               'source': 'rocprofv3 --pmc, separate passes (profiles/collect_r02.sh), mean over the dispatches of `bench.py --roofline-only`',
               **{k: v for k, v in w43.items() if k != 'dispatches'}, 'dispatches': w43['dispatches'],
               'fetch_bytes_corrected': fetch, 'write_bytes': write, 'algorithmic_bytes': 134807552,
               'traffic_bytes_per_launch': fetch + write}, open('profiles/r02_wino43_pmc.json', 'w'), indent=1)
if w.get('FETCH_SIZE') is not None and w.get('WRITE_SIZE') is not None:
    # guide: FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950 wide reads -> x2; WRITE_SIZE (KB) is exact for 16-B stores
    fetch, write = w['FETCH_SIZE'] * 1024 * 2, w['WRITE_SIZE'] * 1024
    json.dump({'kernel': 'conv_wino_v2_kernel<2, 0> fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1)',
               'source': 'rocprofv3 --pmc, separate passes (profiles/collect_r02.sh), mean over the dispatches of `bench.py --roofline-only`',
               **{k: v for k, v in w.items() if k != 'dispatches'}, 'dispatches': w['dispatches'],
               'fetch_bytes_corrected': fetch, 'write_bytes': write, 'algorithmic_bytes': 134807552,
               'traffic_bytes_per_launch': fetch + write}, open('profiles/r02_wino_pmc.json', 'w'), indent=1)
if r.get('WRITE_SIZE') is not None:
    json.dump({'kernel': 'gauss_fwd_reg_kernel [64,128,128,15], nine rotating 62.9 MB outputs (566 MB > Infinity Cache)',
               'source': 'rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes (profiles/collect_r02.sh)',
               **{k: v for k, v in r.items() if k != 'dispatches'}, 'dispatches': r['dispatches'],
               'write_bytes': r['WRITE_SIZE'] * 1024, 'fetch_bytes_corrected': (r.get('FETCH_SIZE') or 0) * 1024 * 2,
               'algorithmic_bytes': 62922240, 'traffic_bytes_per_launch': r['WRITE_SIZE'] * 1024 + (r.get('FETCH_SIZE') or 0) * 1024 * 2},
              open('profiles/r02_render_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
for f in ('profiles/r02_roofline_only_kernel_stats.csv',):
    for row in csv.DictReader(open(f)):
        if any(k in row['Name'] for k in ('wino43', 'wino_v2', 'gauss_fwd', 'bf16_kernel', 'conv_igemm')):
            print(row['Name'][:70], row['Calls'], 'avg_us', float(row['AverageNs']) / 1e3)
PY
