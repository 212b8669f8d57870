#!/bin/bash
# Round-6 profile collection, run on the GPU box from the repo root:   bash profiles/collect_r06.sh
# Writes raw rocprofv3 output under gpurun_out/r06/ and the summaries that are committed under profiles/ (names r06_*).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r06; mkdir -p $O
if [ "$1" != "post" ]; then
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
# 1. the bench lines (headline c1 with graph replay and eager, c3, c4)
python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
KPX_GRAPH=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_eager.json 2>/dev/null
python3 bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2>/dev/null
python3 bench.py --dtype bf16 --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bf16.json 2>/dev/null
# 2. per-kernel statistics of the step alone (13 profiled steps, eager launches: one trace record per kernel either way)
KPX_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/step -o s --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_profiled.json 2>/dev/null
KPX_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/step_c3 -o s --output-format csv -- python3 bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c3_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/step_c4 -o s --output-format csv -- python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4_profiled.json 2>/dev/null
KPX_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/step_bf16 -o s --output-format csv -- python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_bf16_profiled.json 2>/dev/null
# 3. the roofline micro-benchmarks alone (kernel averages must agree with the bench line's avg_launch_ms): c1 legs, and the own-shape legs of c3 / c4
rocprofv3 --kernel-trace --stats -d $O/roof -o s --output-format csv -- python3 bench.py --roofline-only > $O/roofline_only.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/roof_c3 -o s --output-format csv -- python3 bench.py --roofline-only --config c3 > $O/roofline_only_c3.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/roof_c4 -o s --output-format csv -- python3 bench.py --roofline-only --config c4 > $O/roofline_only_c4.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/roof_bf16 -o s --output-format csv -- python3 bench.py --roofline-only --dtype bf16 > $O/roofline_only_bf16.json 2>/dev/null
# 3b. per-layer table of the convolutions of one step, each direction alone (DESIGN sections 6 / 9 quote it)
python3 bench_layers.py > $O/bench_layers.txt 2>/dev/null
python3 bench_layers.py --dtype bf16 > $O/bench_layers_bf16.txt 2>/dev/null
# 4. PMC, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never with --stats / sys-trace)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  d=$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace -d $d -o p --output-format csv -- python3 bench.py --roofline-only > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace -d ${d}_c3 -o p --output-format csv -- python3 bench.py --roofline-only --config c3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace -d ${d}_bf16 -o p --output-format csv -- python3 bench.py --roofline-only --dtype bf16 > /dev/null 2>&1
done
# (gpurun merges only gpurun_out/ back: run this post-processing step again in the build container -- `bash profiles/collect_r06.sh post` --
#  to copy the summaries into profiles/)
fi
python3 - <<'PY'
import csv, glob, collections, json, shutil, os, subprocess
O = 'gpurun_out/r06'
def cp(pattern, dst):
    f = glob.glob(pattern, recursive=True)
    if f: shutil.copy(f[0], 'profiles/' + dst)
if os.path.exists('.git_head'): shutil.copy('.git_head', 'profiles/r06_commit.txt')
cp(O + '/step/**/*kernel_stats.csv', 'r06_step_kernel_stats.csv')
cp(O + '/step_c3/**/*kernel_stats.csv', 'r06_step_c3_kernel_stats.csv')
cp(O + '/step_c4/**/*kernel_stats.csv', 'r06_step_c4_kernel_stats.csv')
cp(O + '/roof/**/*kernel_stats.csv', 'r06_roofline_only_kernel_stats.csv')
cp(O + '/roof_c3/**/*kernel_stats.csv', 'r06_roofline_only_c3_kernel_stats.csv')
cp(O + '/roof_c4/**/*kernel_stats.csv', 'r06_roofline_only_c4_kernel_stats.csv')
cp(O + '/roof_bf16/**/*kernel_stats.csv', 'r06_roofline_only_bf16_kernel_stats.csv')
cp(O + '/step_bf16/**/*kernel_stats.csv', 'r06_step_bf16_kernel_stats.csv')
if os.path.exists(O + '/bench_layers.txt'): shutil.copy(O + '/bench_layers.txt', 'profiles/r06_bench_layers.txt')
if os.path.exists(O + '/bench_layers_bf16.txt'): shutil.copy(O + '/bench_layers_bf16.txt', 'profiles/r06_bench_layers_bf16.txt')
for f, dst in (('bench.json', 'r06_bench.json'), ('bench_eager.json', 'r06_bench_eager.json'), ('bench_c3.json', 'r06_bench_c3.json'), ('bench_c4.json', 'r06_bench_c4.json'), ('bench_bf16.json', 'r06_bench_bf16.json'), ('roofline_only_bf16.json', 'r06_roofline_only_bf16.json'),
               ('roofline_only.json', 'r06_roofline_only.json'), ('roofline_only_c3.json', 'r06_roofline_only_c3.json'), ('roofline_only_c4.json', 'r06_roofline_only_c4.json')):
    if os.path.exists(os.path.join(O, f)): shutil.copy(os.path.join(O, f), 'profiles/' + dst)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    c3 = '_c3/' in f.replace(os.sep, '/')
    b16 = '_bf16/' in f.replace(os.sep, '/')
    for r in csv.DictReader(open(f)):
        if b16:
            for key, tag in (('conv3x3_bf16s_kernel<2, 2, 4, 32, 0>', 'bf16'), ('conv3x3_wgrad_bf16_kernel<2, 4, 1, 1>', 'wgrad_bf16')):
                if key in r['Kernel_Name']:
                    pmc[tag][r['Counter_Name']].append(float(r['Counter_Value']))
            continue
        if c3:
            for key, tag in (('conv_wino43_kernel<0, false>', 'wino43_c3'), ('conv_wino43b_kernel<0, false>', 'wino43b_c3'), ('gauss_fwd', 'render_c3')):
                if key in r['Kernel_Name']:
                    pmc[tag][r['Counter_Name']].append(float(r['Counter_Value']))
            continue
        for key, tag in (('conv_wino_wgrad_kernel<2, 2>', 'wgrad'), ('conv_wino43_kernel<0, false>', 'wino43'), ('conv_wino43b_kernel<0, false>', 'wino43b'), ('conv_wino_v2_kernel<2, 0>', 'wino'), ('gauss_fwd', 'render'), ('conv_igemm_kernel<128, 128', 'direct'),
                         ('conv_gemm3_kernel<128, 128, 2, 4, false, 3,', 'gemm3')):
            if key in r['Kernel_Name']:
                pmc[tag][r['Counter_Name']].append(float(r['Counter_Value']))
mean = lambda v: sum(v) / len(v) if v else None
out = {tag: dict({c: mean(v) for c, v in d.items()}, dispatches={c: len(v) for c, v in d.items()}) for tag, d in pmc.items()}
json.dump(out, open(O + '/pmc_raw.json', 'w'), indent=1)
for tag, name, alg, kernel in (('wino43', 'r06_wino43_pmc.json', 134807552, 'conv_wino43_kernel<0, false> F(4x4,3x3) fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1), fp32 MFMA (the roofline_wino43_f32mfma leg)'),
                               ('wino43b', 'r06_wino43b_pmc.json', 134807552, 'conv_wino43b_kernel<0, false> F(4x4,3x3) bf16x3 fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1)'),
                               ('wino43b_c3', 'r06_wino43b_c3_pmc.json', 2 * 16 * 128 * 128 * 128 * 4 + 9 * 128 * 128 * 4, 'conv_wino43b_kernel<0, false> F(4x4,3x3) bf16x3 fwd 3x3 s1 128->128 @128x128 B=16 (translator conv_3_1 of the 256x256 K=40 network)'),
                               ('wino', 'r06_wino_pmc.json', 134807552, 'conv_wino_v2_kernel<2, 0> fwd 3x3 s1 128->128 @64x64 B=32'),
                               ('render', 'r06_render_pmc.json', 62922240, 'gauss_fwd_reg_kernel [64,128,128,15], nine rotating 62.9 MB outputs'),
                               ('direct', 'r06_direct_pmc.json', 134807552, 'conv_igemm_kernel<128,128,..> fwd 3x3 s1 128->128 @64x64 B=32 (KPX_NO_WINO=1 KPX_NO_GEMM3=1)'),
                               # img_discr conv_3: x 64*18*18*256*4 + w 16*256*512*4 + y 64*10*10*512*4 (the split-K slabs and their reduce are extra, and counted)
                               ('gemm3', 'r06_gemm3_pmc.json', 64 * 18 * 18 * 256 * 4 + 16 * 256 * 512 * 4 + 64 * 10 * 10 * 512 * 4,
                                'conv_gemm3_kernel<128,128,2,4,false,3> fwd 4x4 s2 256->512 @18x18 N=64 (img_discr conv_3), kernel only (split-K slabs written, not reduced)'),
                               # bf16 configuration: x 32*64*64*128*2 read + y the same written + the bf16 filter fragments 9*128*128*2
                               ('bf16', 'r06_bf16_pmc.json', 2 * 32 * 64 * 64 * 128 * 2 + 9 * 128 * 128 * 2, 'conv3x3_bf16s_kernel<2,2,4,32,0> fwd 3x3 s1 128->128 @64x64 B=32 (translator conv_3_1), bf16 tensors in HBM'),
                               ('wgrad_bf16', 'r06_wgrad_bf16_pmc.json', 32 * 64 * 64 * (256 + 128) * 2 + 9 * 256 * 128 * 4,
                                'conv3x3_wgrad_bf16_kernel<2,4,1,1> wgrad 3x3 s1 256->128 @64x64 B=32 (translator conv_3_0), kernel only (split slabs written, not reduced)'),
                               # translator conv_3_0 weight gradient: x 32*64*64*256*4 + dy 32*64*64*128*4 read, split slabs written (their reduce is a separate kernel)
                               ('wgrad', 'r06_wgrad_pmc.json', 32 * 64 * 64 * (256 + 128) * 4 + 9 * 256 * 128 * 4,
                                'conv_wino_wgrad_kernel<2, 2> wgrad 3x3 s1 256->128 @64x64 B=32 (translator conv_3_0), kernel only (split slabs written, not reduced)'),
                               ('wino43_c3', 'r06_wino43_c3_pmc.json', 2 * 16 * 128 * 128 * 128 * 4 + 9 * 128 * 128 * 4, 'conv_wino43_kernel<0, false> F(4x4,3x3) fwd 3x3 s1 128->128 @128x128 B=16 (translator conv_3_1 of the 256x256 K=40 network)'),
                               ('render_c3', 'r06_render_c3_pmc.json', 32 * (256 * 256 * 40 * 4 + 40 * 8), 'gauss_fwd kernel [32,256,256,40] (--config c3), three rotating 335.5 MB outputs')):
    d = out.get(tag, {})
    if d.get('WRITE_SIZE') is None: continue
    # guide: FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950 wide reads -> x2; WRITE_SIZE (KB) is exact for 16-B stores
    fetch, write = (d.get('FETCH_SIZE') or 0) * 1024 * 2, d['WRITE_SIZE'] * 1024
    json.dump(dict({k: v for k, v in d.items() if k != 'dispatches'}, kernel=kernel, dispatches=d['dispatches'],
                   source='rocprofv3 --pmc, separate passes (profiles/collect_r06.sh), mean over the dispatches of `bench.py --roofline-only`',
                   fetch_bytes_corrected=fetch, write_bytes=write, algorithmic_bytes=alg, traffic_bytes_per_launch=fetch + write),
              open('profiles/' + name, 'w'), indent=1)
for csvf, steps, dst in (('profiles/r06_step_kernel_stats.csv', 13, 'profiles/r06_step_breakdown.txt'), ('profiles/r06_step_c3_kernel_stats.csv', 13, 'profiles/r06_step_c3_breakdown.txt'),
                         ('profiles/r06_step_bf16_kernel_stats.csv', 13, 'profiles/r06_step_bf16_breakdown.txt')):
    if os.path.exists(csvf):
        txt = subprocess.run(['python3', 'profiles/step_breakdown.py', csvf, str(steps)], capture_output=True, text=True).stdout
        open(dst, 'w').write(txt); print(txt)
PY
# how the streams overlap inside a step (needs the raw kernel trace, which stays under gpurun_out/)
[ -f $O/step/s_kernel_trace.csv ] && python3 profiles/overlap.py $O/step/s_kernel_trace.csv 13 > profiles/r06_step_overlap.txt
for f in bench bench_eager bench_c3 bench_c4 bench_bf16; do python3 -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().split(chr(10))[-1]); print('$f', d['value'], d['unit'], d['ms_per_step'], 'host', d.get('host_work_ms_per_step_min'))
except Exception as e: print('$f ERR', e)"; done
