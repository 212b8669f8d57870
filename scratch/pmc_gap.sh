#!/bin/bash
# launch-gap study: GRBM_GUI_ACTIVE (kernel duration in cycles per XCD x 8) against SQ_BUSY_CYCLES (x 32 SEs) / SQ_WAVE_CYCLES for a few launch sizes
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for args in "1 64 128 128" "4 64 128 128" "32 64 128 128" "64 64 128 128"; do
  O=gpurun_out/pmc_gap; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace -d $O/p -o p --output-format csv -- python3 scratch/bf16s_one.py $args > $O/out.txt 2>&1
  python3 - "$O" "$args" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
pmc = collections.defaultdict(list)
for f in glob.glob(O + '/p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv3x3_bf16s_kernel' in r['Kernel_Name']: pmc[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for f in glob.glob(O + '/p/**/*kernel_trace.csv', recursive=True):
    dur += [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'conv3x3_bf16s_kernel' in r['Kernel_Name']]
m = {k: sum(v) / len(v) for k, v in pmc.items()}
print(sys.argv[2], 'duration us %.1f' % (sum(dur) / len(dur) / 1e3), 'GRBM/8 %.0f' % (m['GRBM_GUI_ACTIVE'] / 8), 'wave life %.0f' % (m['SQ_WAVE_CYCLES'] * 4 / m['SQ_WAVES']), 'SQ_BUSY raw %.0f' % m['SQ_BUSY_CYCLES'], 'waves', m['SQ_WAVES'])
PY
done
