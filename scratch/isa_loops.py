"""instruction-class strings of every loop with MFMAs in a cross-compiled .s file:  python scratch/isa_loops.py /tmp/conv_gemm3.s [min_mfma] [kernel substring]
   M mfma, v VALU, p packed fp32, a accvgpr move, r ds_read, W ds_write, B vector load, S vector store, D LDS-DMA, w s_waitcnt, n s_nop, | barrier, s other scalar"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
minm = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sub = sys.argv[3] if len(sys.argv) > 3 else ''
fn = None
for i, l in enumerate(lines):
    m = re.match(r'^(_Z\S+):', l)
    if m:
        fn = m.group(1)
    if 'Inner Loop Header' in l and (sub in (fn or '')):
        lab = l.split(':')[0]
        tag = 'Header=' + lab.lstrip('.L')
        end = None
        for j in range(i + 1, len(lines)):
            if re.match(r'^\.LBB\d+_\d+:', lines[j]) and tag not in lines[j]:
                end = j - 1
                break
        if end is None:
            continue
        seg = lines[i:end + 1]
        nm = sum('v_mfma' in x for x in seg)
        if nm < minm:
            continue
        s = ''
        for x in seg:
            t = x.strip()
            if not x.startswith('\t') or t.startswith(';') or t.startswith('.'):
                continue
            op = t.split()[0]
            s += ('M' if op.startswith('v_mfma') else 'p' if re.match(r'v_pk_(add|mul|fma)_f32', op) else 'a' if op.startswith('v_accvgpr') else 'v' if op.startswith('v_') else
                  'r' if op.startswith('ds_read') else 'W' if op.startswith('ds_write') else 'D' if (op.startswith('buffer_load') or op.startswith('global_load')) and ' lds' in t else
                  'B' if op.startswith('buffer_load') or op.startswith('global_load') else 'S' if op.startswith('buffer_store') or op.startswith('global_store') else
                  'w' if op.startswith('s_waitcnt') else 'n' if op.startswith('s_nop') else '|' if op.startswith('s_barrier') else 's')
        print(fn[:100], lab, 'mfma', nm, 'instr', len(s), 'pk', s.count('p'), 'acc', s.count('a'))
        for k in range(0, len(s), 170):
            print('  ' + s[k:k + 170])
