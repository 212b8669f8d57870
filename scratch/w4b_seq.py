"""instruction-class sequence of the first K loop of conv_wino43b_kernel<0> in /tmp/k0.s (written by scratch/w4b_loopstats.sh):
   M mfma, v VALU, p packed-fp32 VALU, a accvgpr move, r ds_read, W ds_write, B buffer_load, D LDS-DMA, w s_waitcnt, n s_nop, | barrier, s other scalar"""
import re, sys
lines = open('/tmp/k0.s').read().split('\n')
hdr = [i for i, l in enumerate(lines) if 'Inner Loop Header' in l]
want = int(sys.argv[1]) if len(sys.argv) > 1 else 0
done = 0
for h in hdr:
    lab = lines[h].split(':')[0]
    tag = 'Header=' + lab.lstrip('.L')      # the loop = the header block + the following blocks marked "in Loop: Header=<it>"
    end = None
    for i in range(h + 1, len(lines)):
        if re.match(r'^\.LBB\d+_\d+:', lines[i]) and tag not in lines[i]:
            end = i - 1
            break
    if end is None:
        continue
    seg = lines[h:end + 1]
    if sum('v_mfma' in l for l in seg) < 100:
        continue
    if done != want:
        done += 1
        continue
    out = ''
    for l in seg:
        t = l.strip()
        if not l.startswith('\t') or t.startswith(';') or t.startswith('.'):
            continue
        op = t.split()[0]
        c = ('M' if op.startswith('v_mfma') else 'p' if op.startswith('v_pk_') else 'a' if op.startswith('v_accvgpr') else 'v' if op.startswith('v_') else
             'r' if op.startswith('ds_read') else 'W' if op.startswith('ds_write') else 'D' if op.startswith('buffer_load') and ' lds' in t else
             'B' if op.startswith('buffer_load') else 'w' if op.startswith('s_waitcnt') else 'n' if op.startswith('s_nop') else '|' if op.startswith('s_barrier') else
             's' if op.startswith('s_') else '?')
        out += c
    print(lab)
    for i in range(0, len(out), 160):
        print(out[i:i + 160])
    break
