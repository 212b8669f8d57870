"""issue-slot model of the K loop of conv_wino43b_kernel<0> (first wavefront role) from /tmp/k0.s (scratch/w4b_loopstats.sh writes it; EXTRA=-DKPX_W4B_STAMP
   for the stamped build, whose s_memtime instructions delimit the phases): every instruction costs 4 cycles of the wavefront's issue (an MFMA 8, an LDS-DMA
   piece 60), an MFMA cannot start before the previous one's 32 cycles are over.  Waits (LDS / filter-fragment latency, barrier) are NOT modelled."""
import re, sys, collections
lines = open('/tmp/k0.s').read().split('\n')
hdr = [i for i, l in enumerate(lines) if 'Inner Loop Header' in l]
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
    seg = [l.strip() for l in lines[h:end + 1] if l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.')]
    if sum('v_mfma' in l for l in seg) < 100:
        continue
    t = 0.0          # wavefront issue clock
    pipe = 0.0       # matrix pipe free at
    gaps = []
    n = 0
    marks = []
    kinds = collections.Counter()
    for l in seg:
        op = l.split()[0]
        if op.startswith('v_mfma'):
            gaps.append(n); n = 0
            t = max(t, pipe)
            pipe = t + 32
            t += 8
        elif op == 's_memtime':
            marks.append((t, dict(kinds))); kinds = collections.Counter()
        else:
            n += 1
            t += 60 if (op.startswith('buffer_load') and ' lds' in l) else 4 + (int(l.split()[1]) if op == 's_nop' else 0)
            kinds['pk' if op.startswith('v_pk') else 'acc' if op.startswith('v_accvgpr') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else
                  'vmem' if op.startswith('buffer') or op.startswith('global') else 'wait' if op == 's_waitcnt' else 'salu'] += 1
    print(lab, 'instructions', len(seg), 'model cycles per K step %.0f' % max(t, pipe), '(108 MFMAs alone: 3456)')
    c = collections.Counter(gaps[1:])
    print(' gap sizes (instructions between consecutive MFMAs): ', sorted(c.items()))
    if marks:
        prev = 0.0
        for m, k in marks:
            print('   phase ending at stamp: %6.0f cycles   %s' % (m - prev, k)); prev = m
        print('   tail: %6.0f cycles   %s' % (max(t, pipe) - prev, dict(kinds)))
    break
