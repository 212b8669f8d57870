#!/usr/bin/env python3
"""How the three streams of a train step overlap, from the rocprofv3 kernel trace of `bench.py --no-roofline --no-cpu-baseline` (eager launches):

    python profiles/overlap.py gpurun_out/r03/step/s_kernel_trace.csv <profiled steps> [<steps to skip>]

Splits the trace into steps at the generator's Adam launch (the last kernel of a step), and prints per step: wall time from the first kernel
start to the last kernel end, time with >= 1 / >= 2 / >= 3 kernels resident, and the summed kernel time per hardware queue."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows = [r for r in csv.DictReader(open(path)) if r['Kind'] == 'KERNEL_DISPATCH']
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # step boundary: every second adam_tf launch (D update, then G update) ends a step
    ends, seen = [], 0
    for r in rows:
        if 'adam_tf' in r['Kernel_Name']:
            seen += 1
            if seen % 2 == 0:
                ends.append(int(r['End_Timestamp']))
    steps, lo = [], 0
    for e in ends:
        cur = [r for r in rows if lo < int(r['End_Timestamp']) <= e]
        lo = e
        steps.append(cur)
    out = []
    for cur in steps[skip:]:
        ev = []
        per_queue = defaultdict(float)
        for r in cur:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            ev.append((s, 1)); ev.append((e, -1))
            per_queue[r['Queue_Id']] += (e - s) / 1e6
        ev.sort()
        t0, t1 = ev[0][0], ev[-1][0]
        depth, last, busy = 0, t0, defaultdict(float)
        for t, d in ev:
            if depth > 0:
                for k in range(1, min(depth, 3) + 1):
                    busy[k] += (t - last) / 1e6
            depth += d
            last = t
        out.append(((t1 - t0) / 1e6, busy[1], busy[2], busy[3], sum(per_queue.values()), dict(per_queue), len(cur)))
    n = len(out)
    avg = lambda i: sum(o[i] for o in out) / n
    print('steps analysed: %d (of %d in the trace)' % (n, len(steps)))
    print('per step: first kernel start -> last kernel end %.2f ms; >= 1 kernel resident %.2f ms, >= 2 %.2f ms, >= 3 %.2f ms' % (avg(0), avg(1), avg(2), avg(3)))
    print('summed kernel time %.2f ms in %.0f launches; per hardware queue: %s' % (
        avg(4), avg(6), ', '.join('%s: %.2f ms' % (q, sum(o[5].get(q, 0.0) for o in out) / n) for q in sorted({q for o in out for q in o[5]}))))


if __name__ == '__main__':
    main()
