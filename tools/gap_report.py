#!/usr/bin/env python3
"""Where the GPU waits inside a step: reads the `*_kernel_trace.csv` of a `rocprofv3 --kernel-trace` run, orders the dispatches
by start time and reports the idle gaps (start of a kernel minus the latest end before it), grouped by the kernel that follows
the gap (usage: tools/gap_report.py <dir> <steps> [min_gap_us=2]).  A gap of a few us is the dispatch latency between dependent
kernels; a gap of 20+ us is the host (a synchronising call, an allocation, Python between two launches)."""
import collections
import csv
import glob
import sys

d, steps = sys.argv[1], int(sys.argv[2])
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]


busy = sum(e - s for s, e, _ in rows)
gaps = collections.defaultdict(lambda: [0, 0.0])
end = rows[0][1]
prev = rows[0][2]
idle = 0.0
hist = collections.Counter()
for s, e, n in rows[1:]:
    g = (s - end) / 1e3
    if g > 0:
        idle += g
        hist[min(int(g) // 5 * 5, 100)] += 1
        if g >= min_gap:
            a = gaps[(short(prev), short(n))]
            a[0] += 1
            a[1] += g
    if e > end:
        end, prev = e, n
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f'# {f}')
print(f'# {len(rows)} dispatches, span {span / 1e3:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle {idle / 1e3:.2f} ms '
      f'(= {idle / steps:.1f} us/step over {steps} steps; the span includes the program outside the timed steps)')
print('# gaps by length (us): ' + ', '.join(f'{k}{"+" if k == 100 else ""}: {v}' for k, v in sorted(hist.items())))
print('%-60s %-60s %7s %9s %10s' % ('after', 'before', 'count', 'avg_us', 'us/step'))
for (p, n), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%-60s %-60s %7d %9.1f %10.1f' % (p, n, c, t / c, t / steps))
