#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` directory into a per-kernel table
(usage: tools/rocprof_summary.py <dir-with-*_kernel_stats.csv> [steps])."""
import csv
import glob
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# {f}')
print(f'# total GPU kernel time {tot / 1e6:.2f} ms' + (f' = {tot / 1e6 / steps:.2f} ms/step over {steps} steps' if steps else ''))
print('%-72s %7s %10s %10s %6s' % ('kernel', 'calls', 'avg_us', 'total_ms', '%'))
for r in rows[:30]:
    name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    name = name.split('(')[0][:72]
    print('%-72s %7s %10.1f %10.2f %6.1f' % (name, r['Calls'], float(r['AverageNs']) / 1e3,
                                             float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
