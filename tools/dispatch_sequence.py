"""Per-dispatch sequence of one step of a `rocprofv3 --kernel-trace --output-format csv` run: start offset, duration and the gap to
the latest end before it, for the window between the last two dispatches of a marker kernel (default `logmel_kernel`: a Whisper
step; `ce_finalize_kernel` cuts a text workload from one loss to the next).  usage: tools/dispatch_sequence.py <trace dir> [marker]"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
# the last step: from the last logmel_kernel dispatch on
marker = sys.argv[2] if len(sys.argv) > 2 else 'logmel_kernel'
starts = [i for i, r in enumerate(rows) if marker in r[2]]
a = starts[-2] if len(starts) > 1 else 0
b = starts[-1] if len(starts) > 1 else len(rows)
t0 = rows[a][0]; end = rows[a][0]
tot_gap = 0
for s, e, n in rows[a:b]:
    gap = (s - end) / 1e3
    tot_gap += max(gap, 0)
    print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {short(n)}')
    end = max(end, e)
print('step span', (rows[b - 1][1] - t0) / 1e3, 'us; idle', tot_gap, 'us; dispatches', b - a)
