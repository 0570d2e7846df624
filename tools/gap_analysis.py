"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace run, attributed to the kernel that FOLLOWS the gap.
Usage: python tools/gap_analysis.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps, cnt, dur = defaultdict(float), defaultdict(int), defaultdict(float)
last_end = rows[0][0]
for s, e, n in rows:
    if s > last_end:
        gaps[n] += s - last_end
    cnt[n] += 1
    dur[n] += e - s
    last_end = max(last_end, e)
print('kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms' % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
print('%-62s %7s %10s %10s %9s' % ('kernel (idle time BEFORE it)', 'calls', 'busy ms', 'idle ms', 'idle/call us'))
for n in sorted(gaps, key=gaps.get, reverse=True)[:25]:
    print('%-62s %7d %10.3f %10.3f %9.2f' % (n, cnt[n], dur[n] / 1e6, gaps[n] / 1e6, gaps[n] / cnt[n] / 1e3))
