"""Kernel sequence of ONE step from a rocprofv3 --kernel-trace --output-format csv run: the last `n` dispatches in start order
with duration and the gap to the previous kernel's end.   Usage: python tools/kernel_order.py <kernel_trace.csv> <kernels per step>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2])
last = rows[-n:]
prev_end = None
tot = gap_tot = 0.0
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = e
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0.0)
    print('%7.2f us  gap %6.2f  %s' % ((e - s) / 1e3, gap, r['Kernel_Name'][:110]))
print('kernels %d  busy %.1f us  gaps %.1f us  span %.1f us' % (n, tot, gap_tot, (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3))
