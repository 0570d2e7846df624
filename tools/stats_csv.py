"""rocprofv3 --kernel-trace --stats results (sqlite .db or *_kernel_stats.csv) -> a profiles/*.csv summary:
Name,Calls,TotalDurationUs,AverageUs,Percentage.   Usage: python tools/stats_csv.py <results.db | kernel_stats.csv> <out.csv>"""
import csv
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = []
if src.endswith('.db'):
    c = sqlite3.connect(src)
    for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        rows.append((name, calls, total, avg, pct))
else:
    for r in csv.DictReader(open(src)):
        rows.append((r['Name'], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, r['Percentage']))
with open(dst, 'w') as f:
    f.write('Name,Calls,TotalDurationUs,AverageUs,Percentage\n')
    for name, calls, total, avg, pct in rows:
        f.write('"%s",%s,%.3f,%.3f,%s\n' % (name, calls, float(total), float(avg), pct))
print('wrote', dst, len(rows), 'kernels')
