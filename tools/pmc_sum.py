"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter): python tools/pmc_sum.py out.txt dir1 dir2 ... [--match substr]"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    args = sys.argv[1:]
    match = None
    if '--match' in args:
        i = args.index('--match'); match = args[i + 1]; del args[i:i + 2]
    out, dirs = args[0], args[1:]
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
                if match and match not in k:
                    continue
                acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
    with open(out, 'w') as fo:
        for k in sorted(acc):
            fo.write('== %s\n' % k[:140])
            wc = acc[k].get('SQ_WAVE_CYCLES', 0) / max(cnt[k].get('SQ_WAVE_CYCLES', 1), 1)
            for c in sorted(acc[k]):
                v = acc[k][c] / cnt[k][c]                      # per dispatch
                fo.write('  %-34s %.5g' % (c, v) + ('   /SQ_WAVE_CYCLES %.3f' % (v / wc) if wc and c.startswith('SQ_') else '') + '\n')
    print(open(out).read())

main()
