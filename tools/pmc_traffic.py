"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/<tag>_pmc_{fetch,write}_size_summary.csv and profiles/<tag>_traffic.json (read by bench.py as roofline.traffic; tag = r2, or
argv[3]).
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>
Units: rocprofv3 reports both counters in KB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size
(MI355X_MICROARCH.md, HBM section) and is doubled; WRITE_SIZE is exact."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV = ('igemm_', 'wino_kernel', 'wino2_kernel', 'wino_filter_kernel', 'wino2_filter_kernel', 'slab_reduce_kernel', 'splitk_reduce_kernel', 'weight_transpose_kernel')   # everything a conv C call launches


def load(path, name):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == name:
            tot[r['Kernel_Name']] += float(r['Counter_Value'])
            n[r['Kernel_Name']] += 1
    return tot, n


def main(fetch_csv, write_csv, tag='r2'):
    out = {}
    for name, path, fn in (('FETCH_SIZE', fetch_csv, tag + '_pmc_fetch_size_summary.csv'), ('WRITE_SIZE', write_csv, tag + '_pmc_write_size_summary.csv')):
        tot, n = load(path, name)
        with open(os.path.join(ROOT, 'profiles', fn), 'w') as f:
            f.write('Kernel_Name,Launches,%s_total_KB,%s_KB_per_launch\n' % (name, name))
            for k in sorted(tot, key=lambda k: -tot[k]):
                f.write('"%s",%d,%.1f,%.1f\n' % (k, n[k], tot[k], tot[k] / n[k]))
        out[name] = (tot, n)
    (F, nF), (W, nW) = out['FETCH_SIZE'], out['WRITE_SIZE']
    is_conv = lambda k: any(t in k for t in CONV)
    main_launch = lambda k: 'igemm_' in k or 'wino_kernel' in k or 'wino2_kernel' in k   # one GEMM kernel per conv C call; reduces / transposes / filter transforms ride along
    launches = sum(nF[k] for k in F if main_launch(k))
    fetch = sum(F[k] for k in F if is_conv(k)) * 1024.0
    write = sum(W[k] for k in W if is_conv(k)) * 1024.0
    js = {
        'kernel': 'igemm_taps_kernel / wino_kernel / wino2_kernel / igemm_wgrad_kernel (+ their slab reduces, filter transforms and the dgrad weight transposes): every conv / linear launch of bench.py',
        'command': 'rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none',
        'bs': 64, 'sz': 224, 'gpus': 1,
        'fetch_size_bytes_per_launch_raw': fetch / launches,
        'fetch_correction': 'x2 (gfx950 FETCH_SIZE reports half of wide coalesced reads; MI355X_MICROARCH.md HBM)',
        'write_size_bytes_per_launch': write / launches,
        'traffic_bytes_per_launch': (2 * fetch + write) / launches,
        'note': 'memory-side (L2 miss) traffic incl. Infinity-Cache hits; per-launch average over %d conv / linear GEMM launches' % launches,
    }
    json.dump(js, open(os.path.join(ROOT, 'profiles', tag + '_traffic.json'), 'w'), indent=1)
    print(json.dumps(js, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:4])
