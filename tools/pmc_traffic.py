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
CONV = ('igemm_', 'wino_kernel', 'wino2_kernel', 'wino2s_kernel', 'wino_filter_kernel', 'wino2_filter_kernel', 'wino2s_filter_kernel', 'wino_filter_multi_kernel', 'wino_wgrad_finish_kernel', 'wino2d_wgrad_finish_kernel', 'slab_reduce_kernel', 'splitk_reduce_kernel', 'weight_transpose_kernel')   # everything a conv C call launches


def load(path, name):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == name:
            tot[r['Kernel_Name']] += float(r['Counter_Value'])
            n[r['Kernel_Name']] += 1
    return tot, n


def main(fetch_csv, write_csv, tag='r2', mfma_csv=None, stamp=None, write_files=True):
    """write_files=False: only compute and return the summary dict (bench.py --counters)"""
    out = {}
    for name, path, fn in (('FETCH_SIZE', fetch_csv, tag + '_pmc_fetch_size_summary.csv'), ('WRITE_SIZE', write_csv, tag + '_pmc_write_size_summary.csv')):
        tot, n = load(path, name)
        if write_files:
            with open(os.path.join(ROOT, 'profiles', fn), 'w') as f:
                f.write('Kernel_Name,Launches,%s_total_KB,%s_KB_per_launch\n' % (name, name))
                for k in sorted(tot, key=lambda k: -tot[k]):
                    f.write('"%s",%d,%.1f,%.1f\n' % (k, n[k], tot[k], tot[k] / n[k]))
        out[name] = (tot, n)
    (F, nF), (W, nW) = out['FETCH_SIZE'], out['WRITE_SIZE']
    is_conv = lambda k: any(t in k for t in CONV)
    main_launch = lambda k: 'igemm_' in k or 'wino_kernel' in k or 'wino2_kernel' in k or 'wino2s_kernel' in k   # one GEMM kernel per conv C call; reduces / transposes / filter transforms ride along
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
    if stamp:
        js['source_stamp'] = stamp          # nnl_source_stamp() of the library the passes profiled: bench.py reports these figures only for that build
    if mfma_csv:
        # MFMA-pipe busy fraction of the same kernels: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), summed over all launches
        B, nB = load(mfma_csv, 'SQ_VALU_MFMA_BUSY_CYCLES')
        G, _ = load(mfma_csv, 'GRBM_GUI_ACTIVE')
        if write_files:
            with open(os.path.join(ROOT, 'profiles', tag + '_pmc_mfma_busy.csv'), 'w') as f:
                f.write('Kernel_Name,Launches,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_fraction\n')
                for k in sorted(B, key=lambda k: -B[k]):
                    if B[k] > 0:
                        f.write('"%s",%d,%.0f,%.0f,%.4f\n' % (k, nB[k], B[k], G[k], B[k] / (G[k] / 8 * 1024)))
        busy = sum(B[k] for k in B if main_launch(k))
        act = sum(G[k] for k in G if main_launch(k))
        js['mfma_busy'] = round(busy / (act / 8 * 1024), 4)
        js['mfma_busy_note'] = 'time-weighted over the GEMM kernels of the conv / linear launches (main_launch kernels)'
    if write_files:
        json.dump(js, open(os.path.join(ROOT, 'profiles', tag + '_traffic.json'), 'w'), indent=1)
        print(json.dumps(js, indent=1))
    return js


if __name__ == '__main__':
    main(*sys.argv[1:6])
