"""Fit the schedule-model constants of csrc/wino2s.hip (w2s_cost_model, through the host-only entry nnl_debug_w2s_model) to forced-schedule
sweeps (tools/wino2s_plan_sweep.py logs).   python tools/wino2s_fit.py LOG [LOG ...]   (runs on the CPU)"""
import ctypes as C, itertools, json, math, sys
sys.path.insert(0, '.')
from neuralnetworklibrary_amd._lib import lib


def cdiv(a, b): return -(-a // b)


def load(files):
    pts = []
    for f in files:
        for line in open(f):
            if not line.startswith('{'):
                continue
            r = json.loads(line)
            N, Cc, K, H = r['N'], r['C'], r['K'], r['H']
            M4 = N * ((H + 1) // 2) ** 2; gn = cdiv(K, 64); T = cdiv(M4, 64) * gn; I = 4 * (Cc // 8)
            for k, v in r.items():
                if ':' not in k:
                    continue
                P, rest = k.split(':'); ks, S = map(int, rest.split('x'))
                if (ks > 1 and I // ks < 8) or (S > 1 and I // S < 4):
                    continue                                    # (the planner refuses these: the launch ran another plan)
                pts.append((r['shape'], N, T, gn, M4, K, I, int(P), ks, S, v))
    return pts


def model(p, prm):
    arr = (C.c_double * 6)(*prm)
    return lib.nnl_debug_w2s_model(p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], arr)


def main():
    pts = load(sys.argv[1:])
    mins = {}
    for p in pts:
        mins[(p[0], p[1])] = min(mins.get((p[0], p[1]), 1e9), p[-1])

    def err(prm, verbose=False, cut=1.5):
        e = n = 0
        for p in pts:
            m = model(p, prm)
            if m < 0 or p[-1] > cut * mins[(p[0], p[1])]:
                continue
            e += math.log(m / p[-1]) ** 2; n += 1
            if verbose:
                print(p[0], p[1], 'P', p[7], 'ks', p[8], 'S', p[9], 'meas', p[-1], 'model', round(m, 1))
        return math.sqrt(e / max(n, 1)), n
    best = None
    for prm in itertools.product((0.7, 0.8, 0.9), (1.2, 1.3, 1.4), (2, 5, 8), (0, 3, 6), (6, 10), (2e6, 4e6, 8e6)):
        e, n = err(prm)
        if best is None or e < best[0]:
            best = (e, prm, n); print(best, flush=True)
    if '-v' in sys.argv:
        err(best[1], verbose=True)
    by = {}
    for p in pts:
        by.setdefault((p[0], p[1]), []).append(p)
    for k, ps in by.items():
        ms = [(model(p, best[1]), p) for p in ps]
        pick = min(m for m in ms if m[0] > 0)[1]
        print(k, 'model picks', pick[7:10], 'measured', pick[-1], 'best measured', mins[k])


if __name__ == '__main__':
    main()
