"""Instruction mix of the innermost MFMA loops of every kernel in gfx950 assembly files (hipcc -S --cuda-device-only ...): MFMA count against
VALU / register-move / s_nop counts per iteration.  On gfx950 VALU work is not hidden behind a dependent MFMA chain (tools/coissue_probe.hip),
so every VALU instruction of such a loop is paid in matrix time.  usage: python tools/loop_valu_survey.py file.s [file.s ...]"""
import collections
import re
import sys


def survey(path):
    txt = open(path).read().split('\n')
    kernels, name, rows = {}, None, []
    for i, l in enumerate(txt):
        m = re.match(r'^(_Z[\w]+):', l)
        if m:
            name = m.group(1); kernels[name] = [i, None]
        elif 's_endpgm' in l and name and kernels[name][1] is None:
            kernels[name][1] = i
    for k, (a, b) in kernels.items():
        if b is None:
            continue
        lines = txt[a:b]
        labels = [(i, l) for i, l in enumerate(lines) if l.startswith('.LBB')]
        loops = collections.defaultdict(list)                 # header name -> label line indices of its blocks
        for n, (i, l) in enumerate(labels):
            nxt = labels[n + 1][0] if n + 1 < len(labels) else len(lines)
            m = re.search(r'Header=(BB\d+_\d+)', l)
            if m:
                loops[m.group(1)].append((i, nxt))
            if 'Loop Header' in l:
                loops[l.split(':')[0].lstrip('.L')].append((i, nxt))
        for h, blocks in loops.items():
            lo, hi = min(x for x, _ in blocks), max(y for _, y in blocks)
            body = [l for l in lines[lo:hi] if l.strip() and not l.strip().startswith(';') and not l.startswith('.')]
            c = collections.Counter(l.split()[0] for l in body)
            n_m = sum(v for kk, v in c.items() if kk.startswith('v_mfma'))
            if n_m < 8:
                continue
            # skip outer loops: another loop of this kernel lies strictly inside with the same MFMA count
            valu = sum(v for kk, v in c.items() if kk.startswith('v_') and 'mfma' not in kk)
            mov = sum(v for kk, v in c.items() if kk.startswith('v_mov') or kk.startswith('v_accvgpr'))
            rows.append({'kernel': k, 'loop': h, 'mfma': n_m, 'valu': valu, 'mov': mov, 's_nop': c['s_nop'],
                         'lds': sum(v for kk, v in c.items() if kk.startswith('ds_')),
                         'vmem': sum(v for kk, v in c.items() if kk.startswith('buffer_') or kk.startswith('global_'))})
    return rows


if __name__ == '__main__':
    for f in sys.argv[1:]:
        print(f)
        for r in survey(f):
            print('  %-10s mfma %3d  VALU %4d (mov %3d)  s_nop %2d  lds %3d  vmem %3d  %s' % (
                r['loop'], r['mfma'], r['valu'], r['mov'], r['s_nop'], r['lds'], r['vmem'], r['kernel'][:110]))
