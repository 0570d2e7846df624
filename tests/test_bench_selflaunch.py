"""bench.py's self-launch: `python bench.py --gpus N` without a launcher starts N ranks itself, before the
parent touches the GPU, and rank 0 prints exactly ONE JSON line.  Rehearsed here on CPU with gloo (`--dry-run`: the same
launch / rendezvous / barrier-bracketed timing / max-over-ranks / JSON code, a torch-CPU toy model instead of the HIP path)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--dry-run', '--steps', '4',
                        '--warmup', '1', '--bs', '8'], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize('n', [1, 2])
def test_self_launch_dry_run(n):
    out = _run(n)
    assert out['dry_run'] is True and out['n_gpus'] == n and out['ranks_seen'] == n
    assert out['steps'] == 4 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['strong']['global_batch'] == 8 and out['strong']['per_gpu_batch'] == 8 // n
    assert out['value'] > 0 and out['ms_per_step'] > 0


def test_parent_does_not_touch_the_gpu_before_launching():
    """the self-launch branch runs before any torch.cuda call (a process that has initialised the GPU must not spawn/exec ranks
    that then fight over it; on the GPU pool an exec from such a process is fatal)"""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main = src[src.index('def main():'):]
    assert main.index('self_launch(args.gpus)') < main.index('worker(args)')
    launch = src[src.index('def self_launch('):src.index('# ====', src.index('def self_launch('))]
    assert 'torch.cuda' not in launch and 'os.exec' not in launch
