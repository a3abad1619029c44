"""bench.py prints one JSON line with the agreed fields -- needs a GPU (small
MovieLens-shaped workload, a few steps)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_has_the_agreed_fields(cuda):
    env = dict(os.environ, BENCH_WATCHDOG_S='240')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'ml1m', '--batch', '256',
                          '--steps', '5', '--warmup', '2', '--cpu-batches', '2', '--train-steps', '3'],
                         capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in r, key
    assert r['n_gpus'] == 1 and r['steps'] == 5 and r['warmup'] == 2 and r['higher_is_better'] is True
    assert r['scaling'] == 'weak' and r['vs_baseline'] is None and r['data'] == 'synthetic' and r['dtype'] == 'f32'
    assert r['value'] > 0 and abs(r['value'] - 256 * 5 / (r['ms_per_step'] * 5e-3)) < 1e-6 * r['value']
    assert 'workload' in r['config'] and 'model' not in r['config']
    roof = r['roofline']
    assert roof['bound'] in ('hbm', 'mfma') and roof['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9 and 'traffic' in roof
    cpu = r['cpu_baseline']
    assert cpu['kind'] in ('port', 'reference') and cpu['cores'] >= 1 and cpu['value'] > 0 and cpu['sample']
    # the secondary timing: the training step (24/19-wide features run on padded copies), or the reason it was skipped
    assert 'skipped' in r['train_step'] or r['train_step']['value'] > 0
