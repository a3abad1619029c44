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


def test_headline_line_carries_measured_traffic(cuda):
    """The default workload (cfg 2): the roofline's traffic comes from the stored counter pass of the kernel
    source that is running (a stale pass would print null), fractions stay below one, the CPU baseline names
    its cores and the whole-step figure prices the bytes of the route taken."""
    env = dict(os.environ, BENCH_WATCHDOG_S='270')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '10', '--warmup', '5', '--cpu-batches', '3',
                          '--train-steps', '5'], capture_output=True, text=True, timeout=290, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r['config']['workload'].startswith('cfg2') and r['config']['batch_per_gpu'] == 4096 and r['dtype'] == 'f32'
    roof = r['roofline']
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0
    assert roof['traffic'] is not None, "profiles/traffic.json holds no counter pass of the current sage2.hip"
    assert 0.3 < roof['frac'] <= 1.0 and 0.3 < roof['cache_aware']['frac'] <= 1.0
    assert 0.5 * roof['bytes_per_launch'] < roof['traffic'] < 1.5 * roof['bytes_per_launch']
    assert roof['launches_timed'] >= 2 and roof['launch_us'] * 1e-3 <= r['ms_per_step'] * 1.05
    assert abs(roof['achieved'] - roof['bytes_per_launch'] / roof['launch_us'] * 1e-3) < 1e-6 * roof['achieved']
    assert r['cpu_baseline']['cores'] == 1 and r['cpu_baseline']['kind'] == 'port'
    assert r['whole_step_gbps'] <= roof['achieved'] * 1.001 and r['layered_accounting']['note']
    assert r['train_step']['value'] > 0 and r['forward_pipelined']['value'] > 0
