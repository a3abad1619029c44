"""The data-parallel training step on a GPU with more than one rank.

The reference trains in one process (``train_movielens.py:200-207``); here the links of a step are sharded over
ranks.  Two ranks (gloo: a one-GPU box cannot run RCCL with two ranks on one device) each run the static plan on
their rank-strided half of every 2B-link batch -- sampling, forward, backward replayed from a captured HIP graph
(and launch by launch), the flat-bucket all-reduce, the scaled update -- and a single process takes the same 2B
links as one batch.  Checked: shards disjoint and complete, a link's walk identical in both runs, both ranks'
weights bit-identical after every step, and equal to the single-process weights up to fp32 summation order.

The ranks are fresh child processes (tests/dist_gpu_worker.py); this process only reads what they wrote.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = os.path.join(ROOT, 'tests', 'dist_gpu_worker.py')
STEPS, B = 3, 64


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(world, out_dir, extra=(), timeout=150, backend='gloo'):
    """Starts ``world`` ranks and waits for all of them.  A rank that ends badly ends the others at once (they
    would sit in a collective until its timeout), and the whole launch has a deadline: a stall fails this test
    with what the ranks printed instead of holding up the suite."""
    import tempfile
    import time
    port = _free_port()
    procs, logs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        log = tempfile.TemporaryFile(mode='w+')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, WORKER, '--out', str(out_dir), '--steps', str(STEPS), '--batch', str(B),
                                       '--backend', backend] + list(extra), env=env, cwd=ROOT,
                                      stdout=log, stderr=subprocess.STDOUT, text=True))

    def output(i):
        logs[i].seek(0)
        return logs[i].read()[-3000:]

    deadline = time.monotonic() + timeout
    while any(p.poll() is None for p in procs):
        bad = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.monotonic() > deadline:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            for q in procs:
                q.wait()
            what = "rank %d of %d failed" % (bad[0], world) if bad else "no end after %d s" % timeout
            raise AssertionError(what + ":\n" + "\n----\n".join(output(i) for i in range(world)))
        time.sleep(0.05)
    for rank, p in enumerate(procs):
        assert p.returncode == 0, "rank %d of %d failed:\n%s" % (rank, world, output(rank))
    return [output(i) for i in range(world)]


@pytest.fixture(scope='module')
def runs(cuda, tmp_path_factory):
    out = tmp_path_factory.mktemp('ddp')
    logs = _launch(2, out, extra=['--time'])
    _launch(1, out)
    return out, logs


def _load(out, world, rank, mode, overlap):
    return torch.load(os.path.join(out, 'w%d_r%d_%s_%s.pt' % (world, rank, mode, overlap)), weights_only=True)


@pytest.mark.parametrize('mode', ['device', 'graph'])
def test_two_ranks_equal_one_process_on_the_union_batch(runs, mode):
    out, _ = runs
    r0, r1 = (_load(out, 2, r, mode, 'none') for r in (0, 1))
    one = _load(out, 1, 0, mode, 'none')
    assert r0['batch'] == r1['batch'] == B and one['batch'] == 2 * B
    # rank 0's initial weights everywhere, and they are the single process's
    assert torch.equal(r0['initial'], r1['initial']) and torch.equal(r0['initial'], one['initial'])
    n0 = 6
    firm = None
    for k in range(STEPS):
        s0, s1, s = r0['steps'][k], r1['steps'][k], one['steps'][k]
        # shards: disjoint, complete, rank-strided in the epoch order
        p0, p1, p = s0['positions'], s1['positions'], s['positions']
        assert not set(p0.tolist()) & set(p1.tolist())
        assert torch.equal(p[0::2], p0) and torch.equal(p[1::2], p1)
        # a link's walks do not depend on the number of ranks: hop-1 ids and the neighbor means of every
        # layer-0 head (sums over the hop-1 / hop-2 rows in walk order) are the same bits
        for rank, sr in ((0, s0), (1, s1)):
            for t in ('user', 'item'):
                for name, width in (('ids', 1), ('means', 64)):
                    got = sr[name][t].reshape(B * (1 + n0), -1)
                    want = s[name][t].reshape(2 * B * (1 + n0), -1)
                    assert got.shape[1] == want.shape[1] == width
                    assert torch.equal(got[:B], want[:2 * B][rank::2])                         # the roots
                    got_h = got[B:].reshape(B, n0, width)
                    want_h = want[2 * B:].reshape(2 * B, n0, width)[rank::2]
                    assert torch.equal(got_h, want_h)                                          # their hop-1 heads
        # both ranks hold the same bits after every step
        assert torch.equal(s0['params'], s1['params'])
        assert torch.equal(s0['grads'], s1['grads'])
        # mean of the two half-batch gradients = the full-batch gradient (MSE is a mean over links)
        scale = float(s['grads'].abs().max())
        tol = 2e-5 if k == 0 else 2e-4                     # later steps start from weights that differ in the last bits
        assert float((s0['grads'] - s['grads']).abs().max()) <= tol * scale
        # predictions of a link: same weights up to rounding, same walks
        assert float((s0['pred'] - s['pred'][0::2]).abs().max()) <= (1e-6 if k == 0 else 1e-4)
        assert float((s1['pred'] - s['pred'][1::2]).abs().max()) <= (1e-6 if k == 0 else 1e-4)
        if k == 0:
            assert torch.equal(s0['pred'], s['pred'][0::2])            # identical weights: identical bits
        assert abs(0.5 * (s0['loss'] + s1['loss']) - s['loss']) <= 1e-5 * max(1.0, s['loss'])
        # weights: Adam's first steps move every weight by about the rate whatever the gradient's size, so a
        # gradient that is rounding noise may flip; compare where the gradient is not noise
        firm_k = s['grads'].abs() > 1e-3 * scale
        firm = firm_k if firm is None else firm & firm_k
        assert float(firm.float().mean()) > 0.3
        diff = (s0['params'] - s['params']).abs()[firm]
        assert float(diff.max()) <= 2e-5 * float(s['params'].abs().max()) * (1 + 4 * k)
    assert not torch.equal(one['final'], one['initial'])


def test_delayed_all_reduce_keeps_ranks_identical_and_equals_one_process(runs):
    """overlap='delayed': step k's gradients are reduced beside step k + 1 and applied after it.  Ranks stay
    bit-identical; the run equals a single process applying its own gradients one step late; and the first two
    steps read the initial weights (nothing has been applied before the second backward)."""
    out, _ = runs
    r0, r1 = (_load(out, 2, r, 'graph', 'delayed') for r in (0, 1))
    one = _load(out, 1, 0, 'graph', 'delayed')
    now = _load(out, 2, 0, 'graph', 'none')
    for k in range(STEPS):
        assert torch.equal(r0['steps'][k]['params'], r1['steps'][k]['params'])
    assert torch.equal(r0['final'], r1['final'])
    assert torch.equal(r0['steps'][0]['params'], r0['initial'])                    # step 0's update is still travelling
    assert not torch.equal(r0['steps'][1]['params'], r0['initial'])                # applied after step 1's backward
    # step 0 and step 1 both ran on the initial weights: step 0 predicts as the undelayed run does
    assert torch.equal(r0['steps'][0]['pred'], now['steps'][0]['pred'])
    # the weights after step 1 are the undelayed weights after step 0 (same gradients, same first update)
    assert torch.equal(r0['steps'][1]['params'], now['steps'][0]['params'])
    # (a weight whose gradient is rounding noise may move the other way under Adam: all but a few agree closely)
    far = (r0['final'] - one['final']).abs() > 1e-4 * float(one['final'].abs().max())
    assert float(far.float().mean()) < 0.05
    assert not torch.equal(r0['final'], r0['steps'][STEPS - 1]['params'])          # flush() applied the last update


def test_measured_cost_of_collective_and_update(runs):
    out, logs = runs
    with open(os.path.join(out, 'w2_timing.json')) as fh:
        t = json.load(fh)
    print("two gloo ranks on one GPU, B = %(batch)d per rank: replay %(replay_us).0f us, all-reduce of %(bucket_bytes)d B "
          "%(all_reduce_us).0f us, update %(update_us).0f us; whole step %(step_us).0f us" % t)
    assert t['world'] == 2 and t['backend'] == 'gloo' and t['replay_us'] > 0 and t['all_reduce_us'] > 0
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'ddp_gloo_timing.json'), 'w') as fh:
        json.dump(t, fh)


def test_bench_with_two_ranks_reports_two_ranks(cuda):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['BENCH_WATCHDOG_S'] = '140'
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '5',
                          '--warmup', '3', '--train-steps', '5'], env=env, capture_output=True, text=True, timeout=160,
                         cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = lines[0]
    assert r['ranks_seen'] == 2 and r['n_gpus'] == 2 and r['backend'] == 'gloo' and r['scaling'] == 'weak'
    assert r['value'] > 0 and abs(r['value'] - 2 * 4096 * 5 / (r['ms_per_step'] * 5e-3)) < 1e-6 * r['value']
    assert r['train_step']['value'] > 0


def test_collective_captured_in_the_graph_with_rccl_at_world_one(cuda, tmp_path):
    """overlap='capture' puts the all-reduce and the update inside the replayed graph.  RCCL refuses two ranks on
    one device, so this rehearses the capture with a process group of ONE rank (the collective is still issued
    and captured): the run must equal the same steps with the collective outside the graph."""
    _launch(1, tmp_path, backend='nccl')
    cap = _load(tmp_path, 1, 0, 'graph', 'capture')
    _launch(1, tmp_path)
    ref = _load(tmp_path, 1, 0, 'graph', 'none')
    for k in range(STEPS):
        assert torch.equal(cap['steps'][k]['params'], ref['steps'][k]['params'])
        assert cap['steps'][k]['loss'] == ref['steps'][k]['loss']
