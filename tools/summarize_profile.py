#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag> directory (tools/profile_gpu.sh) into
profiles/<name>.md: per-kernel time from the kernel trace, and HBM traffic per
launch of our kernels from the FETCH_SIZE / WRITE_SIZE passes.

gfx950 counter corrections: FETCH_SIZE and WRITE_SIZE are reported in KiB.
FETCH_SIZE = read requests x 64 bytes; a wide coalesced read (16 bytes per lane)
travels as 128-byte requests, so for kernels whose reads are such row streams
(WIDE_READERS below) the read side is doubled.  Kernels that read scattered
4- or 8-byte words (the sampler, the adjacency build) are NOT doubled: their
figure is "requests x 64 bytes", the sectors they cost.  A kernel that mixes both
(the one-kernel forward: rows + its own sampling reads) is doubled as a whole, an
upper estimate.

    python tools/summarize_profile.py <gpurun_out/prof_dir> <name> [<workload> <batch> <fanouts like 25x10>]

With the last three arguments the dominant kernels' bytes per launch are also
entered in profiles/traffic.json, each with the git blob hash of the kernel's
source file, so that a reader of the entry can tell whether it still describes
the code."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ours = ('sage2_kernel', 'sage3_seeds_kernel', 'sage2_tail_bwd_kernel', 'sage2_bwd_all_kernel', 'sage3_bwd_seeds_kernel', 'sage3_bwd_subtrees_kernel', 'sage_bwd_reduce_kernel', 'stage_batch_kernel', 'sage_wide_kernel', 'l2norm_inplace_kernel', 'sage_fused_kernel', 'sample_kernel', 'gather_mean_kernel', 'dense_mean_kernel', 'gather_rows_kernel', 'rows_gemm_kernel',
        'rows_gemm_big_kernel', 'wgrad_big_kernel', 'wgrad_kernel', 'wgrad_direct_kernel', 'wgrad_reduce_kernel', 'wgrad_reduce4_kernel', 'l2norm', 'link_head', 'adam_kernel', 'csr_',
        'group_max_bwd_kernel', 'head_')


WIDE_READERS = ('sage2_kernel', 'sage3_seeds_kernel', 'sage_fused_kernel', 'sage_wide_kernel', 'gather_mean_kernel', 'dense_mean_kernel', 'gather_rows_kernel',
                'rows_gemm', 'wgrad', 'sage2_tail_bwd_kernel', 'sage2_bwd_all_kernel', 'sage3_bwd_seeds_kernel', 'sage3_bwd_subtrees_kernel', 'sage_bwd_reduce_kernel', 'link_head', 'adam_kernel', 'l2norm')
SOURCE_OF = {'sage2_kernel': ('sage2.hip', 'sage_tile.h'), 'sage3_seeds_kernel': ('sage3_seeds.hip', 'sage_tile.h'), 'sage_fused_kernel': 'fused.hip', 'sage_wide_kernel': 'wide.hip',
             'sample_kernel': 'sampler.hip', 'gather_mean_kernel': 'gather.hip', 'dense_mean_kernel': 'gather.hip'}


def fetch_factor(kernel):
    return 2.0 if any(w in kernel for w in WIDE_READERS) else 1.0


def blob_hash(path):
    """git's blob id of a file (sha1 over 'blob <size>\\0' + content), computed without git."""
    import hashlib
    data = open(path, 'rb').read()
    return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()


def short(n):
    for o in ours:
        if o in n:
            tail = n[n.index(o):]
            return tail.split('(')[0][:60]
    return n[:60]


lines = ['# rocprofv3 summary: %s' % name, '']
bl = os.path.join(src, 'bench_line.json')
if os.path.exists(bl):
    lines += ['bench line of the traced run:', '', '```', open(bl).read().strip(), '```', '']
stats = sorted(glob.glob(os.path.join(src, 'trace', '*', '*kernel_stats.csv')), key=os.path.getmtime, reverse=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    lines += ['## kernel time (rocprofv3 --kernel-trace --stats), kernels of libgsage.so', '',
              '| kernel | calls | total ms | avg us | min us | max us |', '|---|---|---|---|---|---|']
    for r in rows:
        if any(o in r['Name'] for o in ours):
            lines.append('| %s | %s | %.3f | %.2f | %.2f | %.2f |' % (
                short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
                float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
    lines.append('')

# a kernel launched on shapes of very different size: the average over its large launches (within a factor
# four of the longest), which is what bench.py times for the dominant kernel
traces = sorted(glob.glob(os.path.join(src, 'trace', '*', '*kernel_trace.csv')), key=os.path.getmtime, reverse=True)
if traces:
    dur = defaultdict(list)
    for r in csv.DictReader(open(traces[0])):
        if any(o in r['Kernel_Name'] for o in ours):
            dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    mixed = {k: v for k, v in dur.items() if max(v) > 3.0 * min(v) and max(v) > 100.0}
    if mixed:
        lines += ['kernels launched on shapes of very different size, large launches only (within a factor four of the longest):', '',
                  '| kernel | large launches | avg us | min us | max us |', '|---|---|---|---|---|']
        for k, v in sorted(mixed.items()):
            big = [x for x in v if x >= 0.25 * max(v)]
            lines.append('| %s | %d of %d | %.2f | %.2f | %.2f |' % (k, len(big), len(v), sum(big) / len(big), min(big), max(big)))
        lines.append('')

traffic = {}
for kind, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    files = sorted(glob.glob(os.path.join(src, kind, '*', '*counter_collection.csv')), key=os.path.getmtime, reverse=True)
    if not files:
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r.get('Counter_Name') == counter and any(o in r['Kernel_Name'] for o in ours):
            acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items():
        # a kernel launched on shapes of very different size (the wide kernel on layer 0 and on layer 1):
        # the figure is that of its large launches, those within a factor four of the largest
        big = [x for x in v if x >= 0.25 * max(v)]
        traffic.setdefault(k, {})[kind] = (sum(big) / len(big) * 1024.0, len(big))
if traffic:
    lines += ['## HBM-side traffic per launch (rocprofv3 --pmc, separate passes)', '',
              '| kernel | launches | FETCH_SIZE raw MB (requests x 64 B) | factor | read MB | WRITE_SIZE MB |',
              '|---|---|---|---|---|---|']
    for k, d in sorted(traffic.items()):
        f, nf = d.get('fetch', (0.0, 0))
        w, _ = d.get('write', (0.0, 0))
        lines.append('| %s | %d | %.2f | x%d | %.2f | %.2f |' % (k, nf, f / 1e6, fetch_factor(k), fetch_factor(k) * f / 1e6, w / 1e6))
    lines += ['', 'factor x2: 16-byte-per-lane row streams (128-byte requests tallied at 64 bytes); x1: scattered word reads, '
              'the figure is the sectors they cost.  Launches: those within a factor four of the kernel\'s largest.', '']
sq_files = sorted(glob.glob(os.path.join(src, 'sq', '*', '*counter_collection.csv')), key=os.path.getmtime, reverse=True)
if sq_files:
    # per kernel: the launch with the most wave cycles (the largest shape of that kernel)
    per = defaultdict(lambda: defaultdict(dict))
    for r in csv.DictReader(open(sq_files[0])):
        if any(o in r['Kernel_Name'] for o in ours):
            per[short(r['Kernel_Name'])][r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    lines += ['## wave states and matrix-pipe use (rocprofv3 --pmc, SQ + GRBM pass; largest launch of each kernel)', '',
              'SQ_WAVE_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts '
              'cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE / 8 is the launch length in shader cycles.', '',
              '| kernel | parked (s_waitcnt, barrier) | issue stall | issuing | MFMA pipe busy | LDS bank-conflict cycles / wave cycle |',
              '|---|---|---|---|---|---|']
    for k, d in sorted(per.items()):
        best = max(d.values(), key=lambda c: c.get('SQ_WAVE_CYCLES', 0.0))
        wc = best.get('SQ_WAVE_CYCLES', 0.0)
        if wc <= 0:
            continue
        cyc = best.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        mf = best.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        lines.append('| %s | %.0f %% | %.0f %% | %.0f %% | %s | %.3f |' % (
            k, 100 * best.get('SQ_WAIT_ANY', 0) / wc, 100 * best.get('SQ_WAIT_INST_ANY', 0) / wc,
            100 * best.get('SQ_ACTIVE_INST_ANY', 0) / wc,
            ('%.0f %% of %.0f k cycles' % (100 * mf / (1024 * cyc), cyc / 1e3)) if cyc > 0 and mf > 0 else '-',
            best.get('SQ_LDS_BANK_CONFLICT', 0) / (4 * wc)))
    lines.append('')
os.makedirs(os.path.join(root, 'profiles'), exist_ok=True)
with open(os.path.join(root, 'profiles', name + '.md'), 'w') as fh:
    fh.write('\n'.join(lines) + '\n')
if stats:
    import shutil
    shutil.copy(stats[0], os.path.join(root, 'profiles', name + '_kernel_stats.csv'))
print('\n'.join(lines))
json.dump({k: {kk: vv[0] for kk, vv in d.items()} for k, d in traffic.items()},
          open(os.path.join(root, 'profiles', name + '_traffic_raw.json'), 'w'), indent=1)

if len(sys.argv) >= 6:
    workload, batch, fan = sys.argv[3], int(sys.argv[4]), sys.argv[5]
    tj = os.path.join(root, 'profiles', 'traffic.json')
    rec = json.load(open(tj)) if os.path.exists(tj) else {}
    rec['_comment'] = ("HBM-side bytes per launch of a workload's dominant kernels: FETCH_SIZE x factor + WRITE_SIZE from separate "
                       "rocprofv3 --pmc passes (tools/summarize_profile.py); 'source' holds the git blob hash of the kernel's "
                       "file at the time of the measurement")
    for k, d in traffic.items():
        base = k.split('<')[0]
        if base not in SOURCE_OF or 'fetch' not in d:
            continue
        names = SOURCE_OF[base] if isinstance(SOURCE_OF[base], tuple) else (SOURCE_OF[base],)
        src_files = [os.path.join('graphsage_amd', 'csrc', n) for n in names]
        key = '%s/%s/B%d/%s' % (workload, k, batch, fan)        # k keeps the template arguments: variants stay apart
        rec[key] = {'bytes': fetch_factor(k) * d['fetch'][0] + d.get('write', (0.0, 0))[0],
                    'fetch_raw': d['fetch'][0], 'fetch_factor': fetch_factor(k), 'write': d.get('write', (0.0, 0))[0],
                    'launches': d['fetch'][1], 'source': {f: blob_hash(os.path.join(root, f)) for f in src_files},
                    'profile': 'profiles/%s.md' % name}
    json.dump(rec, open(tj, 'w'), indent=1, sort_keys=True)
