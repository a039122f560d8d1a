#!/usr/bin/env python
"""GPU idle analysis of a rocprofv3 --kernel-trace CSV: busy time vs wall span and the largest gaps
(with the kernels on either side).  Usage: gap_report.py <kernel_trace.csv> [skip_fraction]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:70]


def main(path, skip=0.5):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    rows = rows[int(len(rows) * skip):]  # steady state only
    # whole steps only: from the first voxelize launch of the window up to (not including) the last one
    vox = [i for i, r in enumerate(rows) if 'dynamic_voxelize_' in r[2]]
    if len(vox) >= 2:
        rows = rows[vox[0]:vox[-1]]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    cur_end = rows[0][0]
    gaps = []
    for i, (s, e, n) in enumerate(rows):
        if s > cur_end:
            gaps.append((s - cur_end, rows[i - 1][2], n))
            busy += e - s
        else:
            busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    print(f'kernels {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {100 * (1 - busy / span):.1f} %')
    per = {}
    for st, en, n in rows:
        m = re.match(r'(Cijk_\w+?_MT\d+x\d+x\d+)', n)
        k = ('hipBLASLt ' + m.group(1)) if m else short(n)
        t = per.setdefault(k, [0, 0])
        t[0] += en - st
        t[1] += 1
    nvox = sum(1 for r in rows if 'dynamic_voxelize_' in r[2]) or 1
    # training steps inside the window: attention-backward launches / encoder layers (a bench process also voxelizes for its
    # forward-only side measurements - counting voxelize launches under-states the per-step figures: round 5's "12 % of the step
    # is not kernel time" was that)
    layers = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    nbwd = sum(1 for r in rows if 'sra_bwd' in r[2])
    nsteps = nbwd / layers if nbwd >= layers else nvox
    print(f'steady-state window: {nvox} voxelize launches, {nbwd} attention-backward launches = {nsteps:.2f} training steps of '
          f'{layers} layers;  busy {busy / 1e6 / nsteps:.3f} ms/step, span {span / 1e6 / nsteps:.3f} ms/step')
    for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
        print(f'  {t / 1e6 / nsteps:8.3f} ms/step  {c / nsteps:6.1f} calls/step  avg {t / c / 1e3:7.1f} us  {k}')
    # front of the step: voxelize -> first SRA forward launch (index phase + VFE + first projections)
    starts = [i for i, r in enumerate(rows) if 'dynamic_voxelize_' in r[2]]
    fronts = []
    for a in starts:
        b = next((i for i in range(a, len(rows)) if 'sra_fwd' in rows[i][2]), None)
        if b is None:
            continue
        span_f = rows[b][0] - rows[a][0]
        busy_f = sum(e - st for st, e, _ in rows[a:b])
        fronts.append((span_f, busy_f, b - a))
    if fronts:
        n = len(fronts)
        print(f'front of the step (voxelize .. first SRA launch): span {sum(f[0] for f in fronts) / n / 1e3:.0f} us, '
              f'busy {sum(f[1] for f in fronts) / n / 1e3:.0f} us, {sum(f[2] for f in fronts) / n:.0f} launches')
    # every transition between consecutive launches (also the back-to-back ones): how long the device waits between the end of
    # one kernel and the start of the next, for the transitions that occur most often (the dependent-launch bubble of the stack)
    trans = {}
    for i in range(1, len(rows)):
        k = (short(rows[i - 1][2])[:44], short(rows[i][2])[:44])
        trans.setdefault(k, []).append(rows[i][0] - rows[i - 1][1])
    print('end -> start gap of the most frequent transitions (median / mean us, count):')
    tot_pos = 0
    for k, v in sorted(trans.items(), key=lambda kv: -len(kv[1]))[:14]:
        v2 = sorted(v)
        print(f'  {v2[len(v2) // 2] / 1e3:7.2f} / {sum(v2) / len(v2) / 1e3:7.2f} us  x{len(v2):<4d} {k[0]}  ->  {k[1]}')
    allgaps = [rows[i][0] - rows[i - 1][1] for i in range(1, len(rows))]
    pos = [g for g in allgaps if g > 0]
    print(f'all transitions: {len(allgaps)}, positive gaps {len(pos)}, sum {sum(pos) / 1e6:.3f} ms = {sum(pos) / 1e6 / nsteps:.3f} ms/step; '
          f'gaps below 20 us: {sum(g for g in pos if g < 20000) / 1e6 / nsteps:.3f} ms/step')
    hist = {}
    for g, a, b in gaps:
        k = (short(a), short(b))
        t = hist.setdefault(k, [0, 0])
        t[0] += g
        t[1] += 1
    print('largest idle contributors (total us, count, after -> before):')
    for k, (t, c) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:8]:
        print(f'  {t / 1e3:9.1f} us  x{c:<4d} {k[0]}  ->  {k[1]}')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
