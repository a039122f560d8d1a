#!/bin/bash
# HBM-side traffic (L2 misses: FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950 x2 FETCH correction) of EVERY
# kernel of the SST training step, collected over a short bench run without side legs -> <out>/step_traffic.txt (+ .json):
# per kernel name: launches, bytes read / written per launch.  The whole-stack table of DESIGN.md comes from here.
# Usage (GPU box): bash tools/collect_step_traffic.sh gpurun_out/<tag>/step_traffic [extra bench.py flags]
set -e
OUT=${1:-gpurun_out/step_traffic}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
FLAGS="--steps 4 --warmup 3 --no-cpu-baseline --no-forward-only-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure --no-config-as-is-leg --no-bf16-leg --no-workload-legs --no-voxelize-roofline $@"
rm -rf /tmp/st_f /tmp/st_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/st_f -o f -- python "$R/bench.py" $FLAGS > /tmp/st_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/st_w -o w -- python "$R/bench.py" $FLAGS > /tmp/st_w.log 2>&1
python - "$R/$OUT" <<'PY'
import csv, json, re, sys
from collections import defaultdict
out = sys.argv[1]
def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc
f = per_kernel('/tmp/st_f/f_counter_collection.csv', 'FETCH_SIZE')
w = per_kernel('/tmp/st_w/w_counter_collection.csv', 'WRITE_SIZE')
rows = []
for name in sorted(set(f) | set(w)):
    fv, wv = f.get(name, []), w.get(name, [])
    n = max(len(fv), len(wv))
    rd = 2 * 1024 * sum(fv) / max(len(fv), 1)
    wr = 1024 * sum(wv) / max(len(wv), 1)
    short = name.replace('void ', '').replace('(anonymous namespace)::', '')
    short = re.sub(r'\(.*', '', short)
    rows.append({'kernel': short[:80], 'launches': n, 'read_MB_per_launch': round(rd / 1e6, 2), 'write_MB_per_launch': round(wr / 1e6, 2),
                 'total_MB_all_launches': round((rd + wr) * n / 1e6, 1)})
rows.sort(key=lambda r: -r['total_MB_all_launches'])
json.dump({'correction': 'FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (KB)', 'kernels': rows},
          open(out + '/step_traffic.json', 'w'), indent=1)
with open(out + '/step_traffic.txt', 'w') as fh:
    fh.write('%-82s %8s %14s %14s %14s\n' % ('kernel', 'launches', 'read MB/launch', 'write MB/launch', 'total MB'))
    for r in rows[:60]:
        fh.write('%-82s %8d %14.2f %14.2f %14.1f\n' % (r['kernel'], r['launches'], r['read_MB_per_launch'], r['write_MB_per_launch'],
                                                      r['total_MB_all_launches']))
total = sum(r['total_MB_all_launches'] for r in rows)
nbwd = sum(r['launches'] for r in rows if r['kernel'].startswith('sra_bwd'))
with open(out + '/step_traffic.txt', 'a') as fh:
    fh.write('TOTAL %.1f MB over all launches; %d attention-backward launches = %.1f training steps of 12 encoder layers -> %.2f GB per step\n'
             % (total, nbwd, nbwd / 12.0, total / max(nbwd / 12.0, 1e-9) / 1e3))
print(open(out + '/step_traffic.txt').read()[:6500])
PY
