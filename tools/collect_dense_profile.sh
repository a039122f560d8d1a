#!/bin/bash
# Evidence for the dense kernels of round 2 (csrc/dense_bf16.hip, csrc/dense_f32.hip): rocprofv3 kernel statistics of the
# micro-benchmarks and HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction as in collect_sra_traffic.sh).
# Usage (GPU box): bash tools/collect_dense_profile.sh gpurun_out/dense
set -e
OUT=${1:-gpurun_out/dense}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/d_s /tmp/d_f /tmp/d_w /tmp/d_l
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/d_s -o s -- python "$R/tools/microbench.py" dense_bf16 > "$R/$OUT/microbench_dense_bf16.log" 2>&1
cp /tmp/d_s/s_kernel_stats.csv "$R/$OUT/dense_bf16_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/d_l -o l -- python "$R/tools/lds_linear_only.py" 20 > /dev/null 2>&1
cp /tmp/d_l/l_kernel_stats.csv "$R/$OUT/dense_f32_kernel_stats.csv"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/d_f -o f -- python "$R/tools/microbench.py" dense_bf16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/d_w -o w -- python "$R/tools/microbench.py" dense_bf16 > /dev/null 2>&1
python - "$R/$OUT" <<'PY'
import csv, json, re, sys, collections
out = sys.argv[1]
def avg(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and ('tall_linear_bf16_k' in r['Kernel_Name'] or 'wgrad_' in r['Kernel_Name']):
            name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0]
            acc[name].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = avg('/tmp/d_f/f_counter_collection.csv', 'FETCH_SIZE'), avg('/tmp/d_w/w_counter_collection.csv', 'WRITE_SIZE')
res = {k: {'hbm_read_bytes': 2 * f[k] * 1024, 'hbm_write_bytes': w.get(k, 0) * 1024, 'hbm_bytes_per_launch': 2 * f[k] * 1024 + w.get(k, 0) * 1024}
       for k in sorted(f)}
res['note'] = 'M = 90107 rows; FETCH_SIZE x 2 (gfx950 wide-stream correction) + WRITE_SIZE, KB -> bytes; algorithmic bytes: (K + N) x 2 x M for a linear (+ N x 2 x M per extra [M, N] operand of its epilogue), 299.9 MB for the grouped weight gradient'
json.dump(res, open(out + '/dense_bf16_traffic.json', 'w'), indent=1)
for k, v in res.items():
    if isinstance(v, dict):
        print(k, round(v['hbm_bytes_per_launch'] / 1e6, 1), 'MB')
PY
grep -v amdgpu.ids "$R/$OUT/microbench_dense_bf16.log" | tail -18
