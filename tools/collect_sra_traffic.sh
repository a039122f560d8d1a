#!/bin/bash
# Collects HBM traffic of the SRA forward kernel on the bench workload with rocprofv3 PMC counters, in SEPARATE
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with --sys-trace etc.), and writes
# profiles/<round>/sra_fwd_traffic.json.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half of
# the bytes of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is taken as reported (KB).
# Usage (GPU box): bash tools/collect_sra_traffic.sh gpurun_out/traffic
set -e
OUT=${1:-gpurun_out/traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_f /tmp/tr_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tr_f -o f -- python "$R/tools/sra_only.py" 5 > /tmp/tr_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tr_w -o w -- python "$R/tools/sra_only.py" 5 > /tmp/tr_w.log 2>&1
cp /tmp/tr_f/f_counter_collection.csv "$R/$OUT/fetch_counter_collection.csv"
cp /tmp/tr_w/w_counter_collection.csv "$R/$OUT/write_counter_collection.csv"
python - "$R/$OUT" <<'PY'
import csv, json, re, sys
out = sys.argv[1]
def avg(path, counter, pat):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
            if r['Counter_Name'] == counter and re.search(pat, r['Kernel_Name'])]
    return sum(vals) / len(vals), len(vals)
res = {}
# the fp32 kernels exist in two variants since round 5: <.., false> standard attention, <.., true> scaled cosine attention
pats = {'sra_fwd_wave_k': r'sra_fwd_wave_k<\d+, false>', 'sra_bwd_fused_k': r'sra_bwd_fused_k<\d+, \d+, false>',
        'sra_fwd_wave_k_cosine': r'sra_fwd_wave_k<\d+, true>', 'sra_bwd_fused_k_cosine': r'sra_bwd_fused_k<\d+, \d+, true>',
        'sra_fwd_bf16_k': 'sra_fwd_bf16_k', 'sra_bwd_bf16_k': 'sra_bwd_bf16_k'}
for kern, pat in pats.items():
    try:
        f, nf = avg(out + '/fetch_counter_collection.csv', 'FETCH_SIZE', pat)
        w, nw = avg(out + '/write_counter_collection.csv', 'WRITE_SIZE', pat)
    except ZeroDivisionError:
        continue
    res[kern] = {'FETCH_SIZE_KB_raw': f, 'WRITE_SIZE_KB_raw': w, 'launches_averaged': nf,
                 'hbm_read_bytes': 2 * f * 1024, 'hbm_write_bytes': w * 1024,
                 'hbm_bytes_per_launch': 2 * f * 1024 + w * 1024,
                 'correction': 'FETCH_SIZE x2 (gfx950 wide-stream under-count, MI355X_MICROARCH.md HBM section)'}
res['workload'] = 'tools/sra_only.py: bench frame (116000 points -> 90107 tokens, 1521 windows), d=128, 8 heads; fp32 kernels and the bf16-storage kernels'
json.dump(res, open(out + '/sra_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
