python -m pytest tests/test_gpu_dense.py tests/test_gpu_bf16.py -q -x -k "lds_linear or tall_linear" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/lds_linear_only.py 20 > /tmp/mb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm2 -o m -- python $GRAFT_REPO_ROOT/tools/microbench.py dense_bf16 > /tmp/mb2.log 2>&1
python - <<'PY'
import csv
for f in ('/tmp/pm/m_kernel_stats.csv', '/tmp/pm2/m_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'tall_linear' in r['Name'] or 'wgrad_' in r['Name']:
            print(r['Name'][28:72], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
