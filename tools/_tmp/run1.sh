python -m pytest tests/test_gpu_bf16.py -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/microbench.py dense_bf16 > /tmp/mb.log 2>&1
grep "wgrad\|library form" /tmp/mb.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/pm/m_kernel_stats.csv')))
for r in rows:
    if 'tall_linear' in r['Name'] or 'wgrad' in r['Name']:
        print(r['Name'][28:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
