python -m pytest tests/test_gpu_dense.py -q -x -k "lds_linear" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for v in 8 4; do
rm -rf /tmp/pm
SST_AMD_LDS_LINEAR_WAVES=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/lds_linear_only.py 20 > /tmp/mb.log 2>&1
python - $v <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/pm/m_kernel_stats.csv')):
    if 'tall_linear' in r['Name']:
        print('waves', sys.argv[1], r['Name'][28:72], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
done
cd $GRAFT_REPO_ROOT
SST_AMD_LDS_LINEAR=1 python bench.py --no-cpu-baseline --no-forward-only-leg --no-bf16-leg 2>/dev/null | cut -c1-250
