cd /tmp && export TMPDIR=/tmp
for v in 4 8; do
rm -rf /tmp/pm2
SST_AMD_BF16_LINEAR_WAVES=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm2 -o m -- python $GRAFT_REPO_ROOT/tools/microbench.py dense_bf16 > /tmp/mb2.log 2>&1
python - $v <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/pm2/m_kernel_stats.csv')):
    if 'tall_linear' in r['Name']:
        print('waves', sys.argv[1], r['Name'][28:75], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
done
