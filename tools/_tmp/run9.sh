cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3; do
rm -rf /tmp/pm
SST_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/lds_linear_only.py 10 > /tmp/mb.log 2>&1
python - $d <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/pm/m_kernel_stats.csv')):
    if 'tall_linear' in r['Name']:
        print('dbg', sys.argv[1], r['Name'][28:72], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
done
