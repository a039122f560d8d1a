#!/bin/bash
# quick A/B on the GPU box: a few tests, the headline and the LiDAR-like frame twice, kernel statistics of the headline
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-ab}
mkdir -p $OUT
cd $R
SIDE="--no-cpu-baseline --no-forward-only-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure --no-config-as-is-leg --no-bf16-own-process --no-bf16-leg"
python -m pytest ${TESTS:-tests/test_gpu_dense_f32x6.py tests/test_gpu_cosine.py} -q -x 2>&1 | tail -2
for i in 1 2; do
  python bench.py $SIDE 2>/dev/null > $OUT/sst_$i.json
  python -c "import json;d=json.load(open('$OUT/sst_$i.json'));print('sst',d['value'],d['ms_per_step'],d['step_ms']['median'])"
  python bench.py --cloud lidar $SIDE 2>/dev/null > $OUT/lidar_$i.json
  python -c "import json;d=json.load(open('$OUT/lidar_$i.json'));print('lidar',d['value'],d['ms_per_step'],d['step_ms']['median'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/bench.py --steps 10 --warmup 4 $SIDE > /tmp/prof.log 2>&1
cp /tmp/pp/p_kernel_stats.csv $OUT/sst_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/pp/p_kernel_stats.csv')))
for r in rows[:${TOP:-16}]:
    print("%5.2f%% %5d %9.1f us  %s"%(float(r["Percentage"]),int(r["Calls"]),float(r["AverageNs"])/1e3,r["Name"][:70]))
PY
