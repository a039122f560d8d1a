#!/bin/bash
# same-box A/B of an environment switch: bash tools/_ab_env.sh VAR "0 1" [extra bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VAR=$1; VALS=$2; shift 2
SIDE="--no-cpu-baseline --no-forward-only-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure --no-config-as-is-leg --no-bf16-own-process --no-bf16-leg --no-workload-legs --no-voxelize-roofline"
for i in 1 2 3; do
  for v in $VALS; do
    env $VAR=$v python bench.py $SIDE "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$VAR=$v',d['value'],d['ms_per_step'],d['step_ms']['median'])"
  done
done
