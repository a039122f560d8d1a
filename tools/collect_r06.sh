#!/bin/bash
# Round-6 evidence (GPU box): rocprofv3 kernel statistics + steady-state reports of the SST step (fp32 exact-split, bf16,
# LiDAR-like frame) without the side legs, and the per-kernel HBM traffic of a step.
# Usage: bash tools/collect_r06.sh <tag> [quick]   -> gpurun_out/<tag>/   (copy into profiles/r06/)
TAG=${1:-r06/a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$name -o p -- "$@" > $OUT/${name}_under_rocprof.log 2>&1
  cp /tmp/pp_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  python $R/tools/gap_report.py /tmp/pp_$name/p_kernel_trace.csv 0.65 70 > $OUT/${name}_steady_state_trace_report.txt 2>&1
}
SIDE="--no-cpu-baseline --no-forward-only-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure --no-config-as-is-leg --no-bf16-own-process --no-workload-legs --no-voxelize-roofline"
prof sst python $R/bench.py --steps 16 --warmup 6 $SIDE --no-bf16-leg
if [ "$2" != "quick" ]; then
  prof sst_bf16 python $R/bench.py --precision bf16 --steps 16 --warmup 6 $SIDE
  prof sst_lidar python $R/bench.py --cloud lidar --steps 16 --warmup 6 $SIDE --no-bf16-leg
  cd $R && bash tools/collect_step_traffic.sh gpurun_out/$TAG > $OUT/step_traffic.log 2>&1
fi
ls $OUT
