#!/bin/bash
# Round-5 final evidence (GPU box), after the launch-count work on the layer executors: the driver's bench line with all its
# legs, the other SST workloads, rocprofv3 kernel statistics + steady-state reports of the same commands without the side legs.
# Usage: bash tools/collect_r05_final.sh <tag>   -> gpurun_out/<tag>/   (copy into profiles/r05/ as h_*)
TAG=${1:-r05/h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload sst_center > $OUT/bench_workload_sst_center.json 2> $OUT/bench_center.err
python bench.py --workload sst_bs2 --no-cpu-baseline > $OUT/bench_workload_sst_bs2.json 2> /dev/null
python bench.py --cloud lidar --no-bf16-leg --no-f32x3-leg --no-forward-only-leg --no-traffic-remeasure > $OUT/bench_cloud_lidar.json 2> /dev/null
python tools/step_segments.py 30 > $OUT/step_segments_uniform.json 2> /dev/null
python tools/step_segments.py 30 0 lidar > $OUT/step_segments_lidar.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$name -o p -- "$@" > $OUT/${name}_under_rocprof.log 2>&1
  cp /tmp/pp_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  python $R/tools/gap_report.py /tmp/pp_$name/p_kernel_trace.csv 0.65 70 > $OUT/${name}_steady_state_trace_report.txt 2>&1
}
SIDE="--no-cpu-baseline --no-forward-only-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure --no-config-as-is-leg --no-bf16-own-process"
prof sst python $R/bench.py --steps 16 --warmup 6 $SIDE --no-bf16-leg
prof sst_bf16 python $R/bench.py --precision bf16 --steps 16 --warmup 6 $SIDE
prof sst_lidar python $R/bench.py --cloud lidar --steps 16 --warmup 6 $SIDE --no-bf16-leg
ls -la $OUT | head -40
