#!/bin/bash
# Round-4 evidence set (GPU box): bench lines of the workloads with their CPU legs, rocprofv3 kernel statistics + steady-state
# reports of the same commands (without the side legs), the SST step on the LiDAR-like frame, SRA traffic (PMC passes), the
# dense / sparse micro-benchmarks of the exact-split kernels.
# Usage: bash tools/collect_r04.sh <tag>   -> gpurun_out/<tag>/   (copy into profiles/r04/)
TAG=${1:-r04/f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload fsd > $OUT/bench_workload_fsd.json 2> $OUT/bench_fsd.err
python bench.py --workload fsdv2 > $OUT/bench_workload_fsdv2.json 2> $OUT/bench_fsdv2.err
python bench.py --workload sst_bs2 --no-cpu-baseline > $OUT/bench_workload_sst_bs2.json 2> /dev/null
python bench.py --workload sst_bev > $OUT/bench_workload_sst_bev.json 2> /dev/null
python bench.py --cloud lidar --no-bf16-leg --no-f32x3-leg --no-forward-only-leg --no-traffic-remeasure > $OUT/bench_cloud_lidar.json 2> /dev/null
python tools/dense_x6_bench.py > $OUT/dense_f32_vs_f32x6_microbench.txt 2>&1
python tools/conv_modes.py fsd > $OUT/conv_modes_fsd.txt 2>&1
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$name -o p -- "$@" > $OUT/${name}_under_rocprof.log 2>&1
  cp /tmp/pp_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  python $R/tools/gap_report.py /tmp/pp_$name/p_kernel_trace.csv 0.65 70 > $OUT/${name}_steady_state_trace_report.txt 2>&1
}
SIDE="--no-cpu-baseline --no-forward-only-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure"
prof sst python $R/bench.py --steps 16 --warmup 6 $SIDE --no-bf16-leg
python $R/tools/front_of_step.py /tmp/pp_sst/p_kernel_trace.csv > $OUT/sst_front_of_step.txt 2>&1
prof sst_bf16 python $R/bench.py --precision bf16 --steps 16 --warmup 6 $SIDE
prof sst_lidar python $R/bench.py --cloud lidar --steps 16 --warmup 6 $SIDE --no-bf16-leg
prof fsd python $R/bench.py --workload fsd --steps 5 --warmup 3 --no-cpu-baseline --no-f32x3-leg
gzip -c /tmp/pp_fsd/p_kernel_trace.csv > $OUT/fsd_kernel_trace.csv.gz
python $R/tools/family_cost.py $OUT/fsd_kernel_trace.csv.gz 9 0.0 45 > $OUT/fsd_family_cost.txt 2>&1
rm -f $OUT/fsd_kernel_trace.csv.gz
prof fsdv2 python $R/bench.py --workload fsdv2 --steps 5 --warmup 3 --no-cpu-baseline --no-f32x3-leg
cd $R
bash tools/collect_sra_traffic.sh gpurun_out/$TAG/traffic > $OUT/traffic.log 2>&1
ls -la $OUT | head -50
