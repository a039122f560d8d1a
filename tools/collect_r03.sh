#!/bin/bash
# Round-3 evidence set (GPU box): bench lines of the three workloads with their CPU legs, rocprofv3 kernel statistics of the
# same commands (without the CPU legs), per-layer sparse-convolution table, segmented-reduce and voxelize microbenchmarks.
# Usage: bash tools/collect_r03.sh <tag>   -> gpurun_out/<tag>/   (copy into profiles/r03/)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload fsd > $OUT/bench_workload_fsd.json 2> $OUT/bench_fsd.err
python bench.py --workload fsdv2 > $OUT/bench_workload_fsdv2.json 2> $OUT/bench_fsdv2.err
python bench.py --workload sst_bs2 --no-cpu-baseline > $OUT/bench_workload_sst_bs2.json 2> /dev/null
python bench.py --workload sst_bev > $OUT/bench_workload_sst_bev.json 2> /dev/null
python tools/conv_layers.py fsd > $OUT/conv_layers_fsd.txt 2>&1
python tools/conv_layers.py fsdv2 > $OUT/conv_layers_fsdv2.txt 2>&1
python tools/seg_reduce_only.py 30 > $OUT/seg_reduce_microbench.txt 2>&1
python tools/microbench.py voxelize > $OUT/voxelize_microbench.txt 2>&1
python tools/dense_x3_bench.py > $OUT/dense_f32_vs_f32x3_microbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$name -o p -- "$@" > $OUT/${name}_under_rocprof.log 2>&1
  cp /tmp/pp_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  gzip -c /tmp/pp_$name/p_kernel_trace.csv > $OUT/${name}_kernel_trace.csv.gz
}
prof sst python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-forward-only-leg --no-bf16-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure
python $R/tools/gap_report.py /tmp/pp_sst/p_kernel_trace.csv 0.65 60 > $OUT/sst_steady_state_trace_report.txt
prof sst_f32x3 python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-forward-only-leg --no-bf16-leg --no-lidar-leg --no-traffic-remeasure
prof fsd python $R/bench.py --workload fsd --steps 5 --warmup 3 --no-cpu-baseline
prof fsdv2 python $R/bench.py --workload fsdv2 --steps 5 --warmup 3 --no-cpu-baseline
prof seg_reduce python $R/tools/seg_reduce_only.py 10
rm -f $OUT/sst_f32x3_kernel_trace.csv.gz $OUT/seg_reduce_kernel_trace.csv.gz
python $R/tools/family_cost.py $OUT/fsd_kernel_trace.csv.gz 9 0.0 45 > $OUT/fsd_family_cost.txt 2>&1
ls -la $OUT
