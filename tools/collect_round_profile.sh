#!/bin/bash
# Collects the per-round evidence on the GPU box: bench line, rocprofv3 kernel statistics of the same command,
# steady-state trace report (busy/idle + per-kernel ms/step), SRA HBM traffic (separate PMC passes).
# Usage: bash tools/collect_round_profile.sh <tag>      -> gpurun_out/<tag>/...
set -e
TAG=${1:-g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-forward-only-leg --no-bf16-leg --no-f32x3-leg --no-lidar-leg > $OUT/bench_under_rocprof.log 2>&1 || tail -5 $OUT/bench_under_rocprof.log
cp /tmp/pb/b_kernel_stats.csv $OUT/kernel_stats.csv
python $R/tools/gap_report.py /tmp/pb/b_kernel_trace.csv 0.65 60 > $OUT/steady_state_trace_report.txt
gzip -c /tmp/pb/b_kernel_trace.csv > $OUT/kernel_trace.csv.gz
head -12 $OUT/steady_state_trace_report.txt
# the reduced-precision mode: the same step with --precision bf16
rm -rf /tmp/pc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o c -- python $R/bench.py --precision bf16 --steps 16 --warmup 6 --no-cpu-baseline --no-forward-only-leg --no-f32x3-leg --no-lidar-leg > $OUT/bench_bf16_under_rocprof.log 2>&1 || tail -5 $OUT/bench_bf16_under_rocprof.log
cp /tmp/pc/c_kernel_stats.csv $OUT/bf16_kernel_stats.csv
python $R/tools/gap_report.py /tmp/pc/c_kernel_trace.csv 0.65 60 > $OUT/bf16_steady_state_trace_report.txt
head -6 $OUT/bf16_steady_state_trace_report.txt
bash $R/tools/collect_sra_traffic.sh gpurun_out/$TAG/traffic > $OUT/traffic.log 2>&1 || tail -5 $OUT/traffic.log
tail -30 $OUT/traffic.log | head -12
