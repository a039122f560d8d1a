set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --precision bf16 --steps 16 --warmup 6 --no-cpu-baseline --no-forward-only-leg --no-gemm-tuning > $OUT/bench_bf16_under_rocprof.log 2>&1 || tail -5 $OUT/bench_bf16_under_rocprof.log
tail -1 $OUT/bench_bf16_under_rocprof.log | cut -c1-300
python $R/tools/gap_report.py /tmp/pb/b_kernel_trace.csv 0.65 70 > $OUT/bf16_steady_state_trace_report.txt
head -75 $OUT/bf16_steady_state_trace_report.txt
cd $R
python bench.py --no-cpu-baseline --no-forward-only-leg > $OUT/bench.json 2>$OUT/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2i/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], json.dumps(j['reduced_precision'])[:900])
PY
