python -m pytest tests/test_gpu_bf16.py -q -x 2>&1 | tail -2
mkdir -p gpurun_out/r2j
python bench.py --no-cpu-baseline --no-forward-only-leg > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err || tail -5 gpurun_out/r2j/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2j/bench.json').read().strip().splitlines()[-1])
print('fp32', j['value'], j['ms_per_step']); r=j['reduced_precision']; print('bf16', r['value'], r['ms_per_step'], r['vs_fp32_forward'], r['roofline']['sra_fwd']['avg_launch_ms'], r['roofline']['sra_bwd']['avg_launch_ms'])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 16 --warmup 6 --no-cpu-baseline --no-forward-only-leg > /tmp/b.log 2>&1 || tail -5 /tmp/b.log
python $GRAFT_REPO_ROOT/tools/gap_report.py /tmp/pb/b_kernel_trace.csv 0.65 40 > $GRAFT_REPO_ROOT/gpurun_out/r2j/bf16_steady_state_trace_report.txt
head -48 $GRAFT_REPO_ROOT/gpurun_out/r2j/bf16_steady_state_trace_report.txt | cut -c1-130
