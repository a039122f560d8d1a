python -m pytest tests/test_gpu_bf16.py -q -x 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-forward-only-leg 2>/dev/null > /tmp/bj.json; python - <<'PY'
import json
j=json.loads(open('/tmp/bj.json').read().strip().splitlines()[-1])
print('fp32', j['value'], j['ms_per_step']); r=j['reduced_precision']; print('bf16', r['value'], r['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o c -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 12 --warmup 5 --no-cpu-baseline --no-forward-only-leg > /tmp/c.log 2>&1
python $GRAFT_REPO_ROOT/tools/gap_report.py /tmp/pc/c_kernel_trace.csv 0.65 22 | cut -c1-120
