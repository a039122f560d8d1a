python -m pytest tests/test_gpu_sra.py tests/test_gpu_end_to_end.py tests/test_gpu_dense.py tests/test_gpu_dist.py tests/test_gpu_frame_plan.py -q -x 2>&1 | tail -6
python bench.py --no-cpu-baseline --no-forward-only-leg 2>/dev/null > /tmp/bj.json; python - <<'PY'
import json
j=json.loads(open('/tmp/bj.json').read().strip().splitlines()[-1])
print('fp32', j['value'], j['ms_per_step']); r=j['reduced_precision']; print('bf16', r['value'], r['ms_per_step'], r['vs_fp32_forward'])
PY
