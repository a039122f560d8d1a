python -m pytest tests/test_gpu_bf16.py -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-forward-only-leg 2>/dev/null > /tmp/bj.json; python - <<'PY'
import json
j=json.loads(open('/tmp/bj.json').read().strip().splitlines()[-1])
print('fp32', j['value'], j['ms_per_step']); r=j['reduced_precision']; print('bf16', r['value'], r['ms_per_step'], r['vs_fp32_forward'])
PY
