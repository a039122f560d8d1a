for st in 20 300; do
python bench.py --precision bf16 --steps $st --no-cpu-baseline --no-forward-only-leg 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps', $st, j['value'], j['ms_per_step'])"
done
python bench.py --no-cpu-baseline --no-forward-only-leg --steps 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 then leg', j['value'], j['ms_per_step'], j['reduced_precision']['ms_per_step'])"
python bench.py --no-cpu-baseline --no-forward-only-leg --steps 20 --no-gemm-tuning 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no tuning: fp32 then leg', j['value'], j['ms_per_step'], j['reduced_precision']['ms_per_step'])"
