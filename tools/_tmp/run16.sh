for v in 1 0 1 0; do
SST_AMD_BF16_FUSED_LN=$v python bench.py --precision bf16 --no-cpu-baseline --no-forward-only-leg 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_ln', $v, j['value'], j['ms_per_step'])"
done
