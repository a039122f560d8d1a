cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1 /tmp/p2
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/p1 -o a -- python $GRAFT_REPO_ROOT/tools/lds_linear_only.py 4 > /tmp/p1.log 2>&1 || tail -5 /tmp/p1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/p2 -o a -- python $GRAFT_REPO_ROOT/tools/lds_linear_only.py 4 > /tmp/p2.log 2>&1 || tail -5 /tmp/p2.log
python - <<'PY'
import csv, collections
for d in ('/tmp/p1/a_counter_collection.csv', '/tmp/p2/a_counter_collection.csv'):
    acc = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(d)):
            if 'tall_linear_lds' in r['Kernel_Name']:
                acc[(r['Kernel_Name'][40:62], r['Counter_Name'])].append(float(r['Counter_Value']))
    except Exception as e:
        print('ERR', d, e); continue
    for k, v in sorted(acc.items()):
        print(k, round(sum(v) / len(v)), len(v))
PY
