#!/bin/bash
# SQ counters of the SRA kernels (forward + one-pass backward) on the bench frame: rocprofv3 --pmc passes over
# tools/sra_only.py (counters only; never combined with --sys-trace etc.), summarised per kernel by tools/pmc_summary.py.
# Usage (GPU box): bash tools/collect_sra_pmc.sh gpurun_out/pmc_sra
set -e
OUT=${1:-gpurun_out/pmc_sra}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm1 /tmp/pm2 /tmp/pm3
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pm1 -o p -- python "$R/tools/sra_only.py" 4 > /tmp/pm1.log 2>&1 || tail -3 /tmp/pm1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pm2 -o p -- python "$R/tools/sra_only.py" 4 > /tmp/pm2.log 2>&1 || tail -3 /tmp/pm2.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm3 -o p -- python "$R/tools/sra_only.py" 4 > /tmp/pm3.log 2>&1 || tail -3 /tmp/pm3.log
for i in 1 2 3; do
  f=$(ls /tmp/pm$i/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then grep -E "Kernel_Name|sra_" "$f" > "$R/$OUT/sq_set$i.csv"; fi
done
python "$R/tools/pmc_summary.py" 'sra_(fwd_wave|bwd_fused|bwd_dq|bwd_dkv)_k<[0-9]+>' "$R/$OUT"/sq_set*.csv | tee "$R/$OUT/summary.txt"
