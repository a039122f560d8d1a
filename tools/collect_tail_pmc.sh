#!/bin/bash
# SQ counters of the one-kernel encoder tail (csrc/layer_tail_x6.hip) on the bench frame: three rocprofv3 --pmc passes over
# tools/tail_bench.py (counters only), summarised per kernel by tools/pmc_summary.py.
# Usage (GPU box): bash tools/collect_tail_pmc.sh gpurun_out/pmc_tail [M]
set -e
OUT=${1:-gpurun_out/pmc_tail}
M=${2:-90107}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm1 /tmp/pm2 /tmp/pm3
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pm1 -o p -- python "$R/tools/tail_bench.py" $M > /tmp/pm1.log 2>&1 || tail -3 /tmp/pm1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pm2 -o p -- python "$R/tools/tail_bench.py" $M > /tmp/pm2.log 2>&1 || tail -3 /tmp/pm2.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm3 -o p -- python "$R/tools/tail_bench.py" $M > /tmp/pm3.log 2>&1 || tail -3 /tmp/pm3.log
for i in 1 2 3; do
  f=$(ls /tmp/pm$i/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then grep -E "Kernel_Name|encoder_tail|tall_linear_f32x6_k<128, 128, 1>" "$f" > "$R/$OUT/sq_set$i.csv"; fi
done
python "$R/tools/pmc_summary.py" 'encoder_tail_(fwd|bwd)_x6_k<[0-9]+>|tall_linear_f32x6_k<128, 128, 1>' "$R/$OUT"/sq_set*.csv | tee "$R/$OUT/summary.txt"
