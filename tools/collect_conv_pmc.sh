#!/bin/bash
# SQ counters of the output-stationary sparse convolution on single layers of the FSD bench pass: rocprofv3 --pmc passes over
# tools/conv_only.py (counters only; never combined with --sys-trace etc.), summarised for the LOOP's launches by
# tools/conv_pmc_summary.py.  Usage (GPU box): bash tools/collect_conv_pmc.sh gpurun_out/pmc_conv [layer indices ...]
OUT=${1:-gpurun_out/pmc_conv}
shift
LAYERS=${@:-0 3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
for L in $LAYERS; do
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf /tmp/pc
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pc -o p -- python "$R/tools/conv_only.py" fsd $L 6 > /tmp/pc.log 2>&1 || tail -3 /tmp/pc.log
    f=$(ls /tmp/pc/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then grep -E "Kernel_Name|sp_conv_os_k" "$f" > "$R/$OUT/L${L}_set$i.csv"; fi
  done
done
python "$R/tools/conv_pmc_summary.py" "$R/$OUT" $LAYERS | tee "$R/$OUT/summary.txt"
