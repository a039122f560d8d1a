#!/bin/bash
# same-box A/B of two builds of the library: SST_AMD_LIB=<base .so> against the in-tree one.
# The base build is made by hand before the call (it travels with the snapshot, *.so is git-ignored):
#   mkdir -p sst_amd/csrc/ab && git show HEAD:sst_amd/csrc/<file>.hip > sst_amd/csrc/ab/<file>_base.hip   (+ a copy of common.h with
#   the include path one level deeper), hipcc -c it with the Makefile's flags, link it with the other in-tree objects into
#   sst_amd/csrc/ab/libsst_amd_base.so; remove sst_amd/csrc/ab afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
BASE=$R/sst_amd/csrc/ab/libsst_amd_base.so
SIDE="--no-cpu-baseline --no-forward-only-leg --no-lidar-leg --no-f32x3-leg --no-traffic-remeasure --no-config-as-is-leg --no-bf16-own-process --no-bf16-leg"
python -m pytest ${TESTS:-tests/test_gpu_dense_f32x6.py} -q -x 2>&1 | tail -2
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export SST_AMD_LIB=$BASE; else unset SST_AMD_LIB; fi
    python bench.py $SIDE 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v sst',d['value'],d['ms_per_step'],d['step_ms']['median'])"
    python bench.py --cloud lidar $SIDE 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v lidar',d['value'],d['ms_per_step'],d['step_ms']['median'])"
  done
done
