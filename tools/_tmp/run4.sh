python -m pytest tests/test_gpu_dense.py -q -x -k "lds_linear" 2>&1 | tail -4
python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_sra.py tests/test_gpu_dist.py -q -x 2>&1 | tail -4
mkdir -p gpurun_out/r2l
python bench.py --no-cpu-baseline --no-forward-only-leg --no-bf16-leg > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err || tail -5 gpurun_out/r2l/bench.err
cut -c1-330 gpurun_out/r2l/bench.json
SST_AMD_LDS_LINEAR=0 python bench.py --no-cpu-baseline --no-forward-only-leg --no-bf16-leg 2>/dev/null | cut -c1-330
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-forward-only-leg --no-bf16-leg > /tmp/b.log 2>&1 || tail -5 /tmp/b.log
python $GRAFT_REPO_ROOT/tools/gap_report.py /tmp/pb/b_kernel_trace.csv 0.65 30 > $GRAFT_REPO_ROOT/gpurun_out/r2l/steady_state_trace_report.txt
head -34 $GRAFT_REPO_ROOT/gpurun_out/r2l/steady_state_trace_report.txt | cut -c1-130
