cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-forward-only-leg --no-bf16-leg > /tmp/b.log 2>&1 || tail -5 /tmp/b.log
python $GRAFT_REPO_ROOT/tools/gap_report.py /tmp/pb/b_kernel_trace.csv 0.65 12 | cut -c1-130 | head -14
