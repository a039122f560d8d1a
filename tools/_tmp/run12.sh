cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb
rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-forward-only-leg --no-bf16-leg > /tmp/b.log 2>&1
python $GRAFT_REPO_ROOT/tools/front_of_step.py /tmp/pb/b_kernel_trace.csv
tail -1 /tmp/b.log | cut -c1-200
