cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -o b -- python $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline > /tmp/pb.log 2>&1 || tail -5 /tmp/pb.log
tail -1 /tmp/pb.log | cut -c1-200
python $R/tools/gap_report.py /tmp/pb/b_kernel_trace.csv 0.7 60 > $R/gpurun_out/trace_report.txt; cat $R/gpurun_out/trace_report.txt
