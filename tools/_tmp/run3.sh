python -m pytest tests/test_gpu_virtual_voxel.py -q -x 2>&1 | tail -8
mkdir -p gpurun_out/r2k
for w in fsd fsdv2 sst_bs2; do
  timeout 400 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-forward-only-leg --no-bf16-leg > gpurun_out/r2k/bench_$w.json 2> gpurun_out/r2k/bench_$w.err || tail -8 gpurun_out/r2k/bench_$w.err
  cut -c1-1500 gpurun_out/r2k/bench_$w.json
done
