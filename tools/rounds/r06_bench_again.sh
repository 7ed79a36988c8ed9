# round 6: the driver's bench invocation once more on the final sources (another box: the bf16 modes are power-limited and move ~3 %
# between boxes - leases q and s measured the same train-step kernels at 1079 and 1116 ms)
set -x
mkdir -p gpurun_out/r06t
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06t/bench.out 2> gpurun_out/r06t/bench.err ) 2> gpurun_out/r06t/bench_time.txt
cp bench_detail.json gpurun_out/r06t/
rocm-smi --showpower --showclocks --showtemp > gpurun_out/r06t/smi.txt 2>&1
tail -1 gpurun_out/r06t/bench.out | cut -c1-400
