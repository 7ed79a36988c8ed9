# round 6, seventh lease: the one reworked test and the driver's bench invocation with the final cpu_baseline
set -x
mkdir -p gpurun_out/r06h
timeout 900 python -m pytest tests/test_hip_fwd_bf16.py -q -s -p no:cacheprovider > gpurun_out/r06h/tests.txt 2>&1
tail -3 gpurun_out/r06h/tests.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06h/bench.out 2> gpurun_out/r06h/bench.err ) 2> gpurun_out/r06h/bench_time.txt
cp bench_detail.json gpurun_out/r06h/
tail -1 gpurun_out/r06h/bench.out | cut -c1-1500
cat gpurun_out/r06h/bench_time.txt
