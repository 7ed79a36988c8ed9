# round 6, eighth lease: the two test files after the split (full-size tests moved to tests/test_hip_full_size.py)
set -x
mkdir -p gpurun_out/r06i
timeout 1500 python -m pytest tests/test_hip_full_size.py tests/test_hip_train.py -q -p no:cacheprovider > gpurun_out/r06i/tests.txt 2>&1
tail -4 gpurun_out/r06i/tests.txt
