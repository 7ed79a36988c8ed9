# round 6, eleventh lease: the driver's round-end commands on the committed tree - the whole -m gpu suite, then smoke
set -x
mkdir -p gpurun_out/r06l
( time timeout 1700 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r06l/tests.txt 2>&1 ) 2> gpurun_out/r06l/tests_time.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06l/smoke.txt 2>&1
tail -3 gpurun_out/r06l/tests.txt; tail -4 gpurun_out/r06l/tests_time.txt; tail -2 gpurun_out/r06l/smoke.txt
