#!/bin/bash
# Round profiles on the GPU box (run through gpurun): kernel-trace statistics of bench.py, then SEPARATE counter passes
# (FETCH_SIZE, WRITE_SIZE, SQ busy/clock) as the MI355X guide prescribes (no --pmc together with other trace domains
# than --kernel-trace).  Writes gpurun_out/prof_<tag>/...; tools/pmc_traffic.py turns the counter CSVs into
# profiles/<tag>_hbm_traffic.json.   usage: tools/profile_round.sh r02
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o p --output-format csv -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.log"
ONE="python bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$C" -o p --output-format csv -- $ONE > "$OUT/bench_$C.json" 2> "$OUT/pmc_$C.log"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace -d "$OUT/pmc_SQ" -o p --output-format csv -- $ONE > "$OUT/bench_SQ.json" 2> "$OUT/pmc_SQ.log"
python tools/pmc_traffic.py "$OUT" "$TAG"
ls -la "$OUT"
