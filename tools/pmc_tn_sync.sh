#!/bin/bash
# fabric fetch and speed of the f32 TN kernels with the region tasks' pacing on / off (PN_TN_SYNC)
export TMPDIR=/tmp
ONE="python bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-fast-mode"
for v in 0 1; do
  PN_TN_SYNC=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_sync/$v -o p --output-format csv -- $ONE > gpurun_out/prof_sync_$v.json 2>/dev/null
  PN_TN_SYNC=$v python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-fast-mode > gpurun_out/prof_sync_t$v.json 2>/dev/null
done
python - <<'PY'
import csv, glob, collections, json
for v in ("0","1"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/prof_sync/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if "gemm_tn_fast" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-24:]].append(2*float(r["Counter_Value"])*1024/1e9)
    j = json.load(open(f"gpurun_out/prof_sync_t{v}.json"))
    print("sync", v, {k:[round(x) for x in a] for k,a in agg.items()}, j["ms_per_step"], {k:x["tflops"] for k,x in j["kernels"].items() if k.startswith("tn") and x["tflops"]>120})
PY
rm -rf gpurun_out/prof_sync
