#!/bin/bash
# Fabric-side fetch and timing of the f32 weight-gradient (TN) kernels with the 32-workgroup region-task order on and off
# (PN_TN_TASKS).  usage (through gpurun): tools/pmc_tn_tasks.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_tn
mkdir -p $OUT
ONE="python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-fast-mode"
for D in 0 1; do
  PN_TN_TASKS=$D $ONE > $OUT/bench$D.json 2> $OUT/bench$D.err
  PN_TN_TASKS=$D rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch$D -o p --output-format csv -- $ONE > /dev/null 2> $OUT/fetch$D.log
  PN_TN_TASKS=$D rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/hit$D -o p --output-format csv -- $ONE > /dev/null 2> $OUT/hit$D.log
done
python - <<'PY'
import csv, glob, collections, json
res = {}
for d in (0, 1):
    j = json.load(open(f"gpurun_out/prof_tn/bench{d}.json"))
    res[f"tasks{d}"] = {"ms_per_step": j["ms_per_step"], "kernels": {k: v["tflops"] for k, v in j["kernels"].items() if v["tflops"] > 120}}
    for kind in ("fetch", "hit"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob(f"gpurun_out/prof_tn/{kind}{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                k = r["Kernel_Name"]
                if "gemm_tn" in k or "gemm_nt" in k:
                    k = k.replace("void pn::", "").split("(")[0]
                    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    agg[k]["n_" + r["Counter_Name"]] += 1
        for k, c in agg.items():
            e = res[f"tasks{d}"].setdefault("pmc", {}).setdefault(k, {})
            if "FETCH_SIZE" in c:
                e["fetch_GB_per_launch_x2"] = round(2 * c["FETCH_SIZE"] * 1024 / c["n_FETCH_SIZE"] / 1e9, 1)
                e["launches"] = c["n_FETCH_SIZE"]
            if "TCC_HIT_sum" in c:
                e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), 3)
    res[f"tasks{d}"]["pmc"] = {k: v for k, v in res[f"tasks{d}"]["pmc"].items() if v.get("fetch_GB_per_launch_x2", 0) > 50}
json.dump(res, open("gpurun_out/prof_tn/tn_tasks_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/fetch* $OUT/hit*
