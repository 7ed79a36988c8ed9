#!/bin/bash
# L2 hit rate / fabric fetch of the f32 pair-grid GEMMs with LDS-DMA staging on and off (PN_F32_DMA), one counter
# pass each.  usage (through gpurun): tools/pmc_l2.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_l2
mkdir -p $OUT
ONE="python bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-fast-mode"
for D in 0 1; do
  PN_F32_DMA=$D rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/hit$D -o p --output-format csv -- $ONE > /dev/null 2> $OUT/hit$D.log
  PN_F32_DMA=$D rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch$D -o p --output-format csv -- $ONE > /dev/null 2> $OUT/fetch$D.log
done
python - <<'PY'
import csv, glob, collections, json
res = {}
for d in (0, 1):
    for kind in ("hit", "fetch"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob(f"gpurun_out/prof_l2/{kind}{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                k = r["Kernel_Name"]
                if "gemm_nt" in k or "gemm_tn" in k:
                    k = k.replace("void pn::", "").split("(")[0]
                    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    agg[k]["n_" + r["Counter_Name"]] += 1
        for k, c in agg.items():
            e = res.setdefault(f"dma{d}", {}).setdefault(k, {})
            if "TCC_HIT_sum" in c:
                e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
                e["l2_req_per_launch"] = (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) / c["n_TCC_HIT_sum"]
            if "FETCH_SIZE" in c:
                e["fetch_GB_per_launch_x2"] = 2 * c["FETCH_SIZE"] * 1024 / c["n_FETCH_SIZE"] / 1e9
                e["launches"] = c["n_FETCH_SIZE"]
json.dump(res, open("gpurun_out/prof_l2/l2_summary.json", "w"), indent=1)
for d, ks in res.items():
    for k, e in sorted(ks.items()):
        if e.get("fetch_GB_per_launch_x2", 0) > 20:
            print(d, k[:48], {a: round(b, 3) for a, b in e.items()})
PY
rm -rf $OUT/hit* $OUT/fetch*
