#!/bin/bash
# bf16x3 NT kernels, low-VALU rewrite on / off (PN_B3_FAST): timing, matrix-pipe busy, shader clock, VALU per MFMA.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_b3
mkdir -p $OUT
ONE="python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --math bf16x3 --no-fast-mode"
for D in 0 1; do
  PN_B3_FAST=$D $ONE > $OUT/bench$D.json 2> $OUT/bench$D.err
  PN_B3_FAST=$D rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace -d $OUT/sq$D -o p --output-format csv -- $ONE > /dev/null 2> $OUT/sq$D.log
done
python - <<'PY'
import csv, glob, collections, json
for d in (0, 1):
    j = json.load(open(f"gpurun_out/prof_b3/bench{d}.json"))
    print("fast", d, j["ms_per_step"], {k: v["tflops"] for k, v in j["kernels"].items() if "bf16x3" in k})
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/prof_b3/sq{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"]
            if "bf16x3" in k or "b3_fast" in k:
                k = k.replace("void pn::", "").split("(")[0]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if (k, r["Dispatch_Id"]) not in seen:
                    seen.add((k, r["Dispatch_Id"]))
                    agg[k]["sec"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                    agg[k]["n"] += 1
    for k, c in sorted(agg.items()):
        if c["sec"] / max(c["n"], 1) > 0.05:
            print("  ", k[:60], "avg_ms", round(c["sec"] / c["n"] * 1e3, 1), "busy", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 3),
                  "clock", round(c["GRBM_GUI_ACTIVE"] / 8 / c["sec"] / 1e9, 3), "valu/mfma", round(c["SQ_INSTS_VALU"] / max(c["SQ_INSTS_MFMA"], 1), 2))
PY
rm -rf $OUT/sq0 $OUT/sq1
