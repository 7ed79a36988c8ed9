#!/bin/bash
# Counter passes over one train step with the bf16 backward (pn_set_backward_math(1)); separate --pmc passes, kernel-trace only.
set -u
OUT=gpurun_out/pmc_amp
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
CMD="python tools/amp_step.py"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace -d "$OUT/sq" -o p --output-format csv -- $CMD > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY --kernel-trace -d "$OUT/lds" -o p --output-format csv -- $CMD > "$OUT/lds.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/mem" -o p --output-format csv -- $CMD > "$OUT/mem.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o p --output-format csv -- $CMD > "$OUT/fetch.log" 2>&1
python - <<'PY'
import csv, glob, os, json
from collections import defaultdict
out = {}
for sub in ("sq", "lds", "mem", "fetch"):
    agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmc_amp/{sub}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"]
            if "bf16" not in k or "gemm_" not in k: continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            if d < 20: continue   # full-grid launches only
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); n[k] += 1; dur[k] += d
    out[sub] = {k.replace("void pn::", "")[:70]: {"launches": n[k], "avg_ms": dur[k] / n[k], **{c: v / n[k] for c, v in agg[k].items()}} for k in agg}
json.dump(out, open("gpurun_out/pmc_amp/summary_after.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
grep -i -E "^.*(TCP_|TCC_|SQ_.*LDS|SQ_WAIT)" "$OUT/avail.txt" | head -80 > "$OUT/avail_short.txt"
