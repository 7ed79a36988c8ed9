#!/bin/bash
# Counter passes over the AMP-class eval forward (tools/amp_fwd.py); separate --pmc passes, kernel-trace only (the MI355X guide's
# rule).  Summary -> gpurun_out/pmc_amp_fwd/summary.json: per kernel (the all-DMA single-product GEMMs and k_make_h_bf16) matrix-
# pipe busy, shader clock, VALU per MFMA, L2 hit rate, TCP stall share, fabric-side fetch / write bytes per launch.
set -u
OUT=gpurun_out/pmc_amp_fwd
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/amp_fwd.py"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace -d "$OUT/sq" -o p --output-format csv -- $CMD > "$OUT/sq.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/mem" -o p --output-format csv -- $CMD > "$OUT/mem.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o p --output-format csv -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o p --output-format csv -- $CMD > "$OUT/write.log" 2>&1
python - <<'PY'
import csv, glob, json
from collections import defaultdict
out = {}
for sub in ("sq", "mem", "fetch", "write"):
    agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmc_amp_fwd/{sub}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"]
            if not ("bf16dma" in k or "bf16m16" in k or "make_h" in k): continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            if d < 1.0: continue   # pair-grid chunk launches only (8 ms GEMMs, ~1.5 ms operand passes)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); n[k] += 1; dur[k] += d
    out[sub] = {k.replace("void pn::", "")[:80]: {"launches": n[k], "avg_ms": dur[k] / n[k], **{c: v / n[k] for c, v in agg[k].items()}} for k in agg}
res = {"workload": "eval forward, B = 256 x L = 512 x 32102 labels, bf16x3 base + forward_math = bf16 (tools/amp_fwd.py), per chunk launch "
                   "(2048 labels x 256 proteins = 524288 pair rows: 9.9 TFLOP per GEMM launch)", "kernels": {}}
for k, v in out["sq"].items():
    e = {"launches": v["launches"], "avg_ms": v["avg_ms"]}
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    if gui > 0:
        e["mfma_busy"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)
        e["clock_ghz"] = gui / 8.0 / (v["avg_ms"] * 1e-3) / 1e9
        if v.get("SQ_INSTS_MFMA", 0.0) > 0: e["valu_per_mfma"] = v.get("SQ_INSTS_VALU", 0.0) / v["SQ_INSTS_MFMA"]
    m = out["mem"].get(k, {})
    if m.get("TCC_REQ_sum", 0.0) > 0:
        e["l2_hit"] = m.get("TCC_HIT_sum", 0.0) / m["TCC_REQ_sum"]
    f, w = out["fetch"].get(k, {}), out["write"].get(k, {})
    # MI355X guide: FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE counts half the bytes of wide reads on gfx950 -> doubled
    e["fabric_fetch_GB"] = 2.0 * f.get("FETCH_SIZE", 0.0) * 1024.0 / 1e9
    e["fabric_write_GB"] = w.get("WRITE_SIZE", 0.0) * 1024.0 / 1e9
    res["kernels"][k] = e
import os, sys
sys.path.insert(0, os.getcwd())
from protnote_amd.build import csrc_hash
res["csrc_hash"] = csrc_hash()
json.dump(res, open("gpurun_out/pmc_amp_fwd/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:5000])
PY
