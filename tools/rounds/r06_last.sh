# round 6, last lease: the final sources (source hash e13edefca16d5230 = 8bc82ce832a81f42 + the paired-fragment bf16 store of
# E_STORE_H16): the kernel lab, counter passes + kernel stats of the headline command, smoke, the driver's bench invocation, the
# whole GPU suite, then the AMP train-step counters and the AMP shape sweep (tools/rounds/r06_extra.sh)
set -x
mkdir -p gpurun_out/r06s
timeout 300 tools/lab_bf16_nt.bin > gpurun_out/r06s/lab.txt 2>&1; echo "rc=$?" >> gpurun_out/r06s/lab.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06s/smoke.txt 2>&1
bash tools/profile_round.sh r06 > gpurun_out/r06s/profile_round.log 2>&1
cp gpurun_out/prof_r06/r06_hbm_traffic.json gpurun_out/r06s/ 2>/dev/null
find gpurun_out/prof_r06/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06s/r06_kernel_stats.csv \;
cp gpurun_out/prof_r06/bench_stats.json gpurun_out/r06s/r06_bench_under_rocprof.json
rm -rf gpurun_out/prof_r06
mkdir -p profiles && cp gpurun_out/r06s/r06_hbm_traffic.json profiles/r06_hbm_traffic.json
timeout 900 bash tools/pmc_amp_fwd.sh > gpurun_out/r06s/pmc_amp_fwd.log 2>&1
cp gpurun_out/pmc_amp_fwd/summary.json gpurun_out/r06s/pmc_amp_fwd_summary.json
rm -rf gpurun_out/pmc_amp_fwd/sq gpurun_out/pmc_amp_fwd/mem gpurun_out/pmc_amp_fwd/fetch gpurun_out/pmc_amp_fwd/write
export TMPDIR=/tmp
PN_FORWARD_MATH=bf16 PN_STEPS=4 rocprofv3 --kernel-trace --stats -d gpurun_out/r06s/stats -o p --output-format csv -- python tools/amp_step.py > gpurun_out/r06s/amp_step.log 2>&1
find gpurun_out/r06s/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06s/r06_kernel_stats_amp_full.csv \;
rm -rf gpurun_out/r06s/stats
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06s/bench.out 2> gpurun_out/r06s/bench.err ) 2> gpurun_out/r06s/bench_time.txt
cp bench_detail.json gpurun_out/r06s/
( time timeout 1700 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r06s/tests.txt 2>&1 ) 2> gpurun_out/r06s/tests_time.txt
tail -1 gpurun_out/r06s/bench.out | cut -c1-700
cat gpurun_out/r06s/bench_time.txt gpurun_out/r06s/smoke.txt | tail -6
tail -2 gpurun_out/r06s/tests.txt

mkdir -p gpurun_out/r06s
export TMPDIR=/tmp
OUT=gpurun_out/r06s/pmc
mkdir -p $OUT
CMD="python tools/amp_step.py"
export PN_FORWARD_MATH=bf16 PN_STEPS=2
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
    for f in glob.glob(f"gpurun_out/r06s/pmc/{sub}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"]
            if not ("bf16" in k and "gemm_" in k) or "bf16x3" in k: continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            if d < 2.0: continue   # pair-grid launches only (4 ms chunk GEMMs, 150 ms weight gradients)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); n[k] += 1; dur[k] += d
    out[sub] = {k.replace("void pn::", "")[:80]: {"launches": n[k], "avg_ms": dur[k] / n[k], **{c: v / n[k] for c, v in agg[k].items()}} for k in agg}
res = {"workload": "AMP-class train step (bf16x3 base, forward_math = backward_math = bf16) at the bench shape, tools/amp_step.py, 2 steps; per launch", "kernels": {}}
for k, v in out["sq"].items():
    e = {"launches": v["launches"], "avg_ms": v["avg_ms"]}
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    if gui > 0:
        e["mfma_busy"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)
        e["clock_ghz"] = gui / 8.0 / (v["avg_ms"] * 1e-3) / 1e9
        if v.get("SQ_INSTS_MFMA", 0.0) > 0: e["valu_per_mfma"] = v.get("SQ_INSTS_VALU", 0.0) / v["SQ_INSTS_MFMA"]
    m = out["mem"].get(k, {})
    if m.get("TCC_REQ_sum", 0.0) > 0: e["l2_hit"] = m.get("TCC_HIT_sum", 0.0) / m["TCC_REQ_sum"]
    if "TCP_PENDING_STALL_CYCLES_sum" in m: e["tcp_pending_stall_cycles"] = m["TCP_PENDING_STALL_CYCLES_sum"]
    f, w = out["fetch"].get(k, {}), out["write"].get(k, {})
    e["fabric_fetch_GB"] = 2.0 * f.get("FETCH_SIZE", 0.0) * 1024.0 / 1e9   # (MI355X guide: KB units, wide reads counted at half)
    e["fabric_write_GB"] = w.get("WRITE_SIZE", 0.0) * 1024.0 / 1e9
    res["kernels"][k] = e
from protnote_amd import build
res["csrc_hash"] = build.csrc_hash()
json.dump(res, open("gpurun_out/r06s/r06_pmc_amp_train.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:4000])
PY
rm -rf $OUT/sq $OUT/mem $OUT/fetch $OUT/write
PN_SWEEP_MODE=amp timeout 900 python tools/shape_sweep.py gpurun_out/r06s/r06_shape_sweep_amp.json > gpurun_out/r06s/sweep.log 2>&1
tail -5 gpurun_out/r06s/sweep.log

grep 'PRODUCT\|LAB' gpurun_out/r06s/lab.txt
