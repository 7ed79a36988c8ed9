#!/bin/bash
# SQ wave-state counters of one GEMM microbenchmark (tools/gemm_k_sweep.py): where the waves of the f32 LDS-DMA kernel
# spend their cycles.  usage (through gpurun): tools/pmc_kernel.sh
export TMPDIR=/tmp
OUT=gpurun_out/prof_sq
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/a -o p --output-format csv -- python tools/gemm_k_sweep.py > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/b -o p --output-format csv -- python tools/gemm_k_sweep.py >> $OUT/run.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/prof_sq/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if "gemm_nt_dma" in r["Kernel_Name"]:
                key = r["Grid_Size"]
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                agg[key]["n"] += 1
    for k, c in agg.items():
        print(tag, "grid", k, {a: f"{b:.4g}" for a, b in c.items()})
PY
rm -rf $OUT/a $OUT/b
