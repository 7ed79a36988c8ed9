#!/bin/bash
# where the waves of the bf16x3 NT kernels wait: old kernel vs the low-VALU rewrite (PN_B3_FAST)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_b3s
mkdir -p $OUT
ONE="python bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --math bf16x3 --no-fast-mode"
for D in 0 1; do
  PN_B3_FAST=$D rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/a$D -o p --output-format csv -- $ONE > /dev/null 2> $OUT/a$D.log
  PN_B3_FAST=$D rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/b$D -o p --output-format csv -- $ONE > /dev/null 2> $OUT/b$D.log
done
python - <<'PY'
import csv, glob, collections
for d in (0, 1):
    for tag in ("a", "b"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob(f"gpurun_out/prof_b3s/{tag}{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                k = r["Kernel_Name"]
                if ("nt_bf16x3_kernel<1, 0" in k or "b3_fast_kernel<1, 0" in k):
                    agg["bn_relu"][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, c in agg.items():
            wc = c.get("SQ_WAVE_CYCLES", 0) or 1
            print("fast", d, tag, k, {a: (round(b / wc, 3) if tag == "a" else f"{b:.3g}") for a, b in c.items()})
PY
rm -rf $OUT/a0 $OUT/a1 $OUT/b0 $OUT/b1
