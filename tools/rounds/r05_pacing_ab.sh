set -x
mkdir -p gpurun_out/r05g
python -m protnote_amd.build > /dev/null 2>&1
PN_STEPS=6 python tools/tn_tasks_ab.py gpurun_out/r05g/pace_8.json > /dev/null 2> gpurun_out/r05g/a.err
for N in 4 16 32; do
  export PN_EXTRA_HIPCC_FLAGS=-DPN_TN_SYNC_SLABS_B16=$N
  python -m protnote_amd.build > gpurun_out/r05g/build_$N.log 2>&1
  PN_STEPS=6 python tools/tn_tasks_ab.py gpurun_out/r05g/pace_$N.json > /dev/null 2> gpurun_out/r05g/b_$N.err
done
unset PN_EXTRA_HIPCC_FLAGS
python -m protnote_amd.build > /dev/null 2>&1
PN_STEPS=6 python tools/tn_tasks_ab.py gpurun_out/r05g/pace_8_again.json > /dev/null 2>> gpurun_out/r05g/a.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05g/pace_*.json")):
    d=json.load(open(f)); print(f, d["extra_flags"], round(d["ms_per_step"],1), d["per_launch_ms"].get('tn:plain x bn_relu [bf16, one product]'))
P
