#!/bin/bash
# run bench with several flag sets: tools/abl_bench.sh "<flags1>" "<flags2>" ... -- [bench args]
FLAGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do FLAGS+=("$1"); shift; done
shift
for F in "${FLAGS[@]}"; do
  PN_EXTRA_HIPCC_FLAGS="$F" python -m protnote_amd.build --force >/dev/null 2>&1
  echo -n "[$F]: "
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step %.1f  family %.1f TF |' % (d['ms_per_step'], d['roofline']['achieved']), ' '.join('%s=%.1f'%(k.split(':')[1][:14],v['tflops']) for k,v in d['kernels'].items() if v['ms_total']>50))"
done
