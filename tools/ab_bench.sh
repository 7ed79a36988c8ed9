#!/bin/bash
# A/B two builds of the library on the GPU box: usage tools/ab_bench.sh "<flagsA>" "<flagsB>" [bench args]
FA="$1"; FB="$2"; shift 2
for round in 1 2; do
  for v in A B; do
    if [ $v = A ]; then F="$FA"; else F="$FB"; fi
    PN_EXTRA_HIPCC_FLAGS="$F" python -m protnote_amd.build --force >/dev/null 2>&1
    echo -n "round $round variant $v [$F]: "
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step %.1f  family %.1f TF |' % (d['ms_per_step'], d['roofline']['achieved']), ' '.join('%s=%.1f'%(k.split(':')[1][:14],v['tflops']) for k,v in d['kernels'].items() if v['ms_total']>50))"
  done
done
