#!/bin/bash
# eval pair-head timing under several builds: tools/abl_fwd.sh "<flags1>" "<flags2>" ...   (env PN_MATH_MODE respected)
for F in "$@"; do
  PN_EXTRA_HIPCC_FLAGS="$F" python -m protnote_amd.build --force >/dev/null 2>&1
  echo -n "[$F]: "
  python tools/quick_fwd.py 2>/dev/null | grep -A1 "^pairhead" | tr '\n' ' '
  echo
done
python -m protnote_amd.build --force >/dev/null 2>&1
