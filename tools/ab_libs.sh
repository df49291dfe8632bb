#!/bin/bash
# same box, libraries alternating: tools/ab_libs.sh <chains> <reps> lib1.so lib2.so ...   (tools/nuts_ab.py per library; ms + checksum per run)
C=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for L in "$@"; do
    echo -n "$L: "
    MI_MCMC_LIB=$L python tools/nuts_ab.py $C 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    j = json.loads(ln); print('%.1f ms (exec %.4g, chk %.17g)' % (j['ms'], j['executed'], j['checksum']), end='; ')
print()"
  done
done
