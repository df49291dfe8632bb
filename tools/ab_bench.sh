#!/bin/bash
# A/B of library variants on one box, alternating: tools/ab_bench.sh <config> <steps> <lib-a> <lib-b> [rounds]   (lib = path, or "default")
cd "$(dirname "$0")/.."
CFG=$1; STEPS=$2; A=$3; B=$4; R=${5:-2}
for r in $(seq 1 $R); do
  for L in $A $B; do
    if [ "$L" = "default" ]; then unset MI_MCMC_LIB; else export MI_MCMC_LIB=$L; fi
    python bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --traffic none --no-ess 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L', 'config', $CFG, 'ms_per_step', round(j['ms_per_step'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3), j['roofline']['kernel'])"
  done
done
