#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_gemm_*: kernel stats of `python tools/gemm_time.py short` (hmc, dense Gaussian d = 1024, 65 536 chains, L = 16:
# gemm_samplers.hip), then one --pmc pass per counter group (counters in their own runs, --kernel-trace only), condensed by tools/summarize_prof.py.
#   usage (on the GPU box, from the repo root):  bash tools/profile_gemm.sh <tag>
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r6}
OUT=gpurun_out/prof_${TAG}_gemm
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/gemm_time.py short"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1
python tools/summarize_prof.py "$OUT" 0 "$CMD" > "$OUT/pmc.json"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
python - "$OUT/pmc.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(json.dumps(j.get("derived"), indent=1))
PY
