#!/bin/bash
# rocprofv3 passes behind profiles/: kernel stats of `python bench.py --config N`, then one --pmc pass per counter group
# (counters are collected in their own runs, never together with tracing domains other than --kernel-trace).
#   usage (on the GPU box, from the repo root):  bash tools/profile_bench.sh <tag> <config> [extra bench.py args...]
# Writes gpurun_out/prof_<tag>_c<config>/{stats,fetch,write,sq}/ and, condensed by tools/summarize_prof.py,
# gpurun_out/prof_<tag>_c<config>/{pmc.json,kernel_stats.csv}: copy those to profiles/<tag>_c<config>_{pmc.json,kernel_stats.csv}.
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r2}; CFG=${2:-2}; shift 2 || true
OUT=gpurun_out/prof_${TAG}_c$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --traffic none $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1
grep -h '^{' "$OUT/stats.log" | tail -1 > "$OUT/bench_line.json"
python tools/summarize_prof.py "$OUT" "$CFG" "$CMD" > "$OUT/pmc.json"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
python - "$OUT/pmc.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(json.dumps(j.get("derived"), indent=1))
PY
