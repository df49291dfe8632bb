#!/bin/bash
# rocprofv3 passes behind profiles/: kernel stats of `python bench.py`, then one --pmc pass per counter group
# (counters are collected in their own runs, never together with tracing domains other than --kernel-trace).
#   usage (on the GPU box, from the repo root):  bash tools/profile_bench.sh <tag> [command...]
# Writes gpurun_out/prof_<tag>/{stats,fetch,write,sq}/ and gpurun_out/prof_<tag>/summary.json (tools/summarize_prof.py).
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r1}; shift || true
if [ $# -eq 0 ]; then set -- python bench.py --steps 3 --warmup 1; fi
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$@" > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- "$@" > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- "$@" > "$OUT/write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- "$@" > "$OUT/sq.log" 2>&1
python tools/summarize_prof.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
