#!/bin/bash
# One GPU-box visit: the -m gpu suite, then a bench line per BASELINE config (gpurun_out/<tag>_*.{log,json}).
#   usage: bash tools/gpu_round.sh <tag> [pytest-args...]
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r2}; shift || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err
for c in 3 4 5; do
  timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_c$c.json 2> gpurun_out/${TAG}_bench_c$c.err
done
for n in 8192 16384 32768; do
  timeout 300 python bench.py --chains $n --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_c2_${n}chains.json 2>/dev/null
done
for f in gpurun_out/${TAG}_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print({k: j[k] for k in ("value", "ms_per_step")}, j["roofline"], j.get("cpu_baseline", {}).get("mode_a"), j.get("cpu_baseline", {}).get("mode_b"))
except Exception as e:
    print("no line:", e)
PY
done
