#!/bin/bash
# The rocprofv3 evidence behind profiles/<tag>_*: kernel-trace stats + three PMC passes for every BASELINE config bench.py runs
# (tools/profile_bench.sh), then the plain bench lines.   usage (GPU box, repo root): bash tools/profile_all.sh <tag>
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r2}
mkdir -p gpurun_out/profiles_$TAG
for c in 3 4 5; do
  bash tools/profile_bench.sh $TAG $c > gpurun_out/prof_${TAG}_c$c.log 2>&1
  cp gpurun_out/prof_${TAG}_c$c/pmc.json gpurun_out/profiles_$TAG/${TAG}_c${c}_pmc.json
  cp gpurun_out/prof_${TAG}_c$c/kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_c${c}_kernel_stats.csv
done
bash tools/profile_bench.sh $TAG 2 > gpurun_out/prof_${TAG}_c2.log 2>&1
cp gpurun_out/prof_${TAG}_c2/pmc.json gpurun_out/profiles_$TAG/${TAG}_c2_pmc.json
cp gpurun_out/prof_${TAG}_c2/kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_c2_kernel_stats.csv
mv gpurun_out/prof_${TAG}_c2 gpurun_out/prof_${TAG}_c2_full
# strong-scaling shape of configs[1] at one GPU's share of 65 536 chains over 8 GPUs (hmc_split.hpp)
bash tools/profile_bench.sh $TAG 2 --chains 8192 > gpurun_out/prof_${TAG}_c2s.log 2>&1
mv gpurun_out/prof_${TAG}_c2 gpurun_out/prof_${TAG}_c2_8192
cp gpurun_out/prof_${TAG}_c2_8192/pmc.json gpurun_out/profiles_$TAG/${TAG}_c2_8192chains_pmc.json
cp gpurun_out/prof_${TAG}_c2_8192/kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_c2_8192chains_kernel_stats.csv
# NUTS (configs[3]) at one GPU's share of 65 536 chains over 8 GPUs
mv gpurun_out/prof_${TAG}_c4 gpurun_out/prof_${TAG}_c4_full
bash tools/profile_bench.sh $TAG 4 --chains 8192 > gpurun_out/prof_${TAG}_c4s.log 2>&1
cp gpurun_out/prof_${TAG}_c4/pmc.json gpurun_out/profiles_$TAG/${TAG}_c4_8192chains_pmc.json
cp gpurun_out/prof_${TAG}_c4/kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_c4_8192chains_kernel_stats.csv
