#!/bin/bash
# quick resource check of one kernel instantiation (seconds): tools/dev_compile.sh <dev.hip> [-DMACRO=..]
F=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I/root/repo/mcmc_amd/csrc -I/root/repo/include/mi_mcmc_engine "$@" -Rpass-analysis=kernel-resource-usage -save-temps=obj -c -o /tmp/nuts/$(basename $F .hip).o $F 2>&1 | grep -E "error|warning|VGPRs:|AGPRs|SGPRs Spill|VGPRs Spill|Scratch"
