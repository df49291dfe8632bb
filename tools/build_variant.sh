#!/bin/bash
# A/B library: tools/build_variant.sh <name> <file.hip> [-DFLAGS...] -> mcmc_amd/libmi_<name>.so (same objects, one TU recompiled)
set -e
cd "$(dirname "$0")/../mcmc_amd/csrc"
NAME=$1; TU=$2; shift 2
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include/mi_mcmc_engine"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/${TU%.hip}_$NAME.o $TU
OBJS=$(ls build/*.o | grep -v "_prof.o" | grep -v "build/${TU%.hip}\.o" | grep -v -E "_[a-z0-9]+\.o$" || true)
BASE=""
for o in mi_mcmc gemm_samplers linalg_device literal_launch callback_host stats_collate hmc_launch hmc_general_launch hmc_dense_launch mala_launch nuts_launch nuts_general_launch nuts_bounded_launch nuts_dense_launch rwmh_launch small_launch small_logit_d12 small_logit_d34 small_logit_d56 small_logit_d78 logistic_lds logistic_nuts logistic_hmc_box logistic_hmc_dense_m logistic_mala_dense_m logistic_nuts_box logistic_nuts_dense_m; do
  if [ "$o.hip" = "$TU" ]; then BASE="$BASE build/${o}_$NAME.o"; else BASE="$BASE build/$o.o"; fi
done
/opt/rocm/bin/hipcc $FLAGS -shared -o ../libmi_$NAME.so $BASE
echo built ../libmi_$NAME.so
