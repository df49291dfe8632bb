#!/bin/bash
# registers / scratch of every kernel of one translation unit: tools/resource_usage.sh <file.hip> [filter] [-DFLAGS...]
cd "$(dirname "$0")/../mcmc_amd/csrc"
TU=$1; FILT=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include/mi_mcmc_engine -I../../include "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null $TU 2>&1 | python3 -c "
import sys,re
cur=None; d={}
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur=m.group(1); d={}
    for key,pat in (('v','    VGPRs: '),('a','AGPRs: '),('scratch','ScratchSize \[bytes/lane\]: '),('spill','VGPRs Spill: '),('lds','LDS Size \[bytes/block\]: ')):
        m2=re.search(pat+r'(\d+)',ln)
        if m2 and cur: d[key]=m2.group(1)
    if 'LDS Size' in ln and cur:
        if re.search('$FILT',cur): print(cur, d)
        cur=None
"
