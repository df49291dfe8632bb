cd /root/repo
for v in 1 2 3 4; do echo "variant $v"; MI_MCMC_LIB=tools/bin/libmi_v$v.so timeout 300 python tools/bench_configs.py --algo nuts --chains 65536 --reps 1 2>&1 | tail -1 | cut -c1-70; done
for b in 4 16; do echo "batch $b"; MI_NUTS_BATCH=$b timeout 300 python tools/bench_configs.py --algo nuts --chains 65536 --reps 1 2>&1 | tail -1 | cut -c1-70; done
