cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for b in 4 16; do echo "batch $b"; MI_NUTS_BATCH=$b timeout 300 python tools/bench_configs.py --algo nuts --chains 65536 --reps 1 2>&1 | tail -1 | cut -c1-70; done
timeout 300 python tools/bench_configs.py --algo nuts --chains 65536 --reps 2 2>&1 | tail -1 | cut -c1-200
