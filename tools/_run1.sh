cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity_nuts.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
MI_NUTS_PROF=1 timeout 300 python tools/bench_configs.py --algo nuts --chains 65536 --reps 1 2>&1 | tail -12 | cut -c1-100
