cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-330
timeout 300 python tools/bench_c3.py 65536 10 2>&1 | tail -1 | cut -c1-220
timeout 300 python tools/bench_configs.py --algo nuts --chains 65536 --reps 1 2>&1 | tail -1 | cut -c1-100
timeout 300 python tools/bench_c5.py 2>&1 | tail -1
