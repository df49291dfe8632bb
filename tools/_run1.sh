set -x
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity_mala.py tests/test_gpu_parity_hmc.py tests/test_gpu_parity_nuts.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/bench_c3.py 65536 10 2>&1 | tail -1
timeout 300 python bench.py 2>&1 | tail -1
timeout 300 python tools/bench_configs.py --algo nuts --chains 65536 --reps 1 2>&1 | tail -1
timeout 300 python tools/bench_c5.py 2>&1 | tail -1
