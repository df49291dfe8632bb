cd /root/repo
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/fuzz_parity.py 30 5 | tail -3
