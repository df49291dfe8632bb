cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for s in 7 11 31; do timeout 900 python tools/fuzz_parity.py 150 $s 2>&1 | grep -v "^ok" | tail -12; done
