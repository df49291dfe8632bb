cd /root/repo
timeout 900 python tools/fuzz_parity.py 150 11 2>&1 | grep -v "^ok" | tail -30
