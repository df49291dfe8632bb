cd /root/repo
timeout 300 python tools/_repro.py 2>&1 | tail -25
