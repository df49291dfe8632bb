cd /root/repo
for sk in 0 1; do timeout 120 tools/bin/lb_sk$sk 65536 20 0; timeout 120 tools/bin/lb_sk$sk 65536 5 1 4; done
