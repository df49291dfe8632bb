cd /root/repo
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log; tail -c 1500 gpurun_out/bench_line.json
bash tools/profile_bench.sh r1c > gpurun_out/prof_r1c.log 2>&1
bash tools/profile_bench.sh r1c_c3 python tools/bench_c3.py 65536 10 > gpurun_out/prof_r1c_c3.log 2>&1
bash tools/profile_bench.sh r1c_c4 python tools/bench_configs.py --algo nuts --chains 65536 --reps 1 > gpurun_out/prof_r1c_c4.log 2>&1
timeout 300 python tools/bench_c3.py 65536 10 2>&1 | tail -1 | cut -c1-220
timeout 300 python tools/bench_c5.py 2>&1 | tail -1
