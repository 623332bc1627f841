cd /root/repo
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/gputest_f.txt 2>&1; grep -E "passed|failed" gpurun_out/gputest_f.txt | tail -2
( time timeout 900 python scripts/gpu_fuzz.py 1100000 100000 ) > gpurun_out/fuzz_wide3_r05.txt 2>&1; grep -v "^$" gpurun_out/fuzz_wide3_r05.txt | tail -12
( time FUZZ_EXTRAS=1 timeout 300 python scripts/gpu_fuzz_rollout.py 3000000 15000 ) > gpurun_out/fuzz_rollout_wide3_r05.txt 2>&1; tail -4 gpurun_out/fuzz_rollout_wide3_r05.txt
