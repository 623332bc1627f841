cd /root/repo
( time timeout 1400 python scripts/gpu_fuzz.py 900000 120000 ) > gpurun_out/fuzz_wide2_r05.txt 2>&1
( time FUZZ_EXTRAS=2 timeout 400 python scripts/gpu_fuzz_rollout.py 2000000 20000 ) > gpurun_out/fuzz_rollout_wide2_r05.txt 2>&1
tail -4 gpurun_out/fuzz_wide2_r05.txt; tail -4 gpurun_out/fuzz_rollout_wide2_r05.txt
