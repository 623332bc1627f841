#!/bin/bash
# round 4, first visit: GPU tests, the bench line, the rollout fuzz around the two seeds of round 3's open item
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.txt 2>&1; tail -5 gpurun_out/gputest.txt
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
(timeout 300 python scripts/gpu_fuzz_rollout.py 14000 1500; timeout 300 python scripts/gpu_fuzz_rollout.py 24000 1500) > gpurun_out/fuzz_rollout.txt 2>&1; tail -4 gpurun_out/fuzz_rollout.txt
bash scripts/ab_solvers.sh > gpurun_out/ab_solvers.txt 2>&1; cat gpurun_out/ab_solvers.txt
