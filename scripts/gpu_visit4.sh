#!/bin/bash
# round 4, fourth visit: GPU tests and the bench line after the whole-step kernel learnt constant-row and identity tasks
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.txt 2>&1; tail -5 gpurun_out/gputest.txt
timeout 800 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
python scripts/rollout_bench.py > gpurun_out/rollout_bench.txt 2>&1; tail -2 gpurun_out/rollout_bench.txt
