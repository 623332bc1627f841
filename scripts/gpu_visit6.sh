#!/bin/bash
# round 4, sixth visit: GPU tests, bench line, stages of the pipelined array call after the early-request / lazy-stats changes
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.txt 2>&1; tail -5 gpurun_out/gputest.txt
timeout 800 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
python scripts/prof_pipeline.py > gpurun_out/prof_pipeline.txt 2>&1; head -4 gpurun_out/prof_pipeline.txt; grep ranges gpurun_out/prof_pipeline.txt
