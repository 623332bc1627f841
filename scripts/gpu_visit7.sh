#!/bin/bash
# round 4, seventh visit: GPU tests, smoke, the random-robot fuzz with the round-4 task classes, bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.txt 2>&1; tail -3 gpurun_out/gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
(FUZZ_EXTRAS=1 python scripts/gpu_fuzz_rollout.py 50000 6000; FUZZ_EXTRAS=1 python scripts/gpu_fuzz_rollout.py 90000 6000) > gpurun_out/fuzz_rollout_extras.txt 2>&1; tail -4 gpurun_out/fuzz_rollout_extras.txt
timeout 800 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
