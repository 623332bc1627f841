#!/bin/bash
# round 4, final visit: GPU tests, smoke, the fuzz with the round-4 task classes, then the whole profiling pass
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.txt 2>&1; tail -3 gpurun_out/gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
(FUZZ_EXTRAS=1 python scripts/gpu_fuzz_rollout.py 90000 6000) > gpurun_out/fuzz_rollout_extras.txt 2>&1; tail -2 gpurun_out/fuzz_rollout_extras.txt
bash scripts/gpu_profile.sh
