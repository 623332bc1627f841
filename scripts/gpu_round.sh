#!/bin/bash
# One GPU visit: gpu tests, bench line, rocprof kernel stats.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
ls -R gpurun_out/prof | head -30
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o r01 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> gpurun_out/pmc_$c.err
done
ls -R gpurun_out/pmc_FETCH_SIZE | head
# closed loop on the device: bench line + kernel stats
python scripts/rollout_bench.py > gpurun_out/rollout_bench.txt 2>&1; tail -2 gpurun_out/rollout_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rollout -o r01 -- python scripts/rollout_bench.py > /dev/null 2> gpurun_out/prof_rollout.err
