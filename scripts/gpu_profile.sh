#!/bin/bash
# Profiling visit: rocprofv3 kernel stats of the bench, HBM PMC passes, SQ counters, rollout.  Outputs under gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc gpurun_out/prof_rollout
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o r01 -- python scripts/stack_and_solve_once.py > /dev/null 2> gpurun_out/pmc_$c.err
done
bash scripts/pmc_probe.sh > gpurun_out/sq_counters.txt 2>&1
python scripts/rollout_bench.py > gpurun_out/rollout_bench.txt 2>&1; tail -2 gpurun_out/rollout_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rollout -o r01 -- python scripts/rollout_bench.py > /dev/null 2> gpurun_out/prof_rollout.err
ls gpurun_out/prof gpurun_out/pmc_FETCH_SIZE | head
