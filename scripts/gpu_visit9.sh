#!/bin/bash
# round 4, last visit (after the certificate learnt the sign of the active rows' multipliers): the parity tests of the
# solve kernels, the fuzz range that found it, the PMC / SQ / kernel-stat passes of the headline kernel, then the bench line
# with the traffic of THESE kernel sources (collect_profiles.py runs here first: it writes profiles/traffic_r04.json)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_both_kernels.py -m gpu -q -k "fuzz or parity_suite or both" > gpurun_out/gputest_parity.txt 2>&1; tail -2 gpurun_out/gputest_parity.txt
PINKHIP_FUZZ_KERNELS=sweep python scripts/gpu_fuzz.py 500000 40000 > gpurun_out/fuzz_wide_after.txt 2>&1; tail -3 gpurun_out/fuzz_wide_after.txt
rm -rf gpurun_out/prof gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o r01 -- python scripts/stack_and_solve_once.py > /dev/null 2> gpurun_out/pmc_$c.err
done
bash scripts/pmc_probe.sh gpurun_out/pmc > gpurun_out/sq_counters.txt 2>&1
python scripts/collect_profiles.py r04 > /dev/null 2>&1
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err
