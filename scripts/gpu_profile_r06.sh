#!/bin/bash
# Profiling visit of round 6: the bench line + detail, rocprofv3 kernel stats of the timed kernel alone, of every kernel the
# bench runs and of the stack-only kernel alone (warmed), HBM PMC passes (separate --pmc runs, kernel-trace only), SQ counters
# of the headline, the JVRC-shaped, the nv = 30 + 6 rows and the nv = 33 kernels, section clocks, the closed-loop bench,
# the array API.  Outputs under gpurun_out/; `python scripts/collect_profiles.py r06` copies the summaries into profiles/.
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/prof_all gpurun_out/prof_stack gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc gpurun_out/pmc_jvrc gpurun_out/pmc_draco3b gpurun_out/pmc_nv33 gpurun_out/prof_rollout
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; cp gpurun_out/bench_detail.json gpurun_out/bench_detail_full.json; tail -c 300 gpurun_out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_all -o r01 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_all.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stack -o r01 -- python scripts/stack_only_warm.py > gpurun_out/stack_only_warm.txt 2> gpurun_out/prof_stack.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o r01 -- python scripts/stack_and_solve_once.py > /dev/null 2> gpurun_out/pmc_$c.err
done
bash scripts/pmc_probe.sh gpurun_out/pmc > gpurun_out/sq_counters.txt 2>&1
bash scripts/pmc_probe.sh gpurun_out/pmc_jvrc --config jvrc > gpurun_out/sq_counters_jvrc.txt 2>&1
bash scripts/pmc_probe.sh gpurun_out/pmc_draco3b --config draco3b > gpurun_out/sq_counters_draco3b.txt 2>&1
bash scripts/pmc_probe.sh gpurun_out/pmc_nv33 --config draco3_freeflyer > gpurun_out/sq_counters_nv33.txt 2>&1
if [ -f pink_amd/csrc/libpinkhip_clock.so ]; then
  (python scripts/section_clock.py draco3 tight; python scripts/section_clock.py draco3 kinematic; python scripts/section_clock.py draco3 tracking) > gpurun_out/section_clock.txt 2>&1
fi
python scripts/rollout_bench.py > gpurun_out/rollout_bench.txt 2>&1; tail -2 gpurun_out/rollout_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rollout -o r01 -- python scripts/rollout_bench.py > /dev/null 2> gpurun_out/prof_rollout.err
python scripts/prof_pipeline.py > gpurun_out/prof_pipeline.txt 2>&1
python scripts/ab_api_arrays.py > gpurun_out/ab_api_arrays.txt 2>&1
python scripts/host_latency.py > gpurun_out/host_latency.txt 2>&1
ls gpurun_out/prof gpurun_out/pmc_FETCH_SIZE | head
