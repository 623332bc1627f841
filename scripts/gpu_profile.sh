#!/bin/bash
# Profiling visit (round 4; round 3's plus the A/B of the solve kernels, the stages of the pipelined array call and the fuzz runs): the bench line, rocprofv3 kernel stats of EVERY kernel the bench runs (headline, stack-only,
# UR5, JVRC, whole-step), HBM PMC passes, SQ counters of the headline and of the JVRC-shaped kernel, section clocks,
# host latency, closed-loop rollout.  Outputs under gpurun_out/; scripts/collect_profiles.py r04 copies the summaries
# into profiles/.   Usage (GPU box):  bash scripts/gpu_profile.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/prof_all gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc gpurun_out/pmc_jvrc gpurun_out/prof_rollout
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err
# kernel stats of the bench command's timed kernel alone (the average the roofline of the bench line is checked against) ...
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
# ... and of the whole bench minus the CPU legs: every configuration's kernels (the headline kernel's row then mixes the
# input regimes it is run on)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_all -o r01 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_all.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o r01 -- python scripts/stack_and_solve_once.py > /dev/null 2> gpurun_out/pmc_$c.err
done
bash scripts/pmc_probe.sh gpurun_out/pmc > gpurun_out/sq_counters.txt 2>&1
bash scripts/pmc_probe.sh gpurun_out/pmc_jvrc --config jvrc > gpurun_out/sq_counters_jvrc.txt 2>&1
python scripts/section_clock.py > gpurun_out/section_clock.txt 2>&1
python scripts/section_clock.py draco3 kinematic >> gpurun_out/section_clock.txt 2>&1
python scripts/section_clock.py draco3 tracking >> gpurun_out/section_clock.txt 2>&1
CLOCK_W=64 PINKHIP_CLOCK_LIBRARY=$PWD/pink_amd/csrc/libpinkhip_clock_jvrc.so python scripts/section_clock.py jvrc >> gpurun_out/section_clock.txt 2>&1
PINKHIP_CLOCK_LIBRARY=$PWD/pink_amd/csrc/libpinkhip_clock_draco3b.so python scripts/section_clock.py draco3b >> gpurun_out/section_clock.txt 2>&1
bash scripts/pmc_probe.sh gpurun_out/pmc_draco3b --config draco3b > gpurun_out/sq_counters_draco3b.txt 2>&1
python scripts/host_latency.py > gpurun_out/host_latency.txt 2>&1
python scripts/rollout_bench.py > gpurun_out/rollout_bench.txt 2>&1; tail -2 gpurun_out/rollout_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rollout -o r01 -- python scripts/rollout_bench.py > /dev/null 2> gpurun_out/prof_rollout.err
bash scripts/ab_solvers.sh > gpurun_out/ab_solvers.txt 2>&1
python scripts/prof_pipeline.py > gpurun_out/prof_pipeline.txt 2>&1
(python scripts/gpu_fuzz.py 300000 12000) > gpurun_out/fuzz.txt 2>&1
(python scripts/gpu_fuzz_rollout.py 14000 1500; python scripts/gpu_fuzz_rollout.py 24000 1500; python scripts/gpu_fuzz_rollout.py 40000 3000; FUZZ_EXTRAS=1 python scripts/gpu_fuzz_rollout.py 50000 6000) > gpurun_out/fuzz_rollout.txt 2>&1
ls gpurun_out/prof gpurun_out/pmc_FETCH_SIZE | head
