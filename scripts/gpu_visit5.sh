#!/bin/bash
# round 4, fifth visit: where the launch goes in the tracking regime (section clock), the split of the pipelined array call
export TMPDIR=/tmp
mkdir -p gpurun_out
python scripts/section_clock.py draco3 tight > gpurun_out/section_clock_regimes.txt 2>&1
python scripts/section_clock.py draco3 kinematic >> gpurun_out/section_clock_regimes.txt 2>&1
python scripts/section_clock.py draco3 tracking >> gpurun_out/section_clock_regimes.txt 2>&1
cat gpurun_out/section_clock_regimes.txt
python scripts/prof_pipeline.py > gpurun_out/prof_pipeline.txt 2>&1; head -4 gpurun_out/prof_pipeline.txt; grep ranges gpurun_out/prof_pipeline.txt
