#!/bin/bash
# Short GPU visit: gpu tests + bench lines of the three configs.  Outputs under gpurun_out/.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for cfg in draco3 jvrc ur5; do
  B=65536; [ $cfg = ur5 ] && B=4096
  python bench.py --steps 20 --warmup 3 --config $cfg --batch $B --no-cpu-baseline --headline-only > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err
  tail -2 gpurun_out/bench_$cfg.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$cfg.json"))
    print("$cfg", "value=%.3g" % d["value"], "kernel_ms=%.4f" % d["roofline"]["kernel_ms"], "stack_ms=%.4f" % d["stack_only"]["kernel_ms"], d["solver_stats"])
except Exception as e:
    print("$cfg bench failed", e)
PY
done
