#!/bin/bash
# N > 1 control flow on a 1-GPU box: two ranks share device 0 (RCCL refuses duplicate devices -> the bench falls back
# to the TCP gather and says so); everything else -- rendezvous, per-rank handles, barriers, max-over-ranks timing,
# strong / weak modes -- is the real path.
for mode in "--scaling weak --batch 32768" "--scaling strong --global-batch 65536"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --headline-only --no-cpu-baseline $mode 2> gpurun_out/two_rank.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','n_gpus','scaling','ms_per_step','per_rank_kernel_ms','gather','comm_note')}, d['config']['global_batch'], d['solver_stats'])"
  tail -3 gpurun_out/two_rank.err | grep -v "amdgpu.ids"
done
