#!/usr/bin/env python3
"""Per-call time of `solve_ik_batch(ConfigurationBatch, ...)` at the headline shape (bench.py's api_level_arrays) for a few
shapes of the pipelined call: target arrays uploaded with every call / frozen on the device, range splits, stacks with
dense rows (constraints=, barriers).  GPU box:  python scripts/ab_api_arrays.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pink_amd  # noqa: E402
from pink_amd import rollout  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    rows = []
    only = os.environ.get("AB_ONLY")  # substring of the labels to run
    for label, kw, split in (("pageable", dict(), "1,1,1,1"),
                             ("page-locked", dict(pinned=True), "1,1,1,1"),
                             ("page-locked, translation + quaternion targets", dict(pinned=True, quat=True), "1,1,1,1"),
                             ("page-locked, frozen targets, 4 ranges", dict(pinned=True, freeze=True), "1,1,1,1"),
                             ("page-locked, frozen targets, 3 ranges", dict(pinned=True, freeze=True), "1,1,1"),
                             ("page-locked, frozen targets, 2 ranges", dict(pinned=True, freeze=True), "1,1"),
                             ("page-locked, frozen targets, 1 range", dict(pinned=True, freeze=True), "1"),
                             ("page-locked, frozen targets, 5 ranges", dict(pinned=True, freeze=True), "1,1,1,1,1"),
                             ("page-locked, frozen targets, 6 ranges", dict(pinned=True, freeze=True), "1,1,1,1,1,1"),
                             ("page-locked + constraints=[FrameTask]", dict(pinned=True, extra_task="constraint"), "1,1,1,1"),
                             ("page-locked + spherical + position barrier", dict(pinned=True, extra_task="barriers"), "1,1,1,1"),
                             ("page-locked, frozen + constraints=[FrameTask]", dict(pinned=True, freeze=True, extra_task="constraint"), "1,1,1,1")):
        if only and only not in label:
            continue
        rollout.PIPELINE_SPLIT = tuple(float(v) for v in split.split(","))
        pink_amd.clear_device_cache()
        r = bench.api_level_arrays(B, **kw)
        rows.append((label, r))
        print(f"{label:52s} route {r['route']:8s} {r['ms_per_call']:7.3f} ms per call (best {r['ms_per_call_best']:7.3f})  "
              f"{r['solves_per_s'] / 1e6:6.1f} M solves/s  bytes in {r['bytes_in_per_call'] / 1e6:6.1f} MB  diff {r['max_abs_velocity_difference_vs_host_evaluated_tasks_on_sample']:.1e}",
              flush=True)


if __name__ == "__main__":
    main()
