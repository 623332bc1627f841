"""Copy the rocprofv3 summaries of the last GPU visit from gpurun_out/ (scratch)
into profiles/ (tracked) and derive the per-launch HBM traffic of each kernel.

FETCH_SIZE / WRITE_SIZE are reported in KiB-ish units of 1 KB per the rocprofv3
counter definition; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
reads exactly half the bytes of a coalesced streaming read, so it is doubled.
Both counters were collected in separate --pmc passes.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "prof", "r01_kernel_stats.csv"), os.path.join(dst, f"kernel_stats_{tag}.csv"))
if os.path.exists(os.path.join(src, "prof_all", "r01_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "prof_all", "r01_kernel_stats.csv"), os.path.join(dst, f"kernel_stats_all_configs_{tag}.csv"))
if os.path.exists(os.path.join(src, "prof_rollout", "r01_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "prof_rollout", "r01_kernel_stats.csv"), os.path.join(dst, f"kernel_stats_rollout_{tag}.csv"))
if os.path.exists(os.path.join(src, "rollout_bench.json")):
    shutil.copy(os.path.join(src, "rollout_bench.json"), os.path.join(dst, f"rollout_bench_{tag}.json"))
if os.path.exists(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"bench_{tag}.json"))
if os.path.exists(os.path.join(src, "bench_detail_full.json")):  # (round 5: the stdout line is a bounded headline, this is the full record;
    # saved by gpu_profile_r05.sh right after the bench -- the profiling runs of bench.py that follow overwrite bench_detail.json)
    shutil.copy(os.path.join(src, "bench_detail_full.json"), os.path.join(dst, f"bench_detail_{tag}.json"))
if os.path.exists(os.path.join(src, "prof_stack", "r01_kernel_stats.csv")):  # (round 6: the stack-only kernel alone, warmed)
    shutil.copy(os.path.join(src, "prof_stack", "r01_kernel_stats.csv"), os.path.join(dst, f"kernel_stats_stack_{tag}.csv"))
for name in ("sq_counters.txt", "sq_counters_jvrc.txt", "sq_counters_draco3b.txt", "sq_counters_nv33.txt", "host_latency.txt", "ab_solvers.txt", "prof_pipeline.txt",
             "fuzz.txt", "fuzz_rollout.txt", "ab_api_arrays.txt", "section_clock.txt", "stack_only_warm.txt"):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
        shutil.copy(os.path.join(src, name), os.path.join(dst, name.replace(".txt", f"_{tag}.txt")))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

out = {"units": "bytes per kernel launch",
       "fetch_correction": "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section: exact for 16 B/lane streaming reads, other widths "
                           "to be calibrated on a known byte count).  Calibration for the 8 B/lane row loads of these kernels: "
                           "ik_stack_mfma_kernel reads a known 8 (Kd nv + K) B per QP -- compare stack_kernel_hbm_read_bytes_per_launch "
                           "with stack_kernel_algorithmic_read_bytes below",
       "stack_kernel_algorithmic_read_bytes": 8 * (24 * 30 + 48) * 65536,
       "stack_kernel_algorithmic_write_bytes": 8 * (30 * 30 + 30) * 65536,
       "solve_kernel_algorithmic_bytes": 6868 * 65536,
       "workload": "bench.py default (draco3, B = 65536, tight bounds)",
       "source_hash": g._source_hash(g.HIP_DEPS)}
per = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(src, f"pmc_{c}", "r01_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        per[k][c] = sum(v) / len(v) * 1024.0
    with open(os.path.join(dst, f"pmc_{c}_{tag}.csv"), "w") as f:
        f.write("kernel,launches,avg_counter_value_KB\n")
        for k, v in agg.items():
            f.write(f"\"{k}\",{len(v)},{sum(v)/len(v):.3f}\n")
for k, d in per.items():
    if "ik_solve" in k or "ik_stack" in k:
        name = "solve_kernel" if "ik_solve" in k else "stack_kernel"
        rd = 2.0 * d.get("FETCH_SIZE", 0.0)
        wr = d.get("WRITE_SIZE", 0.0)
        out[f"{name}_hbm_read_bytes_per_launch"] = rd
        out[f"{name}_hbm_write_bytes_per_launch"] = wr
        out[f"{name}_hbm_bytes_per_launch"] = rd + wr
# SQ counters of the headline kernel (scripts/gpu_profile.sh: "p<pass> NAME launches average-per-launch"): the VALU
# instruction count per launch is what bench.py prices against the issue rate of the SIMDs (roofline_valu_issue)
sq = os.path.join(src, "sq_counters.txt")
if os.path.exists(sq):
    vals = {}
    for ln in open(sq):
        f = ln.split()
        if len(f) == 4 and f[0].startswith("p") and f[1].startswith("SQ_"):
            vals[f[1]] = float(f[3])
    for key, name in (("SQ_INSTS_VALU", "solve_kernel_valu_instructions_per_launch"), ("SQ_WAVES", "solve_kernel_waves_per_launch"),
                      ("SQ_ACTIVE_INST_VALU", "solve_kernel_valu_active_quad_cycles_per_launch"),
                      ("SQ_INSTS_VALU_FMA_F64", "solve_kernel_fma_f64_instructions_per_launch"),
                      ("SQ_INSTS_SALU", "solve_kernel_salu_instructions_per_launch"),
                      ("SQ_INSTS_LDS", "solve_kernel_lds_instructions_per_launch")):
        if key in vals:
            out[name] = vals[key]
json.dump(out, open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
