#!/usr/bin/env python3
"""Copies the summaries of one tools/profile_r3.sh run (gpurun_out/prof_r3) into profiles/ under this
round's names, and writes the two small files bench.py quotes counters from."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r3")
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "r3_kernel_stats.csv"))
line = open(os.path.join(src, "trace_bench.json")).read().strip().split("\n")[-1]
json.dump(json.loads(line), open(os.path.join(dst, "r3_bench_under_rocprof.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "trace_c2", "trace_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "trace_c2", "trace_kernel_stats.csv"), os.path.join(dst, "r3_c2_kernel_stats.csv"))
    line = open(os.path.join(src, "trace_c2_bench.json")).read().strip().split("\n")[-1]
    json.dump(json.loads(line), open(os.path.join(dst, "r3_c2_bench_under_rocprof.json"), "w"), indent=1)
summary = json.load(open(os.path.join(src, "summary.json")))
json.dump(summary, open(os.path.join(dst, "r3_pmc_summary.json"), "w"), indent=1)
k = summary["kernels"]
for key, name in (("residual", "r3_pmc_residual_kernel.json"), ("split_chain", "r3_pmc_split_chain_kernel.json"),
                  ("separable", "r3_pmc_separable_kernel.json"), ("inlier", "r3_pmc_inlier_kernel.json"),
                  ("exact_count", "r3_pmc_exact_count_kernel.json"), ("moments", "r3_pmc_moments_kernel.json"),
                  ("point_error", "r3_pmc_point_error_kernel.json")):
    if key in k:
        rec = dict(k[key])
        rec["command"] = summary["command"]
        json.dump(rec, open(os.path.join(dst, name), "w"), indent=1)
for tag, name in (("ingest_vox", "t_kernel_stats.csv"), ("c5_trace", "trace_kernel_stats.csv")):
    f = os.path.join(src, tag, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "r3_%s_kernel_stats.csv" % tag))
log = os.path.join(src, "c5_trace.log")
if os.path.exists(log):
    shutil.copy(log, os.path.join(dst, "r3_c5_kernels_library_timings.txt"))
print("published to", dst)
