#!/usr/bin/env python3
"""Copies the summaries of one tools/profile_r2.sh run (gpurun_out/prof_r2) into profiles/ under this
round's names, and writes the two small files bench.py quotes counters from."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r2")
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "r2_kernel_stats.csv"))
line = open(os.path.join(src, "trace_bench.json")).read().strip().split("\n")[-1]
json.dump(json.loads(line), open(os.path.join(dst, "r2_bench_under_rocprof.json"), "w"), indent=1)
summary = json.load(open(os.path.join(src, "summary.json")))
json.dump(summary, open(os.path.join(dst, "r2_pmc_summary.json"), "w"), indent=1)
k = summary["kernels"]
for key, name in (("residual", "r2_pmc_residual_kernel.json"), ("split_chain", "r2_pmc_split_chain_kernel.json"),
                  ("separable", "r2_pmc_separable_kernel.json")):
    if key in k:
        rec = dict(k[key])
        rec["command"] = summary["command"]
        json.dump(rec, open(os.path.join(dst, name), "w"), indent=1)
for tag in ("ingest_vox", "ingest_fit"):
    f = os.path.join(src, tag, "t_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "r2_%s_kernel_stats.csv" % tag))
print("published to", dst)
