#!/usr/bin/env python3
"""Counters of the last dispatch of one kernel in a rocprofv3 --pmc output directory.
usage: pmc_of.py <dir> <kernel-name substring>"""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    t = {}
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if agg:
        k = sorted(agg, key=int)[-1]
        print(round(t[k], 3), "ms", dict(agg[k]))
