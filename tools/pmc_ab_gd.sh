#!/bin/bash
# VALU instruction counts and wave cycles of the exact path's kernels in the first 40 iterations of the
# benchmark's descent, for two builds of the library (same box): tools/pmc_ab_gd.sh <lib A> <lib B>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ab_gd
mkdir -p $OUT
for lib in "$@"; do
  export PROBE_LIB=cameracalibration_amd/$lib
  export PROBE_MAXIT=40
  timeout -s KILL 120 rocprofv3 --pmc ${PMC_COUNTERS:-SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU} --output-format csv -d $OUT/$lib -o pmc -- python3 tools/probe_gdfull.py 20 300000 > $OUT/$lib.log 2> $OUT/$lib.err || { echo "pmc failed for $lib"; tail -3 $OUT/$lib.err; exit 1; }
  python3 - "$OUT/$lib/pmc_counter_collection.csv" "$lib" <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id']) not in seen:
        seen.add(r['Dispatch_Id']); n[k] += 1
for k in ('split_chain_kernel', 'split_finish_kernel', 'screen_cost_kernel'):
    if k in agg: print(sys.argv[2], k, 'launches', n[k], {c: round(v / n[k]) for c, v in agg[k].items()})
PY
done
