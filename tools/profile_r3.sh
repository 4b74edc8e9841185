#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (r3_*): one kernel-trace/stats run of the default
# benchmark and PMC passes (each counter group in a run of its own, never together with a trace option)
# for the residual kernel, the separable histogram kernel, the exact path's kernels and -- from
# tools/probe_c5_kernels.py -- the RANSAC, moment and point-report kernels.  Run through gpurun from
# the repo root; summaries are written by tools/summarize_pmc.py (which stamps every record with a
# hash of the kernel's sources) and published by tools/publish_profiles.py.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3
mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --cpu-candidates 0"
SHORT="python3 bench.py --steps 1 --warmup 1 --cpu-candidates 0 --f32-steps 1 --north-star 0"
C5="python3 tools/probe_c5_kernels.py 100 1000000"
echo "[profile] kernel trace of the default benchmark"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
echo "[profile] kernel trace of the benchmark without its north-star leg (per-kernel averages of the C2 workload alone)"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -o trace -- python3 bench.py --steps 3 --warmup 1 --cpu-candidates 0 --north-star 0 > $OUT/trace_c2_bench.json 2> $OUT/trace_c2.err || { echo "c2 trace failed"; tail -5 $OUT/trace_c2.err; exit 1; }
echo "[profile] kernel trace of the plane-fit / point-report kernels"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace -o trace -- $C5 > $OUT/c5_trace.log 2> $OUT/c5_trace.err || { echo "c5 trace failed"; tail -5 $OUT/c5_trace.err; exit 1; }
pass() {   # tag, program..., then "--", counters...
  local tag=$1; shift
  local prog=()
  while [ "$1" != "--" ]; do prog+=("$1"); shift; done
  shift
  echo "[profile] pmc $tag: $*"
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$tag -o pmc -- "${prog[@]}" > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $OUT/pmc_$tag.err; return 1; }
}
pass fetch $SHORT -- FETCH_SIZE || exit 1
pass write $SHORT -- WRITE_SIZE || exit 1
pass issue $SHORT -- SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU || exit 1
pass clock $SHORT -- GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum || exit 1
pass c5fetch $C5 -- FETCH_SIZE || exit 1
pass c5write $C5 -- WRITE_SIZE || exit 1
pass c5issue $C5 -- SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU || exit 1
pass c5clock $C5 -- GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum || exit 1
python3 tools/summarize_pmc.py $OUT > $OUT/summary.json && echo "[profile] summary written"
echo "[profile] ingest kernels at 100 clouds x 1e6 points (voxel grid)"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ingest_vox -o t -- python3 tools/probe_voxel.py 100 1000000 > $OUT/ingest_vox.log 2> $OUT/ingest_vox.err || { echo "ingest voxel trace failed"; exit 1; }
# the per-dispatch traces are large and nothing reads them afterwards: the statistics files stay
find $OUT -name "*_kernel_trace.csv" -delete
echo "[profile] done"
