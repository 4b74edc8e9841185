#!/bin/bash
# Collects the rocprofv3 evidence for the residual kernel: kernel-trace stats and PMC passes
# (each counter group in a run of its own).  Run through gpurun from the repo root.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KERNEL=${KERNEL:-bf16x3}
OUT=gpurun_out/prof_$KERNEL
mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --cpu-candidates 0 --gd-iterations 0 --kernel $KERNEL"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -o pmc -- $BENCH > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $OUT/pmc_$tag.err; exit 1; }
done
find $OUT -name "*.csv" | head -40
