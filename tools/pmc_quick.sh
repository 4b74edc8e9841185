#!/bin/bash
# Two PMC passes (issue counters, clock/L2 counters) of a short bench run for one kernel.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KERNEL=${KERNEL:-bf16x3}
OUT=gpurun_out/pmcq_$KERNEL
mkdir -p $OUT
BENCH="python3 bench.py --steps 1 --warmup 0 --cpu-candidates 0 --gd-iterations 0 --kernel $KERNEL"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/a -o pmc -- $BENCH > $OUT/a.json 2> $OUT/a.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o pmc -- $BENCH > $OUT/b.json 2> $OUT/b.err || exit 1
python3 - <<PY
import csv,collections,glob
for f in sorted(glob.glob('$OUT/*/pmc_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); t={}
    for r in csv.DictReader(open(f)):
        if 'residual' in r['Kernel_Name']:
            agg[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value']); t[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    k=sorted(agg)[-1]
    print(f.split('/')[-2], round(t[k],2),'ms', dict(agg[k]))
PY
