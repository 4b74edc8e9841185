#!/bin/bash
# Samples clocks and socket power with rocm-smi while a bench run of one kernel is in flight.
KERNEL=${KERNEL:-bf16x3}
python3 bench.py --steps 30 --warmup 1 --cpu-candidates 0 --gd-iterations 0 --kernel $KERNEL > gpurun_out/power_$KERNEL.json 2>&1 &
BPID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|\(W\)" | tr '\n' ' '; echo
  sleep 1
done
wait $BPID
tail -1 gpurun_out/power_$KERNEL.json | cut -c1-160
