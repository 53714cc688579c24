#!/bin/bash
# HBM-traffic counters of the bench step, one counter per pass (gpurun refuses --pmc together with tracing domains other than kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_bench/$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_bench_$C.log 2>&1
done
find $R/gpurun_out/pmc_bench -name "*counter_collection.csv" | head
