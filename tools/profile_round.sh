#!/bin/bash
# Round profile of the bench command: kernel trace + stats (csv), then the HBM-traffic counters in their own passes.
# usage (on the GPU box): tools/profile_round.sh r01   -> gpurun_out/prof_<tag>/..., summaries printed by the tools afterwards
TAG=${1:-rXX}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o k -- python $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$TAG/$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
done
ls $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_$TAG/*
