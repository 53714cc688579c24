#!/bin/bash
# Round evidence on the GPU box: kernel trace + stats, HBM-traffic passes, MFMA-counter pass, and the bench lines of the other
# configurations.  usage: tools/round_evidence.sh r03     (outputs under gpurun_out/; summaries are made afterwards by
# tools/rocprof_summary.py, tools/pmc_traffic.py, tools/pmc_mfma.py and copied into profiles/)
TAG=${1:-rXX}
R=$(cd "$(dirname "$0")/.." && pwd)
bash $R/tools/profile_round.sh $TAG > $R/gpurun_out/evidence_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_$TAG/MFMA -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_${TAG}_MFMA.log 2>&1
cd $R
timeout 300 python bench.py --model atomnas_a_supernet --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_cfg2_atomnas_a_supernet.json 2> gpurun_out/bench_${TAG}_cfg2.err
timeout 300 python bench.py --shrink 0.3 --steps 40 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_cfg3_shrink0.3.json 2> gpurun_out/bench_${TAG}_cfg3.err
timeout 300 python bench.py --model atomnas_c_plus --batch 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_cfg5_atomnas_c_plus_bs128.json 2> gpurun_out/bench_${TAG}_cfg5.err
ls -la gpurun_out | tail -20
