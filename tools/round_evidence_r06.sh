#!/bin/bash
# Round-6 evidence on the GPU box, everything from ONE build: the full GPU test suite, kernel trace + stats and the counter passes of the
# bench command (tools/profile_round.sh), idle time inside the replayed step (tools/graph_gaps.py), the per-shape timing, the lines of the
# other BASELINE configurations with the DEFAULT protocol (100 timed steps after 20 warm-up; config 3 = AtomNAS-A, forced 30 % shrink,
# --allow-untrained: the cross entropy on the fixed batch rises for the first steps after the shrink), the supernet fed by the GPU input
# pipeline, smoke(), and last the default bench line (after the PMC summaries have been written to profiles/ with this build's digest).
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
T=r06
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/${T}_pytest_full.log
bash tools/profile_round.sh $T > gpurun_out/evidence_$T.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_$T/MFMA -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_${T}_MFMA.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$T/k_kernel_trace.csv > gpurun_out/${T}_bench_bs256_kernel_summary.txt 2>&1
cp $(find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_bench_bs256_kernel_stats.csv 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$T -o k -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
python tools/graph_gaps.py $(dirname $(find gpurun_out/gaps_$T -name "*kernel_trace.csv" | head -1)) 3 > gpurun_out/${T}_graph_gaps.txt 2>&1
rm -rf gpurun_out/gaps_$T
python tools/pmc_traffic.py gpurun_out/pmc_$T > profiles/${T}_pmc_traffic.json 2> gpurun_out/${T}_pmc_traffic.log
python tools/pmc_mfma.py gpurun_out/pmc_$T/MFMA > profiles/${T}_pmc_mfma.json 2> gpurun_out/${T}_pmc_mfma.log
cp profiles/${T}_pmc_traffic.json profiles/${T}_pmc_mfma.json gpurun_out/ 2>/dev/null
DETAIL=1 timeout 300 python tools/bringup.py 256 > gpurun_out/${T}_bs256_per_shape_timing.txt 2>&1
timeout 400 python bench.py --model atomnas_a_supernet --no-cpu-baseline > gpurun_out/${T}_bench_cfg2_atomnas_a_supernet.json 2> gpurun_out/${T}_bench_cfg2.err
timeout 400 python bench.py --model atomnas_a_supernet --shrink 0.3 --allow-untrained --no-cpu-baseline > gpurun_out/${T}_bench_cfg3_atomnas_a_shrink0.3.json 2> gpurun_out/${T}_bench_cfg3.err
timeout 400 python bench.py --model atomnas_c_plus --batch 128 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5_atomnas_c_plus_bs128.json 2> gpurun_out/${T}_bench_cfg5.err
timeout 400 python bench.py --input-pipeline uint8 --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench_input_pipeline_uint8.json 2> gpurun_out/${T}_bench_u8.err
timeout 400 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench_resident_same_call.json 2> gpurun_out/${T}_bench_res.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/${T}_final_bench.json 2> gpurun_out/${T}_final_bench.err
rm -rf gpurun_out/prof_$T/*/*.db gpurun_out/pmc_$T 2>/dev/null   # raw traces stay on the box (the merge-back limit is 64 MiB)
find gpurun_out/prof_$T -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
cat gpurun_out/${T}_pytest_full.log | tail -3; tail -2 gpurun_out/${T}_final_bench.err; cut -c1-420 gpurun_out/${T}_final_bench.json; tail -2 gpurun_out/${T}_smoke.log
for f in cfg2_atomnas_a_supernet cfg3_atomnas_a_shrink0.3 cfg5_atomnas_c_plus_bs128 input_pipeline_uint8 resident_same_call; do echo "$f: $(grep -o '"value": [0-9.]*' gpurun_out/${T}_bench_$f.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${T}_bench_$f.json | head -1)"; done
cat gpurun_out/${T}_graph_gaps.txt | head -4
