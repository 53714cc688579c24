#!/bin/bash
# Round-5 evidence on the GPU box, everything from ONE build: the full GPU test suite, kernel trace + stats and the counter passes of the
# bench command (tools/round_evidence.sh's passes), the per-shape timing, the config lines with the DEFAULT protocol (100 timed steps
# after 20 warm-up; config 3: 40 steps, the cross entropy rises for the first steps after a forced shrink), smoke(), and last the default
# bench line (after the PMC summaries have been written to profiles/ with this build's source digest).
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r05_pytest_full.log
bash tools/profile_round.sh r05 > gpurun_out/evidence_r05.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_r05/MFMA -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_r05_MFMA.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r05/k_kernel_trace.csv > gpurun_out/r05_bench_bs256_kernel_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_r05 > profiles/r05_pmc_traffic.json 2> gpurun_out/r05_pmc_traffic.log
python tools/pmc_mfma.py gpurun_out/pmc_r05/MFMA > profiles/r05_pmc_mfma.json 2> gpurun_out/r05_pmc_mfma.log
cp profiles/r05_pmc_traffic.json profiles/r05_pmc_mfma.json gpurun_out/ 2>/dev/null
DETAIL=1 timeout 300 python tools/bringup.py 256 > gpurun_out/r05_bs256_per_shape_timing.txt 2>&1
timeout 400 python bench.py --model atomnas_a_supernet --no-cpu-baseline > gpurun_out/r05_bench_cfg2_atomnas_a_supernet.json 2> gpurun_out/r05_bench_cfg2.err
timeout 400 python bench.py --shrink 0.3 --steps 40 --warmup 20 --no-cpu-baseline > gpurun_out/r05_bench_cfg3_shrink0.3.json 2> gpurun_out/r05_bench_cfg3.err
timeout 400 python bench.py --model atomnas_c_plus --batch 128 --no-cpu-baseline > gpurun_out/r05_bench_cfg5_atomnas_c_plus_bs128.json 2> gpurun_out/r05_bench_cfg5.err
timeout 400 python bench.py --input-pipeline uint8 --no-cpu-baseline --no-roofline > gpurun_out/r05_bench_input_pipeline_uint8.json 2> gpurun_out/r05_bench_u8.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
cat gpurun_out/r05_pytest_full.log | tail -3; tail -2 gpurun_out/r05_final_bench.err; cut -c1-420 gpurun_out/r05_final_bench.json; tail -2 gpurun_out/r05_smoke.log
for f in cfg2_atomnas_a_supernet cfg3_shrink0.3 cfg5_atomnas_c_plus_bs128 input_pipeline_uint8; do echo "$f: $(grep -o '"value": [0-9.]*' gpurun_out/r05_bench_$f.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05_bench_$f.json | head -1)"; done
