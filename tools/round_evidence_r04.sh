#!/bin/bash
# Round-4 evidence on the GPU box, everything from ONE build: kernel trace + stats and the counter passes of the bench command
# (tools/round_evidence.sh), the per-shape timing, the SE micro-benchmark and the kernel trace of config 5, smoke(), and last the
# default bench line (after the PMC summaries have been written to profiles/ with this build's source digest).
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
mkdir -p gpurun_out
bash tools/round_evidence.sh r04
python tools/rocprof_summary.py gpurun_out/prof_r04/k_kernel_trace.csv > gpurun_out/r04_bench_bs256_kernel_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_r04 > profiles/r04_pmc_traffic.json 2> gpurun_out/r04_pmc_traffic.log
python tools/pmc_mfma.py gpurun_out/pmc_r04/MFMA > profiles/r04_pmc_mfma.json 2> gpurun_out/r04_pmc_mfma.log
cp profiles/r04_pmc_traffic.json profiles/r04_pmc_mfma.json gpurun_out/ 2>/dev/null
DETAIL=1 timeout 300 python tools/bringup.py 256 > gpurun_out/r04_bs256_per_shape_timing.txt 2>&1
timeout 200 python tools/sebench.py > gpurun_out/r04_sebench.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_cfg5 -o p --output-format csv -- python $R/bench.py --model atomnas_c_plus --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r04_cfg5.log 2>&1)
python tools/kstats.py gpurun_out/prof_r04_cfg5 > gpurun_out/r04_cfg5_kernel_summary.txt 2>&1
python tools/kstats.py gpurun_out/prof_r04_cfg5 k_se_ --launches > gpurun_out/r04_cfg5_se_kernels.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
tail -3 gpurun_out/r04_final_bench.err; cat gpurun_out/r04_final_bench.json | cut -c1-400; tail -2 gpurun_out/r04_smoke.log
