#!/bin/bash
# Refresh of the build-dependent evidence after the last kernel change of the round (the two-group instance of the depthwise backward on
# the matrix cores): the depthwise kernel tests, the kernel trace + PMC passes, the per-shape timing and the default bench line.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py tests/test_parity_gpu.py -q -m gpu -k "dwconv or block or blockwise" 2>&1 | tail -4 > gpurun_out/r05b_pytest.log
rm -rf gpurun_out/prof_r05 gpurun_out/pmc_r05
bash tools/profile_round.sh r05 > gpurun_out/evidence_r05.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_r05/MFMA -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_r05_MFMA.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r05/k_kernel_trace.csv > gpurun_out/r05_bench_bs256_kernel_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_r05 > profiles/r05_pmc_traffic.json 2> gpurun_out/r05_pmc_traffic.log
python tools/pmc_mfma.py gpurun_out/pmc_r05/MFMA > profiles/r05_pmc_mfma.json 2> gpurun_out/r05_pmc_mfma.log
cp profiles/r05_pmc_traffic.json profiles/r05_pmc_mfma.json gpurun_out/ 2>/dev/null
DETAIL=1 timeout 300 python tools/bringup.py 256 > gpurun_out/r05_bs256_per_shape_timing.txt 2>&1
timeout 900 python bench.py > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
cat gpurun_out/r05b_pytest.log | tail -2; tail -1 gpurun_out/r05_final_bench.err; cut -c1-300 gpurun_out/r05_final_bench.json
