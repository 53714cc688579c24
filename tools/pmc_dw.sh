#!/bin/bash
# SQ counters of the depthwise kernels on three shapes (one pass per counter set; kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export DWBENCH_CASES="56,144,3,1;56,144,7,1;14,480,7,1;7,1152,7,1" DWBENCH_ITERS=4 DWBENCH_SETS=2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_dw2/sq -o sq -- python $R/tools/dwbench.py both 256 slab > $R/gpurun_out/pmc_dw2_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_dw2/sq2 -o sq2 -- python $R/tools/dwbench.py both 256 slab > $R/gpurun_out/pmc_dw2_sq2.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_dw2 dwconv > $R/gpurun_out/pmc_dw2_summary.txt 2>&1
