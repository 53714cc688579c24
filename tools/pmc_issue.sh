#!/bin/bash
# Instruction-issue counters of the bench step per kernel (two passes; kernel-trace only beside --pmc): which kernels are bound by the
# instructions their waves issue rather than by bytes.  Summary: tools/pmc_issue.py gpurun_out/pmc_issue
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_issue
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_issue/A -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_issue_A.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_issue/B -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_issue_B.log 2>&1
python $R/tools/pmc_issue.py $R/gpurun_out/pmc_issue > $R/gpurun_out/pmc_issue_summary.txt 2>&1
find $R/gpurun_out/pmc_issue -name "*.csv" -size +8M -delete
tail -5 $R/gpurun_out/pmc_issue_A.log; head -50 $R/gpurun_out/pmc_issue_summary.txt
