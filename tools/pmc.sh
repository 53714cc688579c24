#!/bin/bash
# usage: tools/pmc.sh <which> ; collects SQ and TCC counters in separate passes (no tracing domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
W=$1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d /root/repo/gpurun_out/pmc_$W/sq -o sq -- python /root/repo/tools/tnbench.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_WAVES --output-format csv -d /root/repo/gpurun_out/pmc_$W/sq2 -o sq2 -- python /root/repo/tools/tnbench.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /root/repo/gpurun_out/pmc_$W/tcc -o tcc -- python /root/repo/tools/tnbench.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_$W/fetch -o fetch -- python /root/repo/tools/tnbench.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_$W/write -o write -- python /root/repo/tools/tnbench.py $W > /dev/null 2>&1
find /root/repo/gpurun_out/pmc_$W -name "*.csv" | head
