#!/bin/bash
# Builds the round-4 prototype of the late-stage wide-input GEMM on the LDS-DMA queue (csrc/experimental/nt_swg.hip): the product
# library's objects + the prototype's entry point atomnas_exp_nt_swg  ->  atomnas_amd/csrc/build/variants/libntswg.so (git-ignored).
# Measured by tools/experiments/ntswg_bench.py with ATOMNAS_HIP_LIB pointing at it.
set -e
cd "$(dirname "$0")/.."
python -m atomnas_amd.build > /dev/null
B=atomnas_amd/csrc/build
mkdir -p $B/variants/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c atomnas_amd/csrc/experimental/nt_swg.hip -o $B/variants/obj/nt_swg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/variants/libntswg.so $B/variants/obj/nt_swg.o $B/*.o
echo built $B/variants/libntswg.so
