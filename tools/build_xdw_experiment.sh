#!/bin/bash
# Builds the round-4 E-elimination experiment: the product library's objects + csrc/experimental/xdw_cw_bwd.hip (the recomputing backward
# kernel; it lived in dwconv_cw.hip under an #ifdef until round 5) + csrc/experimental/xdw_fused.hip
#   ->  atomnas_amd/csrc/build/variants/libxdw.so (git-ignored).
#   usage: tools/build_xdw_experiment.sh [NAME [extra hipcc flags for xdw_fused.hip]]      e.g.  tools/build_xdw_experiment.sh xdt -DXD_TIMING=1
# Load it with ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libxdw.so (tools/experiments/test_xdw_experimental_gpu.py, tools/xdwbench.py).
set -e
cd "$(dirname "$0")/.."
python -m atomnas_amd.build > /dev/null
NAME=${1:-xdw}; shift || true
B=atomnas_amd/csrc/build
mkdir -p $B/variants/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
/opt/rocm/bin/hipcc $FLAGS -DATOMNAS_EXPERIMENTAL_XDW=1 -c atomnas_amd/csrc/experimental/xdw_cw_bwd.hip -o $B/variants/obj/xdw_cw_bwd.o
/opt/rocm/bin/hipcc $FLAGS -DATOMNAS_EXPERIMENTAL_XDW=1 "$@" -c atomnas_amd/csrc/experimental/xdw_fused.hip -o $B/variants/obj/${NAME}_xdw_fused.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/variants/lib$NAME.so $B/variants/obj/xdw_cw_bwd.o $B/variants/obj/${NAME}_xdw_fused.o $B/*.o
echo built $B/variants/lib$NAME.so
