#!/bin/bash
# Builds the round-4 E-elimination experiment: the product library's objects + dwconv_cw.hip recompiled with the recomputing backward
# kernel (-DATOMNAS_EXPERIMENTAL_XDW) + csrc/experimental/xdw_fused.hip  ->  atomnas_amd/csrc/build/variants/libxdw.so (git-ignored).
#   usage: tools/build_xdw_experiment.sh [NAME [extra hipcc flags for xdw_fused.hip]]      e.g.  tools/build_xdw_experiment.sh xdt -DXD_TIMING=1
# Load it with ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libxdw.so (tools/experiments/test_xdw_experimental_gpu.py, tools/xdwbench.py).
set -e
cd "$(dirname "$0")/.."
python -m atomnas_amd.build > /dev/null
NAME=${1:-xdw}; shift || true
B=atomnas_amd/csrc/build
mkdir -p $B/variants/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
if [ ! -f $B/variants/obj/xdw_dwconv_cw.o ] || [ atomnas_amd/csrc/dwconv_cw.hip -nt $B/variants/obj/xdw_dwconv_cw.o ]; then
  /opt/rocm/bin/hipcc $FLAGS -DATOMNAS_EXPERIMENTAL_XDW=1 -c atomnas_amd/csrc/dwconv_cw.hip -o $B/variants/obj/xdw_dwconv_cw.o
fi
/opt/rocm/bin/hipcc $FLAGS -DATOMNAS_EXPERIMENTAL_XDW=1 "$@" -c atomnas_amd/csrc/experimental/xdw_fused.hip -o $B/variants/obj/${NAME}_xdw_fused.o
OBJS=$(ls $B/*.o | grep -v "/dwconv_cw.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/variants/lib$NAME.so $B/variants/obj/xdw_dwconv_cw.o $B/variants/obj/${NAME}_xdw_fused.o $OBJS
echo built $B/variants/lib$NAME.so
