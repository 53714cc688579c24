#!/bin/bash
# usage: tools/variant.sh NAME file.hip "<extra hipcc flags>"   -> atomnas_amd/csrc/build/variants/libNAME.so (experiment builds; git-ignored)
set -e
cd /root/repo
NAME=$1; SRC=$2; EXTRA=$3
B=atomnas_amd/csrc/build
mkdir -p atomnas_amd/csrc/build/variants/obj
O=atomnas_amd/csrc/build/variants/obj/${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed $EXTRA -c atomnas_amd/csrc/$SRC -o $O
OBJS=$(ls $B/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o atomnas_amd/csrc/build/variants/lib$NAME.so $O $OBJS
echo built atomnas_amd/csrc/build/variants/lib$NAME.so
