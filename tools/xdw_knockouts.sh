# knock-out builds of the fused forward (tools/build_xdw_experiment.sh xkoN -DXD_KO=N): what each part of an item costs
for k in ${KOS:-1 2 4 8 6 15}; do echo "KO $k"; ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libxko$k.so XDWBENCH_CASES="${XDWBENCH_CASES:-56,24,144,3;56,24,144,7}" python tools/xdwbench.py 2>&1 | grep "^H.*k[357]:" | sed 's/bwd.*//'; done
