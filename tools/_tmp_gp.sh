cd /root/repo
timeout 900 python -m pytest tests -q -m gpu -x -k "kernels or late_stages or bench_shapes or xbwd or fused" -p no:cacheprovider 2>&1 | tail -4
OLD=/root/repo/atomnas_amd/csrc/build/variants/libold.so
for i in 1 2; do
echo "== old"; ATOMNAS_HIP_LIB=$OLD timeout 300 python tools/bringup.py 256 2>&1 | grep -E "graph ms|atomnas_pw|atomnas_expand|atomnas_project|atomnas_bn|atomnas_act"
echo "== new"; timeout 300 python tools/bringup.py 256 2>&1 | grep -E "graph ms|atomnas_pw|atomnas_expand|atomnas_project|atomnas_bn|atomnas_act"
done
