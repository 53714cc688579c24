cd /root/repo
OLD=/root/repo/atomnas_amd/csrc/build/variants/libold.so
rm -f gpurun_out/dig_old.txt gpurun_out/dig_new.txt
ATOMNAS_HIP_LIB=$OLD ATOMNAS_SWEEP_DIGESTS=gpurun_out/dig_old.txt timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -q -m gpu -k "pw_gemm or expand_bwd or project_bwd" -p no:cacheprovider 2>&1 | tail -2
ATOMNAS_SWEEP_DIGESTS=gpurun_out/dig_new.txt timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -q -m gpu -k "pw_gemm or expand_bwd or project_bwd" -p no:cacheprovider 2>&1 | tail -2
diff gpurun_out/dig_old.txt gpurun_out/dig_new.txt | grep "^>" | awk '{print $2, $3}' | sort | uniq -c | sort -rn | head -60
