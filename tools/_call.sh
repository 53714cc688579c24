mkdir -p gpurun_out/r2l; O=gpurun_out/r2l
python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "expand_bwd" > $O/kern.log 2>&1; echo "rc=$?" >> $O/kern.log
DETAIL=1 python tools/bringup.py 256 > $O/b_f48.log 2>&1
DETAIL=1 ATOMNAS_FUSED_EXPAND_BWD=24 python tools/bringup.py 256 > $O/b_f24.log 2>&1
DETAIL=1 ATOMNAS_FUSED_EXPAND_BWD=0 python tools/bringup.py 256 > $O/b_f0.log 2>&1
tail -3 $O/kern.log; for f in f48 f24 f0; do echo $f; grep "graph ms" $O/b_$f.log; grep fusedbwd $O/b_$f.log; done
