mkdir -p gpurun_out/r2o; O=gpurun_out/r2o
python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -x -k "project_bwd" > $O/kern.log 2>&1; echo "rc=$?" >> $O/kern.log
python -m pytest tests/test_block_gpu.py tests/test_train_step_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 > $O/blocks.log 2>&1; echo "rc=$?" >> $O/blocks.log
DETAIL=1 python tools/bringup.py 256 > $O/b_pb1.log 2>&1
DETAIL=1 ATOMNAS_FUSED_PROJECT_BWD=0 python tools/bringup.py 256 > $O/b_pb0.log 2>&1
tail -4 $O/kern.log; tail -3 $O/blocks.log; for f in pb1 pb0; do echo $f; grep "graph ms" $O/b_$f.log; grep "fusedbwd" $O/b_$f.log; done
