mkdir -p gpurun_out/r2i; O=gpurun_out/r2i
python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -x -k "dw or depthwise" > $O/kern.log 2>&1; echo "rc=$?" >> $O/kern.log
DETAIL=1 python tools/bringup.py 256 > $O/b_new.log 2>&1
DETAIL=1 ATOMNAS_DW_XCD_LEVEL=1 python tools/bringup.py 256 > $O/b_lvl1.log 2>&1
DETAIL=1 ATOMNAS_DW_FWD_CB=16 ATOMNAS_DW_BWD_CB=16 python tools/bringup.py 256 > $O/b_cb16.log 2>&1
DETAIL=1 ATOMNAS_DW_FWD_CB=32 ATOMNAS_DW_BWD_CB=32 python tools/bringup.py 256 > $O/b_cb32.log 2>&1
DETAIL=1 ATOMNAS_DW_FWD_CB=8 python tools/bringup.py 256 > $O/b_fcb8.log 2>&1
python tools/dwbench.py both 256 slab > $O/dw_base.log 2>&1
python tools/dwbench.py tools/variants/libstage.so fwd 256 slab > $O/dw_stage.log 2>&1
python tools/dwbench.py tools/variants/libminb3.so bwd 256 slab > $O/dw_minb3.log 2>&1
python tools/dwbench.py tools/variants/libdma.so bwd 256 slab > $O/dw_dma.log 2>&1
tail -3 $O/kern.log; for f in new lvl1 cb16 cb32 fcb8; do echo $f; grep "graph ms" $O/b_$f.log; done; tail -2 $O/dw_*.log
