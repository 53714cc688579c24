mkdir -p gpurun_out/r2n; O=gpurun_out/r2n
DETAIL=1 python tools/bringup.py 256 > $O/b_new.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
grep "graph ms" $O/b_new.log; tail -3 $O/bench.err; cut -c1-700 $O/bench.json
