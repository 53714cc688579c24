mkdir -p gpurun_out/r2j; O=gpurun_out/r2j
python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
tail -15 $O/pytest.log; tail -4 $O/bench.err; cut -c1-400 $O/bench.json
