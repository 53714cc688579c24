#!/bin/bash
# same-box A/B of the bench step under environment switches: tools/ab_bench.sh "VAR=a" "VAR=b" ... (each run: 20 timed steps, no cpu baseline)
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  out=$(env $cfg python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("%-40s %8.1f img/s  %7.3f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
done
