#!/bin/bash
# stage times of configs 2, 3, 5 for the library in place and for every alternative build in wordpiece_amd/ab/*.so
#   gpurun -- 'bash profiles/ab_walk.sh <tag>'  ->  gpurun_out/<tag>_<config>_<variant>.json
set -o pipefail
TAG=${1:-ab}
for CFG in ${CFGS:-2 3 5}; do
  for L in main wordpiece_amd/ab/*.so; do
    [ -e "$L" ] || [ "$L" = main ] || continue
    V=$(basename "$L" .so)
    if [ "$L" = main ]; then unset WP_LIB; else export WP_LIB=$PWD/$L; fi
    timeout -k 10 150 python bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_${CFG}_$V.json 2> gpurun_out/${TAG}_${CFG}_$V.err || { echo "FAILED $CFG $V"; tail -3 gpurun_out/${TAG}_${CFG}_$V.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_${CFG}_$V.json"))
print("config $CFG $V", d["ms_per_step"], d["stage_ms_per_step"], "scatter launch ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
PY
  done
done
