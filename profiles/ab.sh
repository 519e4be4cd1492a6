#!/bin/bash
# A/B helper: bash profiles/ab.sh <tag> [ENV=VAL ...]  -> kernel stats of one bench run under rocprofv3
set -eo pipefail
TAG=$1; shift
R=$(pwd)
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ab_$TAG" -o ab -- \
  python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/ab_$TAG.log" 2>&1
cd "$R"
echo "== $TAG $*"
grep -o '"ms_per_step": [0-9.]*' "gpurun_out/ab_$TAG.log" || true
python3 profiles/summarize_stats.py "gpurun_out/ab_$TAG" 14
