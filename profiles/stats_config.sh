#!/bin/bash
# kernel stats of one bench configuration: bash profiles/stats_config.sh <tag> <bench args...>
#   -> gpurun_out/<tag>_stats/ (rocprofv3 --kernel-trace --stats), gpurun_out/<tag>_bench.log
set -eo pipefail
TAG=$1; shift
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_stats" -o st -- \
  python3 "$R/bench.py" --no-cpu-baseline "$@" > "$R/gpurun_out/${TAG}_bench.log" 2>&1
cd "$R"
grep -o '"ms_per_step": [0-9.]*' "gpurun_out/${TAG}_bench.log" || true
python3 profiles/summarize_stats.py "gpurun_out/${TAG}_stats" 25
