#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline block on the GPU box:
#   gpurun -- 'bash profiles/collect.sh <tag>'   ->  gpurun_out/<tag>_*  (copy the summaries to profiles/)
# Kernel trace/stats and each PMC counter run in separate passes (MI355X_MICROARCH.md, HBM section).
set -eo pipefail
TAG=${1:-r03}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -o "$TAG" -- \
  python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/${TAG}_trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_$C" -o "$TAG" -- \
    python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${TAG}_pmc_$C.log" 2>&1
done
cd "$R"
python3 bench.py --steps 10 --warmup 3 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cat "$OUT/${TAG}_bench.json"
python3 profiles/pmc_summary.py "$OUT" "$TAG" > "$OUT/${TAG}_pmc_kernels.txt" || true
python3 profiles/make_traffic_json.py "$OUT" "$TAG" > "$OUT/${TAG}_traffic.json" || true
# the other single-GPU configurations: bench line, kernel stats, PMC traffic per kernel
for CFG in 3 5; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}c${CFG}_trace" -o "$TAG" -- \
    python3 "$R/bench.py" --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${TAG}c${CFG}_trace.log" 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}c${CFG}_pmc_$C" -o "$TAG" -- \
      python3 "$R/bench.py" --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${TAG}c${CFG}_pmc_$C.log" 2>&1
  done
  cd "$R"
  python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_config${CFG}_bench.json" 2> "$OUT/${TAG}_config${CFG}_bench.err"
  python3 profiles/pmc_summary.py "$OUT" "${TAG}c${CFG}" > "$OUT/${TAG}_config${CFG}_pmc_kernels.txt" || true
done
python3 bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_config4_bench.json" 2> "$OUT/${TAG}_config4_bench.err"
