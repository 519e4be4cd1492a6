#!/bin/bash
# Per-kernel HBM traffic of one bench step: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md, HBM section), summarised per kernel name as bytes per launch.
#   gpurun -- 'bash profiles/pmc_kernels.sh <tag>'  ->  gpurun_out/<tag>_pmc_kernels.txt
TAG=${1:-r02}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_$C" -o "$TAG" -- \
    python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${TAG}_pmc_$C.log" 2>&1
  echo "pass $C done" >> "$OUT/${TAG}_pmc_progress.txt"
done
cd "$R"
python3 profiles/pmc_summary.py "$OUT" "$TAG" > "$OUT/${TAG}_pmc_kernels.txt"
cat "$OUT/${TAG}_pmc_kernels.txt"
