#!/bin/bash
# SQ counters of one bench step per kernel (instruction mix / stall shares): gpurun -- 'bash profiles/pmc_sq.sh <tag>'
TAG=${1:-r02sq}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d "$OUT/${TAG}" -o "$TAG" -- \
  python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${TAG}.log" 2>&1
cd "$R"
python3 - "$OUT/$TAG" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.Counter()); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[r["Kernel_Name"]] += 1
names = sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:14]
print("%-44s %5s %10s %10s %9s %11s %9s %9s %9s" % ("kernel", "calls", "VALU/wave", "SALU/wave", "LDS/wave", "cyc/wave(x4)", "wait_any", "wait_inst", "active"))
for k in names:
    a = acc[k]; w = max(a["SQ_WAVES"], 1); wc = max(a["SQ_WAVE_CYCLES"], 1)
    print("%-44s %5d %10.0f %10.0f %9.0f %11.0f %8.0f%% %8.0f%% %8.0f%%" % (k.replace("void wp::", "").replace("wp::", "")[:44], cnt[k], a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w, wc / w, 100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc))
PY
