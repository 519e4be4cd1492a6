#!/bin/bash
# SQ counters per DISPATCH of one kernel name pattern in one bench step (which of a kernel's launches is slow, and why):
#   gpurun -- 'bash profiles/pmc_dispatch.sh <tag> <kernel substring> <bench args...>'
TAG=$1; PAT=$2; shift 2
R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS} --kernel-trace --output-format csv -d "$OUT/${TAG}" -o "$TAG" -- \
  python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > "$OUT/${TAG}.log" 2>&1
cd "$R"
python3 - "$OUT/$TAG" "$PAT" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    key = int(r["Dispatch_Id"])
    rows.setdefault(key, {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
print("%-8s %-46s %9s %9s %9s %12s %9s %9s %12s %9s" % ("dispatch", "kernel", "VALU/wv", "SALU/wv", "LDS/wv", "cyc/wave(x4)", "wait_any", "wait_inst", "bank_confl/wv", "lds_act"))
for k, a in rows.items():
    waves = max(a.get("SQ_WAVE_CYCLES", 1) and 1, 1)
    # (SQ_WAVES is not collected here: normalise by VALU-independent wave count = grid / 64 is not in the csv either;
    #  ratios against WAVE_CYCLES are what matters)
    wc = max(a.get("SQ_WAVE_CYCLES", 1), 1)
    print("%-8d %-46s %9.3g %9.3g %9.3g %12.4g %8.0f%% %8.0f%% %12.3g %8.0f%%" % (
        k, a["name"].replace("void wp::", "")[:46], a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_SALU", 0), a.get("SQ_INSTS_LDS", 0), wc,
        100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_LDS_BANK_CONFLICT", 0),
        100 * a.get("SQ_ACTIVE_INST_LDS", 0) / wc))
PY
# raw values of every collected counter per dispatch (PMC=... chooses other counters than the default eight)
python3 - "$OUT/$TAG" "$PAT" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict(); names = []
for r in csv.DictReader(open(f)):
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
print("dispatch " + " ".join("%22s" % n for n in names))
for k, a in rows.items():
    print("%-8d " % k + " ".join("%22.4g" % a.get(n, 0) for n in names))
PY
