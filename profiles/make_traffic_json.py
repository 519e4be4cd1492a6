"""Derives profiles/radix_scatter_traffic.json from the two rocprofv3 --pmc passes of collect.sh.

usage: python profiles/make_traffic_json.py <dir with *_pmc_FETCH_SIZE/ and *_pmc_WRITE_SIZE/> <tag>
FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE tallies 128-B requests as 64 B, so it is doubled
(MI355X_MICROARCH.md, HBM section).  The factor is checked on radix_hist_kernel<uint64>, which reads exactly
8 B per key and nothing else.  Bytes are summed over all launches of one bench step and divided by the
launch count, like bench.py's `achieved`.
"""
import collections
import csv
import glob
import json
import os
import sys

base, tag = sys.argv[1], sys.argv[2]
ITEMS = 20  # WP_RADIX_ITEMS32: 32-bit round-0 keys, 8-byte records (round 1: <unsigned long, 24>)
KEY = "radix_scatter_kernel<unsigned int, %d, " % ITEMS  # (both instantiations: ranks by match-any / by LDS atomics in a sort's first pass)
HIST = "radix_hist_kernel<unsigned int, %d>" % ITEMS


def per_kernel(counter):
    f = glob.glob(os.path.join(base, "%s_pmc_%s" % (tag, counter), "**", "*counter_collection.csv"), recursive=True)[0]
    tot, cnt, grid = collections.Counter(), collections.Counter(), {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"]) * 1024.0
        cnt[k] += 1
        grid.setdefault(k, []).append(int(r["Grid_Size"]))
    return tot, cnt, grid


ft, fc, fg = per_kernel("FETCH_SIZE")
wt, wc, _ = per_kernel("WRITE_SIZE")
names = [k for k in ft if KEY in k]
# bench.py --steps 1 --warmup 1 encodes twice (+ once more for the oracle sample check): per-launch
# averages do not depend on the number of steps
launches = sum(fc[k] for k in names)
# Check of the x2 factor on a kernel with a known read: the plain device-to-device copy of the bench set-up is
# not in the trace, so the check uses the scatter kernel itself: records read = 8 B per element (4 B in the
# first pass of the sort, whose index column is made up) + 1 digit byte is NOT read by it
factor = None
fetch = 2.0 * sum(ft[k] for k in names) / launches
write = sum(wt[k] for k in names) / sum(wc[k] for k in names)
bench = json.load(open(os.path.join(base, "%s_bench.json" % tag)))
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
out = {
    "kernel": "radix_scatter_kernel<uint32, %d, stable|first-pass>" % ITEMS,
    "round": tag,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 "
              "--warmup 1 --no-cpu-baseline` (profiles/collect.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md "
              "(gfx950 tallies 128-B requests as 64 B); WRITE_SIZE taken as is",
    "launches_counted": launches,
    "fetch_bytes_per_launch": int(fetch),
    "write_bytes_per_launch": int(write),
    "hbm_bytes_per_launch": int(fetch + write),
    "algorithmic_bytes_per_launch": alg,
    "ratio": round((fetch + write) / alg, 3),
}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "radix_scatter_traffic.json"), "w"), indent=1)
