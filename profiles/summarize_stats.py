"""Print the top kernels of a rocprofv3 --kernel-trace --stats CSV (…kernel_stats.csv)."""
import csv
import glob
import sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-72s %6s %12s %6s" % (r["Name"][:72], r["Calls"], r["AverageNs"], r["Percentage"]))
