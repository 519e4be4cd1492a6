"""Per-kernel FETCH_SIZE / WRITE_SIZE of the two --pmc passes of pmc_kernels.sh: MB per launch.
FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md): the doubled figure is printed too."""
import collections, csv, glob, os, sys
base, tag = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(base, "%s_pmc_%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True)[0]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"]) * 1024.0
        cnt[r["Kernel_Name"]] += 1
    res[c] = (tot, cnt)
names = sorted(res["FETCH_SIZE"][0], key=lambda k: -(res["FETCH_SIZE"][0][k] + res["WRITE_SIZE"][0].get(k, 0)))
print("%-64s %6s %12s %12s %12s" % ("kernel", "calls", "fetch MB/call", "x2", "write MB/call"))
for k in names[:40]:
    n = res["FETCH_SIZE"][1][k]
    fe = res["FETCH_SIZE"][0][k] / n / 1e6
    wr = res["WRITE_SIZE"][0].get(k, 0) / max(res["WRITE_SIZE"][1].get(k, 1), 1) / 1e6
    print("%-64s %6d %12.1f %12.1f %12.1f" % (k[:64], n, fe, 2 * fe, wr))
