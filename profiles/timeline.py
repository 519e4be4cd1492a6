"""Prints the kernel timeline of the last encode in a rocprofv3 kernel trace CSV (one line per launch:
start offset us, duration us, gap to the previous launch on the same stream, stream, kernel, workgroups)."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "decode_count" in r["Kernel_Name"]]
s = idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
prev_end = {}
for r in rows[s:]:
    st = int(r["Start_Timestamp"]) - t0
    en = int(r["End_Timestamp"]) - t0
    q = r["Stream_Id"]
    gap = st - prev_end.get(q, 0)
    name = r["Kernel_Name"].replace("void ", "").replace("wp::", "")[:38]
    if en - st < 12000 and len(sys.argv) > 2:
        prev_end[q] = en
        continue
    print("%9.1f %8.1f gap%7.1f s%s %s g=%d" % (st / 1e3, (en - st) / 1e3, gap / 1e3, q, name,
                                               int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
    prev_end[q] = en
