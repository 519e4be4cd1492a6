"""Timeline of the last Linear step of a rocprofv3 kernel trace: start (us), duration (us), queue, kernel, grid.

python profiles/timeline.py gpurun_out/<dir>/<name>_kernel_trace.csv [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "decode_count" in r["Kernel_Name"]]
ranks = [i for i, r in enumerate(rows) if "need_groups" in r["Kernel_Name"] or "round0_rank" in r["Kernel_Name"]]
last = ranks[-1]
s = max(i for i in starts if i < last)
e = min([i for i in starts if i > last] + [len(rows)])
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:e]:
    k = r["Kernel_Name"].replace("void wp::", "").replace("wp::", "").split("(")[0][:48]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < min_us:
        continue
    print("%8.1f %8.1f  q%-3s %-48s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r["Queue_Id"], k, r["Grid_Size_X"]))
