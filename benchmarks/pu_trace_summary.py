"""Per-kernel start / duration table of the last x265gpu_pu_motion_estimate call in a rocprofv3 kernel trace
(csv): python benchmarks/pu_trace_summary.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pu_search_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per_call = 5 if len(rows) % 5 == 0 else 4
last = rows[-per_call:]
t0 = min(int(r["Start_Timestamp"]) for r in last)
for r in last:
    name = r["Kernel_Name"].split("pu_search_kernel")[1].split("(")[0]
    print("%-24s grid %8s lds %6s start %8.3f ms  dur %8.3f ms" % (
        name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"),
        (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
print("span %.3f ms" % ((max(int(r["End_Timestamp"]) for r in last) - t0) / 1e6))
