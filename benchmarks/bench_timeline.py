"""Kernel timeline of the last lookahead step in a rocprofv3 --kernel-trace csv of bench.py (start / end of each kernel
run, same-named consecutive launches folded): python benchmarks/bench_timeline.py "<dir>/*_kernel_trace.csv"
(the last step of a bench.py run is its untimed per-phase pass; the gaps between runs are the host round trips)."""
import csv, sys, glob
f = glob.glob(sys.argv[1])[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take last step: find last lowres_init kernel start
idx = [i for i, r in enumerate(rows) if "lowres_init" in r["Kernel_Name"]]
start = idx[-1]
# step = from that lowres to the next lowres (or end)
seg = rows[start:]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
last_name = None
out = []
for r in seg:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.append((n, s, e))
# compress consecutive search kernels
i = 0
while i < len(out):
    n, s, e = out[i]
    j = i
    mx = e
    while j + 1 < len(out) and out[j + 1][0] == n:
        j += 1
        mx = max(mx, out[j][2])
    print("%-30s x%-4d start %8.1f us  end %8.1f us" % (n, j - i + 1, s / 1e3, mx / 1e3))
    i = j + 1
