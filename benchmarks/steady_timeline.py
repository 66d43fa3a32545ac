"""Timeline of the last slicetypeDecide calls of bench.py's steady workload from a rocprofv3 --kernel-trace
--memory-copy-trace run: python benchmarks/steady_timeline.py <dir>."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:30], r.get("Stream_Id", "")))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[-14:], ""))
ev.sort()
# the steady workload's decisions are the last search launches of the run: from the 4th-last to the last
ss = [i for i, e in enumerate(ev) if "search_flow_kernel" in e[2]]
seg = ev[ss[-4] - 30: ss[-1] + 60]
t0 = seg[0][0]
busy_end = t0
i = 0
while i < len(seg):
    s, e, n, st = seg[i]
    j, mx, tot = i, e, e - s
    while j + 1 < len(seg) and seg[j + 1][2] == n:
        j += 1
        mx = max(mx, seg[j][1])
        tot += seg[j][1] - seg[j][0]
    gap = (s - busy_end) / 1e3
    print("%-32s x%-4d start %8.1f end %8.1f  busy %7.1f us  gap-before %6.1f  stream %s" % (n, j - i + 1, (s - t0) / 1e3, (mx - t0) / 1e3, tot / 1e3, gap, st))
    busy_end = max(busy_end, mx)
    i = j + 1
