"""Timeline of the LAST burst of kernels in a rocprofv3 --kernel-trace CSV (one call of a probe script): start offset and
duration in microseconds per launch, so that what overlaps on the internal streams is visible.
    python tools/kernel_timeline.py <dir or *_kernel_trace.csv> [gap_us=200] [burst=1: the last, 2: the one before, ...]"""
import csv, glob, os, sys

path = sys.argv[1]
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 200e3
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
# bursts, last first; the optional third argument picks one (1 = the last, 2 = the one before, ...)
want = int(sys.argv[3]) if len(sys.argv) > 3 else 1
last, seen = [], 1
for r in reversed(rows):
    if last and int(last[-1]["Start_Timestamp"]) - int(r["End_Timestamp"]) > gap:
        if seen == want:
            break
        seen, last = seen + 1, []
    last.append(r)
last.reverse()
base = int(last[0]["Start_Timestamp"])
busy_end, busy = base, 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    for cut in ("hefx::", "void "):
        n = n.replace(cut, "")
    busy += max(0, e - max(s, busy_end))
    busy_end = max(busy_end, e)
    print("%9.1f %8.1f  q%-3s %s" % ((s - base) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n[:70]))
print("span %.1f us, GPU busy %.1f us" % ((busy_end - base) / 1e3, busy / 1e3))
