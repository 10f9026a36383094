"""Kernel timeline of a few bench iterations: start / end of every launch relative to the first one of the iteration (needs a
rocprofv3 --kernel-trace csv).  usage: trace_gaps.py <kernel_trace.csv> [first kernel name substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "primary_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit("not enough iterations")
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} -> {e/1e3:9.1f} us  ({(e-s)/1e3:7.1f})  gap {max(s - prev_end, 0)/1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
print("iteration:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")
