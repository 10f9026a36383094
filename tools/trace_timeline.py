"""The last launches of the per-iteration kernels of a rocprofv3 --kernel-trace csv, on one time axis (us).
usage: trace_timeline.py <kernel_trace.csv> [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
keys = ("primary", "light_trace", "splat_bin", "splat_scatter", "splat_tiles", "gather_vpl", "gather_vsl")
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in keys)]
for r in sel[-int(sys.argv[2]) if len(sys.argv) > 2 else -24:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%10.1f -> %10.1f  (%6.1f)  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][-28:]))
