#!/bin/bash
# config #4's iteration on one time axis: kernel trace of bench.py --workload ppm, the launches of the last iterations (tools/trace_timeline.py)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/ppm_timeline; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $ROOT/bench.py --workload ppm --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $O/run.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
cd $ROOT
python3 - "$f" <<'PY' > $O/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keys = ("primary_kernel", "light_trace", "splat_bin", "splat_scatter", "splat_tiles", "resolve")
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in keys)]
# the timed region: take launches 40 % .. 60 % into the list (steady state, before the profiled extra passes at the end)
n = len(sel); seg = sel[int(n * 0.45): int(n * 0.45) + 30]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f -> %9.1f (%6.1f us)  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].split("::")[-1][:24]))
PY
cat $O/timeline.txt
