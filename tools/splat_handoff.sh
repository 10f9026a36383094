#!/bin/bash
# The hand-off between the three kernels of a photon-splat pass, measured: rocprofv3 kernel trace of tools/splat_times.py (config #3
# and #4 record sets, three passes each), the launches of the last passes on one time axis -> gaps bin -> scatter -> tiles.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/splat_handoff; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $ROOT/tools/splat_times.py > $O/run.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
cd $ROOT
python3 - "$f" <<'PY' > $O/handoff.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("splat_bin", "splat_scatter", "splat_tiles", "splat_tile_box", "splat_clear", "splat_big"))]
# passes: a pass starts at splat_bin
passes, cur = [], []
for r in sel:
    if "splat_bin" in r["Kernel_Name"] and cur: passes.append(cur); cur = []
    cur.append(r)
if cur: passes.append(cur)
for p in passes:
    t0 = int(p[0]["Start_Timestamp"]); prev_end = None; line = []
    for r in p:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = "" if prev_end is None else " gap %5.1f" % ((s - prev_end) / 1e3)
        line.append("%s%s %6.1f us" % (gap, r["Kernel_Name"].split("(")[0].split("::")[-1][:18], (e - s) / 1e3)); prev_end = e
    print("pass %7.1f us: " % ((prev_end - t0) / 1e3) + " |".join(line))
PY
cat $O/run.log | tail -3; cat $O/handoff.txt
