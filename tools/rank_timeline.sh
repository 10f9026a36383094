#!/bin/bash
# kernel trace of ONE dealt rank's frames (tools/rank_timeline.py) -> the last frame on a time axis.  usage: tools/rank_timeline.sh [out dir]
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}; O=${1:-$ROOT/gpurun_out/rank_timeline}; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $ROOT/tools/rank_timeline.py > $O/run.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("evplp::", "")) for r in rows if "evplp::" in r["Kernel_Name"]]
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith("primary_kernel")]
i0 = starts[-2]; t0 = ev[i0][0]
print("   start        duration   kernel (one frame of the rank: from its G-buffer pass to the next one's)")
for s, e, n in ev[i0:starts[-1] + 1]:
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:60]))
PY
rm -rf $O/kt
