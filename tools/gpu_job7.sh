#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
for sc in easy hard; do
  timeout 300 python3 tools/traversal_stats.py --scene $sc --out $O/stats_${sc}.json > $O/stats_${sc}.log 2>&1
  tail -3 $O/stats_${sc}.log
  python3 -c "
import json
d=json.load(open('$O/stats_${sc}.json')); b=d['beam']; print('$sc', {k:(round(v,2) if isinstance(v,float) else v) for k,v in b.items()})"
done
