#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log
for wl in evplp ppm; do
timeout 600 python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$wl.jsonl 2> $O/bench_$wl.err
python3 -c "
import json
d=json.loads(open('$O/bench_$wl.jsonl').read().strip().splitlines()[-1]); r=d.get('roofline_splat', d['roofline']); print('$wl ms/step',round(d['ms_per_step'],3),'splat pass_ms',round(r['pass_ms'],4),'tiles_ms',round(r['tiles_kernel_ms'],4),'frac',round(r['frac'],4),'pairs',r['pairs_per_frame'])"
done
cd /tmp; for wl in evplp ppm; do rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/kt_$wl.log 2>&1; find $GRAFT_REPO_ROOT/$O/kt_$wl -name "*kernel_stats.csv" | head -1 | xargs cat | grep -i "splat\|Name" | cut -c1-160; find $GRAFT_REPO_ROOT/$O/kt_$wl -name "*kernel_trace.csv" -delete; done
