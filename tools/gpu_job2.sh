#!/bin/bash
# shaft-list gather: parity tests, traversal statistics, bench (hard + easy)
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -15 $O/pytest_parity.log
for sc in easy hard; do
  timeout 300 python3 tools/traversal_stats.py --scene $sc --no-lists --out $O/stats_${sc}_walk.json > $O/stats_${sc}_walk.log 2>&1
  timeout 300 python3 tools/traversal_stats.py --scene $sc --out $O/stats_${sc}_lists.json > $O/stats_${sc}_lists.log 2>&1
done
tail -3 $O/stats_*_walk.log
cat $O/stats_hard_lists.json $O/stats_easy_lists.json
for sc in hard easy; do
timeout 600 python3 bench.py --steps 10 --warmup 2 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_$sc.jsonl 2> $O/bench_ir_$sc.err
EVPLP_NO_SHAFT_LISTS=1 timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_${sc}_nolists.jsonl 2> $O/bench_ir_${sc}_nolists.err
done
for f in $O/bench_ir_*.jsonl; do echo $f; python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('value',round(d['value']),'ms',round(d['ms_per_step'],2),'kernel_ms',round(d['roofline']['kernel_ms'],2),'frac',round(d['roofline']['frac'],4))"; done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -5 $O/pytest_all.log
