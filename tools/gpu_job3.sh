#!/bin/bash
# beam-pass gather: parity tests, traversal statistics, bench (hard + easy)
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -15 $O/pytest_parity.log
for sc in easy hard; do
  timeout 300 python3 tools/traversal_stats.py --scene $sc --out $O/stats_${sc}_beam.json > $O/stats_${sc}_beam.log 2>&1
  tail -2 $O/stats_${sc}_beam.log
done
python3 -c "
import json
for sc in ('easy','hard'):
    try:
        d=json.load(open('$O/stats_%s_beam.json'%sc)); print(sc, json.dumps(d['beam']), d['shaded'], d['kernel_ms_with_counters'])
    except Exception as e: print(sc, 'ERR', e)
"
for sc in hard easy; do
timeout 600 python3 bench.py --steps 10 --warmup 2 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_$sc.jsonl 2> $O/bench_ir_$sc.err
EVPLP_NO_BEAMS=1 timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_${sc}_nobeams.jsonl 2> $O/bench_ir_${sc}_nobeams.err
done
for f in $O/bench_ir_*.jsonl; do echo $f; python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('value',round(d['value']),'ms',round(d['ms_per_step'],2),'kernel_ms',round(d['roofline']['kernel_ms'],2),'frac',round(d['roofline']['frac'],4))"; done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ir -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof_ir.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_ir -name "*kernel_stats.csv" | head -1 | xargs cat | head -14
find $O/prof_ir -name "*kernel_trace.csv" -delete; find $O/prof_ir -name "*_agent_info.csv" -delete
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -5 $O/pytest_all.log
