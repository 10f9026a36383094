#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
for sc in easy hard; do
  timeout 300 python3 tools/traversal_stats.py --scene $sc --out $O/stats_${sc}.json > $O/stats_${sc}.log 2>&1
  python3 -c "
import json
d=json.load(open('$O/stats_${sc}.json')); b=d['beam']; print('$sc', {k:(round(v,2) if isinstance(v,float) else v) for k,v in b.items() if 'heav' not in k})"
  timeout 300 python3 tools/traversal_stats.py --scene $sc --no-beams --out $O/stats_${sc}_walk.json > $O/stats_${sc}_walk.log 2>&1
  python3 -c "
import json
d=json.load(open('$O/stats_${sc}_walk.json')); print('$sc walk', 'walks',d['walks'],'nodes/walk',round(d['node_visits_per_walk'],1),'leaves/walk',round(d['leaf_blocks_per_walk'],2), 'pairs/walk', round(d['tri_pairs_per_walk'],2))"
done
for sc in hard easy; do
timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_${sc}.jsonl 2> $O/bench_ir_${sc}.err
EVPLP_NO_BEAMS=1 timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_${sc}_nobeams.jsonl 2> $O/bench_ir_${sc}_nobeams.err
EVPLP_NO_BEAMS=1 EVPLP_GATHER_K=1 timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_${sc}_nobeams_k1.jsonl 2> $O/bench_ir_${sc}_nobeams_k1.err
for v in "" _nobeams _nobeams_k1; do python3 -c "
import json,sys
d=json.loads(open('$O/bench_ir_${sc}$v.jsonl').read().strip().splitlines()[-1]); print('$sc$v value',round(d['value']),'ms',round(d['ms_per_step'],2),'kernel_ms',round(d['roofline']['kernel_ms'],2),'frac',round(d['roofline']['frac'],4))"; done
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ir -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof_ir.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_ir -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
find $O/prof_ir -name "*kernel_trace.csv" -delete; find $O/prof_ir -name "*_agent_info.csv" -delete
