#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
for sc in easy hard; do
  timeout 300 python3 tools/traversal_stats.py --scene $sc --out $O/stats_${sc}.json > $O/stats_${sc}.log 2>&1
  python3 -c "
import json
d=json.load(open('$O/stats_${sc}.json')); print('$sc walk', 'walks',d['walks'],'nodes/walk',round(d['node_visits_per_walk'],1),'leaves/walk',round(d['leaf_blocks_per_walk'],2), 'pairs/walk', round(d['tri_pairs_per_walk'],2), 'exact pairs/walk', round(d['tri_pairs_to_exact_predicate_per_walk'],2), 'nodes/ray', round(d['nodes_per_ray'],1), 'tris/ray', round(d['tris_per_ray'],2))"
done
for sc in hard easy; do for lib in "" _gw6 _gw8; do for k in 1 2 4; do
L=$PWD/evplp_amd/lib/libevplp_hip$lib.so
EVPLP_LIB=$L EVPLP_GATHER_K=$k timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/b.jsonl 2> $O/b.err
python3 -c "
import json,sys
d=json.loads(open('$O/b.jsonl').read().strip().splitlines()[-1]); print('$sc lib=$lib k=$k value',round(d['value']),'ms',round(d['ms_per_step'],2),'kernel_ms',round(d['roofline']['kernel_ms'],2),'frac',round(d['roofline']['frac'],4))"
done; done; done
