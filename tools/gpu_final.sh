#!/bin/bash
# round-end evidence: bench lines for every workload, traversal statistics, rocprofv3 kernel stats + PMC
tag=${1:-r02}
export TMPDIR=/tmp
O=gpurun_out/final_$tag; mkdir -p $O
# (the counters build that tools/traversal_stats.py loads -- evplp_amd/lib/libevplp_hip_stats.so, `make stats` -- travels with the snapshot)
python3 bench.py > $O/${tag}_bench_ir.jsonl 2> $O/bench_ir.err
python3 bench.py --workload evplp --steps 30 --warmup 3 --no-cpu-baseline > $O/${tag}_bench_evplp.jsonl 2> $O/bench_evplp.err
python3 bench.py --workload ppm --steps 100 --warmup 5 --no-cpu-baseline > $O/${tag}_bench_ppm.jsonl 2> $O/bench_ppm.err
python3 bench.py --workload vsl --steps 2 --warmup 1 --no-cpu-baseline > $O/${tag}_bench_vsl.jsonl 2> $O/bench_vsl.err
EVPLP_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/${tag}_bench_ir_forcedist.jsonl 2> $O/bench_fd.err
for sc in easy hard; do python3 tools/traversal_stats.py --scene $sc --out $O/${tag}_traversal_$sc.json > $O/trav_$sc.log 2>&1; done
bash tools/prof_all.sh $tag > $O/prof_all.log 2>&1
bash tools/prof_pmc_wl.sh vsl $tag > $O/prof_vsl.log 2>&1          # config #5: kernel statistics + PMC summary
cp gpurun_out/pmc_vsl/${tag}_bench_vsl_* $O/ 2>/dev/null; cp gpurun_out/pmc_vsl/pmc_vsl_summary.json $O/ 2>/dev/null
cp gpurun_out/prof_$tag/${tag}_* $O/ 2>/dev/null
cp gpurun_out/prof_$tag/pmc_*_summary.json $O/ 2>/dev/null
for f in $O/${tag}_bench_*.jsonl; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value',round(d['value']),'ms',round(d['ms_per_step'],3), r['kernel'] if 'kernel' in r else '', 'frac',round(r['frac'],4), {k:round(v,4) for k,v in r.items() if k in ('kernel_ms','pass_ms','frac_nominal_pairs')})"; done
tail -40 $O/prof_all.log | cut -c1-220
