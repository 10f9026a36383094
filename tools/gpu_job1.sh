#!/bin/bash
# round-2 first GPU job: test suite, traversal statistics on both scenes, bench baselines
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python3 tools/traversal_stats.py --scene easy --out $O/stats_easy.json > $O/stats_easy.log 2>&1
python3 tools/traversal_stats.py --scene hard --out $O/stats_hard.json > $O/stats_hard.log 2>&1
python3 bench.py --steps 10 --warmup 2 --scene hard > $O/bench_ir_hard.jsonl 2> $O/bench_ir_hard.err
python3 bench.py --steps 5 --warmup 1 --scene hard --workload evplp --no-cpu-baseline > $O/bench_evplp_hard.jsonl 2> $O/bench_evplp_hard.err
python3 bench.py --steps 20 --warmup 2 --scene hard --workload ppm --no-cpu-baseline > $O/bench_ppm_hard.jsonl 2> $O/bench_ppm_hard.err
EVPLP_BENCH_FORCE_DIST=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_forcedist.jsonl 2> $O/bench_forcedist.err
tail -c 1500 $O/bench_ir_hard.jsonl; echo; tail -c 600 $O/bench_evplp_hard.jsonl; echo; tail -c 600 $O/bench_ppm_hard.jsonl; echo; tail -c 300 $O/bench_forcedist.jsonl
tail -5 $O/*.err | tail -40
cat $O/stats_easy.json $O/stats_hard.json | head -120
