#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -16 $O/pytest_all.log
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python3 bench.py > $O/bench_default.jsonl 2> $O/bench_default.err; tail -c 2500 $O/bench_default.jsonl
