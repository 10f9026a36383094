#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/prim; rm -rf $O; mkdir -p $O
cd $ROOT
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/tests_full.txt 2>&1; grep -E "passed|failed|error" $O/tests_full.txt | tail -3
cd /tmp
for wl in ir ppm; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${wl} -- python3 $ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/kt_${wl}.log 2>&1
    f=$(find $O/kt_${wl} -name "*kernel_stats.csv" | head -1)
    echo "== $wl"; grep -E "primary|light_trace|path_trace|gather_lvc" $f | cut -d, -f1-4 | cut -c1-150
    grep '"metric"' $O/kt_${wl}.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
    rm -rf $O/kt_${wl}
done
