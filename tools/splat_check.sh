#!/bin/bash
# splat-pass check: parity tests that touch the splat, then kernel stats of the evplp and ppm workloads.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/splat_ab; rm -rf $O; mkdir -p $O
cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_end_to_end.py -q -x -k "splat or photon or bins or evplp or progressive or ppm or determin" > $O/tests_full.txt 2>&1; grep -E "passed|failed|error" $O/tests_full.txt | tail -3 > $O/tests.txt
cat $O/tests.txt
cd /tmp
for wl in evplp ppm; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${wl} -- python3 $ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/kt_${wl}.log 2>&1
    f=$(find $O/kt_${wl} -name "*kernel_stats.csv" | head -1)
    echo "== $wl"; grep -E "splat|Name" $f | cut -d, -f1-5 | cut -c1-150
    grep '"metric"' $O/kt_${wl}.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('roofline') if d['roofline']['bound']=='hbm' else '', d.get('roofline_splat'))"
    rm -rf $O/kt_${wl}
done
