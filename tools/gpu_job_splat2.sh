#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/splat_ab; rm -rf $O; mkdir -p $O
cd /tmp
for lib in "$@"; do
  for wl in evplp ppm; do
    if [ "$lib" != "default" ]; then export EVPLP_LIB=$ROOT/evplp_amd/lib/libevplp_hip_$lib.so; else unset EVPLP_LIB; fi
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${wl}_$lib -- python3 $ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/kt_${wl}_$lib.log 2>&1
    f=$(find $O/kt_${wl}_$lib -name "*kernel_stats.csv" | head -1)
    echo "== $wl lib=$lib"; grep -E "splat" $f | cut -d, -f1-4 | cut -c1-150
    grep '"metric"' $O/kt_${wl}_$lib.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline_splat') or d['roofline']; print(d['ms_per_step'], r.get('pass_ms'), r['frac'])"
    rm -rf $O/kt_${wl}_$lib
  done
done
