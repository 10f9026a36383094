#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/lane; rm -rf $O; mkdir -p $O
cd /tmp
for lib in "$@"; do
  if [ "$lib" != "default" ]; then export EVPLP_LIB=$ROOT/evplp_amd/lib/libevplp_hip_$lib.so; else unset EVPLP_LIB; fi
  for wl in ppm; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${wl}_$lib -- python3 $ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/kt_${wl}_$lib.log 2>&1
    f=$(find $O/kt_${wl}_$lib -name "*kernel_stats.csv" | head -1)
    echo "== $wl lib=$lib"; grep -E "light_trace" $f | cut -d, -f1-4 | cut -c1-150
    rm -rf $O/kt_${wl}_$lib
  done
  python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gpu PT', d['gpu_path_tracer_mpaths_s'])"
  python3 $ROOT/tools/quick_bench.py --help > /dev/null 2>&1
done
