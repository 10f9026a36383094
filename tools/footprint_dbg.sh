#!/bin/bash
# tile-kernel time of the proxy footprint with parts of it compiled out (EVPLP_PROXY_DBG builds): tools/footprint_dbg.sh <tag> <lib suffixes...>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/$1; mkdir -p $O; shift
cd /tmp
for v in "$@"; do
  lib=$ROOT/evplp_amd/lib/libevplp_hip${v:+_$v}.so
  [ "$v" = "full" ] && lib=$ROOT/evplp_amd/lib/libevplp_hip.so
  for wl in evplp ppm; do
    EVPLP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $ROOT/bench.py --workload $wl --footprint ${FOOTPRINT:-proxy} --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/kt_${wl}_${v}.log 2>&1
    f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
    echo "== $v $wl"; grep -E "splat_tiles" $f | cut -d, -f1-4 | cut -c1-150
    rm -rf $O/kt
  done
done
