#!/bin/bash
# VSL gather time (1024^2, tools/quick_bench.py --vsl) of several builds of the library side by side: tools/lib_ab_vsl.sh <tag> <variants... | full>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/$1; mkdir -p $O; shift
for v in "$@"; do
  lib=$ROOT/evplp_amd/lib/libevplp_hip_$v.so; [ "$v" = "full" ] && lib=$ROOT/evplp_amd/lib/libevplp_hip.so
  echo "== $v"
  EVPLP_LIB=$lib python3 $ROOT/tools/quick_bench.py --iters 3 --vsl 2>&1 | grep "^iter" | sed -e 's/|.*gather/gather/' -e 's/splat.*//'
done
