#!/bin/bash
# gather time of several builds of the library side by side (EVPLP_LIB): tools/lib_ab.sh <tag> <variant suffixes... | full>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/$1; mkdir -p $O; shift
for v in "$@"; do
  lib=$ROOT/evplp_amd/lib/libevplp_hip_$v.so; [ "$v" = "full" ] && lib=$ROOT/evplp_amd/lib/libevplp_hip.so
  for style in hard easy; do
    echo "== $v $style"
    EVPLP_LIB=$lib python3 $ROOT/tools/quick_bench.py --iters 4 --style $style 2>&1 | grep "^iter" | sed -e 's/|.*gather/gather/' -e 's/splat.*//'
  done
done
