#!/bin/bash
# kernel statistics of ONE rank's strip of an n-way partition (config #2): tools/strip_kernels.sh <tag> <n> <rank> <rows>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/$1; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $ROOT/tools/quick_bench.py --iters 4 --strip-count $2 --strip-rank $3 --strip-rows $4 > $O/kt_$2_$3_$4.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
echo "== n=$2 rank=$3 rows=$4"; head -8 $f | cut -d, -f1-4 | cut -c1-140; grep "^iter" $O/kt_$2_$3_$4.log | tail -2 | cut -c1-140
rm -rf $O/kt
