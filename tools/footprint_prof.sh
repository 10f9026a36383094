#!/bin/bash
# kernel statistics of the photon splat under both coverage rules (bench workloads evplp and ppm): tools/footprint_prof.sh <tag>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/${1:-footprint}; mkdir -p $O
cd /tmp
for wl in evplp ppm; do
  for fpr in ideal proxy; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${wl}_${fpr} -- python3 $ROOT/bench.py --workload $wl --footprint $fpr --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/kt_${wl}_${fpr}.log 2>&1
    f=$(find $O/kt_${wl}_${fpr} -name "*kernel_stats.csv" | head -1)
    echo "== $wl $fpr"; grep -E "splat|Name" $f | cut -d, -f1-5 | cut -c1-150
    cp $f $O/${wl}_${fpr}_kernel_stats.csv
    rm -rf $O/kt_${wl}_${fpr}
  done
done
