#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
for w in 4 3; do for sc in easy hard; do
EVPLP_LIB=$PWD/evplp_amd/lib/libevplp_hip_bw$w.so timeout 600 python3 bench.py --steps 3 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/bench_${sc}_bw$w.jsonl 2> $O/bench_${sc}_bw$w.err
python3 -c "
import json,sys
d=json.loads(open('$O/bench_${sc}_bw$w.jsonl').read().strip().splitlines()[-1]); print('$sc beam waves $w value',round(d['value']),'ms',round(d['ms_per_step'],2),'kernel_ms',round(d['roofline']['kernel_ms'],2))"
done; done
