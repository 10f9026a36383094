#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
for rep in 1 2; do for sc in hard easy; do for lib in "" _gw7; do for blk in 3 0; do
L=$PWD/evplp_amd/lib/libevplp_hip$lib.so
EVPLP_LIB=$L EVPLP_TILE_BLOCK_LOG2=$blk timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/b.jsonl 2> $O/b.err
python3 -c "
import json,sys
d=json.loads(open('$O/b.jsonl').read().strip().splitlines()[-1]); print('rep $rep $sc lib=$lib block_log2=$blk kernel_ms',round(d['roofline']['kernel_ms'],2))"
done; done; done; done
