#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2n
timeout 900 python3 tools/debug_vis.py > gpurun_out/r2n/debug_cfg4.log 2>&1
tail -30 gpurun_out/r2n/debug_cfg4.log
