#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q --durations=12 > $O/pytest_cfg.log 2>&1; echo "rc=$?" >> $O/pytest_cfg.log
tail -40 $O/pytest_cfg.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_baseline_configs.py > $O/pytest_rest.log 2>&1; echo "rc=$?" >> $O/pytest_rest.log
tail -5 $O/pytest_rest.log
