#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_group.log 2>&1; echo "rc=$?" >> $O/pytest_group.log
tail -30 $O/pytest_group.log
timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_group.py --deselect tests/test_gpu_parity.py > $O/pytest_rest.log 2>&1; echo "rc=$?" >> $O/pytest_rest.log
tail -5 $O/pytest_rest.log
