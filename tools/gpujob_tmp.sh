mkdir -p gpurun_out/r04e
timeout 500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04e/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/r04e/pytest_gpu.txt | tail -2
EVPLP_CUT_BYTES=50000000 timeout 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh.py tests/test_gpu_group.py -m gpu -x -q > gpurun_out/r04e/pytest_bands.txt 2>&1; grep -E "passed|failed" gpurun_out/r04e/pytest_bands.txt | tail -2
timeout 200 python3 bench.py --no-extras --no-cpu-baseline --steps 20 > gpurun_out/r04e/bench_ir_cuts.jsonl 2> gpurun_out/r04e/bench_ir_cuts.err
EVPLP_CUTS=0 timeout 200 python3 bench.py --no-extras --no-cpu-baseline --steps 20 > gpurun_out/r04e/bench_ir_nocuts.jsonl 2> gpurun_out/r04e/bench_ir_nocuts.err
timeout 200 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --scene easy > gpurun_out/r04e/bench_ir_cuts_easy.jsonl 2> /dev/null
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04e/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r04e/kt.log 2>&1
cd $R; f=$(find gpurun_out/r04e/kt -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04e/ir_kernel_stats.csv; rm -rf gpurun_out/r04e/kt; head -8 gpurun_out/r04e/ir_kernel_stats.csv | cut -c1-160
python3 - <<PY
import json
for f in ["bench_ir_cuts","bench_ir_nocuts","bench_ir_cuts_easy"]:
    try:
        d=json.loads(open("gpurun_out/r04e/"+f+".jsonl").read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"].get("kernel_ms"), d["value"])
    except Exception as e: print(f, "ERR", e)
PY
