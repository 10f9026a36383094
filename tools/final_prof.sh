export TMPDIR=/tmp
mkdir -p gpurun_out/final
python bench.py --steps 5 --warmup 1 > gpurun_out/final/bench_ir.jsonl 2> gpurun_out/final/bench_ir.err
python bench.py --steps 5 --warmup 1 --workload evplp --no-cpu-baseline > gpurun_out/final/bench_evplp.jsonl 2> gpurun_out/final/bench_evplp.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_ir -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/final/prof_ir.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_evplp -- python3 bench.py --steps 5 --warmup 1 --workload evplp --no-cpu-baseline > gpurun_out/final/prof_evplp.log 2>&1
bash tools/prof_pmc.sh gpurun_out/final/pmc_ir > gpurun_out/final/pmc_ir.txt 2>&1
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*_agent_info.csv" -delete
cat gpurun_out/final/bench_ir.jsonl gpurun_out/final/bench_evplp.jsonl
tail -12 gpurun_out/final/pmc_ir.txt
