#!/bin/bash
# Round profile: rocprofv3 kernel statistics for the bench workloads + PMC passes (one counter group per run, never combined
# with trace domains).  usage: tools/prof_all.sh <tag>   -> gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
tag=${1:-r02}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp
for wl in ir evplp ppm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$wl -- python3 $ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/kt_$wl.log 2>&1
  f=$(find $O/kt_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_bench_${wl}_kernel_stats.csv
  find $O/kt_$wl -name "*kernel_trace.csv" -delete; find $O/kt_$wl -name "*_agent_info.csv" -delete
done
pmc() { wl=$1; name=$2; shift 2; rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${wl}_$name -- python3 $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_${wl}_$name.log 2>&1; }
for wl in ir evplp ppm; do
  pmc $wl a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
  pmc $wl b SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY
  pmc $wl c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32
  pmc $wl d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  pmc $wl e FETCH_SIZE
  pmc $wl f WRITE_SIZE
  pmc $wl g SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN
done
cd $ROOT
python3 tools/pmc_summary.py $O ir > $O/${tag}_bench_ir_pmc.txt 2>&1
python3 tools/pmc_summary.py $O evplp > $O/${tag}_bench_evplp_pmc.txt 2>&1
python3 tools/pmc_summary.py $O ppm > $O/${tag}_bench_ppm_pmc.txt 2>&1
find $O -name "*_agent_info.csv" -delete
cat $O/${tag}_bench_ir_pmc.txt $O/${tag}_bench_evplp_pmc.txt | head -60
for wl in ir evplp ppm; do head -12 $O/${tag}_bench_${wl}_kernel_stats.csv | cut -c1-200; done
