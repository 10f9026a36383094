#!/bin/bash
# PMC passes (one counter group per run, never combined with trace domains) + kernel statistics for ONE bench workload.
# usage: tools/prof_pmc_wl.sh <workload> <tag> [bench.py args...]   -> gpurun_out/pmc_<workload>/<tag>_bench_<workload>_*.{txt,csv}
wl=$1; tag=$2; shift 2
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/pmc_$wl; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_bench_${wl}_kernel_stats.csv
rm -rf $O/kt
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${wl}_$name -- python3 $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > $O/pmc_${wl}_$name.log 2>&1; }
EXTRA="$*"
pmc a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
pmc b SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32
pmc d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pmc e FETCH_SIZE
pmc f WRITE_SIZE
pmc g SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN
cd $ROOT
python3 tools/pmc_summary.py $O $wl > $O/${tag}_bench_${wl}_pmc.txt 2>&1
find $O -name "*_agent_info.csv" -delete
head -30 $O/${tag}_bench_${wl}_pmc.txt | cut -c1-220
