#!/bin/bash
# PMC passes over bench.py (one counter group per run; never combined with trace domains).
# usage: tools/prof_pmc.sh <outdir> [bench args...]
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > $out/$name.log 2>&1; }
EXTRA="$*"
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
run b SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY
run c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32
run d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run e FETCH_SIZE
run f WRITE_SIZE
python3 tools/pmc_summary.py $out
