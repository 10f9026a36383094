#!/bin/bash
# PMC passes over the VSL gather (tools/quick_bench.py --vsl); one counter group per run.
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/quick_bench.py --res 512 --paths 256 --vpl-paths 256 --vsl --iters 1 > $out/$name.log 2>&1; }
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
run b SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY
run c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32
run g SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
run e FETCH_SIZE
run f WRITE_SIZE
python3 tools/pmc_summary.py $out
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d.items():
    if "vsl" in k: print({a:b for a,b in v.items() if a!="_regs"})
PY
