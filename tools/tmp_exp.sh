export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_q -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_q/*/*_counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "gather_vpl" in r["Kernel_Name"]]
    last=max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"])==last: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print(dict(agg))
PY
