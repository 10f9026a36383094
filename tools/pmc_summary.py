"""Summarise the PMC passes of tools/prof_all.sh: last (timed) launch of every evplp kernel.  usage: pmc_summary.py <dir> <workload>"""
import csv, glob, collections, json, sys
out, wl = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for f in glob.glob(f"{out}/pmc_{wl}_*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "evplp::" in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            res[k].setdefault("_regs", (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"]))
    for (k, c), v in agg.items():
        res[k][c] = v[-1]      # last launch (timed step)
json.dump(res, open(f"{out}/pmc_{wl}_summary.json", "w"), indent=1)
g0 = lambda d, n: d.get(n, float("nan"))
for k, d in sorted(res.items(), key=lambda kv: -g0(kv[1], "SQ_WAVE_CYCLES") if g0(kv[1], "SQ_WAVE_CYCLES") == g0(kv[1], "SQ_WAVE_CYCLES") else 0):
    if not any(t in k for t in ("gather", "splat", "light_trace", "primary", "path_trace", "compact_vpl", "resolve")): continue
    print("==", k, "(VGPR, SGPR, LDS, grid, wg) =", d.get("_regs"))
    g = lambda n: g0(d, n)
    wc = g("SQ_WAVE_CYCLES")
    print(" waves %.0f  wave_cycles(quad) %.4g  busy_cycles %.4g  gui_active %.4g  level_waves %.4g" % (g("SQ_WAVES"), wc, g("SQ_BUSY_CYCLES"), g("GRBM_GUI_ACTIVE"), g("SQ_LEVEL_WAVES")))
    print(" insts: VALU %.4g  SALU %.4g  SMEM %.4g  LDS %.4g  BRANCH %.4g  TRANS %.4g | VALU lane util %.3f" % (g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_SMEM"), g("SQ_INSTS_LDS"), g("SQ_INSTS_BRANCH"), g("SQ_INSTS_VALU_TRANS_F32"), g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64 + 1e-9)))
    print(" frac of wave cycles: wait_any %.3f wait_inst_any %.3f active_valu %.3f active_sca %.3f active_lds %.3f active_any %.3f" % (g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_SCA") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_ANY") / wc))
    print(" scalar cache: req %.4g hits %.4g misses %.4g hit rate %.4f" % (g("SQC_DCACHE_REQ"), g("SQC_DCACHE_HITS"), g("SQC_DCACHE_MISSES"), g("SQC_DCACHE_HITS") / (g("SQC_DCACHE_HITS") + g("SQC_DCACHE_MISSES") + 1e-9)))
    print(" L2: hit %.4g miss %.4g rate %.4f | FETCH_SIZE KB %.5g (x2 for wide streams: MI355X_MICROARCH.md) WRITE_SIZE KB %.5g" % (g("TCC_HIT_sum"), g("TCC_MISS_sum"), g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum") + 1e-9), g("FETCH_SIZE"), g("WRITE_SIZE")))
    if g("SQ_LDS_IDX_ACTIVE") == g("SQ_LDS_IDX_ACTIVE"):
        print(" LDS: active cycles %.4g  bank-conflict cycles %.4g (%.3f of active)  address-conflict cycles %.4g  atomic-return cycles %.4g" % (
            g("SQ_LDS_IDX_ACTIVE"), g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_BANK_CONFLICT") / (g("SQ_LDS_IDX_ACTIVE") + 1e-9), g("SQ_LDS_ADDR_CONFLICT"), g("SQ_LDS_ATOMIC_RETURN")))
