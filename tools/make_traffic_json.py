"""profiles/traffic_*.json from the PMC summaries of tools/prof_all.sh (FETCH_SIZE / WRITE_SIZE passes).
usage: make_traffic_json.py <prof dir> <tag>"""
import json, os, sys
d, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ir = json.load(open(os.path.join(d, "pmc_ir_summary.json")))
gk = [k for k in ir if "gather_vpl_kernel" in k and "FETCH_SIZE" in ir[k]][-1]      # (round 4: a template, "void evplp::gather_vpl_kernel<true>")
g = ir[gk]
cut = next((ir[k] for k in ir if "gather_cut_kernel" in k and "FETCH_SIZE" in ir[k]), None)
f, w = g["FETCH_SIZE"] * 1024, g["WRITE_SIZE"] * 1024
if cut:      # the entry cuts are part of the gather: their scratch is written by the cut kernel and read by the walks
    f += cut["FETCH_SIZE"] * 1024; w += cut["WRITE_SIZE"] * 1024
json.dump({
    "config": "hard:1024x1024:1024:1",
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_all.sh {tag}) over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras`, last launch; summary profiles/{tag}_bench_ir_pmc.txt",
    "kernel": gk + (" + evplp::gather_cut_kernel" if cut else ""), "round": tag, "FETCH_SIZE_KB": g["FETCH_SIZE"], "WRITE_SIZE_KB": g["WRITE_SIZE"],
    "cut_kernel": ({"FETCH_SIZE_KB": cut["FETCH_SIZE"], "WRITE_SIZE_KB": cut["WRITE_SIZE"]} if cut else None),
    "hbm_bytes_per_launch": f + w, "hbm_bytes_per_launch_if_fetch_x2": 2 * f + w,
    "note": "FETCH_SIZE raw (MI355X_MICROARCH.md: it reads 1/2 of the bytes of wide 16 B/lane streams; this kernel reads 64-byte scalar node / leaf blocks, "
            "an uncalibrated width, so the x2 figure is given beside it); fabric-side counters, Infinity-Cache hits included (nodes + leaves = 41 MB stay resident). "
            "WRITE_SIZE = per-item partial sums (128 / splits_per_wave x 16 MB); round 4: + the entry cuts (256 B per (tile group, VPL slot) written by gather_cut_kernel, read once per tile "
            "by the walks through their LDS ring: four reads per slot). Algorithmic bytes per launch = 1024*1024*(64+16+16) + n_vpl*96 = 101 MB.",
    "l2_hit_rate": g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"]),
    "scalar_cache_hit_rate": g["SQC_DCACHE_HITS"] / (g["SQC_DCACHE_HITS"] + g["SQC_DCACHE_MISSES"]),
    "valu_insts": g["SQ_INSTS_VALU"], "salu_insts": g["SQ_INSTS_SALU"], "lds_insts": g["SQ_INSTS_LDS"],
    # share of the SIMD cycles of the launch in which a vector instruction executes: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the
    # waves; GRBM_GUI_ACTIVE is the launch's duration in cycles summed over the 8 XCDs; 1024 SIMDs
    "valu_active_frac": g["SQ_ACTIVE_INST_VALU"] * 4.0 / (g["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
    "valu_lane_utilisation": g["SQ_THREAD_CYCLES_VALU"] / (g["SQ_ACTIVE_INST_VALU"] * 64.0),
    "valu_insts_per_walk_note": "SQ_INSTS_VALU / (wave, VPL) walks of profiles/<round>_traversal_hard.json",
}, open(os.path.join(ROOT, "profiles", "traffic_gather_vpl.json"), "w"), indent=1)
configs = {}
for wl, key, nrec, px in (("evplp", "evplp:hard:1024x1024:1", 2000000, 1024 * 1024), ("ppm", "ppm:hard:1920x1080:1", 1200000, 1920 * 1080)):
    path = os.path.join(d, f"pmc_{wl}_summary.json")
    if not os.path.exists(path):
        continue
    ev = json.load(open(path))
    tot_f = tot_w = 0.0; per = {}
    # (round 5) the timed pass uses the proxy footprint (splat_tiles_kernel<.., true>); the ideal variants in the trace belong to bench.py's
    # comparison passes after the timed region and are listed but not summed
    for k, v in ev.items():
        if "splat" in k and "FETCH_SIZE" in v:
            per[k] = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"]}
            if "splat_tiles_kernel" in k and "true>" not in k:
                continue
            tot_f += v["FETCH_SIZE"] * 1024; tot_w += v["WRITE_SIZE"] * 1024
    configs[key] = {
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/prof_all.sh {tag}) over `python3 bench.py --workload {wl} --steps 2 --warmup 1 ...`, last launch of every splat kernel; summary profiles/{tag}_bench_{wl}_pmc.txt",
        "kernels": per, "hbm_bytes_per_pass": tot_f + tot_w, "hbm_bytes_per_pass_if_fetch_x2": 2 * tot_f + tot_w,
        "algorithmic_bytes_per_pass": nrec * 96 + px * (64 + 24),
    }
json.dump({"note": "FETCH_SIZE raw; MI355X_MICROARCH.md: it reads 1/2 of the bytes of wide 16 B/lane streams (the x2 figure is given beside it)", "configs": configs},
          open(os.path.join(ROOT, "profiles", "traffic_splat.json"), "w"), indent=1)
print("wrote profiles/traffic_gather_vpl.json, profiles/traffic_splat.json")
