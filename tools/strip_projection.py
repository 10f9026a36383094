"""What would an N-way row-strip partition do?  A SINGLE-GPU PROJECTION, not a scaling measurement: every rank of the partition is run as
a strip context of its own, alone on the one GPU, and the HIP-event times of its passes are recorded -- primary rays, its share of the
light paths, entry cuts + gather, photon splat, composite.  The frame time of the partition is then max over ranks of the rank's sum;
what the projection cannot contain is the exchange (the in-place all-gather of the record buffers, 24 MB per rank at config #3; the
all-gather of the composited strips, 12 MB per frame at 1024 x 1024) and any interference between the GPUs of a node.
Writes profiles/r05_strip_projection.json.   usage: python tools/strip_projection.py [out.json]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evplp_amd as ev

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_strip_projection.json")
W = H = 1024; P = 4
jp = ev.synth_scene("/tmp/evplp_strips_hard", "conf", 331000, 1234, W, H, style="hard")
CONFIGS = {"cfg2_ir": dict(n_light=1024, n_vpl=1024, mis="one", splat=False), "cfg3_evplp": dict(n_light=500000, n_vpl=1024, mis="balance", splat=True)}
PASSES = (("primary", ev.PASS_PRIMARY), ("light_trace", ev.PASS_LIGHT_TRACE), ("gather", ev.PASS_GATHER_VPL), ("splat", ev.PASS_SPLAT), ("present", ev.PASS_RESOLVE))


def run_rank(cfg, n, rows, r):
    nl, nv = cfg["n_light"], cfg["n_vpl"]
    with ev.Context(W, H, nl, nv, P, strip_rank=r, strip_count=n, strip_rows=rows) as c:
        c.load_scene_json(jp)
        cam = c.camera(); bsr, total, _ = c.scene_metrics()
        radius = 0.003 * bsr
        kw = dict(camera_pos=list(cam.origin), mis_mode=cfg["mis"], pdf_mc=(nv / nl) / math.pi / radius ** 2, clamping_value=1.0 / total, photon_radius=radius,
                  num_light_paths=nl, num_vpl_light_paths=nv, photons_per_path=P, do_accumulate=1, splat_footprint="proxy")
        split = n > 1 and nl % n == 0 and nl >= 16384          # evplp_group's rule: large path sets are traced 1 / n per rank and all-gathered
        acc = {k: [] for k, _ in PASSES}
        for it in range(4):
            jitter = tuple(float(v) for v in ev.jitter_sequence(0, it + 1, W, H)[it])
            c.trace_light_paths(it)                              # the whole record set (what the exchange would deliver)
            c.synchronize()
            if split:
                c.trace_light_paths(it, r * (nl // n), nl // n)  # ... and, timed, this rank's share of it (same records)
                c.synchronize()
            lt = c.pass_stats(ev.PASS_LIGHT_TRACE)["ms"]
            c.primary(jitter)
            c.gather_vpl(ev.frame_params(**kw, jitter=jitter))
            if cfg["splat"]:
                c.splat_photons(ev.frame_params(**kw, jitter=jitter))
            c.present(1.0, 1.0, 1.0, mask_emitter=True, gamma=True)
            c.synchronize()
            if it >= 1:
                for k, p in PASSES:
                    acc[k].append(lt if k == "light_trace" else (c.pass_stats(p)["ms"] if (k != "splat" or cfg["splat"]) else 0.0))
        return {k: sum(v) / len(v) for k, v in acc.items()}


result = {"what": "single-GPU projection of the row-strip partition (every rank's strip run alone on one MI355X; per-pass HIP-event times); NOT a scaling measurement: "
                  "no exchange, no second device was involved", "scene": "furnished conference stand-in, 331 k triangles, 1024 x 1024", "configs": {}}
for name, cfg in CONFIGS.items():
    base = run_rank(cfg, 1, 16, 0)
    base_sum = sum(base.values())
    entry = {"one_gpu": {"passes_ms": base, "frame_ms": base_sum}, "partitions": []}
    print(name, "1 GPU:", {k: round(v, 3) for k, v in base.items()}, "sum %.3f" % base_sum, flush=True)
    for n in (2, 4, 8):
        for rows in [int(v) for v in os.environ.get("PROJ_ROWS", "8,16").split(",")]:
            ranks = [run_rank(cfg, n, rows, r) for r in range(n)]
            sums = [sum(x.values()) for x in ranks]
            rec = {"n": n, "strip_rows": rows, "per_rank_passes_ms": ranks, "per_rank_frame_ms": sums, "max_ms": max(sums), "mean_ms": sum(sums) / n,
                   "balance": (sum(sums) / n) / max(sums), "projected_speedup_without_exchange": base_sum / max(sums)}
            entry["partitions"].append(rec)
            print(f"  {name} n={n} rows={rows}: per-rank frame ms {[round(s, 2) for s in sums]} max {max(sums):.2f} balance {rec['balance']:.3f} projected x{rec['projected_speedup_without_exchange']:.2f}", flush=True)
    result["configs"][name] = entry
json.dump(result, open(out_path, "w"), indent=1)
print("wrote", out_path)
