"""Contiguous bands dealt by measured cost (evplp_group's EVPLP_PARTITION_BANDS + evplp_group_rebalance) -- a SINGLE-GPU PROJECTION, not a
scaling measurement: the n ranks' band contexts live on the one GPU and are run ONE AFTER THE OTHER (a rank alone on the device, HIP-event
times of its passes: primary rays, its share of the light paths, cuts + gather, photon splat with the proxy footprint, composite); after
every frame the boundaries move as evplp_group_rebalance moves them (each rank's time spread evenly over its rows, n parts of equal cost,
multiples of 16 rows, twice the equal share at most).  Frame time of the partition = the slowest rank's sum; no exchange, no second device.
Writes profiles/r05_band_projection.json.   usage: python tools/band_projection.py [out.json]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evplp_amd as ev

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_band_projection.json")
W = H = 1024; P = 4; ROUNDS = 5
jp = ev.synth_scene("/tmp/evplp_strips_hard", "conf", 331000, 1234, W, H, style="hard")
CONFIGS = {"cfg2_ir": dict(n_light=1024, n_vpl=1024, mis="one", splat=False), "cfg3_evplp": dict(n_light=500000, n_vpl=1024, mis="balance", splat=True)}
PASSES = (("primary", ev.PASS_PRIMARY), ("light_trace", ev.PASS_LIGHT_TRACE), ("gather", ev.PASS_GATHER_VPL), ("splat", ev.PASS_SPLAT), ("present", ev.PASS_RESOLVE))


def frame_of(c, cfg, n, r, it):
    nl, nv = cfg["n_light"], cfg["n_vpl"]
    cam = c.camera(); bsr, total, _ = c.scene_metrics()
    radius = 0.003 * bsr
    kw = dict(camera_pos=list(cam.origin), mis_mode=cfg["mis"], pdf_mc=(nv / nl) / math.pi / radius ** 2, clamping_value=1.0 / total, photon_radius=radius,
              num_light_paths=nl, num_vpl_light_paths=nv, photons_per_path=P, do_accumulate=1, splat_footprint="proxy")
    split = n > 1 and nl % n == 0 and nl >= 16384
    jitter = tuple(float(v) for v in ev.jitter_sequence(0, it + 1, W, H)[it])
    c.trace_light_paths(it); c.synchronize()
    if split:
        c.trace_light_paths(it, r * (nl // n), nl // n); c.synchronize()
    lt = c.pass_stats(ev.PASS_LIGHT_TRACE)["ms"]
    c.primary(jitter)
    c.gather_vpl(ev.frame_params(**kw, jitter=jitter))
    if cfg["splat"]:
        c.splat_photons(ev.frame_params(**kw, jitter=jitter))
    c.present(1.0, 1.0, 1.0, mask_emitter=True, gamma=True)
    c.synchronize()
    return {k: (lt if k == "light_trace" else (c.pass_stats(p)["ms"] if (k != "splat" or cfg["splat"]) else 0.0)) for k, p in PASSES}


def rebalance(first, cost, cap):      # evplp_group_rebalance (group.cpp), restated
    n = len(cost); total = sum(cost)

    def row_at(target):
        acc = 0.0
        for r in range(n):
            if acc + cost[r] >= target or r == n - 1:
                return first[r] + ((target - acc) / cost[r] if cost[r] > 0 else 0.0) * (first[r + 1] - first[r])
            acc += cost[r]
        return float(H)
    new = [0] * (n + 1); new[n] = H
    for r in range(1, n):
        y = int(row_at(total * r / n) / 16.0 + 0.5) * 16
        y = max(y, new[r - 1] + 16); y = min(y, new[r - 1] + cap // 16 * 16); y = min(y, (H - 1) // 16 * 16 - (n - 1 - r) * 16)
        new[r] = y
    for r in range(n - 1, 0, -1):
        new[r] = max(new[r], (new[r + 1] + 15) // 16 * 16 - cap // 16 * 16)
    return new


result = {"what": "single-GPU projection of evplp_group's bands partition (contiguous bands dealt by measured cost; every rank's band run alone on one MI355X, "
                  "per-pass HIP-event times); NOT a scaling measurement: no exchange, no second device was involved",
          "scene": "furnished conference stand-in, 331 k triangles, 1024 x 1024", "configs": {}}
for name, cfg in CONFIGS.items():
    with ev.Context(W, H, cfg["n_light"], cfg["n_vpl"], P) as c:
        c.load_scene_json(jp)
        for it in range(3):
            base = frame_of(c, cfg, 1, 0, it)
    base_sum = sum(base.values())
    entry = {"one_gpu": {"passes_ms": base, "frame_ms": base_sum}, "partitions": []}
    print(name, "1 GPU: sum %.3f" % base_sum, flush=True)
    for n in (2, 4, 8):
        share = max(16, (H // n + 15) // 16 * 16); cap = min(H, 2 * share)
        first = [min(r * share, H) for r in range(n + 1)]; first[n] = H
        ctxs = []
        for r in range(n):
            c = ev.Context(W, H, cfg["n_light"], cfg["n_vpl"], P, band=(first[r], first[r + 1] - first[r]), band_capacity_rows=cap)
            c.load_scene_json(jp); ctxs.append(c)
        rounds = []
        for rd in range(ROUNDS):
            per = []
            for r, c in enumerate(ctxs):
                frame_of(c, cfg, n, r, 0)                       # (first frame after a band moved: allocations, eye cuts)
                per.append(frame_of(c, cfg, n, r, 1 + rd))
            sums = [sum(x.values()) for x in per]
            rounds.append({"band_first_rows": list(first), "per_rank_frame_ms": sums, "max_ms": max(sums), "balance": (sum(sums) / n) / max(sums),
                           "sum_ms": sum(sums), "projected_speedup_without_exchange": base_sum / max(sums), "per_rank_passes_ms": per})
            print(f"  {name} n={n} round {rd}: bands {first} per-rank ms {[round(s, 2) for s in sums]} max {max(sums):.2f} balance {rounds[-1]['balance']:.3f} x{base_sum / max(sums):.2f}", flush=True)
            first = rebalance(first, sums, cap)
            for r, c in enumerate(ctxs):
                c.set_band(first[r], (first[r + 1] if r + 1 < n else (H + 15) // 16 * 16) - first[r])
        for c in ctxs:
            c.close()
        entry["partitions"].append({"n": n, "rounds": rounds, "best_projected_speedup": max(x["projected_speedup_without_exchange"] for x in rounds)})
    result["configs"][name] = entry
json.dump(result, open(out_path, "w"), indent=1)
print("wrote", out_path)
