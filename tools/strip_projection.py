"""What would an N-way row-strip partition do?  A SINGLE-GPU PROJECTION, not a scaling measurement: every rank of the partition is run as
a strip context of its own, alone on the one GPU, and the HIP-event times of its passes are recorded -- primary rays, its share of the
light paths, entry cuts + gather, photon splat, composite.  The frame time of the partition is then max over ranks of the rank's sum;
what the projection cannot contain is the exchange (the in-place all-gather of the record buffers, 24 MB per rank at config #3; the
all-gather of the composited strips, 12 MB per frame at 1024 x 1024) and any interference between the GPUs of a node.

Two deals of the same blocks: "roundRobin" (block b to rank b % n) and "cost" -- what evplp_group_rebalance does: the ranks of the
round-robin deal clock their blocks in a calibration frame (evplp_calibrate_blocks), evplp_deal_blocks deals them (longest processing
time first + pairwise improvement, capacity 150 % of the equal share), every rank is run again with its table (evplp_set_blocks).

usage: python tools/strip_projection.py [out.json] [--configs cfg2_ir,cfg3_evplp,cfg4_ppm,cfg5_vsl] [--rows 8,16] [--n 2,4,8]"""
import argparse, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import evplp_amd as ev
from evplp_amd import strips

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default=os.path.join(ROOT, "profiles", "r06_strip_projection.json"))
ap.add_argument("--configs", default="cfg2_ir,cfg3_evplp")
ap.add_argument("--rows", default="8,16")
ap.add_argument("--n", default="")
ap.add_argument("--deals", default="roundRobin,cost")
ap.add_argument("--capacity-pct", type=int, default=150)
ap.add_argument("--wall", action="store_true", help="also time every rank's FRAMES by the wall clock with light tracing overlapped (as the technique loop and bench.py run them): frame_wall_ms")
ap.add_argument("--cal-fraction", type=int, default=1, help="the calibration frame gathers 1 / this many of the VPL light paths (the technique loop: 8)")
ap.add_argument("--rounds", type=int, default=1, help="calibrate + deal this many times (every round clocks the blocks under the previous round's deal)")
ap.add_argument("--split-light-paths", default="auto", choices=["auto", "on", "off"], help="as evplp_group_config.split_light_paths (auto: evplp_group_split_model)")
args = ap.parse_args()
P = 4
CONFIGS = {
    "cfg2_ir": dict(W=1024, H=1024, n_light=1024, n_vpl=1024, mis="one", splat=False, gather="vpl", n=(2, 4, 8), frames=4),
    "cfg3_evplp": dict(W=1024, H=1024, n_light=500000, n_vpl=1024, mis="balance", splat=True, gather="vpl", n=(2, 4, 8), frames=4),
    # BASELINE config #4: progressive photon mapping, 1920 x 1080, 300 000 light paths, no gather, 4 GPUs
    "cfg4_ppm": dict(W=1920, H=1080, n_light=300000, n_vpl=0, mis="one", splat=True, gather=None, n=(4,), frames=12),
    # BASELINE config #5: progressive VSL gather (16 384 slots) + photon splat, 2048 x 2048, 8 GPUs
    "cfg5_vsl": dict(W=2048, H=2048, n_light=300000, n_vpl=4096, mis="one", splat=True, gather="vsl", n=(8,), frames=3),
}
PASSES = (("primary", ev.PASS_PRIMARY), ("light_trace", ev.PASS_LIGHT_TRACE), ("gather", None), ("splat", ev.PASS_SPLAT), ("present", ev.PASS_RESOLVE))
scene_cache = {}


def scene(W, H):
    if (W, H) not in scene_cache:
        scene_cache[(W, H)] = ev.synth_scene(f"/tmp/evplp_strips_hard_{W}x{H}", "conf", 331000, 1234, W, H, style="hard")
    return scene_cache[(W, H)]


def run_rank(cfg, n, rows, r, blocks=None, calibrate=False, cap_rows=0):
    """One rank's strip context alone on the GPU: mean HIP-event time of every pass over the frames after the first; with `calibrate` also
    the per-image-block cost its gather kernels clocked."""
    W, H, nl, nv = cfg["W"], cfg["H"], cfg["n_light"], cfg["n_vpl"]
    with ev.Context(W, H, nl, nv, P, strip_rank=r, strip_count=n, strip_rows=rows, strip_capacity_rows=cap_rows if n > 1 else 0,
                    overlap_light_tracing=False) as c:
        c.load_scene_json(scene(W, H))
        if blocks is not None:
            c.set_blocks(blocks)
        cam = c.camera(); bsr, total, _ = c.scene_metrics()
        radius = 0.003 * bsr
        vsl_r = max(0.05 * bsr, 0.008)
        nvp = nv
        kw = dict(camera_pos=list(cam.origin), mis_mode=cfg["mis"], pdf_mc=(nv / nl) / math.pi / radius ** 2, clamping_value=1.0 / total, photon_radius=radius,
                  vsl_radius=vsl_r, vsl_inv_pi_radius2=1.0 / (math.pi * vsl_r * vsl_r),
                  num_light_paths=nl, num_vpl_light_paths=nvp, photons_per_path=P, do_accumulate=1, splat_footprint="proxy")
        # evplp_group's rule: every rank traces all paths, or -- where its cost model expects that to be faster, or on request -- 1 / n of them + all-gather
        split = n > 1 and nl % n == 0 and (args.split_light_paths == "on" or (args.split_light_paths == "auto" and ev.split_model(nl, P, n)[0]))
        acc = {k: [] for k, _ in PASSES}
        if calibrate:
            c.calibrate_blocks(True)
        for it in range(cfg["frames"]):
            jitter = tuple(float(v) for v in ev.jitter_sequence(0, it + 1, W, H)[it])
            c.trace_light_paths(it)                              # the whole record set (what the exchange would deliver)
            c.synchronize()
            if split:
                c.trace_light_paths(it, r * (nl // n), nl // n)  # ... and, timed, this rank's share of it (same records)
                c.synchronize()
            lt = c.pass_stats(ev.PASS_LIGHT_TRACE)["ms"]
            c.primary(jitter)
            fp = ev.frame_params(**kw, jitter=jitter, rng_seed=it)
            if calibrate and args.cal_fraction > 1 and cfg["gather"]:
                fp.num_vpl_light_paths = max(nv // args.cal_fraction, min(nv, 64))
            if cfg["gather"] == "vpl":
                c.gather_vpl(fp)
            elif cfg["gather"] == "vsl":
                c.gather_vsl(fp)
            if cfg["splat"]:
                c.splat_photons(fp)
            c.present(1.0, 1.0, 1.0, mask_emitter=True, gamma=True)
            c.synchronize()
            if it >= 1:
                for k, p in PASSES:
                    if k == "light_trace":
                        v = lt
                    elif k == "gather":
                        v = c.pass_stats(ev.PASS_GATHER_VSL if cfg["gather"] == "vsl" else ev.PASS_GATHER_VPL)["ms"] if cfg["gather"] else 0.0
                    elif k == "splat":
                        v = c.pass_stats(p)["ms"] if cfg["splat"] else 0.0
                    else:
                        v = c.pass_stats(p)["ms"]
                    acc[k].append(v)
        out = {k: sum(v) / len(v) for k, v in acc.items()}
        cost = c.block_costs() if (calibrate and cfg["gather"]) else None
    if args.wall and not calibrate:
        # the same rank again as the loops run it: light tracing on its second stream beside the G-buffer pass, no per-pass waits; wall clock over 6 frames
        import time
        with ev.Context(W, H, nl, nv, P, strip_rank=r, strip_count=n, strip_rows=rows, strip_capacity_rows=cap_rows if n > 1 else 0, overlap_light_tracing=True) as c:
            c.load_scene_json(scene(W, H))
            if blocks is not None:
                c.set_blocks(blocks)

            def frame(it):
                jitter = tuple(float(v) for v in ev.jitter_sequence(0, it + 1, W, H)[it])
                fp = ev.frame_params(**kw, jitter=jitter, rng_seed=it)
                if cfg["gather"]:
                    c.primary(jitter); c.trace_light_paths(it, r * (nl // n), nl // n) if split else c.trace_light_paths(it)
                else:
                    (c.trace_light_paths(it, r * (nl // n), nl // n) if split else c.trace_light_paths(it)); c.primary(jitter)
                if cfg["gather"] == "vpl":
                    c.gather_vpl(fp)
                elif cfg["gather"] == "vsl":
                    c.gather_vsl(fp)
                if cfg["splat"]:
                    c.splat_photons(fp)
                c.present(1.0, 1.0, 1.0, mask_emitter=True, gamma=True)
            nfr = 3 if cfg["gather"] == "vsl" else 6
            for it in range(2):
                frame(it)
            c.synchronize(); t0 = time.perf_counter()
            for it in range(nfr):
                frame(2 + it)
            c.synchronize()
            out["frame_wall"] = (time.perf_counter() - t0) / nfr * 1e3
    return out, cost


def partition_record(n, rows, deal, ranks, base_sum, owner=None, base_wall=None):
    walls = [x.pop("frame_wall") for x in ranks] if all("frame_wall" in x for x in ranks) else None
    sums = [sum(x.values()) for x in ranks]
    rec = {"n": n, "strip_rows": rows, "deal": deal, "per_rank_passes_ms": ranks, "per_rank_frame_ms": sums, "max_ms": max(sums), "mean_ms": sum(sums) / n,
           "sum_ms": sum(sums), "balance": (sum(sums) / n) / max(sums), "projected_speedup_without_exchange": base_sum / max(sums)}
    if walls is not None and base_wall:
        rec["per_rank_frame_wall_ms"] = walls; rec["projected_speedup_wall_without_exchange"] = base_wall / max(walls); rec["balance_wall"] = (sum(walls) / n) / max(walls)
    if owner is not None:
        rec["blocks_per_rank"] = np.bincount(owner, minlength=n).tolist(); rec["owner"] = [int(v) for v in owner]
    return rec


result = {"what": "single-GPU projection of the row-strip partition (every rank's strip run alone on one MI355X; per-pass HIP-event times); NOT a scaling measurement: "
                  "no exchange, no second device was involved", "scene": "furnished conference stand-in, 331 k triangles", "capacity_pct": args.capacity_pct, "split_light_paths": args.split_light_paths, "configs": {}}
for name in args.configs.split(","):
    cfg = CONFIGS[name]
    base, _ = run_rank(cfg, 1, 16, 0)
    base_wall = base.pop("frame_wall", None)
    base_sum = sum(base.values())
    entry = {"resolution": [cfg["W"], cfg["H"]], "one_gpu": {"passes_ms": base, "frame_ms": base_sum, "frame_wall_ms": base_wall}, "partitions": []}
    print(name, "1 GPU:", {k: round(v, 3) for k, v in base.items()}, "sum %.3f" % base_sum, ("wall %.3f" % base_wall) if base_wall else "", flush=True)
    for n in ([int(v) for v in args.n.split(",")] if args.n else cfg["n"]):
        for rows in [int(v) for v in args.rows.split(",")]:
            nb = (cfg["H"] + rows - 1) // rows
            cap = min(nb, (-(-nb // n) * args.capacity_pct + 99) // 100)
            want_cost = "cost" in args.deals and cfg["gather"] is not None
            if "roundRobin" in args.deals:
                rr = [run_rank(cfg, n, rows, r, cap_rows=cap * rows) for r in range(n)]
                rec = partition_record(n, rows, "roundRobin", [x[0] for x in rr], base_sum, base_wall=base_wall)
                entry["partitions"].append(rec)
                print(f"  {name} n={n} rows={rows} roundRobin: per-rank ms {[round(s, 2) for s in rec['per_rank_frame_ms']]} max {rec['max_ms']:.2f} sum {rec['sum_ms']:.1f} "
                      f"balance {rec['balance']:.3f} projected x{rec['projected_speedup_without_exchange']:.2f}" + (f" | wall: max {max(rec['per_rank_frame_wall_ms']):.2f} ms x{rec['projected_speedup_wall_without_exchange']:.2f}" if 'per_rank_frame_wall_ms' in rec else ""), flush=True)
            if want_cost:
                # the calibration frame of every rank under the round-robin deal (self-clocking kernels; its pass times are not used)
                cal = dict(cfg); cal["frames"] = 2
                cost = sum(run_rank(cal, n, rows, r, calibrate=True, cap_rows=cap * rows)[1] for r in range(n))
                owner = ev.deal_blocks(cost, n, cap)
                for _ in range(args.rounds - 1):      # again, under the deal just made
                    cost = sum(run_rank(cal, n, rows, r, blocks=strips.blocks_of_rank(owner, r, cost), calibrate=True, cap_rows=cap * rows)[1] for r in range(n))
                    owner = ev.deal_blocks(cost, n, cap)
                dealt = [run_rank(cfg, n, rows, r, blocks=strips.blocks_of_rank(owner, r, cost), cap_rows=cap * rows)[0] for r in range(n)]
                rec = partition_record(n, rows, "cost", dealt, base_sum, owner, base_wall=base_wall)
                rec["block_cost_ticks"] = [int(v) for v in cost]
                entry["partitions"].append(rec)
                print(f"  {name} n={n} rows={rows} cost      : per-rank ms {[round(s, 2) for s in rec['per_rank_frame_ms']]} max {rec['max_ms']:.2f} sum {rec['sum_ms']:.1f} "
                      f"balance {rec['balance']:.3f} projected x{rec['projected_speedup_without_exchange']:.2f} blocks {rec['blocks_per_rank']}" + (f" | wall: max {max(rec['per_rank_frame_wall_ms']):.2f} ms x{rec['projected_speedup_wall_without_exchange']:.2f}" if 'per_rank_frame_wall_ms' in rec else ""), flush=True)
    result["configs"][name] = entry
    json.dump(result, open(args.out, "w"), indent=1)
print("wrote", args.out)
