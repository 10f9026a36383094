"""Developer probe: traversal counters of the VPL gather on the bench scene (nodes / leaf blocks / triangle pairs per
(wave, VPL) walk and per ray).  Needs the diagnostic build:  make stats  ->  evplp_amd/lib/libevplp_hip_stats.so.

    EVPLP_LIB=evplp_amd/lib/libevplp_hip_stats.so python3 tools/traversal_stats.py [--scene easy|hard]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EVPLP_LIB", os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_stats.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import evplp_amd as ev

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--tris", type=int, default=331000)
ap.add_argument("--paths", type=int, default=1024)
ap.add_argument("--scene", default="easy")
ap.add_argument("--out", default="")
ap.add_argument("--k", type=int, default=0)
ap.add_argument("--order", default="record", help="record | morton (Morton order per split: VPLs i and i + 128 are neighbours)")
a = ap.parse_args()
d = "/tmp/evplp_stats_%s" % a.scene
jp = ev.synth_scene(d, "conf", a.tris, 1234, a.res, a.res, style=a.scene)
P = 4
c = ev.Context(a.res, a.res, a.paths, a.paths, P, gather_splits_per_wave=a.k)
c.load_scene_json(jp)
cam = c.camera()
fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=a.paths, num_vpl_light_paths=a.paths, photons_per_path=P, do_accumulate=0)
c.primary((0, 0)); c.trace_light_paths(0)
if a.order == "morton":
    rec = c.download(ev.BUF_RECORDS).copy()
    raw = rec.view(np.uint8).reshape(-1, 96)
    pos = raw[:, 0:12].copy().view(np.float32).reshape(-1, 3)
    flags = raw[:, 12:16].copy().view(np.uint32).reshape(-1)
    usable = (flags & 1) != 0
    idx = np.nonzero(usable)[0]
    p = pos[idx]; lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.uint64), 0, 1023)
    def ex(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
    srt = idx[np.argsort((ex(q[:, 0]) << 2) | (ex(q[:, 1]) << 1) | ex(q[:, 2]), kind="stable")]
    n = len(srt); per = (n + 127) // 128
    il = np.full(per * 128, -1, np.int64)
    for s_ in range(128):
        seg = srt[s_ * per:(s_ + 1) * per]; il[s_:s_ + 128 * len(seg):128] = seg
    il = il[il >= 0]
    o = np.arange(len(raw)); o[:len(il)] = il; o[len(il):] = np.nonzero(~usable)[0]
    c.upload(ev.BUF_RECORDS, raw[o].reshape(-1).view(rec.dtype).reshape(rec.shape))
c.gather_vpl(fp); c.synchronize()
st = c.pass_stats(ev.PASS_GATHER_VPL)
raw = c.debug_counters(ev.PASS_GATHER_VPL)
rays, nodes = int(st["rays"]), int(raw[1])
hist = raw[4:4 + 32].astype(np.int64)
walks, pairs, all_occ = int(raw[4 + 32]), int(raw[4 + 33]), int(raw[4 + 34])
out = {
    "scene": a.scene, "accel": c.accel_info(), "usable_vpls": st["usable"], "rays": rays, "walks": walks,
    "wave_node_visits": nodes, "node_visits_per_walk": nodes / max(walks, 1), "synthetic_node_visits_per_walk": int(raw[3]) / max(walks, 1),
    "entry_cuts": os.environ.get("EVPLP_CUTS", "1") != "0", "leaf_blocks_per_walk": float((hist * np.arange(32)).sum()) / max(walks, 1),
    "tri_pairs_per_walk": pairs / max(walks, 1),
    # per RAY: a wave-level visit tests the node / triangles for its 64 lanes; lanes that are alive at the start of the walk = rays
    "nodes_per_ray": nodes * 64 / max(rays, 1), "tris_per_ray": pairs * 2 * 64 / max(rays, 1),
    "walks_fully_occluded_frac": all_occ / max(walks, 1),
    "leaf_blocks_per_walk_hist": (hist / max(walks, 1)).round(4).tolist(),
    "node_visits_per_empty_walk": int(raw[4 + 35]) / max(int(hist[0]), 1),
    "node_visits_per_fully_occluded_walk": int(raw[4 + 36]) / max(all_occ, 1), "leaf_blocks_per_fully_occluded_walk": int(raw[4 + 37]) / max(all_occ, 1),
    "node_visit_share": {"empty_walks": int(raw[4 + 35]) / max(nodes, 1), "fully_occluded_walks": int(raw[4 + 36]) / max(nodes, 1)},
    "cache_sim": {"walks_with_cache": int(raw[4 + 42]), "cache_alone_occludes_all": int(raw[4 + 38]), "node_visits_saved": int(raw[4 + 39]),
                  "lanes_killed_by_cache": int(raw[4 + 40]), "lanes_occluded_total": int(raw[4 + 41]),
                  "after_fully_occluded_walk": int(raw[4 + 44]), "hits_after_fully_occluded_walk": int(raw[4 + 43])},
    "node_visits_hist_by_16": (raw[4 + 45:4 + 64].astype(np.int64) / max(walks, 1)).round(4).tolist(),
    "order": a.order,
    "kernel_ms_with_counters": st["dominant_kernel_ms"], "launches": st["launches"], "shaded": st["shaded"],
}
print(json.dumps(out, indent=1))
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
