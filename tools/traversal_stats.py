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
a = ap.parse_args()
d = "/tmp/evplp_stats_%s" % a.scene
jp = ev.synth_scene(d, "conf", a.tris, 1234, a.res, a.res, style=a.scene)
P = 4
c = ev.Context(a.res, a.res, a.paths, a.paths, P, gather_splits_per_wave=a.k)
c.load_scene_json(jp)
cam = c.camera()
fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=a.paths, num_vpl_light_paths=a.paths, photons_per_path=P, do_accumulate=0)
c.primary((0, 0)); c.trace_light_paths(0); c.gather_vpl(fp); c.synchronize()
st = c.pass_stats(ev.PASS_GATHER_VPL)
raw = c.debug_counters(ev.PASS_GATHER_VPL)
rays, nodes = int(st["rays"]), int(raw[1])
hist = raw[4:4 + 32].astype(np.int64)
walks, pairs, all_occ = int(raw[4 + 32]), int(raw[4 + 33]), int(raw[4 + 34])
out = {
    "scene": a.scene, "accel": c.accel_info(), "usable_vpls": st["usable"], "rays": rays, "walks": walks,
    "wave_node_visits": nodes, "node_visits_per_walk": nodes / max(walks, 1), "leaf_blocks_per_walk": float((hist * np.arange(32)).sum()) / max(walks, 1),
    "tri_pairs_per_walk": pairs / max(walks, 1),
    # per RAY: a wave-level visit tests the node / triangles for its 64 lanes; lanes that are alive at the start of the walk = rays
    "nodes_per_ray": nodes * 64 / max(rays, 1), "tris_per_ray": pairs * 2 * 64 / max(rays, 1),
    "walks_fully_occluded_frac": all_occ / max(walks, 1),
    "leaf_blocks_per_walk_hist": (hist / max(walks, 1)).round(4).tolist(),
    "kernel_ms_with_counters": st["dominant_kernel_ms"], "launches": st["launches"], "shaded": st["shaded"],
}
print(json.dumps(out, indent=1))
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
