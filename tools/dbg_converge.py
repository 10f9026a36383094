import sys, math, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import evplp_amd as ev, scenes
W, H, P = 96, 64, 4
room = scenes.box_room(seed=3, n_boxes=5, tess=2, aspect=W / H)
def rel_l2(a, b): return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
def run_pt(n):
    with ev.Context(W, H, 1, 1, 1) as c:
        room.upload(c); c.primary((0, 0), clear_light=True)
        for i in range(n): c.path_trace(room.cam_origin, i, 3, accumulate=True)
        return c.resolve(1.0 / n, 0, 0)[:H]
def run_fam(n, nl, nv, mode, rpct):
    with ev.Context(W, H, nl, nv, P) as c:
        room.upload(c); c.primary((0, 0), clear_light=True)
        bsr, total, _ = c.scene_metrics(); r = rpct * bsr
        kw = dict(camera_pos=room.cam_origin, mis_mode=mode, pdf_mc=(nv / nl / math.pi / r ** 2) if r > 0 else 0.0, clamping_value=1.0 / total, photon_radius=r,
                  num_light_paths=nl, num_vpl_light_paths=nv, photons_per_path=P, do_accumulate=1)
        for i in range(n):
            fp = ev.frame_params(rng_seed=i, **kw)
            c.trace_light_paths(i); c.gather_vpl(fp)
            if r > 0: c.splat_photons(fp)
        return c.resolve(1.0 / n, 1.0 / n, 0)[:H]
pt = run_pt(4096); pt2 = run_pt(1024)
print("pt 4096 vs pt 1024 (different spp, shared seeds):", rel_l2(pt2, pt), "mean", pt.mean())
for name, args in {"ir one": (256, 256, 256, "one", 0.0), "evplp balance": (256, 2048, 256, "balance", 0.02), "evplp max": (256, 2048, 256, "max", 0.02),
                   "geometryClamp+pm": (256, 2048, 256, "geometryClamp", 0.02)}.items():
    im = run_fam(*args)
    print(name, "rel L2 vs pt", rel_l2(im, pt), "energy ratio", im.sum() / pt.sum(), "median ratio", np.median(im.sum(-1)[pt.sum(-1) > 0] / pt.sum(-1)[pt.sum(-1) > 0]))
