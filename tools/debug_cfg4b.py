import sys, os, math, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import evplp_amd as ev, oracle_api as oa, scenes
from test_gpu_end_to_end import MT19937, jitter_of
def rel_l2(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))
P = 4; W, H, NL, ITER = 1920, 1080, 300000, 8
d = "/tmp/dbg4b"
jp = ev.synth_scene(d, "s", 120000, 11, W, H, style="textured")
sd, _ = scenes.load_obj_scene(jp, decode=lambda p: ev.decode_image(p)[0])
rows = [37, 300, 541, 700, 905, 1079]
l = oa.load()
with ev.Context(W, H, NL, 0, P) as c:
    c.load_scene_json(jp)
    cam = c.camera(); sd.fovy = cam.fovy
    osc = oa.Scene(sd)
    bsr, total, _ = c.scene_metrics()
    f32 = np.float32
    obsr = f32(l.evo_scene_bounding_sphere_radius(osc.h)); print("bsr", repr(bsr), repr(float(obsr)), "total", repr(total), repr(float(l.evo_scene_total_area(osc.h))))
    radius = f32(obsr * f32(0.003)); clamp = f32(1.0) / f32(l.evo_scene_total_area(osc.h)); pdf_mc = f32(0); clamp_start = clamp
    pr, pc, pp = float(f32(bsr) * f32(0.003)), 1.0 / total, 0.0
    pm_o = np.zeros((H, W, 4), np.float32); pm_x = np.zeros((H, W, 4), np.float32)
    rng = MT19937(7)
    c.clear_accumulators()
    for it in range(ITER):
        jitter = jitter_of(rng, W, H)
        kw = dict(camera_pos=sd.cam_origin, mis_mode=0, pdf_mc=float(pdf_mc), clamping_value=float(clamp), photon_radius=float(radius),
                  num_light_paths=NL, num_vpl_light_paths=0, photons_per_path=P, do_accumulate=1, rng_seed=it + 7, jitter=jitter)
        pkw = dict(kw); pkw.update(photon_radius=pr, clamping_value=pc, pdf_mc=pp)
        c.primary(jitter); c.trace_light_paths(it + 7); c.splat_photons(ev.frame_params(**pkw))
        rec = c.download(ev.BUF_RECORDS)
        orec = osc.trace_light_paths(it + 7, NL, P)
        g = [np.zeros((H, W, 4), np.float32) for _ in range(5)]
        for y in rows:
            gy = osc.primary(W, H, jitter, rows=(y, y + 1))
            for k in range(5): g[k][y] = gy[k][y]
        before_o = pm_o[rows].copy(); before_x = pm_x[rows].copy()
        for y in rows:
            oa.splat(oa.frame_params(**kw), W, H, g, orec, out=pm_o, rows=(y, y + 1))
            oa.splat(oa.frame_params(**kw), W, H, g, rec, out=pm_x, rows=(y, y + 1))
        got = c.download(ev.BUF_PHOTON_ACCUM)[:H][rows][..., :3]
        print("it", it, "radius", repr(float(radius)), repr(pr), "cum rel_l2 vs oracle(own records)", rel_l2(got, pm_o[rows][..., :3]), "vs oracle(product records)", rel_l2(got, pm_x[rows][..., :3]),
              "| this iteration own-vs-product records", rel_l2((pm_x[rows] - before_x)[..., :3], (pm_o[rows] - before_o)[..., :3]))
        r, c_, p_, vr, vi = (C.c_float(x) for x in (radius, clamp, pdf_mc, 0.0, 0.0))
        l.evo_progressive_step(it + 1, 0.7, float(clamp_start), 0, NL, C.byref(r), C.byref(c_), C.byref(p_), 0, C.byref(vr), C.byref(vi))
        radius, clamp, pdf_mc = f32(r.value), f32(c_.value), f32(p_.value)
        pr, pc, pp, _, _ = ev.progressive_step(it + 1, 0.7, 1.0 / total, 0, NL, pr, pc, pp)
