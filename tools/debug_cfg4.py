import sys, os, math, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import evplp_amd as ev, oracle_api as oa, scenes
def rel_l2(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))
P = 4
for (W, H, style, NL) in ((1920, 1080, "textured", 300000), (1024, 1024, "textured", 300000), (1920, 1080, "hard", 300000), (1920, 1080, "easy", 300000), (480, 270, "textured", 20000)):
    d = "/tmp/dbg_%d_%s" % (W, style)
    jp = ev.synth_scene(d, "s", 120000, 11, W, H, style=style)
    sd, _ = scenes.load_obj_scene(jp, decode=lambda p: ev.decode_image(p)[0])
    osc = oa.Scene(sd)
    rows = [37 * H // 1080, 300 * H // 1080, 541 * H // 1080, 905 * H // 1080]
    with ev.Context(W, H, NL, 0, P, deterministic=True) as c:
        c.load_scene_json(jp)
        cam = c.camera(); print('   fovy product', repr(cam.fovy), 'python', repr(sd.fovy), 'aspect', repr(cam.aspect), repr(sd.aspect)); sd.fovy = cam.fovy; sd.aspect = cam.aspect
        bsr, total, _ = c.scene_metrics(); r = 0.003 * bsr
        jitter = (0.0003, -0.0002)
        kw = dict(camera_pos=sd.cam_origin, mis_mode=0, pdf_mc=0.0, clamping_value=1.0 / total, photon_radius=r, num_light_paths=NL, num_vpl_light_paths=0, photons_per_path=P, jitter=jitter)
        c.primary(jitter); c.trace_light_paths(7)
        c.splat_photons(ev.frame_params(**kw), clear=True)
        pm = c.download(ev.BUF_PHOTON_ACCUM)[:H]
        rec = c.download(ev.BUF_RECORDS)
        gb = [c.download(b)[:H] for b in (ev.BUF_GBUF_POSITION, ev.BUF_GBUF_NORMAL, ev.BUF_GBUF_DIFFUSE, ev.BUF_GBUF_PHONG)]
        pairs = c.pass_stats(ev.PASS_SPLAT)["pairs"]
    orec = osc.trace_light_paths(7, NL, P)
    fd = (rec['flags'] != orec['flags']).reshape(NL, P).any(1); pd_ = (np.abs(rec['pos'] - orec['pos']).max(1) > 1e-4).reshape(NL, P).any(1)
    print('   paths with different flags', int(fd.sum()), 'with a vertex moved > 1e-4', int(pd_.sum()), 'of', NL, '; records bitwise equal:', int((rec.view(np.uint8).reshape(NL*P,96) == orec.view(np.uint8).reshape(NL*P,96)).all(1).sum()), 'of', NL*P)
    print(W, H, style, "records identical:", rec.tobytes() == orec.tobytes(), "flags identical:", np.array_equal(rec["flags"], orec["flags"]),
          "max pos diff", float(np.abs(rec["pos"] - orec["pos"]).max()), "max flux rel", float((np.abs(rec["flux"] - orec["flux"]) / (np.abs(orec["flux"]) + 1e-9)).max()))
    out = np.zeros((H, W, 4), np.float32); out2 = np.zeros((H, W, 4), np.float32)
    og = [np.zeros((H, W, 4), np.float32) for _ in range(5)]
    for y in rows:
        gy = osc.primary(W, H, jitter, rows=(y, y + 1))
        for k in range(5): og[k][y] = gy[k][y]
    print("   gbuf rows identical:", [bool(np.array_equal(gb[k][rows], og[k][rows])) for k in range(4)])
    op = 0
    for y in rows:
        _, n = oa.splat(oa.frame_params(**kw), W, H, og, orec, out=out, rows=(y, y + 1)); op += n
        oa.splat(oa.frame_params(**kw), W, H, gb, rec, out=out2, rows=(y, y + 1))
    print("   rel_l2 product vs oracle(own inputs)", rel_l2(pm[rows][..., :3], out[rows][..., :3]), " vs oracle(product inputs)", rel_l2(pm[rows][..., :3], out2[rows][..., :3]),
          "sum ratio", float(pm[rows][..., :3].sum() / out[rows][..., :3].sum()))
