import sys, os, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import evplp_amd as ev, oracle_api as oa, scenes
P = 4; W = H = 1024; N = 1024
jp = ev.synth_scene("/tmp/dbgvis", "conference_synth", 331000, 1234, W, H, style="hard")
sd, _ = scenes.load_obj_scene(jp); osc = oa.Scene(sd); l = oa.load()
with ev.Context(W, H, N, N, P, strip_rank=5, strip_count=32, strip_rows=8) as c:
    c.load_scene_json(jp)
    c.primary((0.0, 0.0)); c.trace_light_paths(0)
    rows = c.global_rows(); ok = rows < H; rows = rows[ok]
    planes = []
    for b in (ev.BUF_GBUF_POSITION, ev.BUF_GBUF_NORMAL, ev.BUF_GBUF_DIFFUSE, ev.BUF_GBUF_PHONG):
        full = np.zeros((H, W, 4), np.float32); full[rows] = c.download(b)[ok]; planes.append(full)
    rec = c.download(ev.BUF_RECORDS)
    kw = dict(camera_pos=sd.cam_origin, mis_mode=0, num_light_paths=N, num_vpl_light_paths=N, photons_per_path=P)
    def counts(mask):
        r2 = rec.copy(); r2["flags"][~mask] = 0
        c.upload(ev.BUF_RECORDS, r2)
        c.gather_vpl(ev.frame_params(**kw)); st = c.pass_stats(ev.PASS_GATHER_VPL)
        o = osc.gather_counts(oa.frame_params(**kw), W, planes, r2, rows)
        return (st["rays"], st["shaded"]), o
    lo, hi = 0, N * P
    full = np.ones(N * P, bool)
    print("all", counts(full))
    while hi - lo > 1:
        mid = (lo + hi) // 2
        m = np.zeros(N * P, bool); m[lo:mid] = True
        a, b = counts(m)
        print(lo, mid, hi, a, b)
        if a != b: hi = mid
        else: lo = mid
    k = lo
    print("record", k, rec[k])
    m = np.zeros(N * P, bool); m[k] = True
    r2 = rec.copy(); r2["flags"][~m] = 0; r2["flux"][k] = 1.0; r2["rho_d"][k] = 1.0
    c.upload(ev.BUF_RECORDS, r2); c.clear_accumulators()
    c.gather_vpl(ev.frame_params(**kw))
    got = np.zeros((H, W, 4), np.float32); got[rows] = c.download(ev.BUF_VPL_ACCUM)[ok]
    out = np.zeros((H, W, 4), np.float32)
    for y in rows: osc.gather(oa.frame_params(**kw), W, H, planes, r2, out=out, rows=(int(y), int(y) + 1))
    d = (got[..., :3].sum(-1) > 0) != (out[..., :3].sum(-1) > 0)
    ys, xs = np.nonzero(d)
    print("pixels with different lit state", list(zip(ys.tolist(), xs.tolist())))
    for y, x in zip(ys, xs):
        p1 = planes[0][y, x, :3]; o = rec["pos"][k].astype(np.float32); dd = (p1 - o).astype(np.float32)
        oo = np.ascontiguousarray(o); dv = np.ascontiguousarray(dd)
        print(" pixel", y, x, "pos", p1, "gpu lit", got[y, x, :3].sum() > 0, "oracle bvh occluded", l.evo_occluded(osc.h, oa.ptr(oo), oa.ptr(dv), C.c_float(0.0001), C.c_float(1 - 0.0001)),
              "oracle brute occluded", l.evo_occluded_brute(osc.h, oa.ptr(oo), oa.ptr(dv), C.c_float(0.0001), C.c_float(1 - 0.0001)))
