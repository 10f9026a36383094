import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import evplp_amd as evplp, oracle_api as oa, scenes
W, H, NP, P = 96, 64, 64, 4
oa.load()
room = scenes.box_room(seed=3, n_boxes=5, tess=2, aspect=W / H)
osc = oa.Scene(room)
with evplp.Context(W, H, NP, NP, P) as c:
    room.upload(c)
    c.primary((0, 0), clear_light=True)
    cam = osc.sd.cam_origin
    c.clear_accumulators()
    c.path_trace(cam, 3, 3, accumulate=True)
    got = c.download(evplp.BUF_VPL_ACCUM)[:H]
    g4 = [c.download(b)[:H] for b in (evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL, evplp.BUF_GBUF_DIFFUSE, evplp.BUF_GBUF_PHONG)]
ref, n = osc.path_trace(cam, 3, 3, W, H, g4)
g, r = got[..., :3], ref[..., :3]
err = np.abs(g - r) / np.maximum(np.abs(r), 1e-3 * r.max())
e = err.max(-1)
idx = np.argsort(e.ravel())[::-1][:30]
for i in idx:
    y, x = divmod(i, W)
    print(y, x, e[y, x], g[y, x], r[y, x], g4[3][y, x])
