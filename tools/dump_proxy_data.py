"""Developer probe (GPU): the G-buffer tiles and the usable VPL records of the bench configuration, for the CPU walk proxy
(tools/bvh_eval reads them with --data): every `--stride`-th 8x8 tile's 64 positions + normals and every usable record's
position + normal.  The scene is the deterministic generator's, so the proxy rebuilds the same OBJ on the CPU side.

    python3 tools/dump_proxy_data.py --scene hard --out gpurun_out/proxy_hard.npz
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import evplp_amd as ev

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--tris", type=int, default=331000)
ap.add_argument("--paths", type=int, default=1024)
ap.add_argument("--scene", default="hard")
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--out", required=True)
a = ap.parse_args()
d = "/tmp/evplp_proxy_%s" % a.scene
jp = ev.synth_scene(d, "conf", a.tris, 1234, a.res, a.res, style=a.scene)
P = 4
c = ev.Context(a.res, a.res, a.paths, a.paths, P)
c.load_scene_json(jp)
c.primary((0, 0)); c.trace_light_paths(0); c.synchronize()
pos = c.download(ev.BUF_GBUF_POSITION).reshape(a.res, a.res, 4)
nrm = c.download(ev.BUF_GBUF_NORMAL).reshape(a.res, a.res, 4)
rec = c.download(ev.BUF_RECORDS)
raw = rec.view(np.uint8).reshape(-1, 96)
rpos = raw[:, 0:12].copy().view(np.float32).reshape(-1, 3)
flags = raw[:, 12:16].copy().view(np.uint32).reshape(-1)
rnrm = raw[:, 16:28].copy().view(np.float32).reshape(-1, 3)
use = (flags & 1) != 0
T = a.res // 8
tp = pos.reshape(T, 8, T, 8, 4).transpose(0, 2, 1, 3, 4).reshape(T * T, 64, 4)
tn = nrm.reshape(T, 8, T, 8, 4).transpose(0, 2, 1, 3, 4).reshape(T * T, 64, 4)
sel = np.arange(0, T * T, a.stride)
np.savez_compressed(a.out, tile_pos=tp[sel].astype(np.float32), tile_nrm=tn[sel, :, :3].astype(np.float32), tile_index=sel.astype(np.int32),
                    vpl_pos=rpos[use], vpl_nrm=rnrm[use], res=a.res, tris=a.tris)
print("tiles", len(sel), "vpls", int(use.sum()), "->", a.out)
