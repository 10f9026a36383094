"""Experiment: does the ORDER of the VPL records change the gather time?  (Same records, record order vs Morton order vs
random order; config #2 on both scene styles.)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import evplp_amd as ev

W = H = 1024; NV = 1024; P = 4
def morton(p):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.uint64), 0, 1023)
    def ex(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
    return (ex(q[:, 0]) << 2) | (ex(q[:, 1]) << 1) | ex(q[:, 2])
for style in ("hard", "easy"):
    d = f"/tmp/evplp_exp_{style}"
    jp = ev.synth_scene(d, "conf", 331000, 1234, W, H, style=style)
    with ev.Context(W, H, NV, NV, P) as c:
        c.load_scene_json(jp)
        cam = c.camera(); _, total, _ = c.scene_metrics()
        c.primary((0.0, 0.0)); c.trace_light_paths(0)
        rec = c.download(ev.BUF_RECORDS).copy()
        fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", clamping_value=1.0 / total, num_light_paths=NV, num_vpl_light_paths=NV, photons_per_path=P, do_accumulate=0, rng_seed=0)
        raw = rec.view(np.uint8).reshape(-1, 96)
        pos = raw[:, 0:12].copy().view(np.float32).reshape(-1, 3)
        flags = raw[:, 12:16].copy().view(np.uint32).reshape(-1)
        usable = (flags & 1) != 0
        orders = {"record": np.arange(len(raw))}
        idx = np.nonzero(usable)[0]
        srt = idx[np.argsort(morton(pos[idx]), kind="stable")]
        o = np.arange(len(raw)); o[:len(srt)] = srt; o[len(srt):] = np.nonzero(~usable)[0]; orders["morton"] = o
        rng = np.random.RandomState(1); o2 = o.copy(); rng.shuffle(o2[:len(srt)]); orders["random"] = o2
        # morton order, but interleaved so that split s (index % 128) gets a contiguous Morton range
        n = len(srt); per = (n + 127) // 128
        il = np.full(per * 128, -1, np.int64)
        for s in range(128):
            seg = srt[s * per:(s + 1) * per]; il[s:s + 128 * len(seg):128] = seg
        il = il[il >= 0]; o3 = o.copy(); o3[:len(il)] = il; orders["morton_per_split"] = o3
        for name, order in orders.items():
            c.upload(ev.BUF_RECORDS, raw[order].reshape(rec.shape if rec.ndim == 2 else -1).view(rec.dtype).reshape(rec.shape))
            ms = []
            for it in range(4):
                c.gather_vpl(fp); c.synchronize()
                ms.append(c.pass_stats(ev.PASS_GATHER_VPL)["dominant_kernel_ms"])
            st = c.pass_stats(ev.PASS_GATHER_VPL)
            print(style, name, "kernel ms", [round(m, 2) for m in ms[1:]], "rays", st["rays"], flush=True)
