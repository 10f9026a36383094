import sys, os, math, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import evplp_amd as ev
d = "/tmp/evplp_cfg4"
for (W, H) in ((1024, 1024), (1920, 1080)):
    jp = ev.synth_scene(d, "conf", 331000, 1234, W, H)
    N, P = 300000, 4
    with ev.Context(W, H, N, 0, P) as c:
        c.load_scene_json(jp)
        cam = c.camera(); bsr, total, _ = c.scene_metrics(); r = 0.003 * bsr
        kw = dict(camera_pos=list(cam.origin), mis_mode="one", pdf_mc=0.0, photon_radius=r, num_light_paths=N, num_vpl_light_paths=0, photons_per_path=P)
        c.primary((0, 0)); c.trace_light_paths(0)
        c.splat_photons(ev.frame_params(**kw), clear=True)
        st = c.pass_stats(ev.PASS_SPLAT)
        pm = c.download(ev.BUF_PHOTON_ACCUM)
        print(W, H, "pairs", st["pairs"], "ms", st["ms"], "bins", st.get("reserved"), "pm max", pm[..., :3].max(), "mean", pm[..., :3].mean(), "local_rows", c.local_rows)
