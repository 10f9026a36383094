"""Splat pass timing probe: config #3 (1024^2, 500k paths) and config #4 (1920x1080, 300k paths) record sets."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evplp_amd as ev
FOOTPRINT = os.environ.get("FOOTPRINT", "ideal")
d = "/tmp/evplp_splat"
for (W, H, N, mode) in ((1024, 1024, 500000, "balance"), (1920, 1080, 300000, "one")):
    jp = ev.synth_scene(d, "conf", 331000, 1234, W, H)
    P = 4
    with ev.Context(W, H, N, 1024, P) as c:
        c.load_scene_json(jp)
        cam = c.camera(); bsr, total, _ = c.scene_metrics(); r = 0.003 * bsr
        kw = dict(camera_pos=list(cam.origin), mis_mode=mode, pdf_mc=1024 / N / math.pi / r ** 2, photon_radius=r, num_light_paths=N, num_vpl_light_paths=1024, photons_per_path=P, splat_footprint=FOOTPRINT)
        c.primary((0, 0)); c.trace_light_paths(0)
        for it in range(3):
            c.splat_photons(ev.frame_params(**kw), clear=True); c.synchronize()
        st = c.pass_stats(ev.PASS_SPLAT)
        print(FOOTPRINT, "%dx%d N=%d %s: splat %.3f ms (tiles %.3f) pairs %.3e bin entries %d (fullest bin %d) light %.3f primary %.3f" % (W, H, N, mode, st["ms"], st["dominant_kernel_ms"], st["pairs"], st["nodes"] & 0xffffffff, st["nodes"] >> 32, c.pass_stats(ev.PASS_LIGHT_TRACE)["ms"], c.pass_stats(ev.PASS_PRIMARY)["ms"]))
