"""Developer probe: gather cost of each 16-row band of the cfg2 frame (one context per band)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import evplp_amd as ev
d = "/tmp/evplp_synth"; jp = ev.synth_scene(d, "conf", 331000, 1234, 1024, 1024)
N, P, bands = 1024, 4, 64
out = []
for r in range(0, bands, 1):
    with ev.Context(1024, 1024, N, N, P, strip_rank=r, strip_count=bands, strip_rows=16) as c:
        c.load_scene_json(jp); cam = c.camera()
        fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=N, num_vpl_light_paths=N, photons_per_path=P)
        for it in range(2):
            c.primary((0, 0)); c.trace_light_paths(0); c.gather_vpl(fp)
        st = c.pass_stats(ev.PASS_GATHER_VPL)
        out.append((r, st["dominant_kernel_ms"], st["rays"]))
tot = sum(o[1] for o in out)
print("sum of band times %.1f ms" % tot)
for r, ms, rays in out:
    print("band %2d rows %4d-%4d  %.2f ms  rays %.3e  ns/ray*1e3 %.2f" % (r, r * 16, r * 16 + 15, ms, rays, ms * 1e6 / max(rays, 1)))
