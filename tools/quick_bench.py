"""Developer timing probe (not the judged bench): synthetic conference scene, cfg2/cfg3 sizes."""
import argparse, json, os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa
import evplp_amd as ev
import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--tris", type=int, default=331000)
ap.add_argument("--paths", type=int, default=1024)
ap.add_argument("--vpl-paths", type=int, default=1024)
ap.add_argument("--builder", type=int, default=1)
ap.add_argument("--mode", default="one")
ap.add_argument("--splat", action="store_true")
ap.add_argument("--vsl", action="store_true")
ap.add_argument("--pt", action="store_true", help="path tracer: one camera path per pixel per iteration")
ap.add_argument("--lvc", action="store_true")
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--k", type=int, default=0, help="splits per wavefront of the gathers (0 = the library default)")
ap.add_argument("--style", default="hard")
ap.add_argument("--strip-count", type=int, default=1)
ap.add_argument("--strip-rank", type=int, default=0)
ap.add_argument("--strip-rows", type=int, default=16)
a = ap.parse_args()
d = "/tmp/evplp_synth"
t0 = time.time()
jp = ev.synth_scene(d + "_" + a.style, "conf", a.tris, 1234, a.res, a.res, style=a.style)
sd, root = scenes.load_obj_scene(jp)
print("scene load %.1fs tris %d" % (time.time() - t0, sd.triangle_soup()[2].shape[0]))
P = 4
c = ev.Context(a.res, a.res, a.paths, a.vpl_paths, P, bvh_builder=a.builder, gather_splits_per_wave=a.k, strip_rank=a.strip_rank, strip_count=a.strip_count, strip_rows=a.strip_rows)
t0 = time.time(); sd.upload(c); print("upload+build %.2fs" % (time.time() - t0), c.accel_info())
bsr, total, larea = c.scene_metrics()
radius = 0.003 * bsr
kw = dict(camera_pos=sd.cam_origin, mis_mode=a.mode, pdf_mc=a.vpl_paths / a.paths / math.pi / radius**2, clamping_value=1.0 / total,
          photon_radius=radius, vsl_radius=0.05 * bsr, vsl_inv_pi_radius2=1 / (math.pi * (0.05 * bsr) ** 2),
          num_light_paths=a.paths, num_vpl_light_paths=a.vpl_paths, photons_per_path=P, do_accumulate=1)
if a.pt:
    for it in range(a.iters):
        c.primary((0, 0)); c.path_trace(sd.cam_origin, it, 3, accumulate=True); c.synchronize()
        st = c.pass_stats(ev.PASS_PATH_TRACE)
        print("pt iter %d: %.2f ms, %d paths, %d rays -> %.1f Mpaths/s %.1f Mrays/s" % (it, st["ms"], st["pairs"], st["rays"], st["pairs"] / st["ms"] / 1e3, st["rays"] / st["ms"] / 1e3))
    img = c.resolve(1.0 / a.iters, 0.0, 1.0)
    os.makedirs("gpurun_out", exist_ok=True)
    ev.save_image("gpurun_out/quick_pt.png", img[: a.res][::-1])
    sys.exit(0)
if a.lvc:
    for it in range(a.iters):
        fp = ev.frame_params(rng_seed=it, **kw)
        c.primary((0, 0)); c.trace_light_paths(it); c.gather_lvc(fp); c.synchronize()
        st = c.pass_stats(ev.PASS_GATHER_LVC)
        print("lvc iter %d: %.2f ms, %.3e pairs, %.3e rays -> %.1f Mpairs/s" % (it, st["ms"], st["pairs"], st["rays"], st["pairs"] / st["ms"] / 1e3))
    sys.exit(0)
for it in range(a.iters):
    fp = ev.frame_params(rng_seed=it, **kw)
    t0 = time.time()
    c.primary((0, 0)); c.trace_light_paths(it)
    if a.vsl: c.gather_vsl(fp)
    else: c.gather_vpl(fp)
    if a.splat: c.splat_photons(fp)
    c.synchronize()
    wall = time.time() - t0
    names = ["primary", "light", "gather_vpl", "gather_vsl", "splat"]
    st = {n: c.pass_stats(i) for i, n in enumerate(names)}
    g = st["gather_vsl" if a.vsl else "gather_vpl"]
    print("iter %d wall %.1f ms | primary %.2f light %.2f gather %.2f (kernel %.2f) splat %.2f (tiles %.3f) | usable %d pairs %.3e rays %.3e nodes %.3e | %.1f Mpairs/s %.1f Mrays/s nodes/ray(wave) %.1f | splat pairs %.3e" % (
        it, wall * 1e3, st["primary"]["ms"], st["light"]["ms"], g["ms"], g["dominant_kernel_ms"], st["splat"]["ms"], st["splat"]["dominant_kernel_ms"],
        g["usable"], g["pairs"], g["rays"], g["nodes"], g["pairs"] / g["ms"] / 1e3, g["rays"] / g["ms"] / 1e3, g["nodes"] * 64 / max(g["rays"], 1), st["splat"]["pairs"]))
img = c.resolve(1.0 / a.iters, 1.0 / a.iters, 1.0)
print("image mean", img[: a.res].mean(axis=(0, 1)), "finite", np.isfinite(img).all())
os.makedirs("gpurun_out", exist_ok=True)
ev.save_image("gpurun_out/quick_%s.png" % ("vsl" if a.vsl else a.mode), img[: a.res][::-1])
