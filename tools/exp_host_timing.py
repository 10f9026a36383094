"""Where does the host spend its time in the config #4 loop?  (per-call wall time of the ctypes calls, no syncs)"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import evplp_amd as ev
W, H, NL, P = 1920, 1080, 300000, 4
jp = ev.synth_scene("/tmp/evplp_host_t", "conf", 331000, 1234, W, H, style="hard")
with ev.Context(W, H, NL, 0, P, overlap_light_tracing=True) as c:
    c.load_scene_json(jp)
    cam = c.camera(); bsr, total, _ = c.scene_metrics()
    r = 0.003 * bsr
    fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", photon_radius=r, num_light_paths=NL, num_vpl_light_paths=0, photons_per_path=P, do_accumulate=1)
    for it in range(5):
        c.trace_light_paths(it); c.primary((0, 0)); c.splat_photons(fp)
    c.synchronize()
    T = {"lt": [], "pr": [], "sp": []}
    t_all = time.perf_counter()
    for it in range(200):
        t0 = time.perf_counter(); c.trace_light_paths(it); t1 = time.perf_counter(); c.primary((0, 0)); t2 = time.perf_counter(); c.splat_photons(fp); t3 = time.perf_counter()
        T["lt"].append(t1 - t0); T["pr"].append(t2 - t1); T["sp"].append(t3 - t2)
    t_enq = time.perf_counter() - t_all
    c.synchronize()
    t_tot = time.perf_counter() - t_all
    for k, v in T.items():
        v = np.array(v[20:]) * 1e6
        print(k, "host us: median %.1f  mean %.1f  max %.1f" % (np.median(v), v.mean(), v.max()))
    print("enqueue of 200 iterations: %.1f ms; until done: %.1f ms -> %.3f ms per iteration" % (t_enq * 1e3, t_tot * 1e3, t_tot / 200 * 1e3))
