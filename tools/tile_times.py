"""When does every tile's wavefront of splat_tiles start and end, and how full is its bin?  Configs #3 / #4 record sets (tools/splat_times.py's)
through a build with -DEVPLP_TILE_TIMES=1 (make VARIANT=ttimes EXTRA_HIPFLAGS=-DEVPLP_TILE_TIMES=1 evplp_amd/lib/libevplp_hip_ttimes.so)."""
import ctypes as C, os, sys, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EVPLP_LIB"] = os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_ttimes.so")
sys.path.insert(0, ROOT)
import evplp_amd as ev
fpn = sys.argv[1] if len(sys.argv) > 1 else "proxy"
for (W, H, N, mode) in ((1024, 1024, 500000, "balance"), (1920, 1080, 300000, "one")):
    jp = ev.synth_scene("/tmp/evplp_tt", "conf", 331000, 1234, W, H)
    P = 4
    with ev.Context(W, H, N, 1024, P) as c:
        c.load_scene_json(jp)
        cam = c.camera(); bsr, total, _ = c.scene_metrics(); r = 0.003 * bsr
        kw = dict(camera_pos=list(cam.origin), mis_mode=mode, pdf_mc=1024 / N / math.pi / r ** 2, photon_radius=r, num_light_paths=N, num_vpl_light_paths=1024, photons_per_path=P, splat_footprint=fpn)
        c.primary((0, 0)); c.trace_light_paths(0)
        for it in range(3):
            c.splat_photons(ev.frame_params(**kw), clear=True); c.synchronize()
        st = c.pass_stats(ev.PASS_SPLAT)
        n = ((W + 7) // 8) * ((H + 7) // 8)
        buf = (C.c_ulonglong * (3 * n))()
        assert ev.lib().evplp_debug_tile_times(buf, 3 * n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 3).astype(np.float64)
    t = a[:, :2] / 100.0; t -= t[:, 0].min(); cnt = a[:, 2]
    life = t[:, 1] - t[:, 0]; end = t[:, 1].max()
    print(f"{W}x{H} {fpn}: pass {st['ms'] * 1e3:.0f} us; tile kernel first start -> last end {end:.1f} us; {n} tiles, fullest bin {int(cnt.max())}, mean {cnt.mean():.0f}")
    print("   lifetime us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f; sum / 8192 slots %.1f us" % (life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), life.sum() / 8192))
    for q in range(10):
        mid = (q + 0.5) * end / 10
        print("   t %6.1f us: %5d waves in flight (%.2f per SIMD), %5d started" % (mid, int(((t[:, 0] <= mid) & (t[:, 1] > mid)).sum()), ((t[:, 0] <= mid) & (t[:, 1] > mid)).sum() / 1024, int(((t[:, 0] >= q * end / 10) & (t[:, 0] < (q + 1) * end / 10)).sum())))
    slow = np.argsort(-life)[:6]
    print("   slowest:", [(int(cnt[i]), round(float(life[i]), 1), round(float(t[i, 0]), 1)) for i in slow], "(entries, lifetime us, start us)")
