"""Per-workgroup clocks of splat_scatter_kernel (the (slice, bucket) workgroups only), config #3's record set; developer build
make VARIANT=sctimes EXTRA_HIPFLAGS=-DEVPLP_SCATTER_TIMES=1 evplp_amd/lib/libevplp_hip_sctimes.so"""
import ctypes as C, os, sys, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EVPLP_LIB"] = os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_sctimes.so")
sys.path.insert(0, ROOT)
import evplp_amd as ev
for (W, H, N, mode) in ((1024, 1024, 500000, "balance"), (1920, 1080, 300000, "one")):
    jp = ev.synth_scene("/tmp/evplp_tt", "conf", 331000, 1234, W, H)
    with ev.Context(W, H, N, 1024, 4) as c:
        c.load_scene_json(jp)
        cam = c.camera(); bsr, total, _ = c.scene_metrics(); r = 0.003 * bsr
        kw = dict(camera_pos=list(cam.origin), mis_mode=mode, pdf_mc=1024 / N / math.pi / r ** 2, photon_radius=r, num_light_paths=N, num_vpl_light_paths=1024, photons_per_path=4)
        c.primary((0, 0)); c.trace_light_paths(0)
        for it in range(3):
            c.splat_photons(ev.frame_params(**kw), clear=True); c.synchronize()
        n = 65536
        buf = (C.c_ulonglong * (3 * n))()
        assert ev.lib().evplp_debug_scatter_times(buf, 3 * n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 3).astype(np.float64)
    a = a[a[:, 1] > 0]
    t = a[:, :2] / 100.0; t -= t[:, 0].min(); cnt = a[:, 2]
    life = t[:, 1] - t[:, 0]; end = t[:, 1].max()
    print(f"{W}x{H}: {len(t)} (slice, bucket) workgroups; first start -> last end {end:.1f} us; entries per workgroup mean {cnt.mean():.0f} max {cnt.max():.0f}")
    print("   lifetime us: mean %.1f median %.1f p90 %.1f max %.1f; last start %.1f us" % (life.mean(), np.median(life), np.percentile(life, 90), life.max(), t[:, 0].max()))
    slow = np.argsort(-life)[:5]
    print("   slowest:", [(int(cnt[i]), round(float(life[i]), 1), round(float(t[i, 0]), 1)) for i in slow], "(entries, lifetime us, start us)")
