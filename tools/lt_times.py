"""When does every wavefront of light_trace_kernel start and end?  Config #4's 300 000 light paths on the furnished scene through a build with
-DEVPLP_LT_TIMES=1 (make VARIANT=lttimes EXTRA_HIPFLAGS=-DEVPLP_LT_TIMES=1 evplp_amd/lib/libevplp_hip_lttimes.so)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EVPLP_LIB"] = os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_lttimes.so")
sys.path.insert(0, ROOT)
import evplp_amd as ev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
jp = ev.synth_scene("/tmp/evplp_ltt", "living", 331000, 1234, 1920, 1080, style="hard")
with ev.Context(1920, 1080, N, 1024, 4) as c:
    c.load_scene_json(jp)
    for it in range(3):
        c.trace_light_paths(it); c.synchronize()
    ms = c.pass_stats(ev.PASS_LIGHT_TRACE)["ms"]
    n = (N + 63) // 64
    buf = (C.c_ulonglong * (2 * n))()
    assert ev.lib().evplp_debug_lt_times(buf, 2 * n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.float64) / 100.0
t -= t[:, 0].min()
life = t[:, 1] - t[:, 0]; end = t[:, 1].max()
print(f"{N} paths: pass {ms * 1e3:.0f} us by events; first start -> last end {end:.0f} us; {n} waves")
print("lifetime us: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f; started after 10 us: %d waves (median start %.0f us)" % (life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), int((t[:, 0] > 10).sum()), float(np.median(t[t[:, 0] > 10, 0])) if (t[:, 0] > 10).any() else 0.0))
for q in range(10):
    mid = (q + 0.5) * end / 10
    print("t %5.0f us: %5d waves in flight (%.2f per SIMD), %5d started" % (mid, int(((t[:, 0] <= mid) & (t[:, 1] > mid)).sum()), ((t[:, 0] <= mid) & (t[:, 1] > mid)).sum() / 1024, int(((t[:, 0] >= q * end / 10) & (t[:, 0] < (q + 1) * end / 10)).sum())))
late = t[:, 0] > 10
print("last end among the waves that started at once: %.0f us; among the late starters: %.0f us" % (t[~late, 1].max(), t[late, 1].max() if late.any() else 0))
