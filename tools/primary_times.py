"""When does every tile's wavefront of the G-buffer pass start and end?  Config #4's frame (1920 x 1080, the furnished scene) through a
build with -DEVPLP_PRIMARY_TIMES=1 (make VARIANT=ptimes EXTRA_HIPFLAGS=-DEVPLP_PRIMARY_TIMES=1 evplp_amd/lib/libevplp_hip_ptimes.so):
distribution of the waves' lifetimes, waves in flight over time, and what sets the duration of the launch."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EVPLP_LIB"] = os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_ptimes.so")
sys.path.insert(0, ROOT)
import evplp_amd as ev

W, H = 1920, 1080
jp = ev.synth_scene("/tmp/evplp_pt", "living", 331000, 1234, W, H, style="hard")
with ev.Context(W, H, 1024, 1024, 4) as c:
    c.load_scene_json(jp)
    for it in range(3):
        c.primary((0.0003, -0.0002)); c.synchronize()
    ms = c.pass_stats(ev.PASS_PRIMARY)["ms"]
    lib = ev.lib()
    n = ((W + 7) // 8) * ((H + 7) // 8)
    buf = (C.c_ulonglong * (2 * n))()
    assert lib.evplp_debug_primary_times(buf, 2 * n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.float64) / 100.0      # us (100 MHz)
t -= t[:, 0].min()
life = t[:, 1] - t[:, 0]
end = t[:, 1].max()
print(f"pass {ms * 1e3:.1f} us by events; first start -> last end {end:.1f} us; {n} waves")
print("lifetime us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max()))
print("sum of lifetimes / (8192 slots): %.1f us" % (life.sum() / 8192))
edges = np.linspace(0, end, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    inflight = int(((t[:, 0] <= mid) & (t[:, 1] > mid)).sum())
    started = int(((t[:, 0] >= a) & (t[:, 0] < b)).sum())
    print("t %6.1f us: %5d waves in flight (%.2f per SIMD), %5d started in this twentieth" % (mid, inflight, inflight / 1024, started))
tx = (W + 7) // 8
slow = np.argsort(-life)[:10]
print("slowest tiles (tx, ty, lifetime us, start us):", [(int(i % tx), int(i // tx), round(float(life[i]), 1), round(float(t[i, 0]), 1)) for i in slow])
