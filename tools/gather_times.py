"""When does a wavefront (= item: a tile and k splits of the VPL list) of gather_vpl_kernel start and end?  Every 16th item of config #2's
frame through a build with -DEVPLP_GATHER_TIMES=1 (make VARIANT=gtimes EXTRA_HIPFLAGS=-DEVPLP_GATHER_TIMES=1 evplp_amd/lib/libevplp_hip_gtimes.so)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EVPLP_LIB"] = os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_gtimes.so")
sys.path.insert(0, ROOT)
import evplp_amd as ev
style = sys.argv[1] if len(sys.argv) > 1 else "hard"
# optional: strip_count strip_rank strip_rows (one rank of an n-way partition, alone on the GPU)
SC, SRK, SRW = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 0, 16)
jp = ev.synth_scene("/tmp/evplp_gt_" + style, "conf", 331000, 1234, 1024, 1024, style=style)
# BLOCKS=b0,b1,...: the rank's block table (evplp_set_blocks), e.g. what tools/strip_projection.py printed as "owner" for a deal by cost
BLOCKS = [int(v) for v in os.environ["BLOCKS"].split(",")] if os.environ.get("BLOCKS") else None
with ev.Context(1024, 1024, 1024, 1024, 4, strip_count=SC, strip_rank=SRK, strip_rows=SRW, strip_capacity_rows=(len(BLOCKS) * SRW if BLOCKS else 0)) as c:
    c.load_scene_json(jp)
    if BLOCKS:
        c.set_blocks(BLOCKS)
    cam = c.camera()
    fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=1024, num_vpl_light_paths=1024, photons_per_path=4, do_accumulate=0)
    for it in range(3):
        c.primary((0, 0)); c.trace_light_paths(it); c.gather_vpl(fp); c.synchronize()
    st = c.pass_stats(ev.PASS_GATHER_VPL)
    n = 65536
    buf = (C.c_ulonglong * (2 * n))()
    assert ev.lib().evplp_debug_gather_times(buf, 2 * n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.float64) / 100.0
t = t[t[:, 1] > 0]
t -= t[:, 0].min()
life = t[:, 1] - t[:, 0]; end = t[:, 1].max()
print(f"{style} strips {SC}/{SRK}/{SRW}: gather {st['dominant_kernel_ms']:.2f} ms by events; sampled items {len(t)} (every 16th); first start -> last end {end / 1e3:.2f} ms")
print("item lifetime us: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max()))
for q in range(20):
    mid = (q + 0.5) * end / 20
    infl = ((t[:, 0] <= mid) & (t[:, 1] > mid)).sum() * 16
    print("t %6.2f ms: ~%5d waves in flight (%.2f per SIMD)" % (mid / 1e3, infl, infl / 1024))
slow = np.argsort(-life)[:5]
print("slowest sampled items (lifetime us, start ms):", [(round(float(life[i])), round(float(t[i, 0]) / 1e3, 2)) for i in slow])
