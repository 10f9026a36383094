"""Per-rank gather time of an N-way strip partition, simulated on one GPU (one context per rank, run in turn)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evplp_amd as ev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = "/tmp/evplp_strips"
jp = ev.synth_scene(d, "conf", 331000, 1234, 1024, 1024)
for rows in (8, 16, 32):
    times = []
    for r in range(N):
        with ev.Context(1024, 1024, 1024, 1024, 4, strip_rank=r, strip_count=N, strip_rows=rows) as c:
            c.load_scene_json(jp)
            cam = c.camera()
            kw = dict(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=1024, num_vpl_light_paths=1024, photons_per_path=4)
            for it in range(2):
                c.primary((0, 0)); c.trace_light_paths(it); c.gather_vpl(ev.frame_params(**kw)); c.synchronize()
            times.append(c.pass_stats(ev.PASS_GATHER_VPL)["ms"])
    print("strip_rows %2d: per-rank gather ms %s | max %.2f mean %.2f (balance %.2f)" % (rows, " ".join("%.2f" % t for t in times), max(times), sum(times) / N, sum(times) / N / max(times)))
