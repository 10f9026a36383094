"""Timing probe: light tracing alone (1920x1080 context, furnished scene) for growing path counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
import evplp_amd as ev
W, H = 1920, 1080
jp = ev.synth_scene("/tmp/evplp_lt_probe", "conf", 331000, 1234, W, H, style="hard")
for n in [int(x) for x in (sys.argv[1:] or ["75000", "150000", "300000", "600000", "1200000"])]:
    with ev.Context(W, H, n, 0, 4) as c:
        c.load_scene_json(jp)
        ms = []
        for it in range(6):
            c.trace_light_paths(it); c.synchronize(); ms.append(c.pass_stats(ev.PASS_LIGHT_TRACE)["ms"])
        m = sum(ms[2:]) / len(ms[2:])
        print(n, "paths: %.3f ms -> %.0f Mpaths/s, %.2f waves/SIMD" % (m, n / m / 1e3, n / 64 / 1024), c.accel_info(), flush=True)
