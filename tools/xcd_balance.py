"""Developer probe: when does the last item of every XCD end in the VPL gather?  Needs the probe build:
    make VARIANT=xcd EXTRA_HIPFLAGS="-DEVPLP_DEBUG_NAN=1 -DEVPLP_XCD_TIMES=1" evplp_amd/lib/libevplp_hip_xcd.so
    EVPLP_LIB=evplp_amd/lib/libevplp_hip_xcd.so python3 tools/xcd_balance.py [--scene hard|easy]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EVPLP_LIB", os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_xcd.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import evplp_amd as ev

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="hard")
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--paths", type=int, default=1024)
ap.add_argument("--vsl", action="store_true", help="the VSL gather (estimator kernel: first row, walk kernel: second row)")
ap.add_argument("--strip-count", type=int, default=1)
ap.add_argument("--strip-rank", type=int, default=0)
ap.add_argument("--strip-rows", type=int, default=16)
a = ap.parse_args()
jp = ev.synth_scene("/tmp/evplp_xcd_%s" % a.scene, "conf", 331000, 1234, a.res, a.res, style=a.scene)
c = ev.Context(a.res, a.res, a.paths, a.paths, 4, strip_rank=a.strip_rank, strip_count=a.strip_count, strip_rows=a.strip_rows)
c.load_scene_json(jp)
cam = c.camera()
import math
bsr, total, larea = c.scene_metrics()
fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=a.paths, num_vpl_light_paths=a.paths, photons_per_path=4, do_accumulate=0,
                     vsl_radius=0.05 * bsr, vsl_inv_pi_radius2=1 / (math.pi * (0.05 * bsr) ** 2))
for it in range(3):
    c.primary((0, 0)); c.trace_light_paths(it)
    which = ev.PASS_GATHER_VSL if a.vsl else ev.PASS_GATHER_VPL
    (c.gather_vsl if a.vsl else c.gather_vpl)(fp); c.synchronize()
    st = c.pass_stats(which)
    raw = c.debug_counters(which)
    for name, off in ((("estimators", 16), ("walks", 24)) if a.vsl else (("gather", 16),)):
        ends = raw[4 + off:4 + off + 8].astype(np.int64)
        rel = (ends - ends.min()) / 100.0            # microseconds after the first XCD to finish
        print("iter %d %s: pass kernels %.2f ms; last item of XCD 0..7 ends +%s us after the earliest; spread %.0f us = %.2f %% of the pass"
              % (it, name, st["dominant_kernel_ms"], np.round(rel, 0).astype(int).tolist(), rel.max(), rel.max() / 10.0 / st["dominant_kernel_ms"]))
