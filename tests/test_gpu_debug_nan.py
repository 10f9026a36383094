"""The -DEVPLP_DEBUG_NAN build (`make nan` -> evplp_amd/lib/libevplp_hip_nan.so): the gathers and the photon splat count every pixel
whose partial sum comes out non-finite -- the reference's ASSERT under DEBUG (realtimetechniques/all.cuh:10-17) -- run once here,
outside every timed path, in a process of its own (the library is chosen at import)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "evplp_amd", "lib", "libevplp_hip_nan.so")

SCRIPT = r'''
import json, math, sys, os
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import evplp_amd as ev, scenes
W, H, N, P = 96, 64, 64, 4
room = scenes.box_room(seed=5, n_boxes=5, tess=2, aspect=W / H)
NONFINITE = 196          # word of PassCounters (kernels.h)
out = {}
with ev.Context(W, H, N, N, P, device=0) as c:
    room.upload(c)
    c.primary((0.0, 0.0), clear_light=True); c.trace_light_paths(1)
    bsr, total, _ = c.scene_metrics()
    r = 0.05 * bsr
    bad = 0
    for mode in range(6):
        fp = ev.frame_params(camera_pos=room.cam_origin, mis_mode=mode, pdf_mc=1.0 / (math.pi * r * r), clamping_value=1.0 / total, photon_radius=r,
                             vsl_radius=r, vsl_inv_pi_radius2=1.0 / (math.pi * r * r), num_light_paths=N, num_vpl_light_paths=N, photons_per_path=P, rng_seed=1)
        c.gather_vpl(fp); bad += int(c.debug_counters(ev.PASS_GATHER_VPL)[NONFINITE])
        c.splat_photons(fp, clear=True); c.synchronize(); bad += int(c.debug_counters(ev.PASS_SPLAT)[NONFINITE])
    c.gather_vsl(fp); bad += int(c.debug_counters(ev.PASS_GATHER_VSL)[NONFINITE])
    out["clean"] = bad
    # the trap itself: a usable VPL whose flux is NaN must be counted by every pixel it lights (a NaN POSITION is not enough: the cosine
    # test fmaxf(NaN, 0) = 0 drops such a pair, in the reference too)
    rec = c.download(ev.BUF_RECORDS).copy()
    rec["flux"][0] = np.nan              # record 0: the on-light vertex of path 0, always a usable VPL
    c.upload(ev.BUF_RECORDS, rec)
    c.gather_vpl(fp)
    out["poisoned"] = int(c.debug_counters(ev.PASS_GATHER_VPL)[NONFINITE])
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_no_non_finite_partial_sums_in_the_debug_build():
    assert os.path.exists(LIB), "make nan (part of `make all`) builds the debug library"
    env = dict(os.environ, EVPLP_LIB=LIB)
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["clean"] == 0, f"{res['clean']} non-finite partial sums on a clean frame (six MIS modes, VPL + VSL gather, photon splat)"
    assert res["poisoned"] >= 1, "the debug build did not count the pixels lit by a VPL whose flux is NaN"
