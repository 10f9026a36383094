"""Soak of evplp_group (developer check, one GPU): eight virtual ranks, dealt blocks, 3 000 accumulating iterations of gather + splat with the strips
exchanged every iteration, a synchronize every 97th iteration and a resolve every 500th; the final frame must equal the single context's bit for bit
(deterministic mode).  What it is for: the rings, the barrier's verdict and the virtual exchange under sustained load."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import evplp_amd as ev
W, H, NL, NV, P, N = 160, 128, 4096, 64, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
jp = ev.synth_scene("/tmp/evplp_soak", "room", 3000, 9, W, H, style="hard")


def run(runner, group):
    c0 = runner.rank(0) if group else runner
    cam = c0.camera(); bsr, total, _ = c0.scene_metrics(); r = 0.05 * bsr
    kw = dict(camera_pos=list(cam.origin), mis_mode="balance", pdf_mc=(NV / NL) / math.pi / (r * r), clamping_value=1.0 / total, photon_radius=r,
              num_light_paths=NL, num_vpl_light_paths=NV, photons_per_path=P, do_accumulate=1)
    if group:
        runner.calibrate(True); runner.primary((0, 0)); runner.trace_light_paths(0); runner.gather(ev.frame_params(**kw), 0); runner.rebalance()
    runner.clear_accumulators()
    t0 = time.perf_counter()
    for it in range(N):
        fp = ev.frame_params(**kw, rng_seed=it)
        runner.primary((0.001, -0.002)); runner.trace_light_paths(it)
        (runner.gather(fp, 0) if group else runner.gather_vpl(fp)); runner.splat_photons(fp)
        runner.present(1.0 / (it + 1), 1.0 / (it + 1), 1.0, mask_emitter=True, gamma=True)
        if it % 97 == 96:
            runner.synchronize()
        if it % 500 == 499:
            runner.resolve(1.0 / (it + 1), 1.0 / (it + 1), 1.0)
    img = runner.resolve(1.0 / N, 1.0 / N, 1.0)[:H]
    return img, time.perf_counter() - t0


with ev.Context(W, H, NL, NV, P, deterministic=True, overlap_light_tracing=True) as c:
    c.load_scene_json(jp)
    ref, t1 = run(c, False)
with ev.Group(W, H, NL, NV, P, 8, devices=[0] * 8, deterministic=True, overlap_light_tracing=True, split_light_paths=1) as g:
    g.load_scene_json(jp)
    img, t8 = run(g, True)
    owners = g.block_owners().tolist()
print(f"{N} iterations: single context {t1:.1f} s, eight dealt virtual ranks {t8:.1f} s; owners {owners}; frames equal: {img.tobytes() == ref.tobytes()}; max {ref.max():.4g}")
sys.exit(0 if img.tobytes() == ref.tobytes() and ref.max() > 0 else 1)
