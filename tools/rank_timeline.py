"""One rank's frame on a time axis (developer probe; run under rocprofv3 --kernel-trace, see tools/rank_timeline.sh): a strip context of an n-way partition
with a dealt block table renders a few config-#2 frames exactly as the loops do (light tracing overlapped)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evplp_amd as ev
W = H = 1024
BLOCKS = [int(v) for v in os.environ.get("BLOCKS", "33,16,14,41,4,52,54,48").split(",")]
jp = ev.synth_scene("/tmp/evplp_rank_tl", "conf", 331000, 1234, W, H, style="hard")
with ev.Context(W, H, 1024, 1024, 4, strip_rank=0, strip_count=8, strip_rows=16, strip_capacity_rows=16 * len(BLOCKS), overlap_light_tracing=True) as c:
    c.load_scene_json(jp); c.set_blocks(BLOCKS)
    c.profile_passes(False)
    cam = c.camera()
    for it in range(6):
        j = tuple(float(v) for v in ev.jitter_sequence(0, it + 1, W, H)[it])
        fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=1024, num_vpl_light_paths=1024, photons_per_path=4, do_accumulate=1, jitter=j, rng_seed=it)
        c.primary(j); c.trace_light_paths(it); c.gather_vpl(fp); c.present(1.0 / (it + 1), 0.0, 1.0, mask_emitter=True, gamma=True)
    c.synchronize()
