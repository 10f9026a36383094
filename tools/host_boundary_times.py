"""What crossing the boundary costs: the frame read back to the host (evplp_resolve: composite + 3 floats per pixel over PCIe) against the
device-side composite alone (evplp_present), and the one-off scene hand-over (OBJ parse + upload + BVH build).  Config #2's frame, one MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evplp_amd as ev
W = H = 1024
jp = ev.synth_scene("/tmp/evplp_boundary", "conf", 331000, 1234, W, H, style="hard")
t0 = time.perf_counter()
c = ev.Context(W, H, 1024, 1024, 4, overlap_light_tracing=True)
c.load_scene_json(jp); c.synchronize()
print("scene hand-over (OBJ / MTL parse, upload, SAH build, 331 k triangles): %.0f ms, once" % ((time.perf_counter() - t0) * 1e3))
cam = c.camera()
fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", num_light_paths=1024, num_vpl_light_paths=1024, photons_per_path=4, do_accumulate=1)
def frame(read_back):
    c.primary((0, 0)); c.trace_light_paths(0); c.gather_vpl(fp)
    return c.resolve(1.0, 0.0, 1.0) if read_back else c.present(1.0, 0.0, 1.0)
for rb in (False, True, False, True):
    for _ in range(3): frame(rb)
    c.synchronize(); t0 = time.perf_counter()
    for _ in range(20): frame(rb)
    c.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print("frame with %s: %.3f ms" % ("evplp_resolve (frame on the host: %.1f MB over PCIe)" % (W * H * 12 / 1e6) if rb else "evplp_present (frame stays on the device)", dt))
