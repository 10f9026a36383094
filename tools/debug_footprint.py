"""Which (pixel, photon) pairs does the HIP proxy footprint count differently from the oracle?  One config-#4-like iteration on the GPU
(textured room, 1920 x 1080, 300 000 light paths, radius 0.3 %), the oracle's proxy image on a few rows, and for every pixel that
differs: what it shows, by how many whole pair contributions it is off, its per-pair verdicts by a CPU replay of the kernel's arithmetic
(fp32 slabs) next to the oracle's ray / triangle count.  Developer tool (GPU box): python tools/debug_footprint.py [rows...]"""
import ctypes as C
import json
import math
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import evplp_amd as ev          # noqa: E402
import oracle_api as oa         # noqa: E402
import scenes                   # noqa: E402


def main():
    rows = [int(v) for v in sys.argv[1:]] or [300, 905]
    W, H, NL, P = 1920, 1080, 300000, 4
    d = tempfile.mkdtemp()
    jp = ev.synth_scene(d, "living", 120000, 11, W, H, style="textured")
    sd, _ = scenes.load_obj_scene(jp, decode=lambda p: ev.decode_image(p)[0])
    if os.environ.get("LOOP"):
        return loop(jp, sd, rows, W, H, NL, P, int(os.environ["LOOP"]))
    jit = ev.jitter_sequence(7, 3, W, H)[2]
    jitter = (float(jit[0]), float(jit[1]))
    with ev.Context(W, H, NL, 0, P, deterministic=bool(int(os.environ.get('DET', '0')))) as c:
        c.load_scene_json(jp)
        sd.fovy = c.camera().fovy
        bsr, total, _ = c.scene_metrics()
        r = 0.003 * bsr
        kw = dict(camera_pos=sd.cam_origin, mis_mode=0, pdf_mc=0.0, clamping_value=1.0 / total, photon_radius=r, num_light_paths=NL, num_vpl_light_paths=0,
                  photons_per_path=P, jitter=jitter)
        c.primary(jitter); c.trace_light_paths(9)
        c.splat_photons(ev.frame_params(**kw, splat_footprint="proxy"), clear=True)
        got = c.download(ev.BUF_PHOTON_ACCUM)[:H]
        st = c.pass_stats(ev.PASS_SPLAT)
        c.splat_photons(ev.frame_params(**kw), clear=True)
        got_ideal = c.download(ev.BUF_PHOTON_ACCUM)[:H]
        gbuf = [c.download(b)[:H] for b in (ev.BUF_GBUF_POSITION, ev.BUF_GBUF_NORMAL, ev.BUF_GBUF_DIFFUSE, ev.BUF_GBUF_PHONG)]
        rec = c.download(ev.BUF_RECORDS)
    print("product: pairs", st["pairs"], "fragments", st["rays"], "ratio", st["rays"] / max(st["pairs"], 1))
    osc = oa.Scene(sd); l = oa.load()
    l.evo_proxy_faces_in_front.restype = C.c_int
    l.evo_proxy_faces_in_front.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    mv, mf = oa.icosphere42()
    okw = dict(kw)
    f32 = np.float32
    o = np.array(sd.cam_origin, f32); la = np.array(sd.cam_lookat, f32); up = np.array(sd.cam_up, f32)

    def nrm(v):
        return (v / np.sqrt((v * v).sum(dtype=f32), dtype=f32)).astype(f32)
    f = nrm(la - o); s = nrm(np.cross(f, up).astype(f32)); u = np.cross(s, f).astype(f32)
    th = f32(math.tan(sd.fovy / 2)); asp = f32(sd.aspect)
    usable = np.nonzero(rec["flags"] & 2)[0]
    pos = rec["pos"][usable]
    for y in rows:
        ideal, proxy, ost = oa.splat_proxy(oa.frame_params(**okw), osc.camera(), W, H, gbuf, rec, rows=(y, y + 1))
        a = got[y, :, :3].astype(np.float64); b = proxy[y, :, :3].astype(np.float64); ai = got_ideal[y, :, :3].astype(np.float64); bi = ideal[y, :, :3].astype(np.float64)
        rl = np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()); rli = np.sqrt(((ai - bi) ** 2).sum() / (bi ** 2).sum())
        bad = np.nonzero((np.abs(a - b) > 2e-4 * np.maximum(b, 1e-3 * b.max()) + 1e-9).any(-1))[0]
        print(f"row {y}: oracle pairs {int(ost[0])} fragments {int(ost[3])}; rel L2 proxy {rl:.3e} ideal {rli:.3e}; {bad.size} pixels off")
        for x in bad[:12]:
            X = gbuf[0][y, x, :3]
            d2 = ((pos - X) ** 2).sum(-1, dtype=f32)
            near = usable[d2 <= f32(r) * f32(r)]
            cx = f32((x + 0.5) / W * 2 - 1); cy = f32((y + 0.5) / H * 2 - 1)
            dj = (s * f32(f32(cx - f32(jitter[0])) * asp * th) + u * f32(f32(cy - f32(jitter[1])) * th) + f).astype(np.float64)
            e = o.astype(np.float64)
            tsurf = float((X.astype(np.float64) - e) @ f.astype(np.float64))
            counts = [l.evo_proxy_faces_in_front(oa.ptr(mv), oa.ptr(mf), 80, oa.ptr(np.ascontiguousarray(rec["pos"][i])), r, e.ctypes.data, dj.ctypes.data, 0.1, tsurf * (1 + 1e-7)) for i in near]
            off = np.linalg.norm(X - (e + tsurf * dj))
            print(f"   x {x}: product {a[x]} oracle {b[x]} ideal-sum {bi[x]}; {near.size} pairs, oracle face counts {counts}; |X - ray| {off:.2e} depth {tsurf:.3f}; "
                  f"rho_d {gbuf[2][y, x, :3]} rho_s {gbuf[3][y, x]}; |q|/r {[round(float(math.sqrt(v)) / r, 4) for v in d2[d2 <= f32(r) * f32(r)]]}")


def loop(jp, sd, rows, W, H, NL, P, iters):
    """LOOP=n: the first n iterations of config #4's progressive schedule, every iteration's photon image (cleared every time) against
    the oracle's on the given rows -- which iteration, which pixels?"""
    jits = ev.jitter_sequence(7, iters, W, H)
    with ev.Context(W, H, NL, 0, P, overlap_light_tracing=bool(int(os.environ.get("OVERLAP", "1")))) as c:
        c.load_scene_json(jp)
        sd.fovy = c.camera().fovy
        osc = oa.Scene(sd)
        bsr, total, _ = c.scene_metrics()
        radius, clamp, pdf = 0.003 * bsr, 1.0 / total, 0.0
        clamp0 = clamp
        for it in range(iters):
            jitter = (float(jits[it][0]), float(jits[it][1]))
            kw = dict(camera_pos=sd.cam_origin, mis_mode=0, pdf_mc=pdf, clamping_value=clamp, photon_radius=radius, num_light_paths=NL, num_vpl_light_paths=0,
                      photons_per_path=P, jitter=jitter)
            c.trace_light_paths(it + 7); c.primary(jitter)
            c.splat_photons(ev.frame_params(**kw, splat_footprint="proxy"), clear=True)
            got = c.download(ev.BUF_PHOTON_ACCUM)[:H]
            st = c.pass_stats(ev.PASS_SPLAT)
            gbuf = [c.download(b)[:H] for b in (ev.BUF_GBUF_POSITION, ev.BUF_GBUF_NORMAL, ev.BUF_GBUF_DIFFUSE, ev.BUF_GBUF_PHONG)]
            rec = c.download(ev.BUF_RECORDS)
            line = f"it {it} r {radius:.5f} bins max {st['nodes'] >> 32}:"
            for y in rows:
                ideal, proxy, ost = oa.splat_proxy(oa.frame_params(**kw), osc.camera(), W, H, gbuf, rec, rows=(y, y + 1))
                a = got[y, :, :3].astype(np.float64); b = proxy[y, :, :3].astype(np.float64)
                bad = np.nonzero((np.abs(a - b) > 2e-4 * np.maximum(b, 1e-3 * b.max()) + 1e-9).any(-1))[0]
                line += f" row {y} relL2 {np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()):.2e} bad {bad.size} {bad[:6].tolist()}"
                for x in bad[:3]:
                    line += f" [x {x} got {a[x][0]:.5f} want {b[x][0]:.5f} ideal {ideal[y, x, 0]:.5f} rho {gbuf[2][y, x, 0]:.3f}]"
            if "bad 0 [] row" not in line or not line.rstrip().endswith("bad 0 []") or it % 10 == 0:
                print(line, flush=True)
            radius, clamp, pdf, _, _ = ev.progressive_step(it + 1, 0.7, clamp0, 0, NL, radius, clamp, pdf)


if __name__ == "__main__":
    main()
