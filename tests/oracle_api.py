"""ctypes wrapper of oracle/liboracle.so -- the CPU restatement used as the checker.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")

RECORD_DTYPE = np.dtype([
    ("pos", np.float32, 3), ("flags", np.uint32), ("normal", np.float32, 3), ("p_select_lambert", np.float32),
    ("flux", np.float32, 3), ("pad1", np.float32), ("flux_dir", np.float32, 3), ("pad2", np.float32),
    ("rho_d", np.float32, 3), ("pad3", np.float32), ("rho_s", np.float32, 3), ("phong_exp", np.float32)])


class Material(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("ns", C.c_float), ("light", C.c_float * 4),
                ("tex_kd", C.c_int32), ("tex_ks", C.c_int32), ("tex_ns", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("rgba", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3), ("fovy", C.c_float), ("aspect", C.c_float)]


class FrameParams(C.Structure):
    _fields_ = [("camera_pos", C.c_float * 3), ("mis_mode", C.c_uint32), ("pdf_mc", C.c_float),
                ("clamping_value", C.c_float), ("photon_radius", C.c_float), ("vsl_radius", C.c_float),
                ("vsl_inv_pi_radius2", C.c_float), ("num_light_paths", C.c_uint32),
                ("num_vpl_light_paths", C.c_uint32), ("photons_per_path", C.c_uint32),
                ("do_accumulate", C.c_uint32), ("rng_seed", C.c_uint32), ("jitter", C.c_float * 2)]


class Rng(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64), ("s0", C.c_uint32), ("s1", C.c_uint32), ("vsl_draw", C.c_int32), ("reserved", C.c_uint32)]


_lib = None
_P = C.c_void_p


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise ImportError(f"{LIB} missing: run `make -C oracle`")
    l = C.CDLL(LIB)
    l.evo_scene_create.restype = _P
    l.evo_scene_create.argtypes = [C.c_int32, _P, _P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P]
    l.evo_scene_destroy.argtypes = [_P]
    for f in ("evo_scene_light_area", "evo_scene_total_area", "evo_scene_bounding_sphere_radius"):
        getattr(l, f).restype = C.c_float
        getattr(l, f).argtypes = [_P]
    l.evo_set_threads.argtypes = [C.c_int]
    l.evo_get_threads.restype = C.c_int
    l.evo_tri_test.restype = C.c_int
    l.evo_tri_test.argtypes = [_P, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P]
    l.evo_occluded.restype = C.c_int
    l.evo_occluded.argtypes = [_P, _P, _P, C.c_float, C.c_float]
    l.evo_occluded_brute.restype = C.c_int
    l.evo_occluded_brute.argtypes = [_P, _P, _P, C.c_float, C.c_float]
    l.evo_closest.restype = C.c_int
    l.evo_closest.argtypes = [_P, _P, _P, C.c_float, C.c_float, C.c_int, _P, _P, _P]
    l.evo_rng_init.argtypes = [_P, C.c_uint32, C.c_uint32, C.c_uint32]
    l.evo_vsl_rng_init.argtypes = [_P, C.c_uint32, C.c_uint32, C.c_uint32]
    l.evo_vsl_rng_step.argtypes = [_P]
    l.evo_rng_u32.restype = C.c_uint32
    l.evo_rng_u32.argtypes = [_P]
    l.evo_rng_uniform.restype = C.c_float
    l.evo_rng_uniform.argtypes = [_P]
    l.evo_tri_area.restype = C.c_float
    l.evo_tri_area.argtypes = [_P]
    l.evo_math_sincos.argtypes = [C.c_float, _P, _P]
    l.evo_math_sincos_array.argtypes = [_P, C.c_int, _P, _P]
    l.evo_math_pow_array.argtypes = [_P, _P, C.c_int, _P]
    l.evo_math_pow.restype = C.c_float
    l.evo_math_pow.argtypes = [C.c_float, C.c_float]
    l.evo_phong_eval_f.restype = C.c_float
    l.evo_phong_eval_f.argtypes = [_P, _P, _P, C.c_float]
    l.evo_lambert_pdf_a.restype = C.c_float
    l.evo_lambert_pdf_a.argtypes = [_P, _P, _P]
    l.evo_phong_pdf_a.restype = C.c_float
    l.evo_phong_pdf_a.argtypes = [_P, _P, _P, _P, _P, C.c_float]
    l.evo_phong_pdf_w.restype = C.c_float
    l.evo_phong_pdf_w.argtypes = [_P, _P, _P, _P, C.c_float]
    l.evo_primary.argtypes = [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32]
    l.evo_trace_light_paths.argtypes = [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P]
    l.evo_vpl_splat_pair.argtypes = [_P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_int, _P]
    l.evo_vsl_splat_pair.argtypes = [_P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, _P]
    l.evo_gather_vpl.argtypes = [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]
    l.evo_gather_vpl_counts.argtypes = [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, _P]
    l.evo_gather_vsl.argtypes = l.evo_gather_vpl.argtypes
    l.evo_gather_vsl_window.argtypes = [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]
    l.evo_gather_lvc.argtypes = l.evo_gather_vpl.argtypes
    l.evo_photon_frag.restype = C.c_int
    l.evo_photon_frag.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P]
    l.evo_splat_photons.argtypes = [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_uint32, _P, _P]
    l.evo_splat_photons_proxy.argtypes = [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_uint32, _P, _P, _P]
    l.evo_splat_photons_proxy_mesh.argtypes = [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_uint32, _P, _P, C.c_int32, _P, _P, _P]
    l.evo_icosphere42.argtypes = [_P, _P]
    l.evo_resolve.argtypes = [C.c_int32, C.c_int32, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P]
    l.evo_progressive_step.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _P, _P, _P, C.c_int, _P, _P]
    l.evo_path_trace.restype = C.c_uint64
    l.evo_path_trace.argtypes = [_P, _P, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int]
    l.evo_write_pfm.restype = C.c_int
    l.evo_write_pfm.argtypes = [C.c_char_p, C.c_int32, C.c_int32, _P]
    l.evo_png_bytes.argtypes = [C.c_int32, _P, _P]
    l.evo_mse.restype = C.c_double
    l.evo_mse.argtypes = [C.c_int32, _P, _P]
    l.evo_rel_mse.restype = C.c_double
    l.evo_rel_mse.argtypes = [C.c_int32, _P, _P]
    _lib = l
    return l


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f3(v):
    return np.ascontiguousarray(v, dtype=np.float32)


class Scene:
    """Oracle-side scene built from a scenes.SceneData."""

    def __init__(self, sd):
        l = load()
        self.sd = sd
        tris = sd.triangle_soup()
        self.verts, self.uvs, self.mat = tris
        mats = (Material * len(sd.materials))()
        for i, m in enumerate(sd.materials):
            mats[i].kd = (C.c_float * 3)(*m["kd"]); mats[i].ks = (C.c_float * 3)(*m["ks"]); mats[i].ns = m["ns"]
            mats[i].tex_kd = m.get("tex_kd", -1); mats[i].tex_ks = m.get("tex_ks", -1); mats[i].tex_ns = m.get("tex_ns", -1)
        self._tex_keep = [np.ascontiguousarray(t, dtype=np.float32) for t in sd.textures]
        texs = (Texture * max(len(sd.textures), 1))()
        for i, t in enumerate(self._tex_keep):
            texs[i].w = t.shape[1]; texs[i].h = t.shape[0]; texs[i].rgba = t.ctypes.data
        inten = f3(sd.light_intensity)
        self.h = l.evo_scene_create(self.mat.shape[0], ptr(self.verts), ptr(self.uvs), ptr(self.mat), len(sd.materials), mats,
                                    len(sd.textures), texs, sd.light_first, sd.light_count, ptr(inten))
        self.lib = l

    def __del__(self):
        try:
            self.lib.evo_scene_destroy(self.h)
        except Exception:
            pass

    def camera(self):
        c = Camera()
        sd = self.sd
        c.origin = (C.c_float * 3)(*sd.cam_origin); c.lookat = (C.c_float * 3)(*sd.cam_lookat); c.up = (C.c_float * 3)(*sd.cam_up)
        c.fovy = sd.fovy; c.aspect = sd.aspect
        return c

    def primary(self, W, H, jitter=(0.0, 0.0), rows=None, light_unoccluded=False):
        planes = [np.zeros((H, W, 4), dtype=np.float32) for _ in range(5)]
        j = f3(jitter)
        cam = self.camera()
        r0, r1 = rows if rows else (0, H)
        self.lib.evo_primary(self.h, C.byref(cam), W, H, ptr(j), r0, r1, *[ptr(p) for p in planes], int(light_unoccluded))
        return planes

    def trace_light_paths(self, seed, npaths, P, begin=0, count=None, records=None):
        if records is None:
            records = np.zeros(npaths * P, dtype=RECORD_DTYPE)
        self.lib.evo_trace_light_paths(self.h, seed, begin, npaths - begin if count is None else count, P, ptr(records))
        return records

    def gather(self, fp, W, H, gbuf, records, out=None, vsl=False, rows=None, lvc=False):
        if out is None:
            out = np.zeros((H, W, 4), dtype=np.float32)
        pairs = C.c_uint64()
        r0, r1 = rows if rows else (0, H)
        fn = self.lib.evo_gather_lvc if lvc else self.lib.evo_gather_vsl if vsl else self.lib.evo_gather_vpl
        fn(self.h, C.byref(fp), W, H, r0, r1, ptr(gbuf[0]), ptr(gbuf[1]), ptr(gbuf[2]), ptr(gbuf[3]), ptr(records), ptr(out), C.byref(pairs))
        return out, pairs.value

    def gather_vsl_window(self, fp, W, H, gbuf, records, out, rows, xs):
        """VSL gather on the pixels xs = (x0, x1) of the rows (r0, r1) only"""
        pairs = C.c_uint64()
        self.lib.evo_gather_vsl_window(self.h, C.byref(fp), W, H, rows[0], rows[1], xs[0], xs[1], ptr(gbuf[0]), ptr(gbuf[1]), ptr(gbuf[2]), ptr(gbuf[3]),
                                       ptr(records), ptr(out), C.byref(pairs))
        return out, pairs.value

    def gather_counts(self, fp, W, gbuf, records, rows):
        """(shadow rays traced, unoccluded pairs) of the VPL gather over the given image rows."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        rays, lit = C.c_uint64(), C.c_uint64()
        self.lib.evo_gather_vpl_counts(self.h, C.byref(fp), W, ptr(rows), rows.shape[0], ptr(gbuf[0]), ptr(gbuf[1]), ptr(records), C.byref(rays), C.byref(lit))
        return rays.value, lit.value

    def path_trace(self, cam_pos, seed, max_bounces, W, H, gbuf, out=None, accumulate=True, rows=None):
        if out is None:
            out = np.zeros((H, W, 4), dtype=np.float32)
        r0, r1 = rows if rows else (0, H)
        c = f3(cam_pos)
        n = self.lib.evo_path_trace(self.h, ptr(c), seed, max_bounces, W, H, r0, r1, ptr(gbuf[0]), ptr(gbuf[1]), ptr(gbuf[2]), ptr(gbuf[3]), ptr(out), int(accumulate))
        return out, n


def splat(fp, W, H, gbuf, records, out=None, rows=None):
    l = load()
    if out is None:
        out = np.zeros((H, W, 4), dtype=np.float32)
    pairs = C.c_uint64()
    r0, r1 = rows if rows else (0, H)
    l.evo_splat_photons(C.byref(fp), W, H, r0, r1, ptr(gbuf[0]), ptr(gbuf[1]), ptr(gbuf[2]), ptr(gbuf[3]), ptr(records), records.shape[0], ptr(out), C.byref(pairs))
    return out, pairs.value


def icosphere42():
    """the oracle's generated proxy mesh: (vertices float32 [42, 3], triangles int32 [80, 3])"""
    v = np.zeros((42, 3), dtype=np.float32); t = np.zeros((80, 3), dtype=np.int32)
    load().evo_icosphere42(ptr(v), ptr(t))
    return v, t


def splat_proxy(fp, cam, W, H, gbuf, records, rows=None, mesh=None, out=None):
    """Both footprints of the photon splat on the same pixels: (ideal sphere image, reference proxy-mesh image, stats) -- stats =
    pairs inside the radius, of those missed by the proxy, counted twice by it, proxy fragments (oracle/evplp_oracle.c).
    mesh = (vertices, triangles) of the proxy in units of the radius (default: the generated icosphere); out: accumulate the proxy
    image into this array."""
    l = load()
    ideal = np.zeros((H, W, 4), dtype=np.float32); proxy = np.zeros((H, W, 4), dtype=np.float32) if out is None else out
    st = np.zeros(4, dtype=np.uint64)
    r0, r1 = rows if rows else (0, H)
    v, t = mesh if mesh is not None else icosphere42()
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3); t = np.ascontiguousarray(t, dtype=np.int32).reshape(-1, 3)
    l.evo_splat_photons_proxy_mesh(C.byref(fp), C.byref(cam), W, H, r0, r1, ptr(gbuf[0]), ptr(gbuf[1]), ptr(gbuf[2]), ptr(gbuf[3]), ptr(records), records.shape[0],
                                   ptr(v), ptr(t), t.shape[0], ptr(ideal), ptr(proxy), ptr(st))
    return ideal, proxy, st


def frame_params(camera_pos, mis_mode=0, pdf_mc=0.0, clamping_value=0.0, photon_radius=0.0, vsl_radius=0.0,
                 vsl_inv_pi_radius2=0.0, num_light_paths=1, num_vpl_light_paths=1, photons_per_path=1,
                 do_accumulate=0, rng_seed=0, jitter=(0.0, 0.0)):
    fp = FrameParams()
    fp.camera_pos = (C.c_float * 3)(*[float(x) for x in camera_pos])
    fp.mis_mode = int(mis_mode); fp.pdf_mc = pdf_mc; fp.clamping_value = clamping_value; fp.photon_radius = photon_radius
    fp.vsl_radius = vsl_radius; fp.vsl_inv_pi_radius2 = vsl_inv_pi_radius2
    fp.num_light_paths = num_light_paths; fp.num_vpl_light_paths = num_vpl_light_paths
    fp.photons_per_path = photons_per_path; fp.do_accumulate = do_accumulate; fp.rng_seed = rng_seed
    fp.jitter = (C.c_float * 2)(float(jitter[0]), float(jitter[1]))
    return fp
