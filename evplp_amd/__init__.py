"""evplp_amd -- Python binding (ctypes) of libevplp_hip.so, the MI355X implementation of evplp's
radiance-accumulation hot path.

The product is the C-ABI library (include/evplp.h) plus the C++ host side above it
(evplp_amd/csrc/host); this module only exposes that ABI to tests, bench.py and the multi-GPU
strip renderer.  It never computes anything itself and has no CPU fallback: if the shared
library is missing the import fails loudly, and without a HIP device `Context()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVPLP_LIB: developer override (e.g. the -DEVPLP_TRAVERSAL_STATS=1 diagnostic build of tools/traversal_stats.py)
LIB_PATH = os.environ.get("EVPLP_LIB") or os.path.join(_HERE, "lib", "libevplp_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

ABI_VERSION = 4

# evplp_status
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_IO, ERR_PARSE, ERR_OOM = 0, -1, -2, -3, -4, -5, -6
# flags (rt/rtcomphoton/rtphotonrecord.h:9-15)
USABLE_VPL, USABLE_PHOTON, LAMBERT_ONLY, PHONG_ONLY = 1, 2, 4, 8
# EMis (rtcomphoton.h:64-72, string map :1199-1206)
MIS_MODES = {"one": 0, "balance": 1, "max": 2, "power2": 3, "geometryClamp": 4, "geometryBrdfClamp": 5}
BVH_LBVH, BVH_SAH, BVH_SBVH, BVH_LBVH_GPU = 0, 1, 2, 3
(BUF_RECORDS, BUF_GBUF_POSITION, BUF_GBUF_NORMAL, BUF_GBUF_DIFFUSE, BUF_GBUF_PHONG, BUF_LIGHT,
 BUF_VPL_ACCUM, BUF_PHOTON_ACCUM, BUF_COUNT) = range(9)
(PASS_PRIMARY, PASS_LIGHT_TRACE, PASS_GATHER_VPL, PASS_GATHER_VSL, PASS_SPLAT, PASS_RESOLVE, PASS_PATH_TRACE,
 PASS_GATHER_LVC, PASS_COUNT) = range(9)

RECORD_DTYPE = np.dtype([
    ("pos", np.float32, 3), ("flags", np.uint32),
    ("normal", np.float32, 3), ("p_select_lambert", np.float32),
    ("flux", np.float32, 3), ("pad1", np.float32),
    ("flux_dir", np.float32, 3), ("pad2", np.float32),
    ("rho_d", np.float32, 3), ("pad3", np.float32),
    ("rho_s", np.float32, 3), ("phong_exp", np.float32),
])
assert RECORD_DTYPE.itemsize == 96


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("res_x", C.c_int32), ("res_y", C.c_int32),
                ("strip_rank", C.c_int32), ("strip_count", C.c_int32), ("strip_rows", C.c_int32),
                ("num_light_paths", C.c_uint32), ("num_vpl_light_paths", C.c_uint32), ("photons_per_path", C.c_uint32),
                ("bvh_builder", C.c_int32), ("deterministic", C.c_int32), ("gather_splits_per_wave", C.c_int32),
                ("overlap_light_tracing", C.c_int32), ("cut_scratch_bytes", C.c_uint64), ("vsl_mask_bytes", C.c_uint64),
                ("band_first_row", C.c_int32), ("band_rows", C.c_int32), ("band_capacity_rows", C.c_int32), ("strip_capacity_rows", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("ns", C.c_float),
                ("tex_kd", C.c_int32), ("tex_ks", C.c_int32), ("tex_ns", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3),
                ("fovy", C.c_float), ("aspect", C.c_float)]


class FrameParams(C.Structure):
    _fields_ = [("camera_pos", C.c_float * 3), ("mis_mode", C.c_uint32), ("pdf_mc", C.c_float),
                ("clamping_value", C.c_float), ("photon_radius", C.c_float), ("vsl_radius", C.c_float),
                ("vsl_inv_pi_radius2", C.c_float), ("num_light_paths", C.c_uint32),
                ("num_vpl_light_paths", C.c_uint32), ("photons_per_path", C.c_uint32),
                ("do_accumulate", C.c_uint32), ("rng_seed", C.c_uint32), ("jitter", C.c_float * 2),
                ("splat_footprint", C.c_uint32), ("reserved", C.c_uint32)]


FOOTPRINT_IDEAL, FOOTPRINT_PROXY = 0, 1
FOOTPRINTS = {"ideal": FOOTPRINT_IDEAL, "proxy": FOOTPRINT_PROXY}


class GroupConfig(C.Structure):
    _fields_ = [("n_ranks", C.c_int32), ("devices", C.POINTER(C.c_int32)), ("strip_rows", C.c_int32), ("use_rccl", C.c_int32),
                ("partition", C.c_int32), ("strip_capacity_pct", C.c_int32), ("split_light_paths", C.c_int32), ("reserved", C.c_int32)]


PARTITION_STRIPS, PARTITION_BANDS = 0, 1


class PassStats(C.Structure):
    _fields_ = [("ms", C.c_float), ("pairs", C.c_uint64), ("rays", C.c_uint64), ("usable", C.c_uint64),
                ("dominant_kernel_ms", C.c_float), ("reserved", C.c_uint32 * 3), ("shaded", C.c_uint64), ("launches", C.c_uint32),
                ("pad", C.c_uint32)]


# every symbol include/evplp.h declares: (restype, argtypes)
_P = C.c_void_p
_SIGNATURES = {
    "evplp_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "evplp_destroy": (None, [_P]),
    "evplp_last_error": (C.c_char_p, [_P]),
    "evplp_abi_version": (C.c_int, []),
    "evplp_set_stream": (C.c_int, [_P, _P]),
    "evplp_synchronize": (C.c_int, [_P]),
    "evplp_add_texture": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "evplp_add_material": (C.c_int, [_P, C.POINTER(Material)]),
    "evplp_add_mesh": (C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int32, C.c_int32]),
    "evplp_set_arealight": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float * 4)]),
    "evplp_set_camera": (C.c_int, [_P, C.POINTER(Camera)]),
    "evplp_load_scene_json": (C.c_int, [_P, C.c_char_p]),
    "evplp_get_camera": (C.c_int, [_P, C.POINTER(Camera)]),
    "evplp_build_accel": (C.c_int, [_P]),
    "evplp_scene_metrics": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "evplp_primary": (C.c_int, [_P, C.POINTER(C.c_float * 2), C.c_int32]),
    "evplp_trace_light_paths": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "evplp_gather_vpl": (C.c_int, [_P, C.POINTER(FrameParams)]),
    "evplp_gather_vsl": (C.c_int, [_P, C.POINTER(FrameParams)]),
    "evplp_gather_lvc": (C.c_int, [_P, C.POINTER(FrameParams)]),
    "evplp_path_trace": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_int32]),
    "evplp_splat_photons": (C.c_int, [_P, C.POINTER(FrameParams), C.c_int32]),
    "evplp_set_splat_proxy": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32]),
    "evplp_group_set_splat_proxy": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32]),
    "evplp_default_splat_proxy": (C.c_int, [_P, _P]),
    "evplp_resolve": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _P]),
    "evplp_present": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32]),
    "evplp_clear_accumulators": (C.c_int, [_P]),
    "evplp_set_band": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "evplp_set_blocks": (C.c_int, [_P, _P, C.c_int32]),
    "evplp_get_blocks": (C.c_int, [_P, _P, C.c_int32]),
    "evplp_calibrate_blocks": (C.c_int, [_P, C.c_int32]),
    "evplp_block_costs": (C.c_int, [_P, _P, C.c_int32]),
    "evplp_deal_blocks": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "evplp_rank_blocks": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32]),
    "evplp_local_rows": (C.c_int, [_P]),
    "evplp_buffer_info": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "evplp_bind_buffer": (C.c_int, [_P, C.c_int32, _P, C.c_size_t]),
    "evplp_download": (C.c_int, [_P, C.c_int32, _P, C.c_size_t]),
    "evplp_upload": (C.c_int, [_P, C.c_int32, _P, C.c_size_t]),
    "evplp_pass_stats_get": (C.c_int, [_P, C.c_int32, C.POINTER(PassStats)]),
    "evplp_profile_kernels": (C.c_int, [_P, C.c_int32]),
    "evplp_profile_passes": (C.c_int, [_P, C.c_int32]),
    "evplp_group_profile_passes": (C.c_int, [_P, C.c_int32]),
    "evplp_debug_counters": (C.c_int, [_P, C.c_int32, _P, C.c_int32]),
    "evplp_accel_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "evplp_accel_builder": (C.c_int, [_P]),
    "evplp_accel_stack_entries": (C.c_int, [_P]),
    "evplp_selftest": (C.c_int, [_P, C.c_int32, _P, C.c_int32]),
    "evplp_debug_ev_math": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P]),
    "evplp_group_create": (C.c_int, [C.POINTER(Config), _P, C.POINTER(_P)]),
    "evplp_group_destroy": (None, [_P]),
    "evplp_group_last_error": (C.c_char_p, [_P]),
    "evplp_group_size": (C.c_int, [_P]),
    "evplp_group_context": (_P, [_P, C.c_int32]),
    "evplp_group_load_scene_json": (C.c_int, [_P, C.c_char_p]),
    "evplp_group_clear_accumulators": (C.c_int, [_P]),
    "evplp_group_primary": (C.c_int, [_P, C.POINTER(C.c_float * 2), C.c_int32]),
    "evplp_group_trace_light_paths": (C.c_int, [_P, C.c_uint32]),
    "evplp_group_gather": (C.c_int, [_P, C.POINTER(FrameParams), C.c_int32]),
    "evplp_group_splat_photons": (C.c_int, [_P, C.POINTER(FrameParams), C.c_int32]),
    "evplp_group_path_trace": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_int32]),
    "evplp_group_synchronize": (C.c_int, [_P]),
    "evplp_group_host_stats": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double * 3)]),
    "evplp_group_rebalance": (C.c_int, [_P, _P]),
    "evplp_group_calibrate": (C.c_int, [_P, C.c_int32]),
    "evplp_group_split_model": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_double * 2)]),
    "evplp_group_block_owners": (C.c_int, [_P, _P, C.c_int32]),
    "evplp_group_resolve": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _P]),
    "evplp_group_present": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32]),
    "evplp_group_present_ex": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32]),
    "evplp_jitter_sequence": (C.c_int, [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "evplp_json_query": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.c_int32]),
    "evplp_progressive_step": (None, [C.c_int32, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "evplp_save_image": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _P]),
    "evplp_load_pfm": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, C.c_size_t]),
    "evplp_load_image": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, C.c_size_t]),
    "evplp_image_error_heat": (C.c_int, [C.c_int32, _P, _P, C.c_float, C.c_int32, _P]),
    "evplp_decode_image": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, C.c_size_t]),
    "evplp_image_mse": (C.c_double, [C.c_int32, _P, _P]),
    "evplp_image_rel_mse": (C.c_double, [C.c_int32, _P, _P]),
    "evplp_image_rel_mse_masked": (C.c_double, [C.c_int32, _P, _P, _P]),
    "evplp_synth_scene": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_uint32, C.c_int32, C.c_int32]),
    "evplp_synth_scene_ex": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32]),
    "evplp_render_json": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_size_t]),
}

_lib = None


class EvplpError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"evplp status {status}: {message}")
        self.status = status


def lib() -> C.CDLL:
    """Load libevplp_hip.so (built in-tree by `make` / __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()) first; "
                              "evplp_amd has no fallback implementation")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if l.evplp_abi_version() != ABI_VERSION:
            raise ImportError("libevplp_hip.so ABI version mismatch")
        _lib = l
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def frame_params(camera_pos, mis_mode=0, pdf_mc=0.0, clamping_value=0.0, photon_radius=0.0, vsl_radius=0.0,
                 vsl_inv_pi_radius2=0.0, num_light_paths=1, num_vpl_light_paths=1, photons_per_path=1,
                 do_accumulate=0, rng_seed=0, jitter=(0.0, 0.0), splat_footprint=0) -> FrameParams:
    fp = FrameParams()
    fp.camera_pos = (C.c_float * 3)(*[float(x) for x in camera_pos])
    fp.mis_mode = MIS_MODES[mis_mode] if isinstance(mis_mode, str) else int(mis_mode)
    fp.pdf_mc = pdf_mc; fp.clamping_value = clamping_value; fp.photon_radius = photon_radius
    fp.vsl_radius = vsl_radius; fp.vsl_inv_pi_radius2 = vsl_inv_pi_radius2
    fp.num_light_paths = num_light_paths; fp.num_vpl_light_paths = num_vpl_light_paths
    fp.photons_per_path = photons_per_path; fp.do_accumulate = do_accumulate; fp.rng_seed = rng_seed
    fp.jitter = (C.c_float * 2)(float(jitter[0]), float(jitter[1]))
    fp.splat_footprint = FOOTPRINTS[splat_footprint] if isinstance(splat_footprint, str) else int(splat_footprint)
    return fp


def default_splat_proxy():
    """(vertices float32 [42, 3], triangles int32 [80, 3]) of the proxy EVPLP_FOOTPRINT_PROXY uses when no mesh was given."""
    v = np.zeros((42, 3), dtype=np.float32); t = np.zeros((80, 3), dtype=np.int32)
    n = lib().evplp_default_splat_proxy(_ptr(v), _ptr(t))
    assert n == 80
    return v, t


class Context:
    """One GPU's share of a frame (thin RAII wrapper over evplp_context)."""

    def __init__(self, res_x: int, res_y: int, num_light_paths: int, num_vpl_light_paths: int, photons_per_path: int,
                 device: int = 0, strip_rank: int = 0, strip_count: int = 1, strip_rows: int = 16,
                 bvh_builder: int = BVH_SAH, deterministic: bool = False, gather_splits_per_wave: int = 0, overlap_light_tracing: bool = False,
                 band=None, band_capacity_rows: int = 0, strip_capacity_rows: int = 0, cut_scratch_bytes: int = 0, vsl_mask_bytes: int = 0):
        """band = (first_row, rows): the context owns those contiguous image rows instead of interleaved strips."""
        self._lib = lib()
        cfg = Config()
        if band is not None:
            cfg.band_first_row, cfg.band_rows = int(band[0]), int(band[1]); cfg.band_capacity_rows = int(band_capacity_rows)
            strip_rank, strip_count = 0, 1
        cfg.abi_version = ABI_VERSION; cfg.device = device; cfg.res_x = res_x; cfg.res_y = res_y
        cfg.strip_rank = strip_rank; cfg.strip_count = strip_count; cfg.strip_rows = strip_rows
        cfg.num_light_paths = num_light_paths; cfg.num_vpl_light_paths = num_vpl_light_paths
        cfg.photons_per_path = photons_per_path; cfg.bvh_builder = bvh_builder; cfg.deterministic = int(deterministic)
        cfg.gather_splits_per_wave = gather_splits_per_wave
        cfg.overlap_light_tracing = int(overlap_light_tracing); cfg.strip_capacity_rows = int(strip_capacity_rows)
        cfg.cut_scratch_bytes = int(cut_scratch_bytes); cfg.vsl_mask_bytes = int(vsl_mask_bytes)
        self.cfg = cfg
        h = C.c_void_p()
        rc = self._lib.evplp_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise EvplpError(rc, self._lib.evplp_last_error(None).decode())
        self._h = h
        self.W, self.H = res_x, res_y
        self.local_rows = self._lib.evplp_local_rows(h)
        self.num_records = num_light_paths * photons_per_path

    @classmethod
    def borrowed(cls, handle, res_x: int, res_y: int, strip_rank: int = 0, strip_count: int = 1, strip_rows: int = 8,
                 num_light_paths: int = 0, num_vpl_light_paths: int = 0, photons_per_path: int = 0):
        """A view of a context somebody else owns (a rank of an evplp_group): statistics and buffers; close() does not destroy it."""
        self = cls.__new__(cls)
        self._lib = lib(); self._h = C.c_void_p(handle); self._borrowed = True
        self.cfg = Config(); self.cfg.strip_rank = strip_rank; self.cfg.strip_count = strip_count; self.cfg.strip_rows = strip_rows
        self.cfg.abi_version = ABI_VERSION; self.cfg.res_x = res_x; self.cfg.res_y = res_y
        self.cfg.num_light_paths = num_light_paths; self.cfg.num_vpl_light_paths = num_vpl_light_paths; self.cfg.photons_per_path = photons_per_path
        self.num_records = num_light_paths * photons_per_path
        self.W, self.H = res_x, res_y
        self.local_rows = self._lib.evplp_local_rows(self._h)
        return self

    # -- plumbing
    def _check(self, rc: int) -> int:
        if rc < 0:
            raise EvplpError(rc, self._lib.evplp_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.evplp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene
    def add_texture(self, rgba: np.ndarray) -> int:
        rgba = _f32(rgba)
        h, w = rgba.shape[:2]
        return self._check(self._lib.evplp_add_texture(self._h, w, h, _ptr(rgba)))

    def add_material(self, kd, ks, ns, tex_kd=-1, tex_ks=-1, tex_ns=-1) -> int:
        m = Material()
        m.kd = (C.c_float * 3)(*[float(x) for x in kd]); m.ks = (C.c_float * 3)(*[float(x) for x in ks]); m.ns = float(ns)
        m.tex_kd, m.tex_ks, m.tex_ns = tex_kd, tex_ks, tex_ns
        return self._check(self._lib.evplp_add_material(self._h, C.byref(m)))

    def add_mesh(self, vertices, indices, material: int, texcoords=None) -> int:
        v = _f32(vertices).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        t = None if texcoords is None else _f32(texcoords).reshape(-1, 2)
        return self._check(self._lib.evplp_add_mesh(self._h, _ptr(v), _ptr(t), v.shape[0], _ptr(i), i.shape[0], material))

    def set_arealight(self, mesh: int, intensity: Sequence[float]):
        arr = (C.c_float * 4)(*[float(x) for x in intensity])
        self._check(self._lib.evplp_set_arealight(self._h, mesh, C.byref(arr)))

    def set_camera(self, origin, lookat, up, fovy: float, aspect: float):
        cam = Camera()
        cam.origin = (C.c_float * 3)(*[float(x) for x in origin]); cam.lookat = (C.c_float * 3)(*[float(x) for x in lookat])
        cam.up = (C.c_float * 3)(*[float(x) for x in up]); cam.fovy = float(fovy); cam.aspect = float(aspect)
        self._check(self._lib.evplp_set_camera(self._h, C.byref(cam)))

    def build_accel(self):
        self._check(self._lib.evplp_build_accel(self._h))

    def load_scene_json(self, json_path: str):
        rc = self._lib.evplp_load_scene_json(self._h, json_path.encode())
        if rc < 0:
            raise EvplpError(rc, f"evplp_load_scene_json({json_path}): " + self._lib.evplp_last_error(self._h).decode())

    def camera(self) -> Camera:
        cam = Camera()
        self._check(self._lib.evplp_get_camera(self._h, C.byref(cam)))
        return cam

    def scene_metrics(self):
        r, t, l = C.c_float(), C.c_float(), C.c_float()
        self._check(self._lib.evplp_scene_metrics(self._h, C.byref(r), C.byref(t), C.byref(l)))
        return r.value, t.value, l.value

    def accel_info(self):
        n, l, d, ms = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
        self._check(self._lib.evplp_accel_info(self._h, C.byref(n), C.byref(l), C.byref(d), C.byref(ms)))
        b = self._lib.evplp_accel_builder(self._h)
        return {"nodes": n.value, "leaves": l.value, "depth": d.value, "build_ms": ms.value,
                "builder": {0: "lbvh", 1: "sah", 2: "sbvh", 3: "gpu"}.get(b, "none"), "stack4_entries": self._lib.evplp_accel_stack_entries(self._h)}

    def selftest(self, which: int = 0) -> np.ndarray:
        out = np.zeros(8, dtype=np.uint64)
        n = self._check(self._lib.evplp_selftest(self._h, which, _ptr(out), 8))
        return out[:n]

    # -- passes
    def ev_math(self, which: int, x, y=None):
        """ev_math.h on the device: which 0 -> (sin x, cos x), 1 -> x ** y (evplp_debug_ev_math)"""
        x = _f32(x).reshape(-1); y = None if y is None else _f32(y).reshape(-1)
        o0 = np.empty_like(x); o1 = np.empty_like(x)
        self._check(self._lib.evplp_debug_ev_math(self._h, which, _ptr(x), _ptr(y), x.size, _ptr(o0), _ptr(o1)))
        return (o0, o1) if which == 0 else o0

    def set_stream(self, stream_ptr: int):
        self._check(self._lib.evplp_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._check(self._lib.evplp_synchronize(self._h))

    def primary(self, jitter=(0.0, 0.0), clear_light=False, light_unoccluded=False, light_skip=False):
        """clear_light / light_unoccluded / light_skip = EVPLP_LIGHT_CLEAR / _UNOCCLUDED / _SKIP (include/evplp.h)."""
        j = (C.c_float * 2)(float(jitter[0]), float(jitter[1]))
        flags = (1 if clear_light else 0) | (2 if light_unoccluded else 0) | (4 if light_skip else 0)
        self._check(self._lib.evplp_primary(self._h, C.byref(j), flags))

    def trace_light_paths(self, rng_seed: int, path_begin: int = 0, path_count: Optional[int] = None):
        if path_count is None:
            path_count = self.cfg.num_light_paths - path_begin
        self._check(self._lib.evplp_trace_light_paths(self._h, rng_seed, path_begin, path_count))

    def gather_vpl(self, fp: FrameParams):
        self._check(self._lib.evplp_gather_vpl(self._h, C.byref(fp)))

    def gather_vsl(self, fp: FrameParams):
        self._check(self._lib.evplp_gather_vsl(self._h, C.byref(fp)))

    def gather_lvc(self, fp: FrameParams):
        self._check(self._lib.evplp_gather_lvc(self._h, C.byref(fp)))

    def path_trace(self, camera_pos, rng_seed: int, max_bounces: int, accumulate=True):
        cp = (C.c_float * 3)(*[float(v) for v in camera_pos])
        self._check(self._lib.evplp_path_trace(self._h, C.byref(cp), rng_seed, max_bounces, int(accumulate)))

    def splat_photons(self, fp: FrameParams, clear=False):
        self._check(self._lib.evplp_splat_photons(self._h, C.byref(fp), int(clear)))

    def set_splat_proxy(self, vertices=None, triangles=None):
        """The proxy mesh of FOOTPRINT_PROXY (None: the generated icosphere); refused unless closed and convex around the origin."""
        if vertices is None:
            self._check(self._lib.evplp_set_splat_proxy(self._h, None, 0, None, 0)); return
        v = _f32(vertices).reshape(-1, 3); t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        self._check(self._lib.evplp_set_splat_proxy(self._h, _ptr(v), v.shape[0], _ptr(t), t.shape[0]))

    def resolve(self, vpl_scale=1.0, photon_scale=1.0, light_scale=1.0, mask_emitter=False, gamma=False) -> np.ndarray:
        out = np.empty((self.local_rows, self.W, 3), dtype=np.float32)
        self._check(self._lib.evplp_resolve(self._h, vpl_scale, photon_scale, light_scale, int(mask_emitter), int(gamma), _ptr(out)))
        return out

    def present(self, vpl_scale=1.0, photon_scale=1.0, light_scale=1.0, mask_emitter=True, gamma=False):
        """The per-iteration composite (rtcomphoton.h:997-1004): the strip's RGB stays on the device."""
        self._check(self._lib.evplp_present(self._h, vpl_scale, photon_scale, light_scale, int(mask_emitter), int(gamma)))

    def clear_accumulators(self):
        self._check(self._lib.evplp_clear_accumulators(self._h))

    def set_blocks(self, image_blocks=None):
        """row-strip context: own these image blocks (in this local order) instead of the blocks b % strip_count == strip_rank; None = back to that"""
        if image_blocks is None:
            self._check(self._lib.evplp_set_blocks(self._h, None, 0)); return
        b = np.ascontiguousarray(image_blocks, dtype=np.int32)
        self._check(self._lib.evplp_set_blocks(self._h, _ptr(b), b.size))

    def blocks(self) -> np.ndarray:
        """the image blocks this context owns, in local order"""
        n = self._check(self._lib.evplp_get_blocks(self._h, None, 0))
        out = np.zeros(n, dtype=np.int32)
        self._check(self._lib.evplp_get_blocks(self._h, _ptr(out), n))
        return out

    def calibrate_blocks(self, on: bool = True):
        self._check(self._lib.evplp_calibrate_blocks(self._h, int(on)))

    def block_costs(self) -> np.ndarray:
        """clock ticks the gathers' wavefronts spent in every IMAGE block since calibrate_blocks(True) (0 for other ranks' blocks)"""
        sr = self.cfg.strip_rows if self.cfg.strip_count > 1 else (self.H + 7) // 8 * 8
        out = np.zeros((self.H + sr - 1) // sr, dtype=np.uint64)
        self._check(self._lib.evplp_block_costs(self._h, _ptr(out), out.size))
        return out

    def set_band(self, first_row: int, rows: int):
        self._check(self._lib.evplp_set_band(self._h, first_row, rows))
        self.cfg.band_first_row, self.cfg.band_rows = first_row, rows

    # -- buffers
    def buffer_info(self, which: int):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.evplp_buffer_info(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_buffer(self, which: int, device_ptr: int, nbytes: int):
        self._check(self._lib.evplp_bind_buffer(self._h, which, C.c_void_p(device_ptr), nbytes))

    def buffer_bytes(self, which: int) -> int:
        n = C.c_size_t()
        self._check(self._lib.evplp_buffer_info(self._h, which, None, C.byref(n)))   # (no pointer taken: see evplp_buffer_info)
        return n.value

    def download(self, which: int) -> np.ndarray:
        n = self.buffer_bytes(which)
        if which == BUF_RECORDS:
            out = np.empty(n // 96, dtype=RECORD_DTYPE)
        else:
            out = np.empty((self.local_rows, self.W, 4), dtype=np.float32)
        self._check(self._lib.evplp_download(self._h, which, _ptr(out), n))
        return out

    def upload(self, which: int, data: np.ndarray):
        data = np.ascontiguousarray(data)
        self._check(self._lib.evplp_upload(self._h, which, _ptr(data), data.nbytes))

    def pass_stats(self, which: int) -> dict:
        s = PassStats()
        self._check(self._lib.evplp_pass_stats_get(self._h, which, C.byref(s)))
        return {"ms": s.ms, "pairs": s.pairs, "rays": s.rays, "usable": s.usable, "dominant_kernel_ms": s.dominant_kernel_ms,
                "nodes": s.reserved[0] | (s.reserved[1] << 32), "shaded": s.shaded, "launches": s.launches,
                # VSL gather: the same two words carry the sample-iterations of the estimators (lighttracing.cu:632-640)
                "samples": (s.reserved[0] | (s.reserved[1] << 32)) if which == PASS_GATHER_VSL else 0}

    def profile_kernels(self, on: bool = True):
        """Record the events around the photon splat's dominant kernel (pass_stats()["dominant_kernel_ms"]); they sit between its launches."""
        self._check(self._lib.evplp_profile_kernels(self._h, int(on)))

    def profile_passes(self, on: bool = True):
        """Record the two events per pass that pass_stats()["ms"] needs (default on); off, a loop of sub-millisecond iterations runs
        without them and pass_stats reports ms = 0 for passes run meanwhile."""
        self._check(self._lib.evplp_profile_passes(self._h, int(on)))

    def debug_counters(self, which: int) -> np.ndarray:
        out = np.zeros(256, dtype=np.uint64)
        n = self._check(self._lib.evplp_debug_counters(self._h, which, _ptr(out), out.size))
        return out[:n]

    # -- strips
    def global_rows(self) -> np.ndarray:
        """Global image row of every local row (>= H for padding rows)."""
        from . import strips
        if self.cfg.band_rows > 0:
            l = np.arange(self.local_rows)
            rows = min(self.cfg.band_rows, self.H - self.cfg.band_first_row)
            return np.where(l < rows, self.cfg.band_first_row + l, self.H + l)
        if self.cfg.strip_count <= 1:
            return strips.global_rows(self.H, self.cfg.strip_rank, self.cfg.strip_count, self.cfg.strip_rows)
        return strips.rows_of_blocks(self.H, self.blocks(), self.cfg.strip_rows, self.local_rows)      # (the library's table: dealt or round-robin)


def deal_blocks(costs, n_ranks: int, capacity_blocks: int) -> np.ndarray:
    """evplp_deal_blocks: owner rank of every image block for these per-block costs (host only, deterministic)"""
    c = np.ascontiguousarray(costs, dtype=np.uint64)
    owner = np.zeros(c.size, dtype=np.int32)
    rc = lib().evplp_deal_blocks(_ptr(c), c.size, n_ranks, capacity_blocks, _ptr(owner))
    if rc < 0:
        raise EvplpError(rc, "evplp_deal_blocks: the blocks do not fit the ranks' capacity")
    return owner


def split_model(num_light_paths: int, photons_per_path: int, n_ranks: int):
    """evplp_group_split_model: (split expected to be faster?, ms with every rank tracing all paths, ms with a share + the record exchange)"""
    out = (C.c_double * 2)()
    rc = lib().evplp_group_split_model(num_light_paths, photons_per_path, n_ranks, C.byref(out))
    return bool(rc == 1), out[0], out[1]


def rank_blocks(costs, owner, rank: int) -> np.ndarray:
    """evplp_rank_blocks: the blocks a deal gives `rank`, in the order it stores and launches them (most expensive first; costs=None: image order)"""
    o = np.ascontiguousarray(owner, dtype=np.int32)
    c = None if costs is None else np.ascontiguousarray(costs, dtype=np.uint64)
    out = np.zeros(o.size, dtype=np.int32)
    n = lib().evplp_rank_blocks(_ptr(c), _ptr(o), o.size, rank, _ptr(out), out.size)
    if n < 0:
        raise EvplpError(n, "evplp_rank_blocks")
    return out[:n]


class Group:
    """evplp_group: n row-strip ranks driven by one thread (RCCL across distinct GPUs, device copies for virtual ranks)."""

    def __init__(self, res_x, res_y, num_light_paths, num_vpl_light_paths, photons_per_path, n_ranks, devices=None, strip_rows=0,
                 use_rccl=False, deterministic=False, bvh_builder=BVH_SAH, overlap_light_tracing=False, partition="strips", strip_capacity_pct=0, split_light_paths=0, cut_scratch_bytes=0, vsl_mask_bytes=0):
        """strip_rows = 0: the library's choice (16 rows)"""
        self._lib = lib()
        cfg = Config()
        cfg.abi_version = ABI_VERSION; cfg.res_x = res_x; cfg.res_y = res_y
        cfg.num_light_paths = num_light_paths; cfg.num_vpl_light_paths = num_vpl_light_paths; cfg.photons_per_path = photons_per_path
        cfg.bvh_builder = bvh_builder; cfg.deterministic = int(deterministic); cfg.overlap_light_tracing = int(overlap_light_tracing)
        cfg.cut_scratch_bytes = int(cut_scratch_bytes); cfg.vsl_mask_bytes = int(vsl_mask_bytes)
        gc = GroupConfig(); gc.n_ranks = n_ranks; gc.strip_rows = strip_rows; gc.use_rccl = int(use_rccl)
        gc.partition = PARTITION_BANDS if partition == "bands" else PARTITION_STRIPS; gc.strip_capacity_pct = int(strip_capacity_pct); gc.split_light_paths = int(split_light_paths)
        self.partition = partition if n_ranks > 1 else "strips"
        self._devs = (C.c_int32 * n_ranks)(*devices) if devices is not None else None
        gc.devices = C.cast(self._devs, C.POINTER(C.c_int32)) if self._devs is not None else None
        h = C.c_void_p()
        rc = self._lib.evplp_group_create(C.byref(cfg), C.byref(gc), C.byref(h))
        if rc != OK:
            raise EvplpError(rc, self._lib.evplp_group_last_error(None).decode())
        self._h = h; self.W, self.H, self.n = res_x, res_y, n_ranks
        self.strip_rows = strip_rows if strip_rows > 0 else 16
        self._paths = (num_light_paths, num_vpl_light_paths, photons_per_path)

    def _check(self, rc):
        if rc < 0:
            raise EvplpError(rc, self._lib.evplp_group_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.evplp_group_destroy(self._h); self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def load_scene_json(self, path):
        self._check(self._lib.evplp_group_load_scene_json(self._h, path.encode()))

    def clear_accumulators(self):
        self._check(self._lib.evplp_group_clear_accumulators(self._h))

    def primary(self, jitter=(0.0, 0.0), light_flags=0):
        j = (C.c_float * 2)(float(jitter[0]), float(jitter[1]))
        self._check(self._lib.evplp_group_primary(self._h, C.byref(j), light_flags))

    def trace_light_paths(self, seed):
        self._check(self._lib.evplp_group_trace_light_paths(self._h, seed))

    def gather(self, fp, kind=0):
        self._check(self._lib.evplp_group_gather(self._h, C.byref(fp), kind))

    def splat_photons(self, fp, clear=False):
        self._check(self._lib.evplp_group_splat_photons(self._h, C.byref(fp), int(clear)))

    def set_splat_proxy(self, vertices=None, triangles=None):
        if vertices is None:
            self._check(self._lib.evplp_group_set_splat_proxy(self._h, None, 0, None, 0)); return
        v = _f32(vertices).reshape(-1, 3); t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        self._check(self._lib.evplp_group_set_splat_proxy(self._h, _ptr(v), v.shape[0], _ptr(t), t.shape[0]))

    def synchronize(self):
        self._check(self._lib.evplp_group_synchronize(self._h))

    def rebalance(self) -> np.ndarray:
        """bands partition: move the band boundaries to equal measured cost (clears the accumulators); returns the n + 1 boundaries.
        strips partition: deal the blocks by the cost clocked since calibrate() (block_owners() has the result)"""
        out = np.zeros(self.n + 1, dtype=np.int32)
        self._check(self._lib.evplp_group_rebalance(self._h, _ptr(out)))
        self._bands = out.copy()
        return out

    def calibrate(self, on: bool = True):
        """strips partition: the gathers clock their blocks until rebalance() deals them by cost"""
        self._check(self._lib.evplp_group_calibrate(self._h, int(on)))

    def block_owners(self) -> np.ndarray:
        n = self._check(self._lib.evplp_group_block_owners(self._h, None, 0))
        out = np.zeros(n, dtype=np.int32)
        self._check(self._lib.evplp_group_block_owners(self._h, _ptr(out), n))
        return out

    def profile_passes(self, on: bool = True):
        self._check(self._lib.evplp_group_profile_passes(self._h, int(on)))

    def host_stats(self, r: int) -> dict:
        """host time of rank r's worker thread (ms inside pass calls, ms inside exchanges, commands run)"""
        out = (C.c_double * 3)()
        self._check(self._lib.evplp_group_host_stats(self._h, r, C.byref(out)))
        return {"calls_ms": out[0], "exchange_ms": out[1], "commands": int(out[2])}

    def present(self, vpl_scale=1.0, photon_scale=1.0, light_scale=1.0, mask_emitter=False, gamma=False, exchange=True):
        """composite + all-gather of the strips on the devices (the per-frame exchange); nothing comes to the host.  exchange=False: the composite alone"""
        self._check(self._lib.evplp_group_present_ex(self._h, vpl_scale, photon_scale, light_scale, int(mask_emitter), int(gamma), int(exchange)))

    def rank(self, r: int) -> "Context":
        """rank r's context (borrowed): pass statistics, buffers"""
        self._lib.evplp_group_context.restype = C.c_void_p
        h = self._lib.evplp_group_context(self._h, r)
        if not h:
            raise EvplpError(ERR_INVALID, "evplp_group_context: bad rank")
        c = Context.borrowed(h, self.W, self.H, strip_rank=r, strip_count=self.n, strip_rows=self.strip_rows,
                             num_light_paths=self._paths[0], num_vpl_light_paths=self._paths[1], photons_per_path=self._paths[2])
        if self.partition == "bands":
            b = getattr(self, "_bands", None)
            if b is None:
                rows16 = (self.H + 15) // 16 * 16; share = max(16, (rows16 // self.n + 15) // 16 * 16)
                b = np.minimum(np.arange(self.n + 1) * share, self.H); b[self.n] = self.H
            c.cfg.strip_rank, c.cfg.strip_count = 0, 1
            c.cfg.band_first_row = int(b[r]); c.cfg.band_rows = int((b[r + 1] if r + 1 < self.n else (self.H + 15) // 16 * 16) - b[r])
        return c

    def resolve(self, vpl_scale=1.0, photon_scale=1.0, light_scale=1.0, mask_emitter=False, gamma=False):
        out = np.empty((self.H, self.W, 3), dtype=np.float32)
        self._check(self._lib.evplp_group_resolve(self._h, vpl_scale, photon_scale, light_scale, int(mask_emitter), int(gamma), _ptr(out)))
        return out


def jitter_sequence(rng_offset: int, count: int, res_x: int, res_y: int) -> np.ndarray:
    out = np.zeros((count, 2), dtype=np.float32)
    rc = lib().evplp_jitter_sequence(rng_offset, count, res_x, res_y, _ptr(out))
    if rc != OK:
        raise EvplpError(rc, "evplp_jitter_sequence")
    return out


def progressive_step(n: int, alpha: float, clamp_start: float, n_vpl: int, n_light: int, radius: float, clamp: float,
                     pdf_mc: float, force_vsl=False, vsl_radius=0.0, vsl_inv_pi_r2=0.0):
    r, c, p, vr, vi = C.c_float(radius), C.c_float(clamp), C.c_float(pdf_mc), C.c_float(vsl_radius), C.c_float(vsl_inv_pi_r2)
    lib().evplp_progressive_step(n, alpha, clamp_start, n_vpl, n_light, C.byref(r), C.byref(c), C.byref(p), int(force_vsl), C.byref(vr), C.byref(vi))
    return r.value, c.value, p.value, vr.value, vi.value


def save_image(path: str, rgb_top_down: np.ndarray):
    a = _f32(rgb_top_down)
    h, w = a.shape[:2]
    rc = lib().evplp_save_image(path.encode(), w, h, _ptr(a))
    if rc != OK:
        raise EvplpError(rc, f"evplp_save_image({path})")


def load_pfm(path: str) -> np.ndarray:
    w, h = C.c_int32(), C.c_int32()
    rc = lib().evplp_load_pfm(path.encode(), C.byref(w), C.byref(h), None, 0)
    if rc != OK:
        raise EvplpError(rc, f"evplp_load_pfm({path})")
    out = np.empty((h.value, w.value, 3), dtype=np.float32)
    rc = lib().evplp_load_pfm(path.encode(), C.byref(w), C.byref(h), _ptr(out), out.size)
    if rc != OK:
        raise EvplpError(rc, f"evplp_load_pfm({path})")
    return out


def load_image(path: str) -> np.ndarray:
    """FloatImage::LoadPFM / LoadHDR by extension: float32 [h, w, 3], rows top to bottom."""
    w, h = C.c_int32(), C.c_int32()
    rc = lib().evplp_load_image(path.encode(), C.byref(w), C.byref(h), None, 0)
    if rc != OK:
        raise EvplpError(rc, f"evplp_load_image({path})")
    out = np.empty((h.value, w.value, 3), dtype=np.float32)
    rc = lib().evplp_load_image(path.encode(), C.byref(w), C.byref(h), _ptr(out), out.size)
    if rc != OK:
        raise EvplpError(rc, f"evplp_load_image({path})")
    return out


def error_heat(img: np.ndarray, ref: np.ndarray, max_error: float, relative: bool = False) -> np.ndarray:
    a, b = _f32(img), _f32(ref)
    out = np.empty_like(a)
    rc = lib().evplp_image_error_heat(a.shape[0] * a.shape[1], _ptr(a), _ptr(b), max_error, int(relative), _ptr(out))
    if rc != OK:
        raise EvplpError(rc, "evplp_image_error_heat")
    return out


def decode_image(path: str):
    """stbi_load(path, .., 3) of the reference (rt/rtcommon.h:144): (pixels uint8 [h, w, 3] top-down, channels in file)."""
    w, h, ch = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib().evplp_decode_image(path.encode(), C.byref(w), C.byref(h), C.byref(ch), None, C.c_size_t(0))
    if rc != OK:
        raise EvplpError(rc, f"evplp_decode_image({path})")
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    rc = lib().evplp_decode_image(path.encode(), C.byref(w), C.byref(h), C.byref(ch), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes))
    if rc != OK:
        raise EvplpError(rc, f"evplp_decode_image({path})")
    return out, ch.value


SCENE_STYLES = {"easy": 0, "hard": 1, "textured": 2}


def synth_scene(out_dir: str, name: str = "conference_synth", target_triangles: int = 331000, seed: int = 1234,
                res_x: int = 1024, res_y: int = 1024, style=0) -> str:
    """style: 0 / "easy" = tessellated boxes, 1 / "hard" = furnished with curved and thin parts."""
    style = SCENE_STYLES[style] if isinstance(style, str) else int(style)
    rc = lib().evplp_synth_scene_ex(out_dir.encode(), name.encode(), target_triangles, seed, res_x, res_y, style)
    if rc < 0:
        raise EvplpError(rc, "evplp_synth_scene failed")
    return os.path.join(out_dir, name + ".json")


def render_json(json_path: str, overrides: Optional[str] = None, device: int = 0):
    err = C.create_string_buffer(1024)
    rc = lib().evplp_render_json(json_path.encode(), overrides.encode() if overrides else None, device, err, 1024)
    if rc != OK:
        raise EvplpError(rc, err.value.decode())
