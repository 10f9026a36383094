"""Pins the oracle (and the product's host-side output surface) against golden vectors produced by the
REFERENCE'S OWN code (tests/golden/make_golden.py: floatimage.cpp + vendored GLM through oracle/_ref).
No GPU needed.  What is pinned: PFM bytes, PNG pixels, FlipY, MSE/relMSE, the camera model (lookAt /
perspective / jitter translation / fovx->fovy), the bounding-sphere radius, and the texture decoders (JPEG /
PNG pixels as the reference's vendored stb_image returns them).  The device arithmetic
(BRDFs, gather, splat) has no reference-run vectors: "parity unpinned" there."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_api as oa
import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import decode_png_rgb8  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "output_surface.npz"))
TEX = np.load(os.path.join(HERE, "golden", "textures.npz"))
TEX_NAMES = sorted(k[: -len("__file")] for k in TEX.files if k.endswith("__file"))
CAM = np.load(os.path.join(HERE, "golden", "camera.npz"))


@pytest.mark.parametrize("name", ["a", "b"])
def test_pfm_bytes_oracle_and_product(name, oracle, tmp_path, evplp):
    img = G[f"{name}_img"]; h, w = img.shape[:2]
    p1 = str(tmp_path / "o.pfm"); p2 = str(tmp_path / "p.pfm")
    assert oracle.evo_write_pfm(p1.encode(), w, h, oa.ptr(img)) == 0
    evplp.save_image(p2, img)
    want = G[f"{name}_pfm_bytes"].tobytes()
    assert open(p1, "rb").read() == want, "oracle PFM differs from FloatImage::SavePFM"
    assert open(p2, "rb").read() == want, "libevplp_hip PFM differs from FloatImage::SavePFM"
    back = evplp.load_pfm(p2)
    assert np.array_equal(back, img)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_hdr_bytes_product(name, tmp_path, evplp):
    """Radiance RGBE: flat for widths < 8 (image a), run-length encoded otherwise -- byte-identical to
    FloatImage::SaveHDR / rgbe.cpp."""
    img = G[f"{name}_hdr_img"]
    p = str(tmp_path / "x.hdr")
    evplp.save_image(p, img)
    assert open(p, "rb").read() == G[f"{name}_hdr_bytes"].tobytes()


@pytest.mark.parametrize("name", ["a", "b"])
def test_png_pixels_oracle_and_product(name, oracle, tmp_path, evplp):
    img = G[f"{name}_img"]; h, w = img.shape[:2]
    want = G[f"{name}_png_pixels"]
    got = np.zeros(img.size, np.uint8)
    oracle.evo_png_bytes(img.size, oa.ptr(img), oa.ptr(got))
    assert np.array_equal(got.reshape(h, w, 3), want), "oracle PNG quantisation differs from FloatImage::SavePNG"
    p = str(tmp_path / "p.png")
    evplp.save_image(p, img)
    assert np.array_equal(decode_png_rgb8(open(p, "rb").read()), want), "libevplp_hip PNG pixels differ from the reference's"


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_error_heat_images_and_hdr_readback(name, tmp_path, evplp):
    """FloatImage::Compute[Rel]SquareErrorHeatImage through Color::Heat (floatimage.cpp:21-62, math/color.h) and
    FloatImage::LoadHDR (rgbe.cpp RLE reader) -- the product's functions against the reference's own outputs."""
    img, other = G[f"{name}_img"], G[f"{name}_other"]
    for rel in (0, 1):
        got = evplp.error_heat(other, img, 0.004, relative=bool(rel))
        assert np.allclose(got, G[f"{name}_heat{rel}"], rtol=0, atol=2e-6), (rel, float(np.abs(got - G[f"{name}_heat{rel}"]).max()))
    p = str(tmp_path / "x.hdr")
    open(p, "wb").write(G[f"{name}_hdr_bytes"].tobytes())
    back = evplp.load_image(p)
    assert np.array_equal(back, G[f"{name}_hdr_decoded"])
    # PFM through the same entry point
    q = str(tmp_path / "x.pfm")
    open(q, "wb").write(G[f"{name}_pfm_bytes"].tobytes()) if f"{name}_pfm_bytes" in G.files else evplp.save_image(q, img)
    assert np.array_equal(evplp.load_image(q), img)


def test_masked_rel_mse(evplp):
    img, other = G["b_img"], G["b_other"]
    n = img.shape[0] * img.shape[1]
    l = evplp.lib()
    full = l.evplp_image_rel_mse_masked(n, oa.ptr(other), oa.ptr(img), None)
    assert full == l.evplp_image_rel_mse(n, oa.ptr(other), oa.ptr(img)) == float(G["b_relmse"])
    mask = np.full(img.shape, 255, np.uint8); mask[2:5, 3:9] = 0
    keep = mask.any(-1)
    r, d = img[keep].astype(np.float32), (other[keep] - img[keep]).astype(np.float32)
    want = float(np.float32(((d * d).sum(-1) / ((r * r).sum(-1) + np.float32(0.001))).astype(np.float32).sum(dtype=np.float32)) / np.float32(keep.sum()))
    got = l.evplp_image_rel_mse_masked(n, oa.ptr(other), oa.ptr(img), mask.ctypes.data_as(C.c_void_p))
    assert abs(got - want) <= 2e-6 * want and got != full


@pytest.mark.parametrize("name", ["a", "b"])
def test_error_metrics(name, oracle, evplp):
    img, other = G[f"{name}_img"], G[f"{name}_other"]
    n = img.shape[0] * img.shape[1]
    for fn_o, fn_p, key in ((oracle.evo_mse, evplp.lib().evplp_image_mse, "mse"), (oracle.evo_rel_mse, evplp.lib().evplp_image_rel_mse, "relmse")):
        want = float(G[f"{name}_{key}"])
        assert fn_o(n, oa.ptr(other), oa.ptr(img)) == want
        assert fn_p(n, oa.ptr(other), oa.ptr(img)) == want


def test_flip_y_convention():
    for name in ("a", "b"):
        assert np.array_equal(G[f"{name}_img"][::-1], G[f"{name}_flipy"])   # FloatImage::FlipY == reversing rows


@pytest.mark.parametrize("i", range(int(CAM["n"])))
def test_camera_model_against_glm(i, oracle):
    """Oracle primary rays, pushed through the reference's projection*view (and its jitter translation),
    must land on their own pixel centres: pins lookAt / perspective / fovx->fovy / the jitter sign."""
    sd = scenes.box_room(seed=2, n_boxes=0, tess=1)
    # a big inward-facing room around the camera so that every pixel hits something
    o = CAM[f"c{i}_origin"]
    sd = scenes.SceneData()
    m = sd.add_material((0.5, 0.5, 0.5))
    lo, hi = o - 40.0, o + 40.0
    sd.add_box(lo, hi, m, outward=False, n=1)
    lm = sd.add_material((0, 0, 0))
    sd.light_mesh = sd.add_quad(o + [-1, -1, 39.5], [0, 2, 0], [2, 0, 0], lm)
    sd.cam_origin, sd.cam_lookat, sd.cam_up = o.tolist(), CAM[f"c{i}_lookat"].tolist(), CAM[f"c{i}_up"].tolist()
    sd.fovy, sd.aspect = float(CAM[f"c{i}_fovy"]), float(CAM[f"c{i}_aspect"])
    sd.triangle_soup()
    osc = oa.Scene(sd)
    W, H = 24, 16
    for jit, key in (((0.0, 0.0), "mvp"), (tuple(CAM[f"c{i}_jitter"].tolist()), "mvp_jittered")):
        pos = osc.primary(W, H, jit)[0]
        M = CAM[f"c{i}_{key}"].astype(np.float64).T        # column-major GLM -> row-major
        P = np.concatenate([pos[..., :3].astype(np.float64), np.ones((H, W, 1))], axis=-1) @ M.T
        ndc = P[..., :2] / P[..., 3:4]
        px = (ndc[..., 0] * 0.5 + 0.5) * W - 0.5; py = (ndc[..., 1] * 0.5 + 0.5) * H - 0.5
        xs, ys = np.meshgrid(np.arange(W), np.arange(H))
        assert np.abs(px - xs).max() < 2e-3 and np.abs(py - ys).max() < 2e-3, (i, key, np.abs(px - xs).max(), np.abs(py - ys).max())
        depth = P[..., 3]                                   # clip w = view depth: inside [near, far]
        assert depth.min() > 0.1 and depth.max() < 100.0


def test_fovx_to_fovy_and_bounding_sphere(oracle):
    for i in range(int(CAM["n"])):
        fovx, aspect = np.float32(CAM[f"c{i}_fovx"]), np.float32(CAM[f"c{i}_aspect"])
        deg = np.float32(0.01745329251994329576923690768489)
        fovy = np.float32(2.0) * np.arctan2(np.tan(fovx * deg * np.float32(0.5)), aspect)   # rt/rtcommon.h:559
        assert abs(float(fovy) - float(CAM[f"c{i}_fovy"])) <= 2e-7
    pts = CAM["bsr_points"]
    sd = scenes.SceneData(); m = sd.add_material((0.5,) * 3)
    idx = np.arange(pts.shape[0] // 3 * 3).reshape(-1, 3)
    sd.add_mesh(pts[: idx.size], idx, m)
    sd.light_mesh = sd.add_quad([0, 0, 0], [0, 1e-3, 0], [1e-3, 0, 0], m)   # inside the point cloud's bounds
    sd.triangle_soup()
    osc = oa.Scene(sd)
    full = np.concatenate([pts[: idx.size], sd.meshes[1]["verts"]])
    lo, hi = full.min(0), full.max(0)
    want = np.float32(np.sqrt(np.float32(((hi - lo) ** 2).sum(dtype=np.float32)))) / np.float32(2)
    got = oracle.evo_scene_bounding_sphere_radius(osc.h)
    assert abs(got - float(want)) <= 1e-6 * float(want)
    if idx.size == pts.shape[0]:
        assert abs(got - float(CAM["bsr_radius"])) <= 1e-5 * got


def test_reference_build_matches_fixtures_when_present():
    """Where oracle/_ref exists (authoring container / prebuilt on the GPU box) re-run one reference call
    live, so stale fixtures cannot hide."""
    ref_path = os.path.join(oa.ROOT, "oracle", "_ref", "libref_pin.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built here")
    ref = C.CDLL(ref_path)
    ref.ref_mse.restype = C.c_double
    ref.ref_mse.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    img, other = G["b_img"], G["b_other"]
    assert ref.ref_mse(img.shape[1], img.shape[0], oa.ptr(other), oa.ptr(img)) == float(G["b_mse"])


@pytest.mark.parametrize("name", TEX_NAMES)
def test_texture_decoders_match_the_reference_decoder(name, tmp_path, evplp):
    """evplp_decode_image == stbi_load(path, &w, &h, &channel, 3) of the stb_image v2.16 the reference links
    (rt/rtcommon.h:144), byte for byte: baseline / progressive / restart-interval JPEG with every chroma layout,
    PNG of every colour type and bit depth incl. Adam7."""
    ext = ".jpg" if name.endswith("_jpg") else ".png"
    path = str(tmp_path / (name + ext))
    open(path, "wb").write(TEX[name + "__file"].tobytes())
    got, channels = evplp.decode_image(path)
    want = TEX[name + "__pixels"]
    assert got.shape == want.shape
    assert channels == int(TEX[name + "__channels"])
    assert np.array_equal(got, want), f"{int((got != want).sum())} bytes differ, max {int(np.abs(got.astype(int) - want).max())}"


def test_texture_decoder_errors(tmp_path, evplp):
    p = tmp_path / "trunc.jpg"
    data = TEX["base420_q75_jpg__file"].tobytes()
    p.write_bytes(data[: len(data) // 8])                       # header only: no EOI
    with pytest.raises(evplp.EvplpError):
        evplp.decode_image(str(p))
    q = tmp_path / "bad.png"
    png = bytearray(TEX["rgb8_png__file"].tobytes()); png[40] ^= 0xFF   # corrupt the deflate stream
    q.write_bytes(bytes(png))
    with pytest.raises(evplp.EvplpError):
        evplp.decode_image(str(q))
    with pytest.raises(evplp.EvplpError):
        evplp.decode_image(str(tmp_path / "missing.png"))
    r = tmp_path / "not_an_image.jpg"; r.write_bytes(b"hello world, not an image")
    with pytest.raises(evplp.EvplpError):
        evplp.decode_image(str(r))


def test_texture_decoders_on_the_reference_assets_when_present(evplp):
    """The living-room textures and the conference mask shipped with the reference (authoring container only):
    decoded live by the reference's decoder (oracle/_ref) and by this build."""
    ref_path = os.path.join(oa.ROOT, "oracle", "_ref", "libref_pin.so")
    assets = "/root/reference/scene"
    if not os.path.exists(ref_path) or not os.path.isdir(assets):
        pytest.skip("reference assets / oracle/_ref not present here")
    import glob
    ref = C.CDLL(ref_path)
    if not hasattr(ref, "ref_stbi_load"):
        pytest.skip("oracle/_ref predates the decoder pin")
    ref.ref_stbi_load.restype = C.c_void_p
    ref.ref_stbi_load.argtypes = [C.c_char_p] + [C.c_void_p] * 3 + [C.c_int]
    ref.ref_stbi_free.argtypes = [C.c_void_p]
    files = sorted(glob.glob(assets + "/livingroom/textures/*.jpg")) + [assets + "/conference/conference_mask.png"]
    assert len(files) >= 5
    for f in files:
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        ptr = ref.ref_stbi_load(f.encode(), C.byref(w), C.byref(h), C.byref(ch), 0)
        assert ptr, f
        want = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, 3)).copy()
        ref.ref_stbi_free(ptr)
        got, channels = evplp.decode_image(f)
        assert channels == ch.value and np.array_equal(got, want), f


# ---- round 2 pins: the jitter sampler and the triangle area (tests/golden/jitter.npz, areas.npz from oracle/_ref)
JIT = np.load(os.path.join(HERE, "golden", "jitter.npz"))
AREAS = np.load(os.path.join(HERE, "golden", "areas.npz"))


@pytest.mark.parametrize("seed", [int(s) for s in JIT["seeds"]])
def test_jitter_sequence_matches_the_reference_sampler(seed, evplp):
    """IndependentSampler(rngOffset).nextVec2() -> (2 u - 1) / resolution (rtcomphoton.h:887, 949), from the reference's own
    common/rng.h + sampler/independent.h compiled with g++ / libstdc++ (both the float mapping of uniform_real_distribution and
    the order of the two draws are implementation-defined: the authors' MSVC build may have drawn another sequence)."""
    n = int(JIT["n"]); W, H = (int(v) for v in JIT["res"])
    got = evplp.jitter_sequence(seed, n, W, H)
    assert got.tobytes() == JIT[f"ndc_{seed}"].tobytes()
    # the Python twin the oracle loops of the GPU tests use
    from test_gpu_end_to_end import MT19937, jitter_of
    rng = MT19937(seed)
    twin = np.array([jitter_of(rng, W, H) for _ in range(n)], np.float32)
    assert twin.tobytes() == JIT[f"ndc_{seed}"].tobytes()
    # and the raw draws: .y takes the first one
    rng = MT19937(seed)
    first, second = np.float32(np.uint32(rng())) / np.float32(2 ** 32), np.float32(np.uint32(rng())) / np.float32(2 ** 32)
    assert JIT[f"vec2_{seed}"][0, 1] == first and JIT[f"vec2_{seed}"][0, 0] == second


def test_triangle_area_matches_the_reference_expression(oracle):
    """glm::length(glm::cross(b - a, c - a)) / 2 (shapes/trianglemesh.cpp:13-19 on the vendored GLM) feeds the light CDF
    (rtcommon.h:501-531), the light area and totalArea (default clamping value, :759-768)."""
    for t, ref in zip(AREAS["tris"], AREAS["areas"]):
        assert oracle.evo_tri_area(oa.ptr(np.ascontiguousarray(t))) == ref


# ---- round 2 pin: the JSON reader against the reference's vendored nlohmann::json 2.1.1 (tests/golden/json_pins.json)
def test_json_reader_matches_the_reference_reader(evplp):
    """The technique blocks are read with `int v = json["k"]`-style conversions of nlohmann::json 2.1.1 (rtcomphoton.h:114-222,
    main.cpp:105-121).  The product has its own reader: same values, same conversions (a boolean converts to a number, nothing
    converts to a boolean or a string), same size(), a repeated key keeps its last value, \\u escapes become UTF-8, and what is
    not RFC 7159 JSON (NaN, +1, .5, 01, trailing commas, comments, control characters in strings) is rejected."""
    import ctypes as C, json, sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from json_cases import CASES
    pins = json.load(open(os.path.join(HERE, "golden", "json_pins.json")))["pins"]
    assert len(pins) > 700
    q = evplp.lib().evplp_json_query
    for ci, path, want, rc, num, hexstr in pins:
        text = CASES[ci][0]
        got_num = C.c_double(0); buf = C.create_string_buffer(4096)
        got = q(text.encode("utf-8"), path.encode(), want, C.byref(got_num), buf, 4096)
        what = f"case {ci} {text[:40]!r} path {path!r} want {want}"
        assert got == rc, f"{what}: status {got}, the reference's reader says {rc}"
        if rc == 0:
            assert repr(got_num.value) == num, f"{what}: {got_num.value!r} vs {num}"
            if want == 3:
                assert buf.value.hex() == hexstr, what
