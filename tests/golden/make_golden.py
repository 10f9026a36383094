#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libref_pin.so = the reference's
own floatimage.cpp + its vendored GLM, compiled where they lie by oracle/Makefile).  Run in the
authoring container (needs /root/reference); the fixtures -- inputs and expected outputs only -- are
committed so the pins also hold on the GPU box, where the reference does not exist.

Fixtures:
  output_surface.npz  PFM and HDR (RGBE) file bytes, decoded PNG pixels, FlipY, MSE / relMSE, error heat images and
                      LoadHDR read-back of seeded images
                      (common/floatimage/floatimage.cpp:64-128, 178-199, 223-273; rgbe.cpp)
  camera.npz          projection*view matrices (with / without the jitter translation) for seeded
                      cameras incl. the conference camera, fovx->fovy, bounding-sphere radii
                      (rt/rtcommon.h:548-591, 805-814; rt/rtcomphoton/rtcomphoton.h:943-952)
  textures.npz        small JPEG / PNG files (written here with Pillow from seeded images: baseline 4:4:4 / 4:2:2 /
                      4:2:0 / 4:1:1, grey, progressive, optimised tables, restart intervals, odd sizes; PNG grey / RGB /
                      RGBA / palette (+tRNS) / 1-2-4-16 bit / Adam7) and the pixels the reference's decoder returns for
                      them: stbi_load(path, &w, &h, &channel, 3) of the vendored stb_image v2.16 (rt/rtcommon.h:144)
"""
import ctypes as C
import os
import struct
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_pin.so")


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def decode_png_rgb8(data: bytes):
    """Minimal PNG decoder (8-bit RGB, non-interlaced, all five filters)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8]); body = data[pos + 8:pos + 8 + n]; pos += 12 + n
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10]); assert depth == 8 and ctype == 2
        elif typ == b"IDAT":
            idat += body
    raw = zlib.decompress(idat)
    stride = w * 3
    out = np.zeros((h, stride), np.uint8); prev = np.zeros(stride, np.int32)
    for y in range(h):
        f = raw[y * (stride + 1)]; line = np.frombuffer(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)], np.uint8).astype(np.int32)
        cur = np.zeros(stride, np.int32)
        for i in range(stride):
            a = cur[i - 3] if i >= 3 else 0; b = prev[i]; c = prev[i - 3] if i >= 3 else 0
            if f == 0: pred = 0
            elif f == 1: pred = a
            elif f == 2: pred = b
            elif f == 3: pred = (a + b) // 2
            else:
                p = a + b - c; pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[i] = (line[i] + pred) & 255
        out[y] = cur; prev = cur
    return out.reshape(h, w, 3)


def smooth_image(rng, w, h):
    """Photo-like test image: low-frequency colour field + edges + a little noise (so that chroma upsampling,
    clamping and every IDCT path matter)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for k in range(3):
        fx, fy, ph = rng.rand(3) * [0.35, 0.35, 6.28]
        img[..., k] = 127 + 100 * np.sin(fx * x + ph) * np.cos(fy * y + 0.5 * ph)
    img[h // 3: h // 3 + max(h // 5, 1), w // 4: w // 4 + max(w // 3, 1)] = [250, 10, 30]      # saturated block: clamping
    img[:, w // 2] = 0
    img += rng.randn(h, w, 3) * 6
    return np.clip(img, 0, 255).astype(np.uint8)


def adam7_png(rgb):
    """Pillow cannot write interlaced PNGs: build an Adam7 RGB8 file by hand (filter type 0/1/2 alternating)."""
    h, w, _ = rgb.shape
    xs, ys, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    raw = b""
    for p in range(7):
        sub = rgb[ys[p]::dy[p], xs[p]::dx[p]]
        if sub.size == 0:
            continue
        prev = np.zeros(sub.shape[1] * 3, np.int32)
        for r, row in enumerate(sub):
            cur = row.reshape(-1).astype(np.int32); f = r % 3
            if f == 0: enc = cur
            elif f == 1: enc = cur - np.concatenate([np.zeros(3, np.int32), cur[:-3]])
            else: enc = cur - prev
            raw += bytes([f]) + (enc & 255).astype(np.uint8).tobytes(); prev = cur

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def make_textures(ref):
    import io
    from PIL import Image
    ref.ref_stbi_load.restype = C.c_void_p
    ref.ref_stbi_load.argtypes = [C.c_char_p] + [C.c_void_p] * 3 + [C.c_int]
    ref.ref_stbi_free.argtypes = [C.c_void_p]
    rng = np.random.RandomState(4242)
    files = {}

    def jpeg(name, img, mode="RGB", **kw):
        b = io.BytesIO(); Image.fromarray(img, mode).save(b, "JPEG", **kw); files[name + ".jpg"] = b.getvalue()

    def png(name, im, **kw):
        b = io.BytesIO(); im.save(b, "PNG", **kw); files[name + ".png"] = b.getvalue()

    a = smooth_image(rng, 37, 29); b_ = smooth_image(rng, 64, 48); c = smooth_image(rng, 9, 70); d = smooth_image(rng, 1, 1); e = smooth_image(rng, 50, 3)
    jpeg("base444_q95", a, quality=95, subsampling=0)
    jpeg("base422_q85", a, quality=85, subsampling=1)
    jpeg("base420_q75", a, quality=75, subsampling=2)
    jpeg("base420_q30_64x48", b_, quality=30, subsampling=2)
    jpeg("base420_tall", c, quality=90, subsampling=2)
    jpeg("base420_1x1", d, quality=90, subsampling=2)
    jpeg("base422_wide", e, quality=90, subsampling=1)
    jpeg("grey_q80", a[..., 0].copy(), mode="L", quality=80)
    jpeg("prog420_q80", b_, quality=80, subsampling=2, progressive=True)
    jpeg("prog444_q92", a, quality=92, subsampling=0, progressive=True)
    jpeg("prog_grey", c[..., 1].copy(), mode="L", quality=85, progressive=True)
    jpeg("opt420_q60", b_, quality=60, subsampling=2, optimize=True)
    jpeg("rst420_q85", b_, quality=85, subsampling=2, restart_marker_blocks=3)
    jpeg("rst_prog", b_, quality=70, subsampling=2, progressive=True, restart_marker_rows=1)
    jpeg("q100_444", a, quality=100, subsampling=0)
    jpeg("base411", b_, quality=85, subsampling="4:1:1")     # h = 4: the replicating upsampler
    png("rgb8", Image.fromarray(a, "RGB"))
    png("rgba8", Image.fromarray(np.dstack([a, (a[..., 0] // 2 + 100).astype(np.uint8)]), "RGBA"))
    png("grey8", Image.fromarray(a[..., 1].copy(), "L"))
    png("greya8", Image.fromarray(np.dstack([a[..., 1], a[..., 2]]), "LA"))
    pal = Image.fromarray(b_, "RGB").quantize(colors=61)
    png("pal8", pal)
    png("pal8_trns", pal, transparency=5)
    png("pal4", Image.fromarray(a, "RGB").quantize(colors=13), bits=4)
    png("pal2", Image.fromarray(a, "RGB").quantize(colors=4), bits=2)
    png("bilevel", Image.fromarray(a[..., 0] > 127))
    png("grey16", Image.fromarray((a[..., 0].astype(np.uint16) * 257 + 31).astype(np.uint16)))
    png("rgb8_trns", Image.fromarray(a, "RGB"), transparency=(250, 10, 30))
    png("rgb8_opt", Image.fromarray(b_, "RGB"), optimize=True)
    png("rgb8_stored", Image.fromarray(e, "RGB"), compress_level=0)
    files["adam7_rgb8.png"] = adam7_png(a)
    files["adam7_small.png"] = adam7_png(smooth_image(rng, 3, 2))
    out = {}
    tmp = tempfile.mkdtemp()
    for name, data in files.items():
        path = os.path.join(tmp, name); open(path, "wb").write(data)
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        ptr = ref.ref_stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(ch), 0)
        assert ptr, name
        px = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, 3)).copy()
        ref.ref_stbi_free(ptr)
        key = name.replace(".", "_")
        out[key + "__file"] = np.frombuffer(data, np.uint8); out[key + "__pixels"] = px; out[key + "__channels"] = np.int32(ch.value)
    np.savez_compressed(os.path.join(HERE, "textures.npz"), **out)
    print("textures.npz:", len(files), "files,", sum(len(v) for v in files.values()), "bytes of image files")


sys.path.insert(0, HERE)


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/libref_pin.so missing: run `make -C oracle ref` where /root/reference exists")
    ref = C.CDLL(REF)
    ref.ref_save.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    ref.ref_flip_y.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ref.ref_mse.restype = C.c_double; ref.ref_mse.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ref.ref_rel_mse.restype = C.c_double; ref.ref_rel_mse.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ref.ref_error_heat.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    ref.ref_load_hdr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    ref.ref_view_projection.argtypes = [C.c_void_p] * 3 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    ref.ref_fovx_to_fovy.restype = C.c_float; ref.ref_fovx_to_fovy.argtypes = [C.c_float, C.c_float]
    ref.ref_bounding_sphere_radius.restype = C.c_float; ref.ref_bounding_sphere_radius.argtypes = [C.c_int, C.c_void_p]

    make_textures(ref)
    rng = np.random.RandomState(20261002)
    out = {}
    tmp = tempfile.mkdtemp()
    for name, (w, h) in {"a": (4, 2), "b": (17, 9), "c": (40, 5)}.items():
        img = (rng.rand(h, w, 3).astype(np.float32) * np.float32(1.6)).astype(np.float32)   # some values > 1 (PNG clamp)
        img[0, 0] = [0.0, 1.0, 0.5]
        other = (img + rng.randn(h, w, 3).astype(np.float32) * np.float32(0.05)).astype(np.float32)
        pfm = os.path.join(tmp, name + ".pfm"); png = os.path.join(tmp, name + ".png"); hdr = os.path.join(tmp, name + ".hdr")
        assert ref.ref_save(pfm.encode(), w, h, P(img)) == 0 and ref.ref_save(png.encode(), w, h, P(img)) == 0
        hdr_img = img.copy(); hdr_img[h // 2, : w // 2] = hdr_img[h // 2, 0]          # a run, to exercise the RLE
        assert ref.ref_save(hdr.encode(), w, h, P(hdr_img)) == 0
        out[f"{name}_hdr_img"] = hdr_img
        out[f"{name}_hdr_bytes"] = np.frombuffer(open(hdr, "rb").read(), np.uint8)
        flipped = np.zeros_like(img); ref.ref_flip_y(w, h, P(img), P(flipped))
        out[f"{name}_img"] = img; out[f"{name}_other"] = other
        out[f"{name}_pfm_bytes"] = np.frombuffer(open(pfm, "rb").read(), np.uint8)
        out[f"{name}_png_pixels"] = decode_png_rgb8(open(png, "rb").read())
        out[f"{name}_flipy"] = flipped
        for rel in (0, 1):                                                              # floatimage.cpp:21-62
            heat = np.zeros_like(img); ref.ref_error_heat(w, h, P(other), P(img), np.float32(0.004), rel, P(heat))
            out[f"{name}_heat{rel}"] = heat
        back = np.zeros_like(img)
        assert ref.ref_load_hdr(hdr.encode(), w, h, P(back)) == 0                        # FloatImage::LoadHDR of the file written above
        out[f"{name}_hdr_decoded"] = back
        out[f"{name}_mse"] = np.float64(ref.ref_mse(w, h, P(other), P(img)))
        out[f"{name}_relmse"] = np.float64(ref.ref_rel_mse(w, h, P(other), P(img)))
    np.savez(os.path.join(HERE, "output_surface.npz"), **out)

    cams = [dict(origin=[15.56, -4.79, 4.37], lookat=[1.15, 2.28, 1.76], up=[0, 0, 1], fovx=70.0, aspect=1280 / 720),   # conference_vpl.json:16-33
            dict(origin=[15.56, -4.79, 4.37], lookat=[1.15, 2.28, 1.76], up=[0, 0, 1], fovx=70.0, aspect=1.0)]
    for _ in range(4):
        o = rng.randn(3) * 5; cams.append(dict(origin=o.tolist(), lookat=(o + rng.randn(3) * 3).tolist(), up=[0, 0, 1], fovx=float(30 + 60 * rng.rand()), aspect=float(0.5 + 1.5 * rng.rand())))
    cam_out = {"n": np.int32(len(cams))}
    for i, c in enumerate(cams):
        o, l, u = (np.asarray(c[k], np.float32) for k in ("origin", "lookat", "up"))
        fovy = ref.ref_fovx_to_fovy(c["fovx"], c["aspect"])
        jitter = ((2 * rng.rand(2) - 1) / np.array([1024, 1024])).astype(np.float32)
        m0 = np.zeros(16, np.float32); m1 = np.zeros(16, np.float32)
        ref.ref_view_projection(P(o), P(l), P(u), fovy, c["aspect"], None, P(m0))
        ref.ref_view_projection(P(o), P(l), P(u), fovy, c["aspect"], P(jitter), P(m1))
        cam_out.update({f"c{i}_origin": o, f"c{i}_lookat": l, f"c{i}_up": u, f"c{i}_fovx": np.float32(c["fovx"]), f"c{i}_aspect": np.float32(c["aspect"]),
                        f"c{i}_fovy": np.float32(fovy), f"c{i}_jitter": jitter, f"c{i}_mvp": m0.reshape(4, 4), f"c{i}_mvp_jittered": m1.reshape(4, 4)})
    pts = (rng.randn(500, 3) * [10, 6, 3]).astype(np.float32)
    cam_out["bsr_points"] = pts
    cam_out["bsr_radius"] = np.float32(ref.ref_bounding_sphere_radius(pts.shape[0], P(pts)))
    np.savez(os.path.join(HERE, "camera.npz"), **cam_out)

    # jitter sampler (common/rng.h + sampler/independent.h compiled as they are) and triangle areas (GLM expression of
    # shapes/trianglemesh.cpp:13-19, the file itself needs Assimp)
    jit = {"seeds": np.array([0, 1, 7, 12345, 4294967295], np.uint32), "n": np.int32(64), "res": np.array([1280, 720], np.int32)}
    for sd in jit["seeds"]:
        v = np.zeros(2 * 64, np.float32); j = np.zeros(2 * 64, np.float32)
        ref.ref_jitter_vec2(C.c_uint32(int(sd)), 64, P(v)); ref.ref_jitter_ndc(C.c_uint32(int(sd)), 64, C.c_float(1280.0), C.c_float(720.0), P(j))
        jit[f"vec2_{int(sd)}"] = v.reshape(64, 2); jit[f"ndc_{int(sd)}"] = j.reshape(64, 2)
    np.savez(os.path.join(HERE, "jitter.npz"), **jit)
    ref.ref_triangle_area.restype = C.c_float
    tris = (rng.randn(300, 9) * np.repeat(10.0 ** rng.uniform(-3, 1.5, (300, 1)), 9, 1)).astype(np.float32)
    tris[:20, 3:6] = tris[:20, 0:3] + (tris[:20, 6:9] - tris[:20, 0:3]) * np.float32(0.5) + np.float32(1e-4) * rng.randn(20, 3).astype(np.float32)   # skinny
    areas = np.array([ref.ref_triangle_area(P(t[0:3].copy()), P(t[3:6].copy()), P(t[6:9].copy())) for t in tris], np.float32)
    np.savez(os.path.join(HERE, "areas.npz"), tris=tris, areas=areas)

    # the reference's JSON reader (vendored nlohmann::json 2.1.1) on the texts of json_cases.py (written for this repository):
    # conversions (`int v = json["k"]`, float, bool, std::string), size(), kind, missing keys, and what is not JSON at all
    import json as pyjson
    from json_cases import CASES
    ref.ref_json_query.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_int]
    pins = []
    for ci, (text, paths) in enumerate(CASES):
        for path in paths:
            for want in range(6):
                if path.endswith("huge") and want == 0:
                    continue                                   # double -> int out of range: undefined in C++, nothing to pin
                num = C.c_double(0); buf = C.create_string_buffer(4096)
                rc = ref.ref_json_query(text.encode("utf-8"), path.encode(), want, C.byref(num), buf, 4096)
                pins.append([ci, path, want, rc, repr(num.value) if rc == 0 else None, buf.value.hex() if rc == 0 and want == 3 else None])
    with open(os.path.join(HERE, "json_pins.json"), "w") as f:
        pyjson.dump({"source": "oracle/_ref ref_json_query = nlohmann::json 2.1.1 as vendored by the reference (reflectcuts/json/json.hpp)",
                     "columns": ["case index in json_cases.CASES", "path", "want", "rc", "repr(number)", "hex(string)"], "pins": pins}, f, indent=0)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
