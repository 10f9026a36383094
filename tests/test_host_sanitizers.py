"""The host-side readers of the asset pipeline under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; the pool has no GPU
sanitizers): tools/host_fuzz/host_fuzz.cpp compiled with the decoders, the float-image loaders, the JSON reader and the OBJ / MTL reader,
run over the committed fixtures and seeded mutations of them.  A reader may refuse a file; it may not touch memory out of bounds, leave
signed arithmetic undefined, or ask for memory out of proportion to the file.  (Round 5: the first runs found a heap over-read for
sampling factors that are not integer ratios, signed overflow in the IDCT / dequantisation of corrupted coefficients and two
allocations sized by a header alone -- fixed in decode_jpeg.cpp / inflate.cpp; a face index near 2^31 in scene_io.cpp.  A longer campaign of the
same harness after the fixes -- 26 seeds x 6 000 mutations of each of 40 files, 6 M inputs -- ran clean.)"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST = os.path.join(ROOT, "evplp_amd", "csrc", "host")
SOURCES = [os.path.join(ROOT, "tools", "host_fuzz", "host_fuzz.cpp")] + [os.path.join(HOST, f) for f in
                                                                        ("decode_jpeg.cpp", "decode_png.cpp", "inflate.cpp", "images.cpp", "scene_io.cpp")]

OBJ = """# a small scene
mtllib room.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
usemtl wall
f 1/1/1 2/2/1 3/3/1
f 1/1/1 3/3/1 4/4/1
usemtl floor
f 1 2 6 5
f -1 -2 -3
g other
usemtl wall
f 1//1 2//1 6//1
"""
MTL = """newmtl wall
Kd 0.7 0.6 0.5
Ks 0.1 0.1 0.1
Ns 40
map_Kd tex_a.png
newmtl floor
Kd 0.3 0.3 0.3
Ks 0 0 0
Ns 0
map_Ks tex_b.png
map_Ns tex_a.png
"""
SCENE = """{ "camera": { "origin": [0.5, 0.5, 3], "direction": [0.5, 0.5, 0], "up": [0, 1, 0], "fovy": 40, "resolution": [64, 48] },
  "objects": ["room.obj"], "arealight": { "path": "light.obj", "intensity": [10, 10, 10, 1] },
  "photonfam": { "numMaxIteration": 3, "renderMode": "vpl", "misMode": "balance", "alpha": 0.7, "numLightPaths": 1000 } }
"""


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("host_fuzz")
    exe = str(d / "host_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"), "-o", exe] + SOURCES
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    seeds = d / "seeds"; seeds.mkdir()
    tex = np.load(os.path.join(HERE, "golden", "textures.npz"))
    pngs = []
    for k in tex.files:
        if k.endswith("__file"):
            name = k[: -len("__file")]
            ext = ".jpg" if name.endswith("_jpg") else ".png"
            (seeds / (name + ext)).write_bytes(tex[k].tobytes())
            if ext == ".png": pngs.append(tex[k].tobytes())
    (seeds / "tex_a.png").write_bytes(pngs[0]); (seeds / "tex_b.png").write_bytes(pngs[1])
    (seeds / "room.obj").write_text(OBJ); (seeds / "room.mtl").write_text(MTL); (seeds / "scene.json").write_text(SCENE)
    out = np.load(os.path.join(HERE, "golden", "output_surface.npz"))      # the reference's own PFM / HDR bytes of two small images
    for name in ("a", "b"):
        (seeds / (name + ".pfm")).write_bytes(out[f"{name}_pfm_bytes"].tobytes())
        if f"{name}_hdr_bytes" in out.files: (seeds / (name + ".hdr")).write_bytes(out[f"{name}_hdr_bytes"].tobytes())
    return exe, str(seeds)


@pytest.mark.parametrize("rng_seed", [1, 2])
def test_readers_survive_mutated_files_under_asan_and_ubsan(harness, rng_seed):
    exe, seeds = harness
    env = dict(os.environ, ASAN_OPTIONS="max_allocation_size_mb=1024:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, seeds, "400", str(rng_seed)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    # every unmutated fixture decodes (iteration 0 of every seed): at least as many successes as image seeds
    line = r.stdout.strip().splitlines()[-1]
    decoded = int(line.split("images:")[1].split("decoded")[0])
    assert decoded >= len([f for f in os.listdir(seeds) if f.endswith((".jpg", ".png"))]), line


def test_block_deal_under_asan_and_ubsan(tmp_path):
    """evplp_deal_blocks / evplp_rank_blocks (csrc/host/deal.cpp; round 6) on 6 000 random cost tables -- equal costs, zeros, costs up to 2^64,
    no blocks, capacities that just fit and that do not -- under ASan + UBSan, every deal checked for its invariants (one owner per block, nobody
    over capacity, never worse than round robin by its own measure, a rank's blocks listed in falling cost)."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "deal_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tools", "host_fuzz", "deal_fuzz.cpp"), os.path.join(HOST, "deal.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for seed in ("5", "77"):
        r = subprocess.run([exe, "3000", seed], capture_output=True, text=True, timeout=600, env=dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1"))
        assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout[-500:], r.stderr[-3000:])


def test_concave_polygons_are_ear_clipped(tmp_path):
    """aiProcess_Triangulate (rtcommon.h:650-653): the OBJ reader turns a convex polygon into a fan and ear-clips one with a reflex corner
    (round 6: a fan of an L- or arrow-shaped face covers area outside it).  tools/host_fuzz/polygon_check.cpp writes L, arrow, U, star and
    convex faces in a tilted plane, both windings, loads them through the reader under ASan + UBSan and checks n - 2 triangles, the
    polygon's area, the polygon's winding."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "polygon_check")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tools", "host_fuzz", "polygon_check.cpp")] + SOURCES[1:]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1500:], r.stderr[-2000:])


def test_jpeg_refusals_found_by_the_harness(tmp_path):
    """The two refusals decode_jpeg.cpp makes beyond the reference's stb_image v2.16 (round 5): sampling factors that are not integer
    ratios (its resamplers -- and ours -- would read past the end of a component's plane) and a frame header asking for more pixels
    than the data behind it can fill.  Through the product's C ABI (evplp_decode_image; no GPU needed)."""
    sys.path.insert(0, ROOT)
    import evplp_amd as ev
    tex = np.load(os.path.join(HERE, "golden", "textures.npz"))
    name = sorted(k for k in tex.files if k.endswith("_jpg__file") and "420" in k)[0]
    data = bytearray(tex[name].tobytes())
    sof = next(i for i in range(len(data) - 1) if data[i] == 0xFF and data[i + 1] in (0xC0, 0xC2))
    assert data[sof + 9] == 3                                    # three components; component k's sampling byte at sof + 11 + 3 k
    ok = tmp_path / "ok.jpg"; ok.write_bytes(bytes(data))
    img, channels = ev.decode_image(str(ok))
    assert img.shape == tex[name[:-len("__file")] + "__pixels"].shape
    bad = bytearray(data); bad[sof + 11] = 0x31; bad[sof + 14] = 0x21            # H factors 3, 2, 1: 3 / 2 is not an integer
    p = tmp_path / "ratio.jpg"; p.write_bytes(bytes(bad))
    with pytest.raises(ev.EvplpError):
        ev.decode_image(str(p))
    big = bytearray(data); big[sof + 5:sof + 9] = b"\xff\xff\xff\xff"           # 65535 x 65535 pixels behind a 1 KB file
    p = tmp_path / "big.jpg"; p.write_bytes(bytes(big))
    with pytest.raises(ev.EvplpError):
        ev.decode_image(str(p))
