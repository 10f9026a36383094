"""Host-side logic of libevplp_hip.so that needs no GPU: the C-ABI surface, the scene / technique
JSON + OBJ readers, the error behaviour, the strip geometry.  (No compute calls here.)"""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(evplp):
    hdr = open(os.path.join(ROOT, "include", "evplp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(evplp_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(evplp.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in include/evplp.h but not exported: {missing}"
    assert declared == set(evplp._SIGNATURES), declared ^ set(evplp._SIGNATURES)
    assert lib.evplp_abi_version() == 4 == evplp.ABI_VERSION


def test_struct_layouts_match_the_header(evplp, tmp_path):
    """The ctypes mirrors against the header itself: a C program prints sizeof / offsetof of what include/evplp.h declares."""
    assert C.sizeof(evplp.FrameParams) == 72 and C.sizeof(evplp.Config) == 88
    assert C.sizeof(evplp.Material) == 40 and C.sizeof(evplp.Camera) == 44 and C.sizeof(evplp.PassStats) == 64
    assert evplp.RECORD_DTYPE.itemsize == 96
    src = tmp_path / "layout.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "evplp.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(evplp_config), sizeof(evplp_frame_params), sizeof(evplp_material), sizeof(evplp_camera), sizeof(evplp_pass_stats),
           sizeof(evplp_record), sizeof(evplp_group_config));
    printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(evplp_config, cut_scratch_bytes), offsetof(evplp_config, vsl_mask_bytes), offsetof(evplp_frame_params, jitter),
           offsetof(evplp_frame_params, splat_footprint), offsetof(evplp_pass_stats, shaded), offsetof(evplp_config, band_first_row));
    printf("%d %d\\n", EVPLP_ABI_VERSION, EVPLP_MAX_PROXY_PLANES);
    return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    sizes = [int(x) for x in lines[0].split()]
    assert sizes == [C.sizeof(evplp.Config), C.sizeof(evplp.FrameParams), C.sizeof(evplp.Material), C.sizeof(evplp.Camera), C.sizeof(evplp.PassStats), 96, C.sizeof(evplp.GroupConfig)]
    offs = [int(x) for x in lines[1].split()]
    assert offs == [evplp.Config.cut_scratch_bytes.offset, evplp.Config.vsl_mask_bytes.offset, evplp.FrameParams.jitter.offset,
                    evplp.FrameParams.splat_footprint.offset, evplp.PassStats.shaded.offset, evplp.Config.band_first_row.offset]
    assert [int(x) for x in lines[2].split()] == [evplp.ABI_VERSION, 128]


def test_create_fails_loudly_without_a_gpu(evplp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(evplp.EvplpError) as e:
        evplp.Context(16, 16, 4, 4, 4)
    assert e.value.status == evplp.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_create_rejects_bad_configs(evplp):
    cfg = evplp.Config()
    h = C.c_void_p()
    cfg.abi_version = 99
    assert evplp.lib().evplp_create(C.byref(cfg), C.byref(h)) == evplp.ERR_INVALID
    assert b"ABI version" in evplp.lib().evplp_last_error(None)
    cfg.abi_version = evplp.ABI_VERSION; cfg.res_x = cfg.res_y = 8; cfg.num_light_paths = 4; cfg.num_vpl_light_paths = 8; cfg.photons_per_path = 4
    assert evplp.lib().evplp_create(C.byref(cfg), C.byref(h)) == evplp.ERR_INVALID
    assert b"num_vpl_light_paths" in evplp.lib().evplp_last_error(None)
    cfg.num_vpl_light_paths = 4; cfg.strip_capacity_rows = -16
    assert evplp.lib().evplp_create(C.byref(cfg), C.byref(h)) == evplp.ERR_INVALID
    assert b"strip_capacity_rows" in evplp.lib().evplp_last_error(None)
    cfg.strip_capacity_rows = 0
    assert evplp.lib().evplp_create(None, C.byref(h)) == evplp.ERR_INVALID


def test_synth_scene_and_independent_obj_parse(evplp, tmp_path):
    jp = evplp.synth_scene(str(tmp_path), "room", 5000, 7, 64, 48)
    root = json.load(open(jp))
    assert root["resX"] == 64 and root["resY"] == 48 and root["arealight"]["intensity"] == [17, 12, 4, 0]
    assert root["photonfam"]["numLightPaths"] == 1024 and root["photonfam"]["misMode"] == "one"
    sd, _ = scenes.load_obj_scene(jp)
    v, uv, m = sd.triangle_soup()
    assert 4000 < v.shape[0] < 6500 and sd.light_count == 128
    # consistently wound closed room: every non-light surface normal faces the room interior or outward of a box;
    # the light faces down
    lv = v[sd.light_first: sd.light_first + sd.light_count].reshape(-1, 3, 3)
    ln = np.cross(lv[:, 1] - lv[:, 0], lv[:, 2] - lv[:, 0])
    assert (ln[:, 2] < 0).all()
    # same seed -> same files
    jp2 = evplp.synth_scene(str(tmp_path / "again"), "room", 5000, 7, 64, 48)
    assert open(jp.replace(".json", ".obj")).read() == open(jp2.replace(".json", ".obj")).read()


def _render(evplp, path, overrides=None):
    err = C.create_string_buffer(1024)
    rc = evplp.lib().evplp_render_json(path.encode(), overrides.encode() if overrides else None, 0, err, 1024)
    return rc, err.value.decode()


def test_render_json_error_behaviour(evplp, tmp_path):
    rc, msg = _render(evplp, str(tmp_path / "missing.json"))
    assert rc == evplp.ERR_IO and "cannot open" in msg
    bad = tmp_path / "bad.json"; bad.write_text("{ not json")
    rc, msg = _render(evplp, str(bad))
    assert rc == evplp.ERR_PARSE
    jp = evplp.synth_scene(str(tmp_path), "room", 600, 1, 32, 32)
    root = json.load(open(jp))
    # missing required technique key (nlohmann would throw on json["numLightPaths"], rtcomphoton.h:114)
    broken = dict(root); broken["photonfam"] = {k: v for k, v in root["photonfam"].items() if k != "numLightPaths"}
    p = tmp_path / "nokey.json"; p.write_text(json.dumps(broken))
    rc, msg = _render(evplp, str(p))
    assert rc == evplp.ERR_PARSE and "numLightPaths" in msg
    # clampingStart is a hard error (rtcomphoton.h:137-142); unknown misMode / frameMode too (map.at throws)
    for key, val, needle in (("clampingStart", 1.0, "clampingStart"), ("misMode", "bogus", "misMode"), ("frameMode", "bogus", "frameMode")):
        rc, msg = _render(evplp, jp, json.dumps({key: val}))
        assert rc == evplp.ERR_PARSE and needle in msg, (key, rc, msg)
    # no camera
    nocam = {k: v for k, v in root.items() if k != "camera"}
    p = tmp_path / "nocam.json"; p.write_text(json.dumps(nocam))
    rc, msg = _render(evplp, str(p))
    assert rc == evplp.ERR_PARSE and "camera" in msg
    # the pt block (RtPt2::render, rtpt2.h:91-111) has its own required keys
    pt = dict(root); pt["pt"] = {"numMaxBounces": 3}; pt.pop("photonfam")
    p = tmp_path / "pt.json"; p.write_text(json.dumps(pt))
    rc, msg = _render(evplp, str(p))
    assert rc == evplp.ERR_PARSE and "rngOffset" in msg
    # no technique block at all
    none = {k: v for k, v in root.items() if k != "photonfam"}
    p = tmp_path / "none.json"; p.write_text(json.dumps(none))
    rc, msg = _render(evplp, str(p))
    assert rc == evplp.ERR_PARSE and "technique" in msg
    # a valid file reaches context creation: without a GPU that fails loudly, never silently
    import torch
    if not torch.cuda.is_available():
        rc, msg = _render(evplp, jp)
        assert rc != evplp.OK and "no HIP device" in msg


def test_driver_binary_reports_errors(tmp_path):
    exe = os.path.join(ROOT, "evplp_amd", "lib", "evplp-render")
    r = subprocess.run([exe, str(tmp_path / "nope.json")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    r = subprocess.run([exe, "--synth", str(tmp_path), "s", "800"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(tmp_path / "s.json") and os.path.exists(tmp_path / "s_lights.obj")


def test_strip_geometry_partitions_every_row_once():
    from evplp_amd import strips
    for H, count, sr in ((1024, 8, 16), (1024, 4, 16), (1080, 4, 8), (720, 8, 8), (64, 2, 8), (50, 3, 8), (64, 1, 16)):
        seen = np.zeros(H, np.int32)
        lr = strips.local_rows(H, count, sr)
        for r in range(count):
            rows = strips.global_rows(H, r, count, sr)
            assert rows.shape[0] == lr
            seen[rows[rows < H]] += 1
        assert (seen == 1).all(), (H, count, sr)
    b, n, split = strips.path_slice(1024, 3, 8)
    assert (b, n, split) == (384, 128, True)
    assert strips.path_slice(1000, 1, 3) == (0, 1000, False)


def test_image_io_roundtrip_and_errors(evplp, tmp_path):
    img = np.random.RandomState(0).rand(5, 7, 3).astype(np.float32)
    p = str(tmp_path / "x.pfm")
    evplp.save_image(p, img)
    assert np.array_equal(evplp.load_pfm(p), img)
    with pytest.raises(evplp.EvplpError):
        evplp.save_image(str(tmp_path / "x.bmp"), img)        # "unsupported file format" floatimage.cpp:272
    with pytest.raises(evplp.EvplpError):
        evplp.load_pfm(str(tmp_path / "none.pfm"))


def test_shipped_scene_files_carry_every_key_the_techniques_require():
    """All 30 scene JSONs of the reference (authoring container only) parse, and their technique blocks hold the
    keys RtComPhoton::render / RtPt2::render read unconditionally (rtcomphoton.h:114-186, rtpt2.h:91-110)."""
    import glob
    files = sorted(glob.glob("/root/reference/scene/*/*.json"))
    if not files:
        pytest.skip("reference scenes not present here")
    need = {"photonfam": ["numLightPaths", "numVplLightPaths", "numMaxBounces", "radiusPercentage", "numMaxIteration", "timeLimitMs", "frameMode", "rngOffset",
                          "combinedFilename", "weightedPhotonFilename", "weightedVplFilename", "statFilename", "useJitter", "useStat"],
            "pt": ["rngOffset", "numMaxIteration", "timeLimitMs", "frameMode", "outputFilename", "statFilename", "useJitter", "useStat", "numSamplePerPixel", "numMaxBounces"]}
    seen = 0
    for f in files:
        root = json.load(open(f))
        assert {"resX", "resY", "scene", "arealight"} <= set(root) and ("camera" in root or "stablecamera" in root), f
        for block, keys in need.items():
            if block in root:
                seen += 1
                missing = [k for k in keys if k not in root[block]]
                assert not missing, (f, block, missing)
                assert root[block]["frameMode"] in ("accumulate", "cleareveryframe")
                if "misMode" in root[block]:
                    assert root[block]["misMode"] in ("one", "balance", "max", "power2", "geometryClamp", "geometryBrdfClamp")
    assert seen == len(files) == 30


def test_lfs_pointer_meshes_are_reported_as_such(evplp):
    f = "/root/reference/scene/conference/conference_ours.json"
    if not os.path.exists(f):
        pytest.skip("reference scenes not present here")
    rc, msg = _render(evplp, f)
    assert rc == evplp.ERR_IO and "Git-LFS pointer" in msg


def test_deal_blocks_balances_and_is_deterministic(evplp):
    """evplp_deal_blocks (host only): every block gets exactly one owner, nobody exceeds the capacity, the fullest rank is never fuller than under
    the round-robin deal, close to the mean when the blocks allow it, and the same costs always give the same table (every process of a
    multi-process run computes it for itself)."""
    import numpy as np
    rng = np.random.default_rng(7)
    for n, nb, capf in [(2, 8, 1.5), (4, 64, 1.5), (8, 64, 1.5), (8, 128, 1.5), (8, 64, 1.0), (3, 7, 2.0), (64, 64, 1.0)]:
        # a costly belt in the middle of the image (the furnished room's chairs and table), a cheap floor and ceiling, noise
        y = (np.arange(nb) + 0.5) / nb
        cost = (1000 * (1.0 + 8.0 * np.exp(-((y - 0.48) / 0.06) ** 2)) * rng.uniform(0.8, 1.2, nb)).astype(np.uint64)
        share = -(-nb // n); cap = min(nb, int(np.ceil(share * capf)))
        owner = evplp.deal_blocks(cost, n, cap)
        assert owner.shape == (nb,) and owner.min() >= 0 and owner.max() < n
        counts = np.bincount(owner, minlength=n)
        assert counts.max() <= cap and counts.sum() == nb
        loads = np.array([cost[owner == r].sum() for r in range(n)], dtype=np.float64)
        rr = np.array([cost[np.arange(nb) % n == r].sum() for r in range(n)], dtype=np.float64)
        assert loads.max() <= rr.max()
        if nb >= 8 * n and capf > 1.0:
            assert loads.mean() / loads.max() > 0.97, (n, nb, loads.tolist())
        assert evplp.deal_blocks(cost, n, cap).tolist() == owner.tolist()
    # all costs equal: still a valid deal within the capacity
    owner = evplp.deal_blocks(np.full(10, 5, np.uint64), 4, 3)
    assert np.bincount(owner, minlength=4).max() <= 3
    with pytest.raises(evplp.EvplpError):
        evplp.deal_blocks(np.ones(9, np.uint64), 2, 4)                 # 9 blocks do not fit 2 x 4


def test_block_tables_reassemble(evplp):
    """strips.rows_of_blocks / assemble_blocks (what bench.py's rank processes use after a deal by cost) mirror StripDev::global_row with a table."""
    import numpy as np
    from evplp_amd import strips
    H, W, SR = 100, 5, 16                                              # 7 blocks, the last one short
    owner = np.array([2, 0, 0, 1, 2, 0, 1])
    frame = np.arange(H * W * 3, dtype=np.float32).reshape(H, W, 3)
    chunk = 3 * SR
    gathered = np.full((3, chunk, W, 3), -1.0, np.float32)
    for r in range(3):
        blocks = strips.blocks_of_rank(owner, r)
        rows = strips.rows_of_blocks(H, blocks, SR, chunk)
        assert (rows < H).sum() == sum(min(SR, H - b * SR) for b in blocks)
        ok = rows < H
        gathered[r][ok] = frame[rows[ok]]
    assert np.array_equal(strips.assemble_blocks(gathered, H, owner, SR), frame)
    # a rank's storage order: most expensive first, ties by index -- the C function and its Python mirror agree
    cost = np.array([5, 9, 9, 1, 7, 2, 2], np.uint64)
    for r in range(3):
        assert strips.blocks_of_rank(owner, r, cost).tolist() == evplp.rank_blocks(cost, owner, r).tolist()
        assert strips.blocks_of_rank(owner, r).tolist() == evplp.rank_blocks(None, owner, r).tolist()
    assert strips.blocks_of_rank(owner, 0, cost).tolist() == [1, 2, 5]
    # the round-robin deal through the same functions equals strips.global_rows
    for r in range(3):
        rr = strips.rows_of_blocks(H, np.arange(r, 7, 3), SR, strips.local_rows(H, 3, SR))
        g = strips.global_rows(H, r, 3, SR)
        assert np.array_equal(rr[rr < H], g[g < H])


def test_split_model_prefers_redundant_light_tracing_where_the_exchange_costs_more(evplp):
    """evplp_group_split_model (host only): config #4's 300 000 paths on four ranks and config #3's 500 000 on eight are traced in full by every
    rank (the record exchange would cost more than the shorter launch saves); a set of many millions is split."""
    for nl, n in ((1024, 8), (300000, 4), (500000, 8)):
        split, all_ms, shared_ms = evplp.split_model(nl, 4, n)
        assert not split and all_ms <= shared_ms
    split, all_ms, shared_ms = evplp.split_model(64_000_000, 4, 8)
    assert split and shared_ms < all_ms
    assert evplp.split_model(300000, 4, 1)[0] is False
