"""All four acceleration-structure builders (host LBVH, host SAH, host SAH with spatial splits, device LBVH) feed the same
walks: the results must not depend on the tree (visibility and closest hit are exact predicates; ties keep the lowest
original triangle index; the per-pixel summation order is a constant of the build)."""
import numpy as np
import pytest

import oracle_api as oa
import scenes

pytestmark = pytest.mark.gpu

W, H = 96, 64
NPATHS, P = 64, 4
BUILDERS = {"lbvh": 0, "sah": 1, "sbvh": 2, "gpu": 3}


@pytest.fixture(scope="module")
def room():
    return scenes.box_room(seed=11, n_boxes=7, tess=3, aspect=W / H)


def render(evplp, room, builder, mis_mode=1):
    with evplp.Context(W, H, NPATHS, NPATHS, P, bvh_builder=builder, deterministic=True) as c:
        room.upload(c)
        info = c.accel_info()
        c.clear_accumulators()
        c.primary((0.002, -0.001), clear_light=True)
        c.trace_light_paths(7)
        cam = c.camera()
        _, total, _ = c.scene_metrics()
        fp = evplp.frame_params(camera_pos=list(cam.origin), mis_mode=mis_mode, pdf_mc=0.4, clamping_value=1.0 / total, photon_radius=0.3,
                                num_light_paths=NPATHS, num_vpl_light_paths=NPATHS, photons_per_path=P, rng_seed=7)
        c.gather_vpl(fp)
        c.splat_photons(fp, clear=True)
        out = {
            "gbuf": [c.download(b)[:H].copy() for b in (evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL, evplp.BUF_GBUF_DIFFUSE, evplp.BUF_LIGHT)],
            "records": c.download(evplp.BUF_RECORDS).copy(),
            "vpl": c.download(evplp.BUF_VPL_ACCUM)[:H].copy(),
            "photon": c.download(evplp.BUF_PHOTON_ACCUM)[:H].copy(),
            "rays": c.pass_stats(evplp.PASS_GATHER_VPL)["rays"],
            "info": info,
        }
    return out


def test_every_builder_gives_the_same_frame(room, evplp):
    res = {name: render(evplp, room, b) for name, b in BUILDERS.items()}
    ref = res["sah"]
    assert ref["rays"] > 0 and np.isfinite(ref["vpl"]).all() and ref["vpl"][..., :3].max() > 0 and ref["photon"][..., :3].max() > 0
    for name, r in res.items():
        assert r["info"]["nodes"] >= 1 and r["info"]["leaves"] >= 1 and 1 <= r["info"]["depth"] < 62, (name, r["info"])
        for a, b in zip(r["gbuf"], ref["gbuf"]):
            assert np.array_equal(a, b), f"{name}: G-buffer differs from the SAH tree's"
        assert r["records"].tobytes() == ref["records"].tobytes(), f"{name}: light-path records differ"
        assert r["rays"] == ref["rays"], (name, r["rays"], ref["rays"])
        assert r["vpl"].tobytes() == ref["vpl"].tobytes(), f"{name}: VPL gather differs"
        assert r["photon"].tobytes() == ref["photon"].tobytes(), f"{name}: photon splat differs"


def test_device_lbvh_matches_the_oracle_visibility(room, evplp, oracle):
    """The device-built tree against the oracle (its own tree, its own walk): same G-buffer triangles, same lit-pixel counts."""
    osc = oa.Scene(room)
    jitter = (0.002, -0.001)
    ref = osc.primary(W, H, jitter)
    records = osc.trace_light_paths(7, NPATHS, P)
    kw = dict(camera_pos=osc.sd.cam_origin, mis_mode=0, num_light_paths=NPATHS, num_vpl_light_paths=NPATHS, photons_per_path=P)
    want, pairs = osc.gather(oa.frame_params(**kw), W, H, ref, records)
    with evplp.Context(W, H, NPATHS, NPATHS, P, bvh_builder=BUILDERS["gpu"]) as c:
        room.upload(c)
        c.primary(jitter, clear_light=True)
        got = [c.download(b)[:H] for b in (evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL)]
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        c.upload(evplp.BUF_RECORDS, records)
        c.clear_accumulators()
        c.gather_vpl(evplp.frame_params(**kw))
        img = c.download(evplp.BUF_VPL_ACCUM)[:H]
        assert c.pass_stats(evplp.PASS_GATHER_VPL)["pairs"] == pairs
    lit_got, lit_ref = (img[..., :3] > 0).any(axis=-1), (want[..., :3] > 0).any(axis=-1)
    assert np.array_equal(lit_got, lit_ref)
    err = np.abs(img[..., :3].astype(np.float64) - want[..., :3]).max()
    assert err <= 2e-4 * np.abs(want[..., :3]).max()


@pytest.mark.parametrize("ntri", [0, 1, 2, 4, 9])
def test_tiny_scenes_build_on_the_device(evplp, ntri):
    """ntri floor triangles + the light (one triangle when ntri = 0, else two): 1, 3 or 4 valid triangles = one leaf block under
    a wrapper root; 6 and 11 = the smallest real hierarchies.  A degenerate triangle (zero area) is dropped as meshBound does
    (rt/triangleintersect.cu:62-81)."""
    quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    verts, idx = [quad + np.array([0, 2.0, 0], np.float32)], [np.array([[0, 1, 2], [0, 2, 3]][:2 if ntri else 1], np.int32)]          # the light overhead
    # floor pieces: ntri triangles + one degenerate
    fl = []
    for k in range(ntri):
        x = -1.0 + 2.0 * k / ntri
        fl.append([[x, -1.0, -1.0], [x + 2.0 / ntri, -1.0, -1.0], [x + 1.0 / ntri, -1.0, 1.0]])
    fl.append([[0, -1, 0], [0, -1, 0], [0, -1, 0]])
    fl = np.array(fl, np.float32).reshape(-1, 3)
    results = {}
    for name in ("sah", "gpu"):
        with evplp.Context(32, 32, 16, 16, 4, bvh_builder=BUILDERS[name]) as c:
            m = c.add_material((0.6, 0.6, 0.6), (0.0, 0.0, 0.0), 1.0)
            light = c.add_mesh(verts[0], idx[0], m)
            c.add_mesh(fl, np.arange(len(fl), dtype=np.int32).reshape(-1, 3), m)
            c.set_arealight(light, (10.0, 10.0, 10.0, 0.0))
            c.set_camera((0.0, 0.5, 3.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.9, 1.0)
            c.build_accel()
            c.primary((0.0, 0.0), clear_light=True)
            c.trace_light_paths(1)
            results[name] = (c.download(evplp.BUF_GBUF_POSITION).tobytes(), c.download(evplp.BUF_GBUF_NORMAL).tobytes(), c.download(evplp.BUF_RECORDS).tobytes())
    assert results["gpu"] == results["sah"]
    assert np.frombuffer(results["sah"][0], np.float32).any() or ntri == 0, "the floor should be visible"
