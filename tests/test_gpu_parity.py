"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (stated here, used below):
  * geometric predicates (visibility, closest hit): bit-exact -> identical record flags, identical
    G-buffer triangle choice; a per-test mismatch budget covers fp ties only and is asserted to be 0
    where the arithmetic is exact by construction.
  * radiance (fp32 with powf/sqrt/div differences between glibc and ROCm's ocml): per-image relative
    L2 <= 1e-5 and per-pixel relative error <= 2e-4 (+1e-7 absolute).
"""
import math

import numpy as np
import pytest

import oracle_api as oa
import scenes

pytestmark = pytest.mark.gpu

REL_L2 = 1e-5
PIX_REL = 2e-4


def rel_l2(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))


def assert_image_close(a, b, rel=REL_L2, pix=PIX_REL, what=""):
    assert np.isfinite(a).all(), f"{what}: non-finite values in the HIP result"
    r = rel_l2(a, b)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tol = pix * np.maximum(np.abs(b), 1e-3 * scale) + 1e-7
    bad = int((err > tol).sum())
    assert r <= rel and bad == 0, f"{what}: rel L2 {r:.3e} (bar {rel}), {bad} pixels over the per-pixel bar, max err {err.max():.3e}"


W, H = 96, 64
NPATHS, P = 64, 4


@pytest.fixture(scope="module")
def room():
    return scenes.box_room(seed=3, n_boxes=5, tess=2, aspect=W / H)


@pytest.fixture(scope="module")
def oscene(room, oracle):
    return oa.Scene(room)


@pytest.fixture(scope="module")
def ctx(room, evplp):
    c = evplp.Context(W, H, NPATHS, NPATHS, P, deterministic=True)
    room.upload(c)
    yield c
    c.close()


@pytest.fixture(scope="module")
def inputs(oscene):
    gbuf = oscene.primary(W, H, (0.003, -0.002))
    records = oscene.trace_light_paths(5, NPATHS, P)
    return gbuf, records


def test_scene_metrics(ctx, oscene):
    r, total, light = ctx.scene_metrics()
    l = oscene.lib
    assert r == l.evo_scene_bounding_sphere_radius(oscene.h)
    assert light == l.evo_scene_light_area(oscene.h)
    assert abs(total - l.evo_scene_total_area(oscene.h)) <= 1e-4 * total   # per-mesh vs flat float summation


@pytest.mark.parametrize("builder", [0, 1])
def test_primary_gbuffer(room, oscene, evplp, builder):
    jitter = (0.003, -0.002)
    with evplp.Context(W, H, NPATHS, NPATHS, P, bvh_builder=builder) as c:
        room.upload(c)
        c.primary(jitter, clear_light=True)
        got = [c.download(b)[:H] for b in (evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL, evplp.BUF_GBUF_DIFFUSE, evplp.BUF_GBUF_PHONG, evplp.BUF_LIGHT)]
    ref = oscene.primary(W, H, jitter)
    # closest hit is exact: same triangle everywhere -> normals / materials identical
    assert np.array_equal(got[1], ref[1]), f"{int((got[1] != ref[1]).any(axis=-1).sum())} pixels picked another triangle"
    assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4])
    assert np.array_equal(got[0], ref[0]), float(np.abs(got[0] - ref[0]).max())   # positions too: unfused ray set-up + hit point
    assert (ref[4][..., 0] > 0).any(), "the light should be visible in this view"


def test_primary_with_a_jitter_larger_than_the_cuts_pyramid(room, oscene, evplp):
    """The eye's entry cuts hold for a jitter of up to one pixel; the ABI takes any float.  A translation of several pixels (here 2.4 and
    1.6 pixels) must still give the oracle's G-buffer, bit for bit: that call walks from the root (evplp_primary)."""
    jitter = (0.05, -0.05)
    assert abs(jitter[0]) > 2.0 / W and abs(jitter[1]) > 2.0 / H
    with evplp.Context(W, H, NPATHS, NPATHS, P) as c:
        room.upload(c)
        for j in (jitter, (0.003, -0.002), jitter):                          # (cuts built by the call in the middle must not leak into the third)
            c.primary(j)
            ref = oscene.primary(W, H, j)
            for b, want in zip((evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL, evplp.BUF_GBUF_DIFFUSE, evplp.BUF_GBUF_PHONG), ref):
                assert c.download(b)[:H].tobytes() == want.tobytes(), (j, b)
        with pytest.raises(evplp.EvplpError):
            c.primary((float("nan"), 0.0))


def test_exact_reciprocal_of_the_triangle_predicates(ctx):
    """The triangle predicates divide by n . d with a 5-instruction refinement of v_rcp_f32 instead of the compiler's 11-instruction
    IEEE division.  Checked on all 2^32 float bit patterns on this GPU: the bits differ only for zero / denormal / infinite inputs and
    for normal inputs of biased exponent >= 253 (|x| >= 2^126) -- inputs for which the predicate is false under either arithmetic
    (device_common.hpp rcp_exact)."""
    r = ctx.selftest(0)
    total, denorm, infnan, normal, lo, hi = (int(v) for v in r)
    assert denorm <= 2 ** 24 and infnan <= 2 ** 24 and total == denorm + infnan + normal
    assert normal == 0 or lo >= 253, (normal, lo, hi)


def test_hardware_power_function_of_the_phong_lobes(ctx):
    """The Phong lobes of the VPL gather and of the photon splat are exp2(e log2 d) on v_log_f32 / v_exp_f32 (the library powf is
    ~170 instructions).  Checked on the part against the double-precision pow: relative error <= 2e-6 for every exponent up to
    10 000 wherever the lobe is at least 1e-4 of its peak -- a fifth of the image bar (1e-5), a hundredth of the pixel bar (2e-4)."""
    err = ctx.selftest(1).astype(np.float64) * 1e-12
    assert len(err) == 6 and (err > 0).all() and err.max() <= 2e-6, err


def test_direction_sampling_functions_give_the_oracles_bits_on_the_device(ctx, oracle):
    """What the byte-equality of the light-path records does NOT prove by itself: the oracle #includes the product's csrc/ev_math.h, so the
    sampled directions' sin / cos / pow agree by construction IF that header computes the same bits under gcc on the host and under clang on
    the GPU (it uses only + - * / fma, rint and integer operations, contraction off).  Here that is evidenced on the part: evm_sincosf and
    evm_powf run on the device (evplp_debug_ev_math) over dense grids of the callers' input ranges -- phi = 2 pi u for 2^22 values of u in
    (0, 1] and the same number of random ones, negative and large arguments; x^y for u^(1 / (e + 1)) and cos^e with Phong exponents from 0
    to 10 000 -- and every result is compared bit for bit with the oracle's gcc build of the same header."""
    lib = oracle
    rng = np.random.default_rng(11)
    n = 1 << 22
    u = np.concatenate([(np.arange(1, n + 1, dtype=np.float64) / n).astype(np.float32), rng.random(n, dtype=np.float32)])
    phi = (np.float32(2.0) * np.float32(3.14159265358979323846)) * u
    xs = np.concatenate([phi, -phi[: n // 4], rng.uniform(-8000.0, 8000.0, n // 4).astype(np.float32), np.array([0.0, 1e-30, 8191.0], np.float32)])
    s_dev, c_dev = ctx.ev_math(0, xs)
    s_ref = np.empty_like(xs); c_ref = np.empty_like(xs)
    lib.evo_math_sincos_array(xs.ctypes.data, xs.size, s_ref.ctypes.data, c_ref.ctypes.data)
    assert s_dev.tobytes() == s_ref.tobytes() and c_dev.tobytes() == c_ref.tobytes(), (int((s_dev.view(np.uint32) != s_ref.view(np.uint32)).sum()), int((c_dev.view(np.uint32) != c_ref.view(np.uint32)).sum()))
    assert np.abs(s_dev[:n].astype(np.float64) - np.sin(phi[:n].astype(np.float64))).max() < 5e-7      # (and they ARE sines)
    es = np.array([0.0, 0.5, 1.0, 5.0, 20.0, 100.0, 1000.0, 10000.0], np.float32)
    m = 1 << 19
    xp, yp = [], []
    for e in es:
        x = np.concatenate([(np.arange(1, m + 1, dtype=np.float64) / m).astype(np.float32), rng.random(m, dtype=np.float32)])
        xp += [x, x]; yp += [np.full(x.size, np.float32(1.0) / (e + np.float32(1.0)), np.float32), np.full(x.size, e, np.float32)]   # sample_phong: u^(1 / (e + 1)); its pdf: cos^e
    xp = np.concatenate(xp + [np.array([0.0, 1.0, 0.25, 1e-38, 0.3], np.float32)]); yp = np.concatenate(yp + [np.array([2.0, 77.0, 0.5, 3.0, 0.0], np.float32)])
    p_dev = ctx.ev_math(1, xp, yp)
    p_ref = np.empty_like(xp)
    lib.evo_math_pow_array(xp.ctypes.data, yp.ctypes.data, xp.size, p_ref.ctypes.data)
    assert p_dev.tobytes() == p_ref.tobytes(), int((p_dev.view(np.uint32) != p_ref.view(np.uint32)).sum())
    ok = p_ref > 1e-30
    assert np.abs(p_dev[ok].astype(np.float64) / np.power(xp[ok].astype(np.float64), yp[ok].astype(np.float64)) - 1.0).max() < 2e-7


def test_light_image_flags(room, oscene, evplp):
    """rtcomphoton.h:985-995: run.lightRender = false leaves the light image alone; cleareveryframe clears the depth buffer the
    light pass shares with the deferred pass, so the emitter is drawn without a depth test."""
    with evplp.Context(W, H, NPATHS, NPATHS, P) as c:
        room.upload(c)
        c.clear_accumulators()
        c.primary((0.0, 0.0), light_skip=True)
        assert not c.download(evplp.BUF_LIGHT)[:H].any()
        c.primary((0.0, 0.0), clear_light=True)
        tested = c.download(evplp.BUF_LIGHT)[:H]
        gb = [c.download(b)[:H] for b in (evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL)]
        c.primary((0.0, 0.0), clear_light=True, light_unoccluded=True)
        free = c.download(evplp.BUF_LIGHT)[:H]
        gb2 = [c.download(b)[:H] for b in (evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL)]
    ref_t = oscene.primary(W, H, (0.0, 0.0))[4]; ref_f = oscene.primary(W, H, (0.0, 0.0), light_unoccluded=True)[4]
    assert np.array_equal(tested, ref_t) and np.array_equal(free, ref_f)
    assert (free[..., 0] > 0).sum() >= (tested[..., 0] > 0).sum() > 0
    assert all(np.array_equal(a, b) for a, b in zip(gb, gb2)), "the G-buffer stays depth-correct"


def test_light_tracing_records(ctx, oscene, evplp):
    ctx.trace_light_paths(5)
    got = ctx.download(evplp.BUF_RECORDS)
    ref = oscene.trace_light_paths(5, NPATHS, P)
    assert np.array_equal(got["flags"], ref["flags"]), "path structure differs (closest hit / RNG stream)"
    used = ref["flags"] != 0
    assert used.sum() > NPATHS
    # the feeders are compiled without contraction and sample directions with the shared ev_math.h: every field of every
    # record is the oracle's, bit for bit.  (The oracle #includes that header, so for sin / cos / pow of the sampled directions this
    # equality holds by construction -- it vouches for the walk and the draw order; the header's own bits on the device are checked by
    # test_direction_sampling_functions_give_the_oracles_bits_on_the_device below.)
    assert got.tobytes() == ref.tobytes()
    # a sliced trace (multi-GPU: each rank a range of paths) writes the same records
    ctx.upload(evplp.BUF_RECORDS, np.zeros_like(got))
    ctx.trace_light_paths(5, 0, NPATHS // 2)
    ctx.trace_light_paths(5, NPATHS // 2, NPATHS - NPATHS // 2)
    again = ctx.download(evplp.BUF_RECORDS)
    assert got.tobytes() == again.tobytes()


def upload_inputs(ctx, evplp, gbuf, records):
    for b, plane in zip((evplp.BUF_GBUF_POSITION, evplp.BUF_GBUF_NORMAL, evplp.BUF_GBUF_DIFFUSE, evplp.BUF_GBUF_PHONG), gbuf):
        pad = np.zeros((ctx.local_rows, W, 4), np.float32); pad[:H] = plane
        ctx.upload(b, pad)
    ctx.upload(evplp.BUF_RECORDS, records)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_gather_vpl_modes(ctx, oscene, evplp, inputs, mode):
    gbuf, records = inputs
    upload_inputs(ctx, evplp, gbuf, records)
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=mode, pdf_mc=0.35, clamping_value=0.02, photon_radius=0.05,
              num_light_paths=NPATHS, num_vpl_light_paths=NPATHS, photons_per_path=P, do_accumulate=0, rng_seed=5)
    ctx.clear_accumulators()
    ctx.gather_vpl(evplp.frame_params(**kw))
    got = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    ref, pairs = oscene.gather(oa.frame_params(**kw), W, H, gbuf, records)
    st = ctx.pass_stats(evplp.PASS_GATHER_VPL)
    assert st["pairs"] == pairs and st["usable"] == int((records["flags"] & 1).astype(bool).sum())
    # shadow rays traced and pairs left unoccluded: identical counts = identical cosine tests and visibility bits
    assert (st["rays"], st["shaded"]) == oscene.gather_counts(oa.frame_params(**kw), W, gbuf, records, np.arange(H))
    assert ref[..., :3].max() > 0
    assert_image_close(got[..., :3], ref[..., :3], what=f"gather_vpl mode {mode}")


@pytest.mark.parametrize("mode", [0, 1, 5])
def test_gather_vpl_is_bitwise_independent_of_item_size(room, oscene, evplp, inputs, mode):
    """The work decomposition is not part of the result: any k splits per wavefront must give the same bits (fixed summation
    tree), the same shadow-ray count and the same unoccluded pairs."""
    gbuf, records = inputs
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=mode, pdf_mc=0.35, clamping_value=0.02, photon_radius=0.05,
              num_light_paths=NPATHS, num_vpl_light_paths=NPATHS, photons_per_path=P, do_accumulate=0, rng_seed=5)
    outs = {}
    for k in (1, 2, 4, 16, 32, 0):
        with evplp.Context(W, H, NPATHS, NPATHS, P, gather_splits_per_wave=k) as c:
            room.upload(c)
            upload_inputs(c, evplp, gbuf, records)
            c.gather_vpl(evplp.frame_params(**kw))
            st = c.pass_stats(evplp.PASS_GATHER_VPL)
            outs[k] = (c.download(evplp.BUF_VPL_ACCUM)[:H].tobytes(), st["rays"], st["shaded"])
    first = outs[1]
    assert first[1] > 0 and first[2] > 0
    for key, val in outs.items():
        assert val == first, f"gather differs for {key} splits per wavefront"


def test_gather_vpl_accumulates(ctx, oscene, evplp, inputs):
    gbuf, records = inputs
    upload_inputs(ctx, evplp, gbuf, records)
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=0, num_light_paths=NPATHS, num_vpl_light_paths=NPATHS // 2,
              photons_per_path=P, do_accumulate=1)
    ctx.clear_accumulators()
    ctx.gather_vpl(evplp.frame_params(**kw)); ctx.gather_vpl(evplp.frame_params(**kw))
    got = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    ref, _ = oscene.gather(oa.frame_params(**kw), W, H, gbuf, records)
    ref, _ = oscene.gather(oa.frame_params(**kw), W, H, gbuf, records, out=ref)
    assert_image_close(got[..., :3], ref[..., :3], what="accumulate x2 with numVplLightPaths < numLightPaths")


def test_gather_lvc_window(ctx, oscene, evplp, inputs):
    """lvcphotonfam: per-pixel random window of light paths (rt/lvclighttracing.cu:348-384)."""
    gbuf, records = inputs
    upload_inputs(ctx, evplp, gbuf, records)
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=1, pdf_mc=0.35, photon_radius=0.05, num_light_paths=NPATHS,
              num_vpl_light_paths=NPATHS // 4, photons_per_path=P, do_accumulate=0, rng_seed=9)
    ctx.clear_accumulators()
    ctx.gather_lvc(evplp.frame_params(**kw))
    got = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    ref, pairs = oscene.gather(oa.frame_params(**kw), W, H, gbuf, records, lvc=True)
    st = ctx.pass_stats(evplp.PASS_GATHER_LVC)
    assert st["pairs"] == pairs > 0
    assert ref[..., :3].max() > 0
    assert_image_close(got[..., :3], ref[..., :3], what="gather_lvc")
    # another seed moves the windows
    ctx.gather_lvc(evplp.frame_params(**{**kw, "rng_seed": 10}))
    assert not np.array_equal(ctx.download(evplp.BUF_VPL_ACCUM)[:H], got)


def test_path_trace(ctx, oscene, evplp, inputs):
    """The "pt" technique's device pass against the oracle's restatement of rt/pathtracing.cu.  Paths make
    discrete choices (lobe, Russian roulette, visibility, the reference's 1e-5 self-intersection epsilon)
    on values that differ in the last ulps between CPU and GPU libm, so a few pixels per thousand take another
    path (measured: ~1.5 per thousand per pass); every other pixel agrees to fp32 round-off, and the image
    energy agrees."""
    gbuf, _ = inputs
    upload_inputs(ctx, evplp, gbuf, np.zeros(NPATHS * P, dtype=evplp.RECORD_DTYPE))
    cam = oscene.sd.cam_origin
    ctx.clear_accumulators()
    for seed in (3, 4):
        ctx.path_trace(cam, seed, 3, accumulate=True)
    got = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    ref, n = oscene.path_trace(cam, 3, 3, W, H, gbuf)
    ref, n2 = oscene.path_trace(cam, 4, 3, W, H, gbuf, out=ref)
    st = ctx.pass_stats(evplp.PASS_PATH_TRACE)
    assert st["pairs"] == n2 == int((gbuf[0][..., 3] != 0).sum()) and st["rays"] >= n2
    assert ref[..., :3].max() > 0
    g, r = got[..., :3], ref[..., :3]
    bad = (np.abs(g - r) > 2e-4 * np.maximum(np.abs(r), 1e-3 * r.max())).any(-1)
    assert bad.mean() <= 8e-3, f"{int(bad.sum())} of {bad.size} pixels took another path"
    ok = ~bad
    assert rel_l2(g[ok], r[ok]) <= 1e-5
    assert abs(float(g.sum()) / float(r.sum()) - 1.0) <= 5e-3
    # cleareveryframe: doAccumulate = 0 overwrites
    ctx.path_trace(cam, 3, 3, accumulate=False)
    single = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    ref1, _ = oscene.path_trace(cam, 3, 3, W, H, gbuf)
    bad1 = (np.abs(single[..., :3] - ref1[..., :3]) > 2e-4 * np.maximum(np.abs(ref1[..., :3]), 1e-3 * ref1[..., :3].max())).any(-1)
    assert bad1.mean() <= 4e-3


def test_visibility_is_bit_exact(ctx, oscene, evplp, inputs):
    """A 'flux = 1, white, mode one' gather differs between HIP and oracle only through visibility:
    count the pixels whose lit-VPL sets differ by comparing against per-VPL oracle visibility."""
    gbuf, records = inputs
    rec = records.copy()
    one = rec[:8].copy()
    upload_inputs(ctx, evplp, gbuf, rec)
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=0, num_light_paths=NPATHS, num_vpl_light_paths=2, photons_per_path=P)
    ctx.clear_accumulators()
    ctx.gather_vpl(evplp.frame_params(**kw))
    got = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    ref, _ = oscene.gather(oa.frame_params(**kw), W, H, gbuf, rec)
    # a visibility flip changes a pixel by a whole VPL contribution (>> 2e-4 relative)
    lit_ref = ref[..., :3].sum(-1) > 0; lit_got = got[..., :3].sum(-1) > 0
    assert np.array_equal(lit_ref, lit_got)
    assert_image_close(got[..., :3], ref[..., :3], what="two-path gather")
    assert one.shape[0] == 8


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_photon_splat_modes(ctx, oscene, evplp, inputs, mode):
    gbuf, records = inputs
    upload_inputs(ctx, evplp, gbuf, records)
    ctx.primary((0.003, -0.002))   # the splat needs the camera of the frame; G-buffer re-uploaded below
    upload_inputs(ctx, evplp, gbuf, records)
    radius = 0.35
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=mode, pdf_mc=NPATHS / NPATHS / math.pi / radius ** 2, clamping_value=0.02,
              photon_radius=radius, num_light_paths=NPATHS, num_vpl_light_paths=NPATHS, photons_per_path=P, jitter=(0.003, -0.002))
    ctx.splat_photons(evplp.frame_params(**kw), clear=True)
    got = ctx.download(evplp.BUF_PHOTON_ACCUM)[:H]
    ref, pairs = oa.splat(oa.frame_params(**kw), W, H, gbuf, records)
    st = ctx.pass_stats(evplp.PASS_SPLAT)
    assert pairs > 200 and st["pairs"] == pairs, (st["pairs"], pairs)
    assert_image_close(got[..., :3], ref[..., :3], what=f"splat mode {mode}")
    # additive blend: a second splat doubles the buffer (rtcomphoton.h:792-795)
    ctx.splat_photons(evplp.frame_params(**kw), clear=False)
    twice = ctx.download(evplp.BUF_PHOTON_ACCUM)[:H]
    assert np.allclose(twice[..., :3], 2 * got[..., :3], rtol=1e-6, atol=1e-9)


def _proxy_meshes():
    """Convex proxies beside the generated icosphere: a cube (coplanar triangle pairs, three slabs), a tetrahedron (no opposite
    faces: four open slabs), the hull of 40 random points (no symmetry at all), an icosphere squashed and pushed off centre."""
    from scipy.spatial import ConvexHull
    ico_v, ico_t = oa.icosphere42()
    c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * np.float32(0.55)
    cube_t = ConvexHull(c).simplices.astype(np.int32)
    tet = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], np.float32) * np.float32(0.5)
    tet_t = np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]], np.int32)
    rng = np.random.default_rng(7)
    pts = rng.normal(size=(40, 3)); pts = (pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.7, 1.0, (40, 1))).astype(np.float32)
    hull = ConvexHull(pts)
    keep = np.unique(hull.simplices); remap = -np.ones(40, np.int64); remap[keep] = np.arange(keep.size)
    squashed = (ico_v * np.array([1.0, 0.6, 0.8], np.float32) + np.array([0.1, 0.05, -0.15], np.float32)).astype(np.float32)
    return {"icosphere": (ico_v, ico_t), "cube": (c, cube_t), "tetrahedron": (tet, tet_t),
            "hull40": (pts[keep], remap[hull.simplices].astype(np.int32)), "squashed": (squashed, ico_t)}


def _splat_both_ways(ctx, evplp, oscene, gbuf, records, kw, mesh, cam=None):
    fp = evplp.frame_params(**kw, splat_footprint="proxy")
    ctx.splat_photons(fp, clear=True)
    got = ctx.download(evplp.BUF_PHOTON_ACCUM)[:H]
    st = ctx.pass_stats(evplp.PASS_SPLAT)
    ideal, proxy, ost = oa.splat_proxy(oa.frame_params(**kw), cam or oscene.camera(), W, H, gbuf, records, mesh=mesh)
    return got, st, ideal, proxy, ost


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_photon_splat_proxy_footprint_modes(ctx, oscene, evplp, inputs, mode):
    """EVPLP_FOOTPRINT_PROXY: the reference's coverage rule -- one fragment per face of the radius-scaled proxy mesh that the pixel's eye
    ray crosses in front of the visible surface, un-culled, depth test LEQUAL (rtcomphoton.h:653-655, 789-837;
    photonsplatinstanced.vert:28-33, .geom:16-32) -- against the oracle's ray / triangle count over the same mesh
    (evo_splat_photons_proxy_mesh): equal pair and fragment counts, images to the splat's bar."""
    gbuf, records = inputs
    ctx.primary((0.003, -0.002)); upload_inputs(ctx, evplp, gbuf, records)
    ctx.set_splat_proxy()                                                   # the generated icosphere
    radius = 0.35
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=mode, pdf_mc=NPATHS / NPATHS / math.pi / radius ** 2, clamping_value=0.02,
              photon_radius=radius, num_light_paths=NPATHS, num_vpl_light_paths=NPATHS, photons_per_path=P, jitter=(0.003, -0.002))
    got, st, ideal, proxy, ost = _splat_both_ways(ctx, evplp, oscene, gbuf, records, kw, oa.icosphere42())
    assert int(ost[0]) > 200 and st["pairs"] == int(ost[0]) and int(ost[1]) > 0 and int(ost[2]) > 0
    # a pair whose eye ray meets an edge of the proxy to within rounding may fall either way: none here
    assert st["rays"] == int(ost[3]), (st["rays"], ost)
    assert_image_close(got[..., :3], proxy[..., :3], what=f"proxy splat mode {mode}")
    assert mode == 2 or rel_l2(proxy[..., :3], ideal[..., :3]) > 1e-3      # ... and the two rules do differ (the max heuristic keeps only pairs deep inside their proxies here)
    # the ideal rule is untouched by a proxy being set
    ctx.splat_photons(evplp.frame_params(**kw), clear=True)
    assert_image_close(ctx.download(evplp.BUF_PHOTON_ACCUM)[:H][..., :3], ideal[..., :3], what=f"ideal splat mode {mode}")


@pytest.mark.parametrize("name", ["cube", "tetrahedron", "hull40", "squashed"])
def test_photon_splat_proxy_meshes(ctx, oscene, evplp, inputs, name):
    """evplp_set_splat_proxy with meshes that are not the icosphere: coplanar faces, faces without an opposite one, no symmetry, the
    origin off centre -- and radii that bring the near plane into the proxies (every pair takes the full test)."""
    gbuf, records = inputs
    ctx.primary((0.003, -0.002)); upload_inputs(ctx, evplp, gbuf, records)
    mesh = _proxy_meshes()[name]
    ctx.set_splat_proxy(*mesh)
    try:
        for radius in (0.3, 1.5):
            kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=1, pdf_mc=1.0 / math.pi / radius ** 2, photon_radius=radius, num_light_paths=NPATHS,
                      num_vpl_light_paths=NPATHS, photons_per_path=P, jitter=(0.003, -0.002))
            got, st, ideal, proxy, ost = _splat_both_ways(ctx, evplp, oscene, gbuf, records, kw, mesh)
            assert st["pairs"] == int(ost[0]) and int(ost[0]) > 200
            assert abs(st["rays"] - int(ost[3])) <= max(1, int(ost[3]) // 50000), (name, radius, st["rays"], ost)
            if st["rays"] == int(ost[3]):
                assert_image_close(got[..., :3], proxy[..., :3], what=f"proxy {name} r={radius}")
            else:
                assert rel_l2(got[..., :3], proxy[..., :3]) <= 1e-3
    finally:
        ctx.set_splat_proxy()


def test_photon_splat_proxy_one_wave_per_tile(room, oscene, evplp, inputs):
    """A context that is not `deterministic` takes the tile kernel with one wave per tile while its bins are short (the module's context
    always takes the four-wave variant): the same fragments, the same image up to the order of the sums."""
    gbuf, records = inputs
    radius = 0.35
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=1, pdf_mc=1.0 / math.pi / radius ** 2, photon_radius=radius, num_light_paths=NPATHS,
              num_vpl_light_paths=NPATHS, photons_per_path=P, jitter=(0.003, -0.002))
    with evplp.Context(W, H, NPATHS, NPATHS, P) as c:
        room.upload(c)
        c.primary((0.003, -0.002)); upload_inputs(c, evplp, gbuf, records)
        for _ in range(2):
            got, st, ideal, proxy, ost = _splat_both_ways(c, evplp, oscene, gbuf, records, kw, None)
            assert st["pairs"] == int(ost[0]) and st["rays"] == int(ost[3])
            assert_image_close(got[..., :3], proxy[..., :3], what="proxy splat, one wave per tile")


def test_photon_splat_mixed_tile_launch(room, oscene, evplp, inputs, monkeypatch):
    """The MIXED launch of the tile kernel (tiles whose bins hold many entries get a workgroup of four waves, the others one wave; chosen
    when the previous pass had full bins): forced here with a heavy threshold of 4 entries, so that this small frame has tiles of both
    kinds -- same pairs and fragments as the oracle, the same image up to the order of the sums; and again with a heavy list of two
    entries, which overflows (the tiles that do not fit stay one-wave tiles)."""
    gbuf, records = inputs
    radius = 0.35
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=1, pdf_mc=1.0 / math.pi / radius ** 2, photon_radius=radius, num_light_paths=NPATHS,
              num_vpl_light_paths=NPATHS, photons_per_path=P, jitter=(0.003, -0.002))
    monkeypatch.setenv("EVPLP_TILE_HEAVY", "4"); monkeypatch.setenv("EVPLP_TILE_MIXED_TRIGGER", "0")
    for cap in ("2048", "2"):
        monkeypatch.setenv("EVPLP_TILE_HEAVY_CAP", cap)
        with evplp.Context(W, H, NPATHS, NPATHS, P) as c:
            room.upload(c)
            c.primary((0.003, -0.002)); upload_inputs(c, evplp, gbuf, records)
            for _ in range(2):
                got, st, ideal, proxy, ost = _splat_both_ways(c, evplp, oscene, gbuf, records, kw, None)
                assert st["nodes"] >> 32 >= 4 and (st["nodes"] & 0xffffffff) > 50, "no bin reaches the heavy threshold: the test does not exercise the four-wave workgroups"
                assert st["pairs"] == int(ost[0]) and st["rays"] == int(ost[3])
                assert_image_close(got[..., :3], proxy[..., :3], what="proxy splat, mixed launch")
                c.splat_photons(evplp.frame_params(**kw), clear=True)                  # ... and the plain radius rule through the same launch
                st_i = c.pass_stats(evplp.PASS_SPLAT)
                assert st_i["pairs"] == int(ost[0])
                assert_image_close(c.download(evplp.BUF_PHOTON_ACCUM)[:H][..., :3], ideal[..., :3], what="ideal splat, mixed launch")


def test_splat_proxy_mesh_is_validated(ctx, evplp):
    """A mesh whose fragment count is not the entry / exit rule of a convex body is refused, with the reason."""
    v, t = oa.icosphere42()
    dent = v.copy(); dent[20] *= 0.5                                        # a vertex pulled inside: not convex
    far = v + np.float32(2.0)                                               # the photon outside its own proxy
    big_v, big_t = None, None
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(1); pts = rng.normal(size=(120, 3)); pts = (pts / np.linalg.norm(pts, axis=1, keepdims=True)).astype(np.float32)
    big_t = ConvexHull(pts).simplices.astype(np.int32)                      # 236 faces, no two parallel
    for verts, tris, why in ((dent, t, "convex"), (v, t[:-1], "closed"), (far, t, "origin"), (pts, big_t, "planes"), (v[:3], t[:1], "4 vertices")):
        with pytest.raises(evplp.EvplpError) as e:
            ctx.set_splat_proxy(verts, tris)
        assert e.value.status == evplp.ERR_INVALID and why in str(e.value), (why, str(e.value))
    dv, dt = evplp.default_splat_proxy()                                    # the product's generated mesh is the oracle's
    assert np.array_equal(dt, t) and np.abs(dv - v).max() <= 1e-7
    ctx.set_splat_proxy(dv, dt)
    ctx.set_splat_proxy()


def test_photon_bins_grow_and_the_pass_reruns(room, oscene, evplp, inputs, monkeypatch):
    """The splat is enqueued without knowing the bin sizes; a pass whose bins overflow writes nothing and is run again by the
    next call with larger bins (context.cpp settle_splat).  Force that path with two-slot bins."""
    gbuf, records = inputs
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=1, pdf_mc=0.35, photon_radius=0.35, num_light_paths=NPATHS,
              num_vpl_light_paths=NPATHS, photons_per_path=P)
    outs = []
    for cap in (None, "2"):
        if cap:
            monkeypatch.setenv("EVPLP_BIN_STRIDE", cap)
        with evplp.Context(W, H, NPATHS, NPATHS, P, deterministic=True) as c:
            room.upload(c)
            upload_inputs(c, evplp, gbuf, records)
            c.splat_photons(evplp.frame_params(**kw), clear=True)
            c.splat_photons(evplp.frame_params(**kw), clear=False)         # accumulates on top: a dropped or doubled pass shows
            st = c.pass_stats(evplp.PASS_SPLAT)
            outs.append((c.download(evplp.BUF_PHOTON_ACCUM)[:H].tobytes(), st["pairs"]))
        monkeypatch.delenv("EVPLP_BIN_STRIDE", raising=False)
    assert outs[0][1] > 1000 and outs[0] == outs[1]


def test_gather_vsl(ctx, oscene, evplp, inputs):
    gbuf, records = inputs
    upload_inputs(ctx, evplp, gbuf, records)
    r = 0.3
    kw = dict(camera_pos=oscene.sd.cam_origin, vsl_radius=r, vsl_inv_pi_radius2=1.0 / (math.pi * r * r), num_light_paths=NPATHS,
              num_vpl_light_paths=16, photons_per_path=P, rng_seed=9)
    ctx.clear_accumulators()
    ctx.gather_vsl(evplp.frame_params(**kw))
    got = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    ref, pairs = oscene.gather(oa.frame_params(**kw), W, H, gbuf, records, vsl=True)
    assert ctx.pass_stats(evplp.PASS_GATHER_VSL)["pairs"] == pairs
    assert ref[..., :3].max() > 0
    # the estimators branch on sampled directions against cone / hemisphere thresholds: a 1-ulp
    # difference in sinf/cosf/powf can flip one of up to 303 sample terms of a pair, so the bar is
    # statistical per pixel (2%) and tight in aggregate (rel L2 1e-3)
    assert_image_close(got[..., :3], ref[..., :3], rel=1e-3, pix=2e-2, what="gather_vsl")


def test_gather_vsl_with_a_light_exactly_on_a_pixel(ctx, oscene, evplp, inputs):
    """A VSL that sits exactly ON a G-buffer point: distance 0, the cone set-up divides by it.  The reference drops the pair
    (fmaxf(NaN, 0) * fmaxf(NaN, 0) = 0 <= 1e-9, lighttracing.cu:619); the walk kernel's pre-test must drop it too instead of
    letting a NaN through to the estimators (rsq(0) = inf would stay in the accumulating image for good)."""
    gbuf, records = inputs
    rec = records.copy()
    y0, x0 = H // 2, W // 3
    assert gbuf[0][y0, x0, 3] != 0
    rec["pos"][1] = gbuf[0][y0, x0, :3]; rec["normal"][1] = gbuf[1][y0, x0, :3]; rec["flags"][1] |= 1
    upload_inputs(ctx, evplp, gbuf, rec)
    r = 0.3
    kw = dict(camera_pos=oscene.sd.cam_origin, vsl_radius=r, vsl_inv_pi_radius2=1.0 / (math.pi * r * r), num_light_paths=NPATHS,
              num_vpl_light_paths=4, photons_per_path=P, rng_seed=9)
    ctx.clear_accumulators()
    ctx.gather_vsl(evplp.frame_params(**kw))
    got = ctx.download(evplp.BUF_VPL_ACCUM)[:H]
    assert np.isfinite(got).all(), "a zero-distance (pixel, VSL) pair put a non-finite value into the image"
    ref, _ = oscene.gather(oa.frame_params(**kw), W, H, gbuf, rec, vsl=True)
    assert np.isfinite(ref).all() and ref[..., :3].max() > 0
    assert_image_close(got[..., :3], ref[..., :3], rel=1e-3, pix=2e-2, what="gather_vsl with a VSL on a pixel")


def test_row_strips_reassemble_bitwise(room, evplp, oscene, inputs):
    """N interleaved strips (as N ranks would own them) == the 1-GPU frame, bit for bit (gather) and
    to fp32 reorder tolerance 0 in deterministic splat mode."""
    gbuf, records = inputs
    kw = dict(camera_pos=oscene.sd.cam_origin, mis_mode=1, pdf_mc=0.35, photon_radius=0.35, num_light_paths=NPATHS,
              num_vpl_light_paths=NPATHS, photons_per_path=P, jitter=(0.003, -0.002))
    frames = {}
    for count in (1, 2, 4):
        vpl = np.zeros((H, W, 4), np.float32); pm = np.zeros((H, W, 4), np.float32)
        lvc = np.zeros((H, W, 4), np.float32); pt = np.zeros((H, W, 4), np.float32); vsl = np.zeros((H, W, 4), np.float32)
        for rank in range(count):
            with evplp.Context(W, H, NPATHS, NPATHS, P, strip_rank=rank, strip_count=count, strip_rows=8, deterministic=True) as c:
                room.upload(c)
                c.primary((0.003, -0.002), clear_light=True)
                c.trace_light_paths(5)
                c.gather_vpl(evplp.frame_params(**kw))
                c.splat_photons(evplp.frame_params(**kw), clear=True)
                rows = c.global_rows()
                ok = rows < H
                vpl[rows[ok]] = c.download(evplp.BUF_VPL_ACCUM)[ok]
                pm[rows[ok]] = c.download(evplp.BUF_PHOTON_ACCUM)[ok]
                # the per-pixel RNG streams of the other techniques are keyed by the GLOBAL pixel id
                c.gather_lvc(evplp.frame_params(**{**kw, "num_vpl_light_paths": NPATHS // 4, "rng_seed": 3}))
                lvc[rows[ok]] = c.download(evplp.BUF_VPL_ACCUM)[ok]
                c.gather_vsl(evplp.frame_params(**{**kw, "num_vpl_light_paths": 8, "vsl_radius": 0.4, "vsl_inv_pi_radius2": 1 / (math.pi * 0.16), "rng_seed": 4}))
                vsl[rows[ok]] = c.download(evplp.BUF_VPL_ACCUM)[ok]
                c.path_trace(oscene.sd.cam_origin, 6, 3, accumulate=False)
                pt[rows[ok]] = c.download(evplp.BUF_VPL_ACCUM)[ok]
        frames[count] = (vpl, pm, lvc, vsl, pt)
    for count in (2, 4):
        for k, name in enumerate(("gather", "splat", "lvc gather", "vsl gather", "path tracer")):
            assert frames[count][k].tobytes() == frames[1][k].tobytes(), f"{name} differs with {count} strips"
    assert all(f[..., :3].max() > 0 for f in frames[1])


def test_resolve_composite(ctx, evplp, oracle):
    rng = np.random.RandomState(0)
    planes = [rng.rand(ctx.local_rows, W, 4).astype(np.float32) for _ in range(3)]
    planes[2][..., :] *= (rng.rand(ctx.local_rows, W, 1) > 0.8)   # sparse emitter mask
    for b, p in zip((evplp.BUF_VPL_ACCUM, evplp.BUF_PHOTON_ACCUM, evplp.BUF_LIGHT), planes):
        ctx.upload(b, p)
    for mask, gamma in ((0, 0), (1, 0), (1, 1)):
        got = ctx.resolve(0.5, 0.25, 1.0, mask_emitter=bool(mask), gamma=bool(gamma))
        ref = np.zeros((ctx.local_rows, W, 3), np.float32)
        oracle.evo_resolve(W, ctx.local_rows, oa.ptr(planes[0]), oa.ptr(planes[1]), oa.ptr(planes[2]), 0.5, 0.25, 1.0, mask, gamma, oa.ptr(ref))
        assert np.allclose(got, ref, rtol=4e-6, atol=1e-7), (mask, gamma)


def test_tile_boxes_from_the_primary_pass_equal_the_fallback_kernel(room, evplp):
    """evplp_primary writes the per-tile position boxes the photon splat culls with; a G-buffer that came in another way
    (evplp_upload, a bound buffer, a pointer handed out by evplp_buffer_info) has them rebuilt by splat_tile_box_kernel.
    Same bins, same image."""
    kw = dict(camera_pos=room.cam_origin, mis_mode=1, pdf_mc=0.35, photon_radius=0.3, num_light_paths=NPATHS, num_vpl_light_paths=NPATHS, photons_per_path=P)
    outs = []
    for reupload in (False, True):
        with evplp.Context(W, H, NPATHS, NPATHS, P, deterministic=True) as c:
            room.upload(c)
            c.primary((0.001, 0.002), clear_light=True)
            c.trace_light_paths(3)
            if reupload:
                c.upload(evplp.BUF_GBUF_POSITION, c.download(evplp.BUF_GBUF_POSITION))
            c.splat_photons(evplp.frame_params(**kw), clear=True)
            st = c.pass_stats(evplp.PASS_SPLAT)
            outs.append((c.download(evplp.BUF_PHOTON_ACCUM)[:H].tobytes(), st["pairs"]))
    assert outs[0][1] > 500 and outs[0] == outs[1]


def test_overlapped_light_tracing_changes_no_bit(room, evplp):
    """evplp_config.overlap_light_tracing: light tracing on a second stream beside the G-buffer pass, ordered behind the last
    reader of the records and in front of the next one.  Several iterations of the technique loop, with and without."""
    outs = []
    for overlap in (False, True):
        with evplp.Context(W, H, NPATHS, NPATHS, P, deterministic=True, overlap_light_tracing=overlap) as c:
            room.upload(c)
            c.clear_accumulators()
            recs = []
            for it in range(4):
                kw = dict(camera_pos=room.cam_origin, mis_mode=1, pdf_mc=0.35, photon_radius=0.3, num_light_paths=NPATHS, num_vpl_light_paths=NPATHS,
                          photons_per_path=P, do_accumulate=1, rng_seed=it)
                c.primary((0.001 * it, -0.002), clear_light=True)
                c.trace_light_paths(10 + it)
                c.gather_vpl(evplp.frame_params(**kw))
                c.splat_photons(evplp.frame_params(**kw))
                if it == 2:
                    recs.append(c.download(evplp.BUF_RECORDS).tobytes())
            st = c.pass_stats(evplp.PASS_LIGHT_TRACE)
            assert st["ms"] > 0
            outs.append((c.download(evplp.BUF_VPL_ACCUM)[:H].tobytes(), c.download(evplp.BUF_PHOTON_ACCUM)[:H].tobytes(), recs[0]))
    assert outs[0] == outs[1]


def test_pass_events_can_be_switched_off(room, evplp):
    """evplp_profile_passes(ctx, 0): the two HIP events per pass are not recorded (a loop of sub-millisecond iterations saves the command
    processor's time between its dispatches); the passes do the same work -- every accumulator bit, every counter -- and their event
    time reads 0 until the next pass recorded with events."""
    outs = []
    for profiled in (True, False):
        with evplp.Context(W, H, NPATHS, NPATHS, P, deterministic=True, overlap_light_tracing=True) as c:
            room.upload(c)
            c.clear_accumulators()
            c.profile_passes(profiled)
            for it in range(3):
                kw = dict(camera_pos=room.cam_origin, mis_mode=1, pdf_mc=0.35, photon_radius=0.3, num_light_paths=NPATHS, num_vpl_light_paths=NPATHS,
                          photons_per_path=P, do_accumulate=1, rng_seed=it)
                c.trace_light_paths(10 + it)
                c.primary((0.001 * it, -0.002), clear_light=True)
                c.gather_vpl(evplp.frame_params(**kw))
                c.splat_photons(evplp.frame_params(**kw))
                c.present(1.0 / (it + 1), 1.0 / (it + 1), 1.0)
            st = {p: c.pass_stats(p) for p in (evplp.PASS_PRIMARY, evplp.PASS_LIGHT_TRACE, evplp.PASS_GATHER_VPL, evplp.PASS_SPLAT)}
            for p, v in st.items():
                assert (v["ms"] > 0) == profiled, (p, profiled, v["ms"])
            outs.append((c.download(evplp.BUF_VPL_ACCUM)[:H].tobytes(), c.download(evplp.BUF_PHOTON_ACCUM)[:H].tobytes(),
                         st[evplp.PASS_GATHER_VPL]["rays"], st[evplp.PASS_SPLAT]["pairs"], c.resolve(1.0, 1.0, 1.0).tobytes()))
            if not profiled:                      # ... and on again
                c.profile_passes(True)
                c.primary((0.0, 0.0))
                assert c.pass_stats(evplp.PASS_PRIMARY)["ms"] > 0
    assert outs[0][2] > 0 and outs[0][3] > 0 and outs[0] == outs[1]


def test_bins_grow_with_two_passes_in_flight(room, evplp, monkeypatch):
    """overlap_light_tracing keeps up to two photon splats pending.  With two-slot bins every pass overflows and runs again.  In
    deterministic mode a pass never stays pending behind a younger one, so the re-runs keep their place in the stream and the
    accumulated image is the same bit for bit (run twice: reproducible; against the serial context with roomy bins: identical);
    without deterministic mode the older pass may run again after the younger one and only fp32 round-off may differ."""
    imgs = []
    for cap, overlap, det in ((None, False, True), ("2", True, True), ("2", True, True), ("2", True, False)):
        if cap:
            monkeypatch.setenv("EVPLP_BIN_STRIDE", cap)
        with evplp.Context(W, H, NPATHS, NPATHS, P, deterministic=det, overlap_light_tracing=overlap) as c:
            room.upload(c)
            c.clear_accumulators()
            for it in range(5):
                kw = dict(camera_pos=room.cam_origin, mis_mode=1, pdf_mc=0.35, photon_radius=0.3 - 0.02 * it, num_light_paths=NPATHS,
                          num_vpl_light_paths=NPATHS, photons_per_path=P, do_accumulate=1, rng_seed=it)
                c.trace_light_paths(20 + it)
                c.primary((0.001 * it, 0.0), clear_light=True)
                c.splat_photons(evplp.frame_params(**kw))
            imgs.append(c.download(evplp.BUF_PHOTON_ACCUM)[:H].astype(np.float64))
        monkeypatch.delenv("EVPLP_BIN_STRIDE", raising=False)
    assert imgs[0].max() > 0
    assert (imgs[0] == imgs[1]).all() and (imgs[1] == imgs[2]).all()
    assert np.abs(imgs[0] - imgs[3]).max() <= 1e-5 * imgs[0].max()


def test_entry_cuts_do_not_change_a_bit(evplp, monkeypatch):
    """The entry cuts (kernels_cut.hip: a frustum per (tile group, VPL) descends the tree to a cut, the packet walks start there) only
    skip subtrees no segment of the group can reach: the VPL and the VSL gather must give the same bits with the cuts, without them
    (EVPLP_CUTS=0: every walk from the root) and with the cut scratch so small that the frame is gathered in several bands of tile
    blocks (three at this size)."""
    w, h, n = 96, 192, 96
    tall = scenes.box_room(seed=3, n_boxes=5, tess=2, aspect=w / h)
    kw = dict(camera_pos=tall.cam_origin, mis_mode=1, pdf_mc=0.35, clamping_value=0.02, photon_radius=0.05, vsl_radius=0.3,
              vsl_inv_pi_radius2=1.0 / (math.pi * 0.09), num_light_paths=n, num_vpl_light_paths=n, photons_per_path=P, do_accumulate=0, rng_seed=5)
    images = {}
    for name, env in (("cuts", {}), ("root", {"EVPLP_CUTS": "0"}), ("bands", {"EVPLP_CUT_BYTES": "1000000"})):
        for k in ("EVPLP_CUTS", "EVPLP_CUT_BYTES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with evplp.Context(w, h, n, n, P) as c:
            tall.upload(c)
            c.primary((0.003, -0.002)); c.trace_light_paths(5)
            c.gather_vpl(evplp.frame_params(**kw))
            vpl = c.download(evplp.BUF_VPL_ACCUM)[:h].copy()
            st = c.pass_stats(evplp.PASS_GATHER_VPL)
            c.gather_vsl(evplp.frame_params(**kw))
            vsl = c.download(evplp.BUF_VPL_ACCUM)[:h].copy()
            images[name] = (vpl, vsl, st["rays"], st["shaded"])
    assert images["cuts"][0].max() > 0 and images["cuts"][1].max() > 0 and images["cuts"][2] > 0
    for name in ("root", "bands"):
        assert images[name][2:] == images["cuts"][2:], (name, images[name][2:], images["cuts"][2:])
        assert images[name][0].tobytes() == images["cuts"][0].tobytes(), f"VPL gather: {name} differs from cuts"
        assert images[name][1].tobytes() == images["cuts"][1].tobytes(), f"VSL gather: {name} differs from cuts"


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_entry_cuts_with_adversarial_vpl_positions(evplp, monkeypatch, seed):
    """Entry cuts against walks from the root on record sets the light tracer would never produce: VPLs sitting exactly on visible surface
    points (the apex of the cut's pyramid lies ON its tile group: no pyramid, the end-point box alone), at the eye, at the centre of the
    room, far outside it, and with flipped normals.  Same bits, same shadow-ray and unoccluded-pair counts."""
    w, h, n = 96, 96, 64
    room = scenes.box_room(seed=seed, n_boxes=6, tess=2, aspect=1.0)
    kw = dict(camera_pos=room.cam_origin, mis_mode=0, pdf_mc=0.35, clamping_value=0.02, photon_radius=0.05, vsl_radius=0.3,
              vsl_inv_pi_radius2=1.0 / (math.pi * 0.09), num_light_paths=n, num_vpl_light_paths=n, photons_per_path=P, do_accumulate=0, rng_seed=seed)
    out = {}
    for name, env in (("cuts", {}), ("root", {"EVPLP_CUTS": "0"})):
        monkeypatch.delenv("EVPLP_CUTS", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with evplp.Context(w, h, n, n, P) as c:
            room.upload(c)
            c.primary((0.0, 0.0)); c.trace_light_paths(seed)
            gpos = c.download(evplp.BUF_GBUF_POSITION)[:h].reshape(-1, 4)
            rec = c.download(evplp.BUF_RECORDS).copy()
            rng = np.random.RandomState(seed)
            usable = np.nonzero(rec["flags"] & evplp.USABLE_VPL)[0]
            vis = np.nonzero(gpos[:, 3] != 0)[0]
            lo, hi = gpos[vis, :3].min(0), gpos[vis, :3].max(0)
            pick = rng.permutation(usable)
            for j, i in enumerate(pick[:40]):
                kind = j % 5
                if kind == 0: rec["pos"][i] = gpos[rng.choice(vis), :3]                    # exactly on a visible surface point
                elif kind == 1: rec["pos"][i] = np.asarray(room.cam_origin, np.float32)      # at the eye
                elif kind == 2: rec["pos"][i] = (0.5 * (lo + hi)).astype(np.float32)         # centre of what the camera sees
                elif kind == 3: rec["pos"][i] = (hi + 1000.0 * (hi - lo)).astype(np.float32)  # far outside
                else: rec["normal"][i] = -rec["normal"][i]                                  # facing into its own surface
            c.upload(evplp.BUF_RECORDS, rec)
            c.gather_vpl(evplp.frame_params(**kw))
            img = c.download(evplp.BUF_VPL_ACCUM)[:h].copy()
            st = c.pass_stats(evplp.PASS_GATHER_VPL)
            out[name] = (img, st["rays"], st["shaded"])
    assert out["cuts"][1] > 0 and np.isfinite(out["cuts"][0]).all()
    assert out["root"][1:] == out["cuts"][1:], (out["root"][1:], out["cuts"][1:])
    assert out["root"][0].tobytes() == out["cuts"][0].tobytes()
