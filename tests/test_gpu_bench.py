"""bench.py on the GPU box: the JSON contract, the collective code path on one rank, strip assembly through the product."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env=None):
    e = dict(os.environ); e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, f"bench.py must print ONE line on stdout, got {len(lines)}: {lines[0][:80]!r} ..."
    return json.loads(lines[0])


def test_bench_line_contract_small():
    d = run_bench(["--steps", "2", "--warmup", "1", "--res", "256", "--tris", "20000", "--no-cpu-baseline", "--no-extras"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "valu" and 0 < r["frac"] < 1 and r["kernel"] == "gather_vpl_kernel" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["config"]["shadow_rays_per_frame"] <= d["config"]["pairs_nominal_per_frame"]


def test_bench_collective_path_on_one_rank():
    """EVPLP_BENCH_FORCE_DIST=1: init_process_group("nccl") with one rank, the record all-gather (split light tracing) and the
    framebuffer all-gather run through RCCL exactly as with N ranks; same value definition, n_gpus from the process group."""
    a = run_bench(["--steps", "2", "--warmup", "1", "--res", "256", "--tris", "20000", "--no-cpu-baseline", "--no-extras"], {"EVPLP_BENCH_FORCE_DIST": "1"})
    b = run_bench(["--steps", "2", "--warmup", "1", "--res", "256", "--tris", "20000", "--no-cpu-baseline", "--no-extras"])
    assert a["n_gpus"] == 1 and a["config"]["shadow_rays_per_frame"] == b["config"]["shadow_rays_per_frame"]
    assert a["config"]["usable_vpl_records"] == b["config"]["usable_vpl_records"]


@pytest.mark.parametrize("wl", ["evplp", "ppm"])
def test_bench_photon_workloads_through_the_collective_path(wl):
    """The photon workloads with EVPLP_BENCH_FORCE_DIST=1: light paths traced by path range + in-place record all-gather, both
    framebuffers all-gathered -- the same photon-pixel pairs as without collectives."""
    args = ["--workload", wl, "--steps", "2", "--warmup", "1", "--tris", "20000", "--no-cpu-baseline", "--no-extras"] + (["--res", "256"] if wl == "evplp" else [])
    a = run_bench(args, {"EVPLP_BENCH_FORCE_DIST": "1"})
    b = run_bench(args)
    ra, rb = (a.get("roofline_splat") or a["roofline"]), (b.get("roofline_splat") or b["roofline"])
    assert a["n_gpus"] == 1 and ra["pairs_per_frame"] > 0 and ra["pairs_per_frame"] == rb["pairs_per_frame"]
    assert ra["bound"] == "hbm" and 0 < ra["frac"] < 1


def test_bench_refuses_a_rank_count_it_cannot_start():
    """--gpus 2 on a one-GPU box: bench.py starts two ranks itself; the second has no device and the parent reports failure
    instead of silently measuring one GPU."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--res", "128", "--tris", "5000",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=180)
    assert p.returncode != 0


def test_two_strip_contexts_assemble_to_the_single_frame(evplp, tmp_path):
    from evplp_amd import strips
    W, H, N, P = 96, 64, 48, 4
    jp = evplp.synth_scene(str(tmp_path), "room", 3000, 9, W, H, style="hard")
    fp = evplp.frame_params(camera_pos=(15.56, -4.79, 4.37), mis_mode="one", num_light_paths=N, num_vpl_light_paths=N, photons_per_path=P)

    def render(rank, count):
        with evplp.Context(W, H, N, N, P, strip_rank=rank, strip_count=count, strip_rows=8) as c:
            c.load_scene_json(jp)
            c.primary(); c.trace_light_paths(2); c.gather_vpl(fp)
            return c.download(evplp.BUF_VPL_ACCUM)
    gathered = np.stack([render(r, 2) for r in range(2)])            # what an all-gather of the two strips delivers
    full = strips.assemble(gathered, H, 2, 8)
    ref = render(0, 1)[:H]
    assert ref[..., :3].max() > 0 and full.tobytes() == ref.tobytes()


SMALL = ["--steps", "2", "--warmup", "1", "--res", "256", "--tris", "20000", "--no-cpu-baseline", "--no-extras"]


@pytest.mark.parametrize("wl", ["ir", "evplp"])
def test_bench_two_ranks_share_one_gpu_over_gloo(wl, tmp_path):
    """EVPLP_BENCH_BACKEND=gloo: `bench.py --gpus 2` starts its two ranks, both on the one GPU of the box, with the collectives staged
    through the host (RCCL refuses two ranks on one device).  Everything else is the N > 1 code: strip contexts, split light
    tracing + record all-gather (evplp), framebuffer all-gathers, max / sum over ranks.  Same shadow rays and photon-pixel pairs as
    one rank; the assembled VPL frame is bit-identical, the photon frame to fp32 round-off (bin order)."""
    args = ["--workload", wl] + SMALL
    two = run_bench(args + ["--gpus", "2", "--dump-frame", str(tmp_path / "two.npy")], {"EVPLP_BENCH_BACKEND": "gloo"})
    one = run_bench(args + ["--dump-frame", str(tmp_path / "one.npy")])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["config"]["physical_gpus"] == 1
    assert two["config"]["shadow_rays_per_frame"] == one["config"]["shadow_rays_per_frame"] > 0
    assert two["config"]["usable_vpl_records"] == one["config"]["usable_vpl_records"]
    a, b = np.load(tmp_path / "two.npy"), np.load(tmp_path / "one.npy")
    assert b[0][..., :3].max() > 0 and a[0].tobytes() == b[0].tobytes()
    if wl == "evplp":
        assert two["roofline_splat"]["pairs_per_frame"] == one["roofline_splat"]["pairs_per_frame"] > 0
        assert b[1][..., :3].max() > 0 and np.abs(a[1] - b[1]).max() <= 1e-5 * b[1].max()


def test_bench_group_front_end(tmp_path):
    """--front-end group: one process drives evplp_group (the native multi-GPU entry of the C ABI); on a one-GPU box its two ranks
    share the device.  Same rays, bit-identical VPL frame."""
    two = run_bench(SMALL + ["--gpus", "2", "--front-end", "group", "--dump-frame", str(tmp_path / "two.npy")])
    one = run_bench(SMALL + ["--dump-frame", str(tmp_path / "one.npy")])
    assert two["n_gpus"] == 2 and "evplp_group" in two["config"]["front_end"]
    assert two["config"]["shadow_rays_per_frame"] == one["config"]["shadow_rays_per_frame"] > 0
    a, b = np.load(tmp_path / "two.npy"), np.load(tmp_path / "one.npy")
    assert a[0].tobytes() == b[0].tobytes()


def test_bench_default_extras_small():
    """The default line's extra objects (here at a small size): the path through evplp_render_json agrees with the timed loop, and
    configs #3 / #4 / #5 ride along with their rooflines."""
    d = run_bench(["--steps", "3", "--warmup", "1", "--tris", "20000", "--no-cpu-baseline"])
    rj = d["render_json"]
    assert rj["iterations"] == 20 and rj["ms_per_iteration"] > 0
    assert 0.8 <= rj["ratio_to_ms_per_step"] <= 1.25, rj
    assert d["evplp"]["roofline_splat"]["bound"] == "hbm" and 0 < d["evplp"]["roofline_splat"]["frac"] < 1
    p = d["ppm"]
    assert p["roofline"]["bound"] == "hbm" and p["ms_per_iteration"] > 0 and p["feeders"]["light_trace"]["ms"] > 0 and p["feeders"]["primary"]["ms"] > 0
    assert 0.7 <= p["render_json"]["ratio_to_ms_per_step"] <= 1.4, p["render_json"]
    v = d["vsl"]["roofline"]
    assert v["kernel"].startswith("gather_vsl_walk_kernel") and v["sample_iterations_per_frame"] > v["lit_pairs_per_frame"] > 0
