#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X:
"Mpaths/s + ms/frame, conference 1024^2, 4096 VPLs, 1/2/4/8 GPU" (BASELINE.json).

A step = one frame of the technique loop (rt/rtcomphoton/rtcomphoton.h:936-1068) for BASELINE
config #2 (Instant Radiosity, 4096 VPL record slots = 1024 light paths x 4 vertices, 1024x1024,
misMode "one", photon splat off): jittered G-buffer, light tracing, VPL gather with one shadow
ray per (pixel, usable VPL) pair, and -- with N > 1 -- the RCCL all-gathers.  The scene is the
procedural conference stand-in (the reference's meshes are Git-LFS stubs): "data": "synthetic".

N GPUs: one process per GPU (torch.distributed, backend nccl = RCCL).  The image is cut into
interleaved 16-row strips (rank r owns row blocks b with b % N == r); the scene and the BVH are
replicated; each rank traces 1/N of the light paths and the record set is shared by an in-place
all-gather; each rank gathers its own pixels; the framebuffer strips are all-gathered every
frame.  Total work is fixed ("scaling": "strong").

Path = one (pixel, usable VPL record) pair = one shadow ray + one contribution (BASELINE.md
section 3).  value = pairs of the whole frame / frame time.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 125.0          # vplSplat, misMode one (lighttracing.cu:282-312)
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--paths", type=int, default=1024, help="numLightPaths = numVplLightPaths (x4 record slots)")
    ap.add_argument("--tris", type=int, default=331000)
    ap.add_argument("--workload", default="ir", choices=["ir", "evplp"], help="ir = config #2 (headline); evplp = config #3 (+2M photon splat)")
    ap.add_argument("--bvh", default="sah", choices=["sah", "lbvh"], help="acceleration-structure builder (same flattened node format)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0, help="path-tracer iterations of the CPU baseline sample (0 = auto, ~15 s)")
    return ap.parse_args()


def cpu_baseline(json_path, res, iters_hint):
    """The oracle's restatement of the reference path tracer (pathtracing.cu:240-377), timed on the host
    cores of this box on a bounded sample of the same scene/camera: a res x res G-buffer, then
    1-spp iterations until ~15 s of path tracing.  Baseline only (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_api as oa
    import scenes
    sd, _ = scenes.load_obj_scene(json_path)
    osc = oa.Scene(sd)
    cores = oa.load().evo_get_threads()
    sample_res = min(res, 1024)
    sd.aspect = 1.0
    g = osc.primary(sample_res, sample_res, (0.0, 0.0))
    out = np.zeros((sample_res, sample_res, 4), np.float32)
    paths, it, t0 = 0, 0, time.time()
    budget = 15.0
    while True:
        _, n = osc.path_trace(sd.cam_origin, it, 3, sample_res, sample_res, g, out=out, accumulate=True)
        paths += n; it += 1
        if (iters_hint and it >= iters_hint) or (not iters_hint and time.time() - t0 >= budget):
            break
    dt = time.time() - t0
    return {"value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": int(cores), "kind": "port",
            "sample": f"oracle path tracer (NEE+MIS, <=3 bounces), {sample_res}x{sample_res} window of the same scene/camera, "
                      f"{it} iterations x 1 spp = {paths} camera paths in {dt:.1f} s",
            "ms_per_frame_at_sample_res": dt / it * 1e3}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import evplp_amd as ev

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # EVPLP_BENCH_FORCE_DIST=1 runs the collective code path with a single rank (1-GPU smoke of the N>1 path)
    use_dist = world > 1 or os.environ.get("EVPLP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    # ---- inputs: procedural conference stand-in, written once per node
    scene_dir = os.path.join("/tmp", f"evplp_bench_{os.getuid()}_{a.tris}_{a.res}")
    json_path = os.path.join(scene_dir, "conference_synth.json")
    if local_rank == 0:
        ev.synth_scene(scene_dir, "conference_synth", a.tris, 1234, a.res, a.res)
    if use_dist:
        dist.barrier()

    W = H = a.res
    P = 4                                   # numMaxBounces 3 -> 4 record slots per path
    n_light = a.paths if a.workload == "ir" else 500000
    n_vpl = a.paths
    strip_rows = 8      # finest interleave (tiles are 8 rows): slowest / mean rank time 10.9 / 10.4 ms at 8 ranks vs 11.8 / 10.4 with 16-row strips
    ctx = ev.Context(W, H, n_light, n_vpl, P, device=local_rank, strip_rank=rank, strip_count=world, strip_rows=strip_rows,
                     bvh_builder=ev.BVH_SAH if a.bvh == "sah" else ev.BVH_LBVH)
    ctx.load_scene_json(json_path)
    cam = ctx.camera()
    bsr, total_area, _ = ctx.scene_metrics()
    radius = 0.003 * bsr if a.workload == "evplp" else 0.0
    # one explicit (non-null) HIP stream carries the kernels AND orders the collectives: torch.distributed
    # synchronises its RCCL work with the current stream, so kernels must be launched on that stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    # torch owns the buffers that take part in collectives
    nrec = n_light * P
    records = torch.zeros(nrec * 96 // 4, dtype=torch.float32, device=dev)
    strip = torch.zeros(ctx.local_rows * W * 4, dtype=torch.float32, device=dev)
    ctx.bind_buffer(ev.BUF_RECORDS, records.data_ptr(), records.numel() * 4)
    ctx.bind_buffer(ev.BUF_VPL_ACCUM, strip.data_ptr(), strip.numel() * 4)
    full = torch.zeros(world * strip.numel(), dtype=torch.float32, device=dev) if use_dist else strip
    # a light-tracing launch is latency-bound (0.26 ms for 1024 paths, 0.25 ms for 128): small path counts are traced
    # redundantly by every rank (identical records, no exchange); large ones are split and all-gathered
    split_paths = use_dist and n_light % world == 0 and (n_light >= 16384 or os.environ.get("EVPLP_BENCH_FORCE_DIST") == "1")
    per_rank = n_light // world if split_paths else n_light

    mis = "one" if a.workload == "ir" else "balance"
    pdf_mc = (n_vpl / n_light) / math.pi / (radius * radius) if radius > 0 else 0.0
    jrng = np.random.RandomState(0)

    def frame(it):
        u = jrng.rand(2)
        jitter = ((2 * u[0] - 1) / W, (2 * u[1] - 1) / H)
        fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode=mis, pdf_mc=pdf_mc, clamping_value=1.0 / total_area,
                             photon_radius=radius, num_light_paths=n_light, num_vpl_light_paths=n_vpl, photons_per_path=P,
                             do_accumulate=1, rng_seed=it, jitter=jitter)
        ctx.primary(jitter)
        if split_paths:
            ctx.trace_light_paths(it, rank * per_rank, per_rank)
            chunk = records.numel() // world
            # out-of-place send buffer (a copy of this rank's slice): no aliasing between input and output
            dist.all_gather_into_tensor(records, records[rank * chunk:(rank + 1) * chunk].clone())
        else:
            ctx.trace_light_paths(it)
        ctx.gather_vpl(fp)
        if a.workload == "evplp":
            ctx.splat_photons(fp)
        if use_dist:
            dist.all_gather_into_tensor(full, strip)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(a.warmup):
        frame(i)
    sync_all()
    kernel_ms, pairs_local, rays_local, splat_ms, splat_tiles_ms, splat_pairs = [], 0, 0, [], [], 0
    t0 = time.perf_counter()
    for i in range(a.steps):
        frame(a.warmup + i)
        st = ctx.pass_stats(ev.PASS_GATHER_VPL)      # HIP events on the launch stream; syncs this rank's stream
        kernel_ms.append(st["dominant_kernel_ms"]); pairs_local += st["pairs"]; rays_local += st["rays"]
        if a.workload == "evplp":
            ss = ctx.pass_stats(ev.PASS_SPLAT)
            splat_ms.append(ss["ms"]); splat_tiles_ms.append(ss["dominant_kernel_ms"]); splat_pairs += ss["pairs"]
    sync_all()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, float(pairs_local), float(rays_local), float(splat_pairs), sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64, device=dev)
    if use_dist:
        mx = stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0]); pairs = float(sm[1]); rays = float(sm[2]); spairs = float(sm[3]); kms = float(mx[4])
    else:
        pairs, rays, spairs, kms = float(pairs_local), float(rays_local), float(splat_pairs), float(stats[4])

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        total_pairs = pairs + spairs
        value = total_pairs / dt / 1e6
        pairs_per_launch = pairs_local / a.steps               # this rank's kernel
        achieved = pairs_per_launch * FLOP_PER_PAIR / (kms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_gather_vpl.json")
        if os.path.exists(tpath) and world == 1 and a.res == 1024 and a.paths == 1024:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        out = {
            "metric": "Mpaths/s", "value": value, "unit": "Mpaths/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("Instant Radiosity, 4096 VPL record slots (1024 light paths x 4), misMode one, photon splat off "
                                    "(BASELINE config #2)" if a.workload == "ir" else
                                    "EVPLP: 4096 VPL slots + 2M photon record slots, misMode balance (BASELINE config #3)"),
                       "scene": f"procedural conference stand-in, {a.tris} target triangles, seed 1234 (reference meshes are LFS stubs)",
                       "resolution": [W, H], "num_light_paths": n_light, "num_vpl_light_paths": n_vpl, "photons_per_path": P,
                       "usable_vpl_records": int(pairs / a.steps / (W * H) + 0.5), "partition": f"{world} x interleaved {strip_rows}-row strips",
                       "path_definition": "gather: (pixel, usable VPL record) pair = 1 shadow ray; splat: (photon, covered pixel) pair",
                       "mrays_per_s": rays / dt / 1e6, "bvh_builder": a.bvh},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": traffic, "kernel": "gather_vpl_kernel", "kernel_ms": kms,
                         "note": "fp32 VECTOR-ALU bound (BVH packet traversal + shading, not GEMM-shaped; the f32 MFMA peak equals the "
                                 "vector peak, 157.3 TFLOP/s). achieved = 125 algorithmic flop per evaluated pair (SURVEY 8d) x pairs per "
                                 "launch / HIP-event kernel time; traversal flops are overhead and not counted."},
        }
        # the same launch against the HBM roofline (north star: "fraction of HBM roofline"): the gather is not HBM-bound, so
        # the fraction is tiny by construction -- algorithmic bytes W*H*(64 + 16 + 16) + n_vpl*96, and the bytes
        # the PMC counters saw (per-item partial sums, BVH lines missing the per-XCD L2s)
        n_usable = pairs / a.steps / (W * H)
        alg_bytes = (W * H * (64 + 16 + 16)) / max(world, 1) + n_usable * 96
        out["roofline_hbm"] = {"bound": "hbm", "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": alg_bytes / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
                               "traffic_frac": (traffic / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS) if traffic else None, "kernel": "gather_vpl_kernel"}
        if a.workload == "evplp" and splat_ms:
            nrec_bytes = nrec * 96 + W * H * 64 + W * H * 24          # SURVEY 8(d) algorithmic bytes per frame
            sms = sum(splat_ms) / len(splat_ms)
            out["roofline_splat"] = {"bound": "hbm", "achieved": nrec_bytes / (sms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": nrec_bytes / (sms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None, "pass_ms": sms,
                                     "tiles_kernel_ms": sum(splat_tiles_ms) / len(splat_tiles_ms), "pairs_per_frame": spairs / a.steps}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(json_path, a.res, a.cpu_iters)
    else:
        out = None
    ctx.close()
    # The JSON line must be the LAST thing on stdout: RCCL prints a version banner through C stdio, which is
    # block-buffered on a pipe and would otherwise surface after it at exit.  Every rank pushes its C buffers
    # out, all ranks meet, then rank 0 prints.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        if world > 1:
            time.sleep(1.0)       # let the other ranks' processes drain whatever they still print while exiting
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
