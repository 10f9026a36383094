#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X:
"Mpaths/s + ms/frame, conference 1024^2, 4096 VPLs, 1/2/4/8 GPU" (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload ir|evplp|ppm|vsl] [--scene hard|easy]

A step = one iteration of the technique loop (rt/rtcomphoton/rtcomphoton.h:936-1068):
  ir    BASELINE config #2 (default, the headline): Instant Radiosity, 1024 light paths x 4 vertices = 4096 VPL record
        slots, 1024x1024, misMode "one", photon splat off: jittered G-buffer, light tracing, VPL gather with one shadow
        ray per (pixel, usable VPL) pair.
  evplp config #3: config #2 + 500 000 light paths (2 M photon record slots) splatted with radius 0.3 %, misMode balance.
  ppm   config #4: progressive photon mapping, 1920x1080, 300 000 light paths, no VPLs, radius shrinking by the
        Knaus-Zwicker schedule (alpha 0.7) from iteration to iteration.
  vsl   config #5: progressive VSL gather (4096 VPL paths = 16 384 record slots, forceVsl) + photons, 2048x2048.
The scene is a procedural conference stand-in (the reference's meshes are Git-LFS stubs): "data": "synthetic";
--scene hard (default) is furnished with curved and thin parts, --scene easy is the room of tessellated boxes.

N GPUs: one process per GPU (torch.distributed, backend nccl = RCCL).  Launched by the driver as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; when no launcher set WORLD_SIZE,
`python bench.py --gpus N` starts the N rank processes itself (before anything touches a GPU) and relays rank 0's
JSON line.  The image is cut into interleaved 8-row strips (rank r owns row blocks b with b % N == r); the scene and
the BVH are replicated; large light-path sets are traced 1/N per rank and shared by an all-gather of the record
buffer, small ones are traced redundantly; each rank gathers / splats its own pixels; the framebuffer strips are
all-gathered every frame.  Total work is fixed ("scaling": "strong").

Path = one evaluated light-transport sample: gather -> one (pixel, usable VPL record) pair
that passes the cosine test and traces its shadow ray; splat -> one (photon, covered pixel) pair.
value = paths of the whole frame / frame time.  `pairs_nominal` (pixels x usable records, the loop count of
splatColor) is reported beside it.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 125.0          # vplSplat, misMode one (lighttracing.cu:282-312)
FLOP_PER_PAIR_MIS = 195.0      # balance / max / power2
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md: FP32 vector peak
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    "ir": "Instant Radiosity, 4096 VPL record slots (1024 light paths x 4), misMode one, photon splat off (BASELINE config #2)",
    "evplp": "EVPLP: 4096 VPL slots + 2M photon record slots (500k light paths), radius 0.3 %, misMode balance (BASELINE config #3)",
    "ppm": "Progressive photon mapping, 1920x1080, 300k light paths x 4 = 1.2M photon record slots, no VPLs, alpha 0.7 (BASELINE config #4)",
    "vsl": "Progressive VSL gather (forceVsl, 4096 VPL paths = 16384 record slots, radius 5 %) + 300k-path photon splat, 2048x2048 (BASELINE config #5)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--res", type=int, default=0, help="override the square resolution of ir / evplp / vsl")
    ap.add_argument("--paths", type=int, default=0, help="override numVplLightPaths (x4 record slots)")
    ap.add_argument("--tris", type=int, default=331000)
    ap.add_argument("--workload", default="ir", choices=sorted(WORKLOADS))
    ap.add_argument("--scene", default="hard", choices=["hard", "easy"])
    ap.add_argument("--mis", default="", help="override misMode")
    ap.add_argument("--bvh", default="sah", choices=["sah", "sbvh", "lbvh", "gpu"], help="acceleration-structure builder (same flattened node format)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (other scene, 16384-slot variant, GPU path tracer)")
    ap.add_argument("--cpu-iters", type=int, default=0, help="path-tracer iterations of the CPU baseline sample (0 = auto, ~12 s)")
    a = ap.parse_args()
    heavy = a.workload == "vsl"
    if a.steps is None:
        a.steps = 2 if heavy else 60       # default timed region: several seconds
    if a.warmup is None:
        a.warmup = 1 if heavy else 3
    return a


def spawn_ranks(a):
    """`bench.py --gpus N` without a launcher: start N rank processes (one per GPU) from this process, which has not
    imported torch or touched a GPU, relay rank 0's stdout, fail if any rank fails."""
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # wait for all ranks; if one dies the others would sit in the rendezvous / a collective until a timeout: stop them (these are
    # this process's own children, by handle)
    import threading
    buf = []
    t = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    t.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for i, p in enumerate(procs):
            if p.poll() is not None and p.returncode != 0:
                failed = i
                break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    t.join(timeout=5)
    sys.stdout.write((buf[0] if buf else b"").decode(errors="replace"))
    sys.stdout.flush()
    bad = [i for i, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py: rank(s) {bad} failed (exit codes {rcs})")


def cpu_baseline(json_path, res, iters_hint):
    """The oracle's restatement of the reference path tracer (pathtracing.cu:240-377), timed on the host
    cores of this box on a bounded sample of the same scene/camera: a res x res G-buffer, then
    1-spp iterations until ~12 s of path tracing.  Baseline only (kind "port")."""
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
    budget = 12.0
    while True:
        _, n = osc.path_trace(sd.cam_origin, it, 3, sample_res, sample_res, g, out=out, accumulate=True)
        paths += n; it += 1
        if (iters_hint and it >= iters_hint) or (not iters_hint and time.time() - t0 >= budget):
            break
    dt = time.time() - t0
    base = {"value": paths / dt / 1e6, "unit": "M camera paths/s", "cores": int(cores), "kind": "port",
            "sample": f"oracle path tracer (NEE+MIS, <=3 bounces), {sample_res}x{sample_res} window of the same scene/camera, "
                      f"{it} iterations x 1 spp = {paths} camera paths in {dt:.1f} s",
            "ms_per_frame_at_sample_res": dt / it * 1e3,
            "note": "the north star's baseline (reference CPU path tracer); its unit is camera paths, NOT the (pixel, VPL) pairs of `value` -- "
                    "the same-unit figures are cpu_baseline_like_for_like (oracle gather) and gpu_path_tracer_mpaths_s"}
    # the same unit as `value`: the oracle's VPL gather (splatColor restatement) on a band of rows of the same frame
    recs = osc.trace_light_paths(0, 1024, 4)
    fp = oa.frame_params(camera_pos=sd.cam_origin, mis_mode=0, num_light_paths=1024, num_vpl_light_paths=1024, photons_per_path=4)
    rows, pairs_done, t1 = 4, 0, time.time()
    r0 = sample_res // 3
    while time.time() - t1 < 8.0 and r0 + rows <= sample_res:
        _, n = osc.gather(fp, sample_res, sample_res, g, recs, rows=(r0, r0 + rows))
        pairs_done += n; r0 += rows
        rows = min(rows * 2, 64)
    dt2 = time.time() - t1
    like = {"value": pairs_done / dt2 / 1e6, "unit": "Mpaths/s (pixel x usable-VPL pairs, oracle loop count)", "cores": int(cores), "kind": "port",
            "sample": f"oracle VPL gather (splatColor), rows {sample_res // 3}..{r0} of the {sample_res}x{sample_res} frame, {pairs_done} pairs in {dt2:.1f} s"}
    return base, like


def main():
    a = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and world_env is None:
        spawn_ranks(a)
        return
    import numpy as np
    import torch
    import torch.distributed as dist
    import evplp_amd as ev

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # EVPLP_BENCH_FORCE_DIST=1 runs the collective code path with a single rank (1-GPU smoke of the N>1 path)
    force_dist = os.environ.get("EVPLP_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group has {dist.get_world_size()} ranks")

    wl = a.workload
    if wl == "ppm":
        W, H = 1920, 1080
    elif wl == "vsl":
        W = H = a.res or 2048
    else:
        W = H = a.res or 1024
    P = 4                                   # numMaxBounces 3 -> 4 record slots per path
    n_vpl = {"ir": 1024, "evplp": 1024, "ppm": 0, "vsl": 4096}[wl]
    if a.paths and wl != "ppm":
        n_vpl = a.paths
    n_light = {"ir": n_vpl, "evplp": 500000, "ppm": 300000, "vsl": 300000}[wl]
    mis = a.mis or {"ir": "one", "evplp": "balance", "ppm": "one", "vsl": "one"}[wl]
    progressive = wl in ("ppm", "vsl")

    # ---- inputs: procedural conference stand-in, written once per node
    def scene_json(style):
        d = os.path.join("/tmp", f"evplp_bench_{os.getuid()}_{style}_{a.tris}_{W}x{H}")
        if local_rank == 0:
            ev.synth_scene(d, "conference_synth", a.tris, 1234, W, H, style=style)
        return os.path.join(d, "conference_synth.json")
    json_path = scene_json(a.scene)
    if use_dist:
        dist.barrier()

    strip_rows = 8      # finest interleave (tiles are 8 rows)
    builder = {"sah": ev.BVH_SAH, "sbvh": ev.BVH_SBVH, "lbvh": ev.BVH_LBVH, "gpu": ev.BVH_LBVH_GPU}[a.bvh]

    def make_ctx(path, nl, nv):
        c = ev.Context(W, H, nl, nv, P, device=local_rank, strip_rank=rank, strip_count=world, strip_rows=strip_rows, bvh_builder=builder, overlap_light_tracing=True)
        c.load_scene_json(path)
        return c
    ctx = make_ctx(json_path, n_light, n_vpl)
    cam = ctx.camera()
    bsr, total_area, _ = ctx.scene_metrics()
    radius0 = 0.003 * bsr if wl != "ir" else 0.0
    vsl_radius0 = max(0.05 * bsr, 0.008) if wl == "vsl" else 0.0
    # one explicit (non-null) HIP stream carries the kernels AND orders the collectives: torch.distributed
    # synchronises its RCCL work with the current stream, so kernels must be launched on that stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    # torch owns the buffers that take part in collectives
    nrec = n_light * P
    records = torch.zeros(nrec * 96 // 4, dtype=torch.float32, device=dev)
    strip = torch.zeros(ctx.local_rows * W * 4, dtype=torch.float32, device=dev)
    pstrip = torch.zeros(ctx.local_rows * W * 4, dtype=torch.float32, device=dev)
    if use_dist:     # (one GPU: the library keeps the records and double-buffers them behind the overlapped light tracing)
        ctx.bind_buffer(ev.BUF_RECORDS, records.data_ptr(), records.numel() * 4)
    ctx.bind_buffer(ev.BUF_VPL_ACCUM, strip.data_ptr(), strip.numel() * 4)
    ctx.bind_buffer(ev.BUF_PHOTON_ACCUM, pstrip.data_ptr(), pstrip.numel() * 4)
    full = torch.zeros(world * strip.numel(), dtype=torch.float32, device=dev) if use_dist else strip
    pfull = torch.zeros(world * pstrip.numel(), dtype=torch.float32, device=dev) if (use_dist and wl != "ir") else pstrip
    # a light-tracing launch is latency-bound (0.26 ms for 1024 paths, 0.25 ms for 128): small path counts are traced
    # redundantly by every rank (identical records, no exchange); large ones are split and all-gathered
    split_paths = use_dist and n_light % world == 0 and (n_light >= 16384 or force_dist)
    per_rank = n_light // world if split_paths else n_light
    send = torch.empty(records.numel() // world, dtype=torch.float32, device=dev) if split_paths else None   # reused out-of-place send slice

    jrng = np.random.RandomState(0)
    sched = {"radius": radius0, "clamp": 1.0 / total_area, "pdf_mc": (n_vpl / n_light) / math.pi / (radius0 * radius0) if radius0 > 0 else 0.0,
             "vsl_radius": vsl_radius0, "vsl_inv": (1.0 / (math.pi * vsl_radius0 * vsl_radius0)) if vsl_radius0 > 0 else 0.0}
    clamp_start = sched["clamp"]

    def frame(it):
        u = jrng.rand(2)
        jitter = ((2 * u[0] - 1) / W, (2 * u[1] - 1) / H)
        fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode=mis, pdf_mc=sched["pdf_mc"], clamping_value=sched["clamp"],
                             photon_radius=sched["radius"], vsl_radius=sched["vsl_radius"], vsl_inv_pi_radius2=sched["vsl_inv"],
                             num_light_paths=n_light, num_vpl_light_paths=n_vpl, photons_per_path=P,
                             do_accumulate=1, rng_seed=it, jitter=jitter)
        def light_paths():
            if split_paths:
                ctx.trace_light_paths(it, rank * per_rank, per_rank)
                chunk = records.numel() // world
                send.copy_(records[rank * chunk:(rank + 1) * chunk])
                dist.all_gather_into_tensor(records, send)
            else:
                ctx.trace_light_paths(it)
        # Light tracing runs on the context's second stream.  Pure photon mapping: light paths first -- they go to the record buffer
        # nobody reads (double-buffered), start while the previous iteration's splat still runs, and the G-buffer pass (whose call
        # waits for the verdict of the previous photon bins) follows.  With a gather in the frame the order of the reference is
        # better: light paths beside the G-buffer pass, not beside the 90 ms gather whose CUs they would share (+0.5 ms measured).
        if wl == "ppm":
            light_paths(); ctx.primary(jitter)
        else:
            ctx.primary(jitter); light_paths()
        if wl in ("ir", "evplp"):
            ctx.gather_vpl(fp)
        elif wl == "vsl":
            ctx.gather_vsl(fp)
        if wl != "ir":
            ctx.splat_photons(fp)
            if os.environ.get("EVPLP_DUMP_SPLAT_HIST"):   # stats build: rectangle classes of the photons (tools/debug_splat_hist.py)
                print("HIST", it, ctx.debug_counters(ev.PASS_SPLAT)[4:4 + 28].tolist(), flush=True)
        if use_dist:
            if wl != "ppm":
                dist.all_gather_into_tensor(full, strip)
            if wl != "ir":
                dist.all_gather_into_tensor(pfull, pstrip)
        if progressive:   # rtcomphoton.h:1033-1063 after numIterations++
            r, c, p, vr, vi = ev.progressive_step(it + 1, 0.7, clamp_start, n_vpl, n_light, sched["radius"], sched["clamp"], sched["pdf_mc"],
                                                  wl == "vsl", sched["vsl_radius"], sched["vsl_inv"])
            sched.update(radius=r, clamp=c, pdf_mc=p, vsl_radius=vr, vsl_inv=vi)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    gather_pass = ev.PASS_GATHER_VSL if wl == "vsl" else ev.PASS_GATHER_VPL
    for i in range(a.warmup):
        frame(i)
    sync_all()
    kernel_ms, nominal_local, rays_local, shaded_local, splat_ms, splat_tiles_ms, splat_pairs, usable = [], 0, 0, 0, [], [], 0, 0
    # Pass statistics synchronise the stream.  The gather workloads (>= 90 ms per step) read them every step; config #4's 0.9 ms
    # iterations read the splat's HIP events on ten steps spread over the timed region and count pairs with the library's device-side
    # running total, read before and after (a read-back per step cost 15 % of the iteration).
    sample_every = max(1, a.steps // 10) if wl == "ppm" else 1
    pairs_before = ctx.pass_stats(ev.PASS_SPLAT)["shaded"] if wl != "ir" else 0
    sync_all()
    t0 = time.perf_counter()
    for i in range(a.steps):
        frame(a.warmup + i)
        if wl != "ppm":
            st = ctx.pass_stats(gather_pass)      # HIP events on the launch stream; syncs this rank's stream
            kernel_ms.append(st["dominant_kernel_ms"]); nominal_local += st["pairs"]; rays_local += st["rays"]; shaded_local += st.get("shaded", 0)
            usable = st["usable"]
        if wl != "ir" and i % sample_every == 0:
            ss = ctx.pass_stats(ev.PASS_SPLAT)
            splat_ms.append(ss["ms"]); splat_tiles_ms.append(ss["dominant_kernel_ms"])
    sync_all()
    dt = time.perf_counter() - t0
    if wl != "ir":
        splat_pairs = ctx.pass_stats(ev.PASS_SPLAT)["shaded"] - pairs_before
    kms_local = sum(kernel_ms) / len(kernel_ms) if kernel_ms else 0.0
    stats = torch.tensor([dt, float(nominal_local), float(rays_local), float(splat_pairs), kms_local, float(shaded_local)], dtype=torch.float64, device=dev)
    if use_dist:
        mx = stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0]); nominal = float(sm[1]); rays = float(sm[2]); spairs = float(sm[3]); kms = float(mx[4]); shaded = float(sm[5])
        n_ranks = dist.get_world_size()
    else:
        nominal, rays, spairs, kms, shaded = float(nominal_local), float(rays_local), float(splat_pairs), kms_local, float(shaded_local)
        n_ranks = 1

    out = None
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        total_paths = rays + spairs
        value = total_paths / dt / 1e6
        acc = ctx.accel_info()
        out = {
            "metric": "Mpaths/s", "value": value, "unit": "Mpaths/s", "n_gpus": n_ranks, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[wl],
                       "scene": f"procedural conference stand-in, style {a.scene} ({'curved / thin furniture, rotated clutter, ~2400 small occluders' if a.scene == 'hard' else 'tessellated boxes'}), "
                                f"{a.tris} target triangles, seed 1234 (reference meshes are LFS stubs)",
                       "resolution": [W, H], "num_light_paths": n_light, "num_vpl_light_paths": n_vpl, "photons_per_path": P, "mis_mode": mis,
                       "usable_vpl_records": int(usable), "partition": f"{n_ranks} x interleaved {strip_rows}-row strips",
                       "path_definition": "gather: (pixel, usable VPL record) pair that passes the cosine test = 1 shadow ray; splat: (photon, covered pixel) pair",
                       "pairs_nominal_per_frame": nominal / a.steps, "mpairs_nominal_per_s": nominal / dt / 1e6,
                       "shadow_rays_per_frame": rays / a.steps, "unoccluded_pairs_per_frame": shaded / a.steps,
                       "mrays_per_s": rays / dt / 1e6, "bvh_builder": a.bvh, "bvh": acc},
        }
        if wl != "ppm":
            flop = FLOP_PER_PAIR if mis in ("one", "geometryClamp", "geometryBrdfClamp") else FLOP_PER_PAIR_MIS
            rays_per_launch = rays_local / a.steps              # this rank's kernel
            kname = "gather_vsl_kernel" if wl == "vsl" else "gather_vpl_kernel"
            achieved = rays_per_launch * flop / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
            out["roofline"] = {
                "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": None, "traffic_source": "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/prof_all.sh); not measured inside this run",
                "kernel": kname, "kernel_ms": kms, "flop_per_pair": flop,
                "frac_nominal_pairs": (nominal_local / a.steps) * flop / (kms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if kms > 0 else 0.0,
                "note": "fp32 VECTOR-ALU bound (BVH packet traversal + shading; not GEMM-shaped). achieved = algorithmic flop per evaluated pair "
                        "(SURVEY 8d) x pairs that trace a shadow ray per launch / HIP-event kernel time; traversal flops are overhead and not counted. "
                        "frac_nominal_pairs prices every (pixel, usable VPL) loop iteration instead (round-1 definition)."
                        + (" VSL: 125 flop counts one vplSplat-equivalent per pair; the estimator's per-sample flops are not counted." if wl == "vsl" else "")}
            tpath = os.path.join(ROOT, "profiles", "traffic_gather_vpl.json")
            if os.path.exists(tpath) and wl == "ir":
                tj = json.load(open(tpath))
                if tj.get("config") == f"{a.scene}:{W}x{H}:{n_vpl}:{n_ranks}":
                    out["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                    out["roofline"]["traffic_source"] = "profiles/traffic_gather_vpl.json (committed PMC summary of this configuration; not measured inside this run)"
            n_usable = usable
            alg_bytes = (W * H * (64 + 16 + 16)) / max(n_ranks, 1) + n_usable * 96
            out["roofline_hbm"] = {"bound": "hbm", "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": alg_bytes / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS, "kernel": kname,
                                   "note": "the gather priced against HBM (north star: fraction of HBM roofline): tiny by construction, the kernel is issue-bound"}
        if wl != "ir" and splat_ms:
            nrec_bytes = nrec * 96 + (W * H * 64 + W * H * 24) / max(n_ranks, 1)          # SURVEY 8(d) algorithmic bytes per frame (per rank)
            sms = sum(splat_ms) / len(splat_ms)
            rs = {"bound": "hbm", "achieved": nrec_bytes / (sms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": nrec_bytes / (sms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None, "kernel": "photon splat pass (tile boxes + bin + scatter + tiles)", "pass_ms": sms,
                  "tiles_kernel_ms": sum(splat_tiles_ms) / len(splat_tiles_ms), "pairs_per_frame": spairs / a.steps, "algorithmic_bytes": nrec_bytes}
            tpath = os.path.join(ROOT, "profiles", "traffic_splat.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath)).get("configs", {}).get(f"{wl}:{a.scene}:{W}x{H}:{n_ranks}")
                if tj:
                    rs["traffic"] = tj.get("hbm_bytes_per_pass"); rs["traffic_source"] = "profiles/traffic_splat.json (committed PMC summary; not measured inside this run)"
            if wl == "ppm":
                out["roofline"] = rs
            else:
                out["roofline_splat"] = rs
    ctx.close()

    # ---- secondary measurements (rank 0, one GPU): the other scene style, the 16 384-slot reading of "4096 VPLs", the GPU path tracer
    if rank == 0 and world == 1 and wl == "ir" and not a.no_extras and not force_dist:
        def quick_ir(path, nv, steps=5):
            c = make_ctx(path, nv, nv)
            c.set_stream(stream.cuda_stream)
            cm = c.camera()
            _, ta, _ = c.scene_metrics()
            ms, rr, nom, us = [], 0, 0, 0
            for it in range(steps + 1):
                fp = ev.frame_params(camera_pos=list(cm.origin), mis_mode=mis, clamping_value=1.0 / ta, num_light_paths=nv, num_vpl_light_paths=nv,
                                     photons_per_path=P, do_accumulate=1, rng_seed=it)
                torch.cuda.synchronize(dev); t = time.perf_counter()
                c.primary((0.0, 0.0)); c.trace_light_paths(it); c.gather_vpl(fp); c.synchronize()
                if it:
                    ms.append((time.perf_counter() - t) * 1e3)
                    s = c.pass_stats(ev.PASS_GATHER_VPL); rr += s["rays"]; nom += s["pairs"]; us = s["usable"]
            res = {"ms_per_frame": sum(ms) / len(ms), "mpaths_per_s": rr / (sum(ms) * 1e-3) / 1e6, "mpairs_nominal_per_s": nom / (sum(ms) * 1e-3) / 1e6, "usable_vpl_records": int(us)}
            return c, res
        other = "easy" if a.scene == "hard" else "hard"
        c2, r2 = quick_ir(scene_json(other), n_vpl)
        out["scene_" + other] = r2
        c2.close()
        c3, r3 = quick_ir(json_path, 4 * n_vpl, steps=3)
        out["slots_16384_variant"] = dict(r3, note="'4096 VPLs' read as 4096 light PATHS (16384 record slots), SURVEY 8d")
        # GPU path tracer on the same context (same unit as cpu_baseline)
        pt_ms, pt_paths = [], 0
        for it in range(6):
            c3.primary((0.0, 0.0)); c3.path_trace(list(cam.origin), it, 3, accumulate=True); c3.synchronize()
            s = c3.pass_stats(ev.PASS_PATH_TRACE)
            if it:
                pt_ms.append(s["ms"]); pt_paths += s["pairs"]
        out["gpu_path_tracer_mpaths_s"] = pt_paths / (sum(pt_ms) * 1e-3) / 1e6
        c3.close()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        base, like = cpu_baseline(json_path, min(W, H), a.cpu_iters)
        out["cpu_baseline"] = base
        out["cpu_baseline_like_for_like"] = like

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
