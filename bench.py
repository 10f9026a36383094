#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X:
"Mpaths/s + ms/frame, conference 1024^2, 4096 VPLs, 1/2/4/8 GPU" (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload ir|evplp|ppm|vsl] [--scene hard|easy] [--front-end ranks|group]

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

The default run (no flags: config #2 on one GPU) also carries, as extra objects of the same JSON line, short runs of the other
configurations -- `evplp` (config #3, roofline of the photon splat), `ppm` (config #4: iteration time, splat roofline, the
light-tracing and G-buffer kernels), `vsl` (config #5) -- the other scene style, the 16 384-slot reading of "4096 VPLs", the GPU
path tracer, and `render_json`: the same configuration run through evplp_render_json, the entry a maintainer of the reference
binds (its per-iteration time must agree with ms_per_step).

N GPUs, two front ends (--front-end auto picks `group` whenever the process sees N devices, `ranks` otherwise):
  group  one process, `evplp_group` -- the native multi-GPU entry of the C ABI, what a maintainer of the reference binds (one host
        thread, N contexts, RCCL opened by the library; ranks on one device exchange by device copies).  Under the driver's launcher
        (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) rank 0 drives the group over the N devices and
        the other N - 1 processes only meet it at a host-side barrier; if the group cannot be opened all of them fall back to `ranks`.
  ranks  one process per GPU (torch.distributed, backend nccl = RCCL); when no launcher set WORLD_SIZE,
        `python bench.py --gpus N --front-end ranks` starts the N rank processes itself (before anything touches a GPU) and relays
        rank 0's JSON line.  EVPLP_BENCH_BACKEND=gloo stages the collectives through the host and lets the ranks share GPUs (rank r
        uses device r mod #devices): the N > 1 logic with the real kernels on a one-GPU box.
Either way the image is cut into interleaved blocks of 16 rows; the scene and the BVH are replicated; the light paths are traced by
every rank (same seed, identical records) unless the library's cost model expects 1/N per rank + an all-gather of the record buffer to be
faster (--split-light-paths); each rank gathers / splats its own pixels; the framebuffer strips are all-gathered every frame
(--exchange-every k: every k-th).  (round 6) --deal cost (the default where the workload has a gather): one calibration frame in front
of the warm-up in which the gathers clock their blocks, then the blocks are dealt by cost -- evplp_group_rebalance, or, with one process
per GPU, evplp_block_costs + an all-reduce of the per-block costs + evplp_deal_blocks in every process -- and every rank launches its most
expensive blocks first; --deal roundRobin: block b to rank b % N.  Total work is fixed ("scaling": "strong").

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
FLOP_PER_VSL_SAMPLE = 540.0    # 3 estimators x ~180 flop per sample-iteration (+ the sample's random numbers, not counted)
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md: FP32 vector peak
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    "ir": "Instant Radiosity, 4096 VPL record slots (1024 light paths x 4), misMode one, photon splat off (BASELINE config #2)",
    "evplp": "EVPLP: 4096 VPL slots + 2M photon record slots (500k light paths), radius 0.3 %, misMode balance (BASELINE config #3)",
    "ppm": "Progressive photon mapping, 1920x1080, 300k light paths x 4 = 1.2M photon record slots, no VPLs, alpha 0.7 (BASELINE config #4)",
    "vsl": "Progressive VSL gather (forceVsl, 4096 VPL paths = 16384 record slots, radius 5 %) + 300k-path photon splat, 2048x2048 (BASELINE config #5)",
}
P = 4                                   # numMaxBounces 3 -> 4 record slots per path
STRIP_ROWS = 16                         # two tile rows per strip block: a rank's entry-cut groups stay 2 x 2 tiles (8-row strips: 2 x 1, twice the cuts per pixel)


def strip_rows_for(a):
    """--strip-rows, or the library's own default (evplp_group_create): 16-row blocks"""
    return a.strip_rows if a.strip_rows > 0 else STRIP_ROWS


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
    ap.add_argument("--footprint", default="proxy", choices=["proxy", "ideal"],
                    help="coverage rule of the photon splat: the reference's proxy mesh (default, as in evplp_render_json) or the radius test alone")
    ap.add_argument("--bvh", default="sah", choices=["sah", "sbvh", "lbvh", "gpu"], help="acceleration-structure builder (same flattened node format)")
    ap.add_argument("--front-end", default="auto", choices=["auto", "ranks", "group"],
                    help="N > 1: `group` = one process driving evplp_group, the library's own multi-GPU entry (auto: whenever this process sees N devices); "
                         "`ranks` = one process per GPU over torch.distributed (auto: fewer devices than ranks, or EVPLP_BENCH_BACKEND=gloo)")
    ap.add_argument("--strip-rows", type=int, default=0, help="N > 1: height of a row block (multiple of 8); 0 = 16")
    ap.add_argument("--deal", default="cost", choices=["cost", "roundRobin"],
                    help="N > 1: row blocks dealt by the cost a calibration frame clocks (evplp_group_rebalance / evplp_deal_blocks; the default where the "
                         "workload has a gather) or block b to rank b %% N")
    ap.add_argument("--split-light-paths", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: every rank traces 1/N of the light paths + in-place all-gather of the records (on), all of them (off), or what evplp_group_split_model expects to be faster (auto)")
    ap.add_argument("--exchange-every", type=int, default=1, help="N > 1: the composited strips are all-gathered every k-th frame (0 = never inside the timed loop)")
    ap.add_argument("--scratch-gb", type=float, default=0.0, help="evplp_config.cut_scratch_bytes in GB (0 = the library's default of 8 GB; config #5 needs 68 GB for one band)")
    ap.add_argument("--mask-gb", type=float, default=0.0, help="evplp_config.vsl_mask_bytes in GB (0 = the library's default of 2 GB; config #5 needs 8.6 GB for one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (configs #3-#5, other scene, 16384-slot variant, GPU path tracer, render_json)")
    ap.add_argument("--cpu-iters", type=int, default=0, help="path-tracer iterations of the CPU baseline sample (0 = auto, ~12 s)")
    ap.add_argument("--dump-frame", default="", help="rank 0 writes the assembled VPL + photon accumulators of the last frame to this .npy file (tests)")
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


class Env:
    """What every measurement of one process shares: the modules, the process group, this rank's device and stream."""

    def __init__(self, a):
        import numpy as np
        import torch
        import torch.distributed as dist
        import evplp_amd as ev
        self.a, self.np, self.torch, self.dist, self.ev = a, np, torch, dist, ev
        world_env = os.environ.get("WORLD_SIZE")
        self.world = int(world_env or "1")
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.group_front_end = a.front_end == "group" and a.gpus > 1
        if not self.group_front_end and self.world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
        ndev = torch.cuda.device_count()
        self.backend = os.environ.get("EVPLP_BENCH_BACKEND", "nccl")
        if self.backend not in ("nccl", "gloo"):
            raise SystemExit("EVPLP_BENCH_BACKEND must be nccl or gloo")
        # RCCL refuses two ranks on one device; the host-staged gloo backend lets the ranks share GPUs
        self.device_index = local_rank % ndev if self.backend == "gloo" else local_rank
        torch.cuda.set_device(self.device_index)
        self.dev = torch.device("cuda", self.device_index)
        self.ndev = ndev
        # EVPLP_BENCH_FORCE_DIST=1 runs the collective code path with a single rank (1-GPU smoke of the N>1 path)
        self.force_dist = os.environ.get("EVPLP_BENCH_FORCE_DIST") == "1"
        self.use_dist = (self.world > 1 or self.force_dist) and not self.group_front_end
        if self.use_dist:
            if "MASTER_ADDR" not in os.environ:
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo")
            if dist.get_world_size() != a.gpus:
                raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group has {dist.get_world_size()} ranks")
        # one explicit (non-null) HIP stream carries the kernels AND orders the collectives: torch.distributed
        # synchronises its RCCL work with the current stream, so kernels must be launched on that stream
        self.stream = torch.cuda.Stream(device=self.dev)
        torch.cuda.set_stream(self.stream)

    # ---- collectives (nccl: on the device, in stream order; gloo: staged through the host)
    def all_gather(self, full, part):
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(full, part)
            return
        self.torch.cuda.synchronize(self.dev)
        h = part.cpu()
        parts = [self.torch.empty_like(h) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(parts, h)
        full.copy_(self.torch.cat(parts))

    def all_reduce(self, t, op):
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=op)
            return t
        h = t.cpu()
        self.dist.all_reduce(h, op=op)
        return h.to(t.device)

    def sync_all(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def scene_json(self, style, W, H):
        d = os.path.join("/tmp", f"evplp_bench_{os.getuid()}_{style}_{self.a.tris}_{W}x{H}")
        if self.rank == 0 or not self.use_dist:
            self.ev.synth_scene(d, "conference_synth", self.a.tris, 1234, W, H, style=style)
        if self.use_dist:
            self.dist.barrier()
        return os.path.join(d, "conference_synth.json")


def workload_shape(a, wl, primary):
    if wl == "ppm":
        W, H = 1920, 1080
    elif wl == "vsl":
        W = H = (a.res if primary and a.res else 2048)
    else:
        W = H = (a.res if primary and a.res else 1024)
    n_vpl = {"ir": 1024, "evplp": 1024, "ppm": 0, "vsl": 4096}[wl]
    if primary and a.paths and wl != "ppm":
        n_vpl = a.paths
    n_light = {"ir": n_vpl, "evplp": 500000, "ppm": 300000, "vsl": 300000}[wl]
    mis = (a.mis if primary and a.mis else {"ir": "one", "evplp": "balance", "ppm": "one", "vsl": "one"}[wl])
    return W, H, n_vpl, n_light, mis


def run_workload(env, wl, steps, warmup, scene, primary=True):
    """`steps` timed iterations of one configuration on this process's rank (or on an evplp_group); rank 0 returns the result object."""
    a, np, torch, dist, ev = env.a, env.np, env.torch, env.dist, env.ev
    W, H, n_vpl, n_light, mis = workload_shape(a, wl, primary)
    progressive = wl in ("ppm", "vsl")
    use_dist, world, rank, dev = env.use_dist, (env.world if env.use_dist else 1), (env.rank if env.use_dist else 0), env.dev
    json_path = env.scene_json(scene, W, H)
    builder = {"sah": ev.BVH_SAH, "sbvh": ev.BVH_SBVH, "lbvh": ev.BVH_LBVH, "gpu": ev.BVH_LBVH_GPU}[a.bvh]
    nrec = n_light * P
    group = None
    SR = strip_rows_for(a)
    # room for a deal by cost: 150 % of the equal share of blocks (what evplp_group_create gives its ranks)
    nblocks = (H + SR - 1) // SR
    cap_blocks = min(nblocks, (-(-nblocks // max(a.gpus, 1)) * 150 + 99) // 100)
    deal_by_cost = a.deal == "cost" and a.gpus > 1 and wl in ("ir", "evplp", "vsl")
    SPLIT = {"auto": 0, "on": 1, "off": -1}[a.split_light_paths]
    owner = [None]                          # the dealt table (owner rank of every image block), once it exists
    block_cost = [None]                     # ... and the clocked costs it was dealt from (a rank stores its blocks most expensive first)
    if env.group_front_end:
        # one process, the native multi-GPU entry: a.gpus ranks on distinct devices (RCCL) or, when the box has fewer, all on device 0
        devices = list(range(a.gpus)) if env.ndev >= a.gpus else [0] * a.gpus
        group = ev.Group(W, H, n_light, n_vpl, P, a.gpus, devices=devices, strip_rows=SR, bvh_builder=builder, overlap_light_tracing=True, split_light_paths=SPLIT,
                         cut_scratch_bytes=int(a.scratch_gb * 2 ** 30), vsl_mask_bytes=int(a.mask_gb * 2 ** 30))
        group.load_scene_json(json_path)
        ranks = [group.rank(r) for r in range(a.gpus)]
        ctx = ranks[0]
        n_ranks = a.gpus
    else:
        ctx = ev.Context(W, H, n_light, n_vpl, P, device=env.device_index, strip_rank=rank, strip_count=world, strip_rows=SR,
                         bvh_builder=builder, overlap_light_tracing=True, strip_capacity_rows=cap_blocks * SR if world > 1 else 0,
                         cut_scratch_bytes=int(a.scratch_gb * 2 ** 30), vsl_mask_bytes=int(a.mask_gb * 2 ** 30))
        ctx.load_scene_json(json_path)
        ctx.set_stream(env.stream.cuda_stream)
        ranks = [ctx]
        n_ranks = world
    cam = ctx.camera()
    bsr, total_area, _ = ctx.scene_metrics()
    radius0 = 0.003 * bsr if wl != "ir" else 0.0
    vsl_radius0 = max(0.05 * bsr, 0.008) if wl == "vsl" else 0.0

    if group is None:
        # torch owns the buffers that take part in collectives
        records = torch.zeros(nrec * 96 // 4, dtype=torch.float32, device=dev)
        strip = torch.zeros(ctx.local_rows * W * 4, dtype=torch.float32, device=dev)
        pstrip = torch.zeros(ctx.local_rows * W * 4, dtype=torch.float32, device=dev)
        if use_dist:     # (one GPU: the library keeps the records and double-buffers them behind the overlapped light tracing)
            ctx.bind_buffer(ev.BUF_RECORDS, records.data_ptr(), records.numel() * 4)
        ctx.bind_buffer(ev.BUF_VPL_ACCUM, strip.data_ptr(), strip.numel() * 4)
        ctx.bind_buffer(ev.BUF_PHOTON_ACCUM, pstrip.data_ptr(), pstrip.numel() * 4)
        full = torch.zeros(world * strip.numel(), dtype=torch.float32, device=dev) if use_dist else strip
        pfull = torch.zeros(world * pstrip.numel(), dtype=torch.float32, device=dev) if (use_dist and wl != "ir") else pstrip
        # a light-tracing launch is latency-bound (0.26 ms for 1024 paths, 0.25 ms for 128): the paths are traced redundantly by every rank
        # (identical records, no exchange) unless the library's cost model expects a share + the exchange to be faster (--split-light-paths)
        split_paths = use_dist and n_light % world == 0 and (SPLIT > 0 or (SPLIT == 0 and ev.split_model(n_light, P, world)[0]) or env.force_dist)
        per_rank = n_light // world if split_paths else n_light
        chunk = records.numel() // world
        # in place, as evplp_group does it: rank r traced its paths into slice r of its own record buffer
        send = records[rank * chunk:(rank + 1) * chunk] if split_paths else None

    # the jitter of rtcomphoton.h:887,946-952: IndependentSampler(rngOffset = 0), pinned to the reference's sampler (tests/golden/jitter.npz)
    jitters = ev.jitter_sequence(0, warmup + steps + 11, W, H)
    sched = {"radius": radius0, "clamp": 1.0 / total_area, "pdf_mc": (n_vpl / n_light) / math.pi / (radius0 * radius0) if radius0 > 0 else 0.0,
             "vsl_radius": vsl_radius0, "vsl_inv": (1.0 / (math.pi * vsl_radius0 * vsl_radius0)) if vsl_radius0 > 0 else 0.0}
    clamp_start = sched["clamp"]

    # the photon splat's coverage rule: the reference's proxy mesh (the technique loop's default) unless --footprint ideal
    footprint = [a.footprint]
    last_jitter = [(0.0, 0.0)]      # (the G-buffer on the device belongs to this jitter: the proxy rule's eye rays go through it)

    def frame(it, advance=True):
        jitter = (float(jitters[it][0]), float(jitters[it][1]))
        exchange = a.exchange_every > 0 and (it + 1) % a.exchange_every == 0
        last_jitter[0] = jitter
        fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode=mis, pdf_mc=sched["pdf_mc"], clamping_value=sched["clamp"],
                             photon_radius=sched["radius"], vsl_radius=sched["vsl_radius"], vsl_inv_pi_radius2=sched["vsl_inv"],
                             num_light_paths=n_light, num_vpl_light_paths=n_vpl, photons_per_path=P,
                             do_accumulate=1, rng_seed=it, jitter=jitter, splat_footprint=footprint[0])
        if group is not None:
            # the technique loop of host/technique.cpp on the group, + the per-frame exchange of the strips
            if wl == "ppm":
                group.trace_light_paths(it); group.primary(jitter)
            else:
                group.primary(jitter); group.trace_light_paths(it)
            if wl in ("ir", "evplp"):
                group.gather(fp, 0)
            elif wl == "vsl":
                group.gather(fp, 1)
            if wl != "ir":
                group.splat_photons(fp)
            group.present(1.0 / (it + 1), 1.0 / (it + 1), 1.0, mask_emitter=True, gamma=True, exchange=exchange)      # runFinalProgram(param, param, 1, true), rtcomphoton.h:997-1004
        else:
            def light_paths():
                if split_paths:
                    ctx.trace_light_paths(it, rank * per_rank, per_rank)
                    env.all_gather(records, send)
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
            # the frame ends with the composite (rtcomphoton.h:997-1004), as the group front end's present() does
            ctx.present(1.0 / (it + 1), 1.0 / (it + 1), 1.0, mask_emitter=True, gamma=True)
            if use_dist and exchange:
                if wl != "ppm":
                    env.all_gather(full, strip)
                if wl != "ir":
                    env.all_gather(pfull, pstrip)
        if progressive and advance:   # rtcomphoton.h:1033-1063 after numIterations++
            r, c, p, vr, vi = ev.progressive_step(it + 1, 0.7, clamp_start, n_vpl, n_light, sched["radius"], sched["clamp"], sched["pdf_mc"],
                                                  wl == "vsl", sched["vsl_radius"], sched["vsl_inv"])
            sched.update(radius=r, clamp=c, pdf_mc=p, vsl_radius=vr, vsl_inv=vi)

    def sync_all():
        if group is not None:
            group.synchronize()
        else:
            env.sync_all()

    gather_pass = ev.PASS_GATHER_VSL if wl == "vsl" else ev.PASS_GATHER_VPL

    def profile_passes(on):
        # the two HIP events per pass behind pass_stats()["ms"]: config #4's sub-millisecond iterations run their timed region without them
        # (as the technique loop of evplp_render_json does for all iterations but its last); the pass times of the line come from the
        # extra iterations after the timed region, recorded with them
        if group is not None:
            group.profile_passes(on)
        else:
            for c in ranks:
                c.profile_passes(on)
    if deal_by_cost:
        # one calibration frame with the self-clocking gather kernels, then the deal (the accumulators are cleared; the schedule does not move)
        if group is not None:
            group.calibrate(True); frame(0, advance=False); group.rebalance()
            owner[0] = group.block_owners()
            ranks = [group.rank(r) for r in range(a.gpus)]; ctx = ranks[0]
        elif use_dist:
            from evplp_amd import strips as _strips
            ctx.calibrate_blocks(True); frame(0, advance=False); env.sync_all()
            cost = torch.from_numpy(ctx.block_costs().astype(np.int64)).to(dev)
            cost = env.all_reduce(cost, dist.ReduceOp.SUM).cpu().numpy().astype(np.uint64)
            owner[0] = ev.deal_blocks(cost, world, cap_blocks)              # (every process computes the same table)
            block_cost[0] = cost
            ctx.set_blocks(_strips.blocks_of_rank(owner[0], rank, cost)); ctx.calibrate_blocks(False)
    if wl == "ppm":
        profile_passes(False)
    for i in range(warmup):
        frame(i)
    sync_all()
    kernel_ms, nominal_local, rays_local, shaded_local, samples_local, splat_ms, splat_tiles_ms, usable = [], 0, 0, 0, 0, [], [], 0
    feeder_ms = {"light_trace": [], "primary": []}
    other_ms, same_ms, frag_stats = [], [], []
    # Pass statistics synchronise the stream.  The gather workloads (>= 80 ms per step) read them every step.  Config #4's 0.7 ms
    # iterations run their timed region without a single read-back (the host stays an iteration ahead of the GPU, as in the
    # technique loop of evplp_render_json); photon-pixel pairs come from the library's device-side running total, read before and
    # after, and the HIP-event times of the passes from ten more iterations after the timed region.
    pairs_before = sum(c.pass_stats(ev.PASS_SPLAT)["shaded"] for c in ranks) if wl != "ir" else 0

    def read_pass_times(profiled=False):
        # the splat's pass time comes from steps WITHOUT the events around its dominant kernel (they sit between its launches:
        # evplp_profile_kernels), the dominant kernel's own time from a few extra steps with them
        if wl != "ir":
            ss = ctx.pass_stats(ev.PASS_SPLAT)
            if profiled:
                splat_tiles_ms.append(ss["dominant_kernel_ms"])
            else:
                splat_ms.append(ss["ms"])
        if not profiled:
            feeder_ms["light_trace"].append(ctx.pass_stats(ev.PASS_LIGHT_TRACE)["ms"]); feeder_ms["primary"].append(ctx.pass_stats(ev.PASS_PRIMARY)["ms"])
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        frame(warmup + i)
        if wl != "ppm":
            for c in ranks:
                st = c.pass_stats(gather_pass)      # HIP events on the launch stream; syncs that rank's stream
                if c is ranks[0]:
                    kernel_ms.append(st["dominant_kernel_ms"]); usable = st["usable"]
                nominal_local += st["pairs"]; rays_local += st["rays"]; shaded_local += st.get("shaded", 0); samples_local += st.get("samples", 0)
            if i < 10:
                read_pass_times()
    sync_all()
    dt = time.perf_counter() - t0
    if wl == "ppm":
        profile_passes(True)
    splat_pairs = (sum(c.pass_stats(ev.PASS_SPLAT)["shaded"] for c in ranks) - pairs_before) if wl != "ir" else 0
    extra = min(10, len(jitters) - (warmup + steps))
    if wl == "ppm":
        for i in range(extra - 3):
            frame(warmup + steps + i); read_pass_times()
        sync_all()
    if wl != "ir":                     # three more steps with the events around the splat's dominant kernel
        for c in ranks:
            c.profile_kernels(True)
        for i in range(max(extra - (1 if wl == "vsl" else 3), 0), extra):
            frame(warmup + steps + i); read_pass_times(profiled=True)
        sync_all()
        for c in ranks:
            c.profile_kernels(False)
        # ... and the pass under the other coverage rule, on the frames just rendered (three passes, HIP events of the pass)
        footprint[0] = "ideal" if a.footprint == "proxy" else "proxy"
        for i in range(0 if a.dump_frame else 3):          # (not before a frame dump: these passes add to the photon accumulator)
            for c in ranks:
                c.splat_photons(ev.frame_params(camera_pos=list(cam.origin), mis_mode=mis, pdf_mc=sched["pdf_mc"], clamping_value=sched["clamp"], photon_radius=sched["radius"],
                                                num_light_paths=n_light, num_vpl_light_paths=n_vpl, photons_per_path=P, do_accumulate=1, jitter=last_jitter[0],
                                                splat_footprint=footprint[0]))
            sync_all()
            other_ms.append(ctx.pass_stats(ev.PASS_SPLAT)["ms"])
        # (the same pass, same photons and radius, under the timed rule: the like-for-like denominator of the ratio)
        for i in range(0 if a.dump_frame else 3):
            for c in ranks:
                c.splat_photons(ev.frame_params(camera_pos=list(cam.origin), mis_mode=mis, pdf_mc=sched["pdf_mc"], clamping_value=sched["clamp"], photon_radius=sched["radius"],
                                                num_light_paths=n_light, num_vpl_light_paths=n_vpl, photons_per_path=P, do_accumulate=1, jitter=last_jitter[0],
                                                splat_footprint=a.footprint))
            sync_all()
            same_ms.append(ctx.pass_stats(ev.PASS_SPLAT)["ms"])
            frag_stats.append((ctx.pass_stats(ev.PASS_SPLAT)["pairs"], ctx.pass_stats(ev.PASS_SPLAT)["rays"]))
        footprint[0] = a.footprint
    kms_local = sum(kernel_ms) / len(kernel_ms) if kernel_ms else 0.0
    stats = torch.tensor([dt, float(nominal_local), float(rays_local), float(splat_pairs), kms_local, float(shaded_local), float(samples_local)], dtype=torch.float64, device=dev)
    if use_dist:
        mx = env.all_reduce(stats.clone(), dist.ReduceOp.MAX)
        sm = env.all_reduce(stats.clone(), dist.ReduceOp.SUM)
        dt = float(mx[0]); nominal = float(sm[1]); rays = float(sm[2]); spairs = float(sm[3]); kms = float(mx[4]); shaded = float(sm[5]); samples = float(sm[6])
        n_ranks = dist.get_world_size()
    else:
        nominal, rays, spairs, kms, shaded, samples = float(nominal_local), float(rays_local), float(splat_pairs), kms_local, float(shaded_local), float(samples_local)

    if a.dump_frame and primary:
        # the assembled accumulators of the last frame (tests: N ranks == 1 rank, bit for bit)
        if group is not None:
            rows = [c.global_rows() for c in ranks]
            vp = np.zeros((H, W, 4), np.float32); pp = np.zeros((H, W, 4), np.float32)
            for c, gr in zip(ranks, rows):
                ok = gr < H
                vp[gr[ok]] = c.download(ev.BUF_VPL_ACCUM)[ok]; pp[gr[ok]] = c.download(ev.BUF_PHOTON_ACCUM)[ok]
        else:
            def assemble(t):
                t = t.cpu().numpy().reshape(n_ranks if use_dist else 1, ctx.local_rows, W, 4)
                img = np.zeros((H, W, 4), np.float32)
                from evplp_amd import strips as _strips
                for r in range(t.shape[0]):
                    blocks = _strips.blocks_of_rank(owner[0], r, block_cost[0]) if owner[0] is not None else np.arange(r, nblocks, t.shape[0])
                    gr = _strips.rows_of_blocks(H, blocks, SR, ctx.local_rows) if t.shape[0] > 1 else np.arange(ctx.local_rows)
                    ok = gr < H
                    img[gr[ok]] = t[r][ok]
                return img
            if use_dist and (wl == "ppm" or a.exchange_every != 1):
                env.all_gather(full, strip)
                if wl != "ir":
                    env.all_gather(pfull, pstrip)
            if use_dist and wl == "ir":
                pfull = torch.zeros(world * pstrip.numel(), dtype=torch.float32, device=dev); env.all_gather(pfull, pstrip)
            torch.cuda.synchronize(dev)
            vp, pp = assemble(full), assemble(pfull)
        if rank == 0:
            np.save(a.dump_frame, np.stack([vp, pp]))

    out = None
    if rank == 0:
        ms_per_step = dt / steps * 1e3
        total_paths = rays + spairs
        value = total_paths / dt / 1e6
        acc = ctx.accel_info()
        physical = len(set(devices)) if group is not None else (min(n_ranks, env.ndev) if env.backend == "gloo" else n_ranks)
        out = {
            "metric": "Mpaths/s", "value": value, "unit": "Mpaths/s", "n_gpus": n_ranks, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[wl],
                       "scene": f"procedural conference stand-in, style {scene} ({'curved / thin furniture, rotated clutter, ~2400 small occluders' if scene == 'hard' else 'tessellated boxes'}), "
                                f"{a.tris} target triangles, seed 1234 (reference meshes are LFS stubs)",
                       "resolution": [W, H], "num_light_paths": n_light, "num_vpl_light_paths": n_vpl, "photons_per_path": P, "mis_mode": mis,
                       "usable_vpl_records": int(usable), "partition": f"{n_ranks} x interleaved {SR}-row strips" + ((", blocks dealt by clocked cost: " + str(np.bincount(owner[0], minlength=n_ranks).tolist()) + " blocks per rank") if owner[0] is not None else ""),
                       "exchange_every": a.exchange_every,
                       "front_end": "evplp_group (one process, C ABI)" if group is not None else ("one process per rank, torch.distributed " + env.backend if use_dist else "one context"),
                       "physical_gpus": physical,
                       "path_definition": "gather: (pixel, usable VPL record) pair that passes the cosine test = 1 shadow ray; splat: (photon, covered pixel) pair",
                       "pairs_nominal_per_frame": nominal / steps, "mpairs_nominal_per_s": nominal / dt / 1e6,
                       "shadow_rays_per_frame": rays / steps, "unoccluded_pairs_per_frame": shaded / steps,
                       "mrays_per_s": rays / dt / 1e6, "bvh_builder": acc.get("builder", a.bvh), "bvh": acc},
        }
        if physical < n_ranks:
            out["config"]["note"] = f"{n_ranks} ranks share {physical} physical GPU(s): the N > 1 logic with the real kernels, not a scaling measurement"
        feeders = {k: (sum(v) / len(v) if v else None) for k, v in feeder_ms.items()}
        rays_lt = n_light * 1.0       # at least the first segment of every light path; (<= numMaxBounces per path)
        out["feeders"] = {
            "primary": {"ms": feeders["primary"], "grays_per_s": (2.0 * W * H / max(n_ranks, 1)) / (feeders["primary"] * 1e-3) / 1e9 if feeders["primary"] else None,
                        "note": "primary_kernel: jittered scene ray + un-jittered light-mesh ray per pixel of this rank's strip; HIP events of the pass "
                                "(with overlap_light_tracing it may share the GPU with light tracing: the figure is then an upper bound of its cost)"},
            "light_trace": {"ms": feeders["light_trace"], "paths": n_light, "waves_per_simd": (n_light / max(n_ranks if (group is not None or (use_dist and split_paths)) else 1, 1) / 64.0) / 1024.0,
                            "mpaths_per_s": (n_light / 1e6) / (feeders["light_trace"] * 1e-3) if feeders["light_trace"] else None,
                            "note": "light_trace_kernel (tracePhotons + closest hit, lighttracing.cu:192-250): one light path per lane, <= 3 rays per path; "
                                    "waves_per_simd = wavefronts of the launch / 1024 SIMDs -- far below the ~8 a latency-bound walk wants"},
        }
        # the feeders against the only roofline their outputs give them (their work is the BVH walk, which has no algorithmic byte or
        # flop count: SURVEY 8d prices neither): bytes they must write / HIP-event time of the pass
        px_rank = W * H / max(n_ranks, 1)
        for key, nbytes, what in (("primary", px_rank * 80.0, "G-buffer 4 x 16 B + light plane 16 B per pixel of this rank's strip"),
                                  ("light_trace", n_light * float(P) * 96.0 / max(n_ranks if (group is not None or (use_dist and split_paths)) else 1, 1), "96-B record slots written by this rank")):
            ms = out["feeders"][key]["ms"]
            if ms:
                gbs = nbytes / (ms * 1e-3) / 1e9
                out["feeders"][key]["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                                   "algorithmic_bytes": nbytes, "traffic": None,
                                                   "note": what + "; the pass is a latency-bound BVH walk (one wave per 8x8 tile / one light path per lane), far from this bound by construction"}
        if wl != "ppm":
            if wl == "vsl":
                flop_total = samples / steps / max(n_ranks, 1) * FLOP_PER_VSL_SAMPLE      # this rank's launch
                flop = None
            else:
                flop = FLOP_PER_PAIR if mis in ("one", "geometryClamp", "geometryBrdfClamp") else FLOP_PER_PAIR_MIS
                flop_total = (rays_local / len(ranks)) / steps * flop
            kname = "gather_vsl_walk_kernel + gather_vsl_shade_kernel" if wl == "vsl" else "gather_vpl_kernel"
            achieved = flop_total / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
            out["roofline"] = {
                "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": None, "traffic_source": "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/prof_all.sh); not measured inside this run",
                "kernel": kname, "kernel_ms": kms}
            if wl == "vsl":
                out["roofline"].update({
                    "flop_per_sample_iteration": FLOP_PER_VSL_SAMPLE, "sample_iterations_per_frame": samples / steps, "lit_pairs_per_frame": shaded / steps,
                    "note": "fp32 VECTOR-ALU bound.  achieved = sample-iterations of the three MIS-combined estimators counted ON THE DEVICE "
                            "(lighttracing.cu:632-640) x 540 flop (SURVEY 8d: 3 x ~180; the three generator steps per iteration and the BVH walks are "
                            "overhead, not counted) / HIP-event kernel time."})
            else:
                out["roofline"].update({
                    "flop_per_pair": flop,
                    "frac_nominal_pairs": (nominal_local / len(ranks) / steps) * flop / (kms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if kms > 0 else 0.0,
                    "note": "fp32 VECTOR-ALU bound (BVH packet traversal + shading; not GEMM-shaped). achieved = algorithmic flop per evaluated pair "
                            "(SURVEY 8d) x pairs that trace a shadow ray per launch / HIP-event kernel time; traversal flops are overhead and not counted. "
                            "frac_nominal_pairs prices every (pixel, usable VPL) loop iteration instead (round-1 definition)."})
            n_usable = usable
            alg_bytes = (W * H * (64 + 16 + 16)) / max(n_ranks, 1) + n_usable * 96
            tpath = os.path.join(ROOT, "profiles", "traffic_gather_vpl.json")
            if os.path.exists(tpath) and wl == "ir":
                tj = json.load(open(tpath))
                if tj.get("config") == f"{scene}:{W}x{H}:{n_vpl}:{n_ranks}":
                    out["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                    out["roofline"]["traffic_ratio"] = tj.get("hbm_bytes_per_launch") / alg_bytes if tj.get("hbm_bytes_per_launch") else None
                    out["roofline"]["traffic_source"] = f"profiles/traffic_gather_vpl.json ({tj.get('round', 'committed')} PMC summary of this configuration; not measured inside this run)"
                    if tj.get("valu_active_frac") is not None:      # "issue-saturated" as a number: SIMD cycles in which a vector instruction executes
                        out["roofline"]["valu_active_frac"] = tj["valu_active_frac"]; out["roofline"]["valu_lane_utilisation"] = tj.get("valu_lane_utilisation")
            out["roofline"]["algorithmic_bytes"] = alg_bytes
            out["roofline_hbm"] = {"bound": "hbm", "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": alg_bytes / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS, "kernel": kname,
                                   "note": "the gather priced against HBM (north star: fraction of HBM roofline): tiny by construction, the kernel is issue-bound"}
        if wl != "ir" and splat_ms:
            nrec_bytes = nrec * 96 + (W * H * 64 + W * H * 24) / max(n_ranks, 1)          # SURVEY 8(d) algorithmic bytes per frame (per rank)
            sms = sum(splat_ms) / len(splat_ms)
            rs = {"bound": "hbm", "achieved": nrec_bytes / (sms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": nrec_bytes / (sms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None, "kernel": "photon splat pass (bin + scatter + tiles)", "pass_ms": sms,
                  "tiles_kernel_ms": (sum(splat_tiles_ms) / len(splat_tiles_ms)) if splat_tiles_ms else None, "pairs_per_frame": spairs / steps, "algorithmic_bytes": nrec_bytes,
                  "footprint": a.footprint}
            if other_ms and same_ms:
                # both coverage rules on the same frame (last radius of the run): pass time by HIP events, mean of three
                o_ms, s_ms = sum(other_ms) / len(other_ms), sum(same_ms) / len(same_ms)
                other = "ideal" if a.footprint == "proxy" else "proxy"
                rs["footprints"] = {a.footprint: {"pass_ms": s_ms, "frac": nrec_bytes / (s_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
                                    other: {"pass_ms": o_ms, "frac": nrec_bytes / (o_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
                                    "proxy_over_ideal": (s_ms / o_ms) if a.footprint == "proxy" else (o_ms / s_ms),
                                    "pairs": frag_stats[-1][0], "proxy_fragments": frag_stats[-1][1] if a.footprint == "proxy" else None,
                                    "note": "proxy = the reference's rule: one fragment per face of the radius-scaled proxy mesh (generated 42-vertex icosphere) the eye ray "
                                            "crosses in front of the surface, un-culled, depth-tested (rtcomphoton.h:653-655, 789-837); ideal = the radius test alone"}
            tpath = os.path.join(ROOT, "profiles", "traffic_splat.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath)).get("configs", {}).get(f"{wl}:{scene}:{W}x{H}:{n_ranks}")
                if tj:
                    rs["traffic"] = tj.get("hbm_bytes_per_pass"); rs["traffic_ratio"] = tj.get("hbm_bytes_per_pass") / nrec_bytes if tj.get("hbm_bytes_per_pass") else None
                    rs["traffic_source"] = "profiles/traffic_splat.json (committed PMC summary; not measured inside this run)"
            if wl == "ppm":
                out["roofline"] = rs
            else:
                out["roofline_splat"] = rs
    if group is not None:
        group.close()
    else:
        ctx.close()
    return out, json_path, (W, H, n_vpl, n_light, mis)


def technique_block(wl, n_vpl, n_light, mis, iterations, tag, footprint="proxy"):
    """The `photonfam` block of the scene file for this configuration (keys of rtcomphoton.h:107-223)."""
    block = {"rngOffset": 0, "numMaxIteration": iterations, "timeLimitMs": 1000000000, "frameMode": "accumulate", "renderMode": "vpl", "misMode": mis,
             "combinedFilename": f"{tag}_combined.pfm", "weightedPhotonFilename": f"{tag}_weightedpm.pfm", "weightedVplFilename": f"{tag}_weightedvpl.pfm",
             "statFilename": f"{tag}_stat.json", "useJitter": True, "useStat": True,
             "numLightPaths": n_light, "numVplLightPaths": n_vpl, "numMaxBounces": 3, "radiusPercentage": 0.0 if wl == "ir" else 0.003,
             "splatFootprint": footprint}
    if wl == "ir":
        block["run"] = {"photonSplat": False}
    if wl in ("ppm", "vsl"):
        block.update(DoProgressive=True, AlphaProgressive=0.7)
    if wl == "vsl":
        block.update(forceVsl=True, vslRadiusPercentage=0.05)
    return block


import contextlib


@contextlib.contextmanager
def c_stdout_to_stderr():
    """The technique loop prints the reference's progress lines with printf (rtcomphoton.h:1057-1062): keep them off this process's
    stdout, which carries ONE JSON line."""
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush(); libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def render_json_time(env, wl, json_path, shape, iterations):
    """The same configuration through evplp_render_json -- what a maintainer of the reference binds in place of
    RtComPhoton::render (host/technique.cpp: the technique loop on an evplp_group of one rank).  Time per iteration from the
    stat file the loop writes ({time, numIterations}, rtcomphoton.h:1109-1119)."""
    W, H, n_vpl, n_light, mis = shape
    root = json.load(open(json_path))
    tag = f"bench_{wl}_{os.getpid()}"
    root["photonfam"] = technique_block(wl, n_vpl, n_light, mis, iterations, tag, env.a.footprint)
    for k in ("pt", "lvcphotonfam"):
        root.pop(k, None)
    d = os.path.dirname(json_path)
    jp = os.path.join(d, tag + ".json")
    json.dump(root, open(jp, "w"))
    t0 = time.perf_counter()
    with c_stdout_to_stderr():
        env.ev.render_json(jp, None, env.device_index)
    wall = time.perf_counter() - t0
    stat = json.load(open(os.path.join(d, f"{tag}_stat.json")))
    for f in (jp, os.path.join(d, f"{tag}_stat.json"), os.path.join(d, f"{tag}_combined.pfm"), os.path.join(d, f"{tag}_weightedpm.pfm"), os.path.join(d, f"{tag}_weightedvpl.pfm")):
        try:
            os.remove(f)
        except OSError:
            pass
    res = {"ms_per_iteration": stat["time"] / max(stat.get("numIterations", iterations), 1), "iterations": stat.get("numIterations", iterations),
           "loop_ms": stat["time"], "call_wall_ms": wall * 1e3,
           "note": "evplp_render_json on the same scene file with this configuration's photonfam block: time of the technique loop (first iteration's "
                   "allocations included) / numIterations, from the stat file; call_wall_ms adds scene load, BVH build and the three PFM outputs"}
    res["last_pass_ms"] = {k[:-2]: v for k, v in stat.items() if k.endswith("Ms")}     # HIP-event time of each pass of the LAST iteration
    return res


def count_devices():
    """GPUs this process could use, counted WITHOUT touching the HIP runtime in THIS process (spawn_ranks' contract: the parent has not
    initialised a GPU): a short-lived child process asks the runtime; the kernel driver's topology is the fallback."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:      # noqa: BLE001
        pass
    n = 0
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:          # (CPU nodes have none)
                n += 1
    except OSError:
        n = 0
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
    if vis is not None and vis.strip() != "":
        n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
    return n


def group_under_launcher(a):
    """`--front-end group` when a launcher has started one process per GPU (the driver's N > 1 command): the measurement is rank 0's
    -- one host thread driving evplp_group over the N devices, as a maintainer's binding would -- and the other processes only keep
    the launcher's contract (they meet rank 0 at a host-side barrier before and after, and exit 0).  Returns True when this process
    is done (ranks > 0), False when it is rank 0 and should go on.  If rank 0 cannot open the group (RCCL communicator across the N
    devices), every process falls back to the `ranks` front end."""
    import torch.distributed as dist
    import datetime
    rank = int(os.environ.get("RANK", "0"))
    # (the idle ranks wait at a barrier while rank 0 runs the whole benchmark -- extras, CPU baseline, render_json: hours, not the
    # default 30 minutes, before they may give up and take rank 0 down with them)
    dist.init_process_group("gloo", timeout=datetime.timedelta(hours=6))
    ok = [1]
    if rank == 0:
        try:
            import evplp_amd as ev
            g = ev.Group(64, 64, 64, 64, P, a.gpus, devices=list(range(a.gpus)), strip_rows=strip_rows_for(a))      # opens RCCL on the N devices
            g.close()
        except Exception as e:      # noqa: BLE001 -- any failure means "use the other front end"
            sys.stderr.write(f"bench.py: evplp_group on {a.gpus} devices failed ({e}); falling back to --front-end ranks\n")
            ok = [0]
    dist.broadcast_object_list(ok, src=0)
    if not ok[0]:
        dist.destroy_process_group()
        a.front_end = "ranks"
        return False
    if rank != 0:
        dist.barrier()              # rank 0 has finished its measurement and printed
        dist.destroy_process_group()
        return True
    a._launcher_group = dist       # rank 0 releases the others at the end
    return False


def main():
    a = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if a.front_end == "auto":
        if a.gpus == 1 or os.environ.get("EVPLP_BENCH_BACKEND") == "gloo" or os.environ.get("EVPLP_BENCH_FORCE_DIST") == "1":
            a.front_end = "ranks"
        else:
            a.front_end = "group" if count_devices() >= a.gpus else "ranks"
    if a.gpus > 1 and world_env is None and a.front_end == "ranks":
        spawn_ranks(a)
        return
    if a.gpus > 1 and world_env is not None and a.front_end == "group":
        if group_under_launcher(a):
            return
        if a.front_end == "group":          # rank 0: the group ignores the launcher's process group
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
                os.environ.pop(k, None)
    # This process's stdout carries ONE line, the JSON.  Everything else that writes to file descriptor 1 -- RCCL's version banner at
    # communicator start-up, the technique loop's progress lines (printf), Python prints -- goes to stderr: descriptor 1 is pointed
    # at stderr for the whole run and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    env = Env(a)
    wl = a.workload
    out, json_path, shape = run_workload(env, wl, a.steps, a.warmup, a.scene)
    rank0 = env.rank == 0 or not env.use_dist
    single = rank0 and not env.use_dist and not env.group_front_end

    # ---- secondary measurements (one GPU): the path a maintainer binds, configs #3 / #4 / #5, the other scene style, the
    # 16 384-slot reading of "4096 VPLs", the GPU path tracer
    if single and not a.no_extras and not env.force_dist:
        ev, torch = env.ev, env.torch
        it = {"ir": 20, "evplp": 10, "ppm": 400, "vsl": 2}[wl]
        rj = render_json_time(env, wl, json_path, shape, it)
        rj["ratio_to_ms_per_step"] = rj["ms_per_iteration"] / out["ms_per_step"]
        out["render_json"] = rj
    if single and wl == "ir" and not a.no_extras and not env.force_dist:
        def brief(o, keys=("ms_per_step", "value", "steps", "roofline", "roofline_splat", "feeders")):
            r = {k: o[k] for k in keys if k in o}
            r["config"] = {k: o["config"][k] for k in ("workload", "resolution", "num_light_paths", "num_vpl_light_paths", "mis_mode", "usable_vpl_records")}
            return r
        o3, jp3, sh3 = run_workload(env, "evplp", 20, 2, a.scene, primary=False)
        out["evplp"] = brief(o3)
        o4, jp4, sh4 = run_workload(env, "ppm", 100, 5, a.scene, primary=False)
        out["ppm"] = brief(o4)
        out["ppm"]["ms_per_iteration"] = o4["ms_per_step"]
        rj4 = render_json_time(env, "ppm", jp4, sh4, 400)
        rj4["ratio_to_ms_per_step"] = rj4["ms_per_iteration"] / o4["ms_per_step"]
        out["ppm"]["render_json"] = rj4
        o5, _, _ = run_workload(env, "vsl", 2, 1, a.scene, primary=False)
        out["vsl"] = brief(o5)
        out["vsl"]["ms_per_iteration"] = o5["ms_per_step"]

        W, H, n_vpl, n_light, mis = shape
        builder = {"sah": ev.BVH_SAH, "sbvh": ev.BVH_SBVH, "lbvh": ev.BVH_LBVH, "gpu": ev.BVH_LBVH_GPU}[a.bvh]

        def quick_ir(path, nv, steps=5):
            c = ev.Context(W, H, nv, nv, P, device=env.device_index, strip_rows=STRIP_ROWS, bvh_builder=builder, overlap_light_tracing=True)
            c.load_scene_json(path)
            c.set_stream(env.stream.cuda_stream)
            cm = c.camera()
            _, ta, _ = c.scene_metrics()
            ms, rr, nom, us = [], 0, 0, 0
            for it in range(steps + 1):
                fp = ev.frame_params(camera_pos=list(cm.origin), mis_mode=mis, clamping_value=1.0 / ta, num_light_paths=nv, num_vpl_light_paths=nv,
                                     photons_per_path=P, do_accumulate=1, rng_seed=it)
                torch.cuda.synchronize(env.dev); t = time.perf_counter()
                c.primary((0.0, 0.0)); c.trace_light_paths(it); c.gather_vpl(fp); c.synchronize()
                if it:
                    ms.append((time.perf_counter() - t) * 1e3)
                    s = c.pass_stats(ev.PASS_GATHER_VPL); rr += s["rays"]; nom += s["pairs"]; us = s["usable"]
            res = {"ms_per_frame": sum(ms) / len(ms), "mpaths_per_s": rr / (sum(ms) * 1e-3) / 1e6, "mpairs_nominal_per_s": nom / (sum(ms) * 1e-3) / 1e6, "usable_vpl_records": int(us)}
            return c, res
        other = "easy" if a.scene == "hard" else "hard"
        c2, r2 = quick_ir(env.scene_json(other, W, H), n_vpl)
        out["scene_" + other] = r2
        c2.close()
        c3, r3 = quick_ir(json_path, 4 * n_vpl, steps=3)
        out["slots_16384_variant"] = dict(r3, note="'4096 VPLs' read as 4096 light PATHS (16384 record slots), SURVEY 8d")
        # GPU path tracer on the same context (same unit as cpu_baseline)
        cam = c3.camera()
        pt_ms, pt_paths = [], 0
        for it in range(6):
            c3.primary((0.0, 0.0)); c3.path_trace(list(cam.origin), it, 3, accumulate=True); c3.synchronize()
            s = c3.pass_stats(ev.PASS_PATH_TRACE)
            if it:
                pt_ms.append(s["ms"]); pt_paths += s["pairs"]
        out["gpu_path_tracer_mpaths_s"] = pt_paths / (sum(pt_ms) * 1e-3) / 1e6
        c3.close()
    if single and not a.no_cpu_baseline:
        base, like = cpu_baseline(json_path, min(shape[0], shape[1]), a.cpu_iters)
        out["cpu_baseline"] = base
        out["cpu_baseline_like_for_like"] = like

    # A compact digest of every roofline of the line: inside `roofline` (which the driver's record keeps) and once more as the LAST key
    # (the driver's 2 KB stdout tail), so that the fractions of configs #3 / #4 / #5 are legible without the long notes in between.
    if rank0 and out is not None:
        def r3(x):
            return None if x is None else float(f"{x:.4g}")
        digest = {}
        for name, o in ((wl, out),) + tuple((k, out[k]) for k in ("evplp", "ppm", "vsl") if k in out and k != wl):
            d = {"ms_per_step": r3(o.get("ms_per_step"))}
            rf = o.get("roofline")
            if rf:
                d.update(bound=rf.get("bound"), frac=r3(rf.get("frac")), kernel_ms=r3(rf.get("kernel_ms", rf.get("pass_ms"))))
                if rf.get("frac_nominal_pairs") is not None:
                    d["frac_nominal_pairs"] = r3(rf.get("frac_nominal_pairs"))
            rs = o.get("roofline_splat")
            if rs:
                d.update(splat_frac=r3(rs.get("frac")), splat_ms=r3(rs.get("pass_ms")))
            fd = o.get("feeders") or {}
            if name == "ppm" and fd:
                d.update(light_trace_ms=r3((fd.get("light_trace") or {}).get("ms")), primary_ms=r3((fd.get("primary") or {}).get("ms")))
            digest[name] = d
        if "roofline" in out:
            out["roofline"]["digest"] = digest
        out["summary"] = digest

    # The JSON line must be the LAST thing on stdout: RCCL prints a version banner through C stdio, which is
    # block-buffered on a pipe and would otherwise surface after it at exit.  Every rank pushes its C buffers
    # out, all ranks meet, then rank 0 prints.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if env.use_dist:
        env.dist.barrier()
        env.dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank0:
        if env.use_dist and env.world > 1:
            time.sleep(1.0)       # let the other ranks' processes drain whatever they still print while exiting
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if getattr(a, "_launcher_group", None) is not None:      # --front-end group under a launcher: let the idle ranks go
        a._launcher_group.barrier()
        a._launcher_group.destroy_process_group()


if __name__ == "__main__":
    main()
