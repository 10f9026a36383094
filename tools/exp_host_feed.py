"""Host time of config #4's iteration through evplp_group, for 1 / 4 / 8 VIRTUAL ranks on one GPU (profiles/r05_host_feed.txt).
What it measures: how long the CALLER's thread is busy per iteration (wall time of the five group calls, no synchronisation in
between) against the time an iteration takes until it is done on the GPU.  Virtual ranks share one device, so the GPU time per
iteration is the sum of all strips' work, NOT what n devices would take: the number that carries over to n GPUs is the host's
time per iteration, which must stay well below a rank's GPU time per iteration (0.15-0.6 ms at config #4).
(round 6) EXCHANGE_EVERY=k in the environment: the strips are all-gathered in every k-th iteration's composite only (0 = never in the loop;
evplp_group_present_ex, the technique loop's "device": {"exchangeEvery": k}) -- the other iterations composite locally, no host barrier.
usage: [EXCHANGE_EVERY=k] python tools/exp_host_feed.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import evplp_amd as ev

W, H, NL, P = 1920, 1080, 300000, 4
label = sys.argv[1] if len(sys.argv) > 1 else ""
EX = int(os.environ.get("EXCHANGE_EVERY", "1"))
jp = ev.synth_scene("/tmp/evplp_host_t", "conf", 331000, 1234, W, H, style="hard")
print(f"# {label}: config #4 (1920 x 1080, 300 000 light paths, photon splat, composite per iteration, all-gather of the strips every {EX} iteration(s) (0 = never)), evplp_group, virtual ranks on device 0")
for n in (1, 4, 8):
    with ev.Group(W, H, NL, 0, P, n, devices=[0] * n, strip_rows=16, overlap_light_tracing=True) as g:
        g.load_scene_json(jp)
        c0 = g.rank(0)
        cam = c0.camera(); bsr, total, _ = c0.scene_metrics()
        fp = ev.frame_params(camera_pos=list(cam.origin), mis_mode="one", photon_radius=0.003 * bsr, num_light_paths=NL, num_vpl_light_paths=0, photons_per_path=P,
                             do_accumulate=1, splat_footprint="proxy")

        def iteration(it):
            g.trace_light_paths(it); g.primary((0.0, 0.0)); g.splat_photons(fp); g.present(1.0 / (it + 1), 1.0 / (it + 1), 1.0, mask_emitter=True, gamma=True, exchange=EX > 0 and (it + 1) % EX == 0)
        for it in range(10):
            iteration(it)
        g.synchronize()
        # (a) what a group call costs the caller while the rings have room: 12 iterations = 48 posted commands per rank, right after a drain
        t0 = time.perf_counter()
        for it in range(12):
            iteration(it)
        post_ms = (time.perf_counter() - t0) / 12 * 1e3
        g.synchronize()
        before = [g.host_stats(r) for r in range(n)] if hasattr(g, "host_stats") else None
        N = 200
        host = []
        t_all = time.perf_counter()
        for it in range(N):
            t0 = time.perf_counter(); iteration(it); host.append(time.perf_counter() - t0)
        t_enq = time.perf_counter() - t_all
        g.synchronize()
        t_tot = time.perf_counter() - t_all
        h = np.array(host[20:]) * 1e3
        if before is not None:
            after = [g.host_stats(r) for r in range(n)]
            calls = [(a["calls_ms"] - b["calls_ms"]) / N for a, b in zip(after, before)]; exch = [(a["exchange_ms"] - b["exchange_ms"]) / N for a, b in zip(after, before)]
            print(f"ranks {n}: caller posts an iteration in {post_ms * 1e3:.1f} us (rings not full); per-rank worker inside its pass calls {np.mean(calls) * 1e3:.1f} us per iteration "
                  f"(max over ranks {np.max(calls) * 1e3:.1f}), inside exchanges {np.mean(exch) * 1e3:.1f} us (virtual ranks: includes the stream waits of the device copies)")
        else:
            print(f"ranks {n}: caller issues an iteration in {post_ms * 1e3:.1f} us when it does not have to wait for the GPU")
        print(f"ranks {n}: caller busy per iteration median {np.median(h):.4f} ms mean {h.mean():.4f} ms max {h.max():.3f} ms | {N} iterations posted in {t_enq * 1e3:.1f} ms, "
              f"done after {t_tot * 1e3:.1f} ms = {t_tot / N * 1e3:.3f} ms per iteration on the one GPU all ranks share")
