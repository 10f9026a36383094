"""The N > 1 path on CPU: two processes over the gloo backend run the same strip / collective
orchestration as bench.py (each rank traces its slice of the light paths, in-place all-gather of the
record set, each rank gathers its own interleaved row strips, all-gather of the framebuffer strips).
The per-strip compute is done by the CPU oracle here (it is the checker standing in for the kernel on a
box without GPUs); the assembled frame must equal the single-process frame bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

W, H, NPATHS, P, STRIP = 48, 32, 32, 4, 8


def _frame_params(oa, room):
    return dict(camera_pos=room.cam_origin, mis_mode=1, pdf_mc=0.4, photon_radius=0.4, num_light_paths=NPATHS,
                num_vpl_light_paths=NPATHS, photons_per_path=P, rng_seed=4, jitter=(0.002, 0.001))


def _worker(rank, world, port, out_path, deal="roundRobin"):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_api as oa
    import scenes
    from evplp_amd import strips
    oa.load().evo_set_threads(2)
    room = scenes.box_room(seed=5, n_boxes=3, tess=1, aspect=W / H)
    osc = oa.Scene(room)
    kw = _frame_params(oa, room)
    # 1. light tracing: each rank a slice, in-place all-gather of the record buffer
    begin, count, split = strips.path_slice(NPATHS, rank, world)
    records = np.zeros(NPATHS * P, oa.RECORD_DTYPE)
    osc.trace_light_paths(4, NPATHS, P, begin=begin, count=count, records=records)
    rec_t = torch.from_numpy(records.view(np.float32))
    if split:
        chunk = rec_t.numel() // world
        dist.all_gather_into_tensor(rec_t, rec_t[rank * chunk:(rank + 1) * chunk].clone())
    # 2. this rank's strips (rows it owns), compact local buffers
    rows = strips.global_rows(H, rank, world, STRIP)
    lr = strips.local_rows(H, world, STRIP)
    owner = None
    if deal == "cost":
        # the deal by cost as bench.py's rank processes make it: every rank has clocked ITS blocks (here: a stand-in cost, the lit pixels of
        # the block's rows), the per-block costs are summed over the ranks, and every process computes the same table from them with the
        # library's own evplp_deal_blocks (host only); capacity = 150 % of the equal share
        import evplp_amd
        nblocks = (H + STRIP - 1) // STRIP
        g0 = osc.primary(W, H, kw["jitter"])
        mine = np.zeros(nblocks, np.int64)
        for r in rows[rows < H]:
            mine[int(r) // STRIP] += 1 + 37 * int((g0[0][int(r), :, 3] != 0).sum()) * (1 + (int(r) // STRIP) % 3)
        cost_t = torch.from_numpy(mine)
        dist.all_reduce(cost_t)
        cap = min(nblocks, ((nblocks + world - 1) // world * 150 + 99) // 100)
        block_cost = cost_t.numpy().astype(np.uint64)
        owner = evplp_amd.deal_blocks(block_cost, world, cap)
        my_blocks = strips.blocks_of_rank(owner, rank, block_cost)            # (most expensive first: the order a rank stores and launches them)
        assert my_blocks.tolist() == evplp_amd.rank_blocks(block_cost, owner, rank).tolist()
        lr = int(max((owner == q).sum() for q in range(world))) * STRIP          # equal chunks: the fullest rank's rows
        rows = strips.rows_of_blocks(H, my_blocks, STRIP, lr)
    g = osc.primary(W, H, kw["jitter"])            # replicated G-buffer (cheap here); only own rows are used below
    vpl_full = np.zeros((H, W, 4), np.float32); pm_full = np.zeros((H, W, 4), np.float32)
    for r in rows[rows < H]:
        osc.gather(oa.frame_params(**kw), W, H, g, records, out=vpl_full, rows=(int(r), int(r) + 1))
        oa.splat(oa.frame_params(**kw), W, H, g, records, out=pm_full, rows=(int(r), int(r) + 1))
    local = np.zeros((lr, W, 8), np.float32)
    ok = rows < H
    local[ok, :, :4] = vpl_full[rows[ok]]; local[ok, :, 4:] = pm_full[rows[ok]]
    # 3. framebuffer all-gather (equal, padded chunks) + de-interleave
    gathered = torch.zeros(world * local.size, dtype=torch.float32)        # flat, like bench.py
    dist.all_gather_into_tensor(gathered, torch.from_numpy(local).reshape(-1))
    if owner is None:
        frame = strips.assemble(gathered.numpy().reshape((world,) + local.shape), H, world, STRIP)
    else:
        frame = strips.assemble_blocks(gathered.numpy().reshape((world,) + local.shape), H, owner, STRIP, block_cost)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world, deal", [(2, "roundRobin"), (2, "cost")])
def test_two_rank_strips_equal_single_process(world, deal, tmp_path):
    sys.path.insert(0, HERE)
    import oracle_api as oa
    import scenes
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + (7 if deal == "cost" else 0)
    mp.spawn(_worker, args=(world, port, out, deal), nprocs=world, join=True)
    frame = np.load(out)
    # single-process reference of the same frame
    room = scenes.box_room(seed=5, n_boxes=3, tess=1, aspect=W / H)
    osc = oa.Scene(room)
    kw = _frame_params(oa, room)
    records = osc.trace_light_paths(4, NPATHS, P)
    g = osc.primary(W, H, kw["jitter"])
    vpl, _ = osc.gather(oa.frame_params(**kw), W, H, g, records)
    pm, _ = oa.splat(oa.frame_params(**kw), W, H, g, records)
    assert vpl[..., :3].max() > 0 and pm[..., :3].max() > 0
    assert frame[..., :4].tobytes() == vpl.tobytes(), "VPL strips do not reassemble to the single-process frame"
    assert frame[..., 4:].tobytes() == pm.tobytes(), "photon strips do not reassemble to the single-process frame"
