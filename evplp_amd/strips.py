"""Row-strip geometry of the multi-GPU partition (pure index math, no GPU needed).

Mirrors evplp::StripDev::global_row (evplp_amd/csrc/evplp_types.h): the image is cut into blocks
of `strip_rows` rows; rank r owns the blocks b with b % count == r and stores them compactly -- or, after a deal
by cost (evplp_deal_blocks / evplp_set_blocks), the blocks its table lists (rows_of_blocks, assemble_blocks).
Every rank's buffer has the same (padded) number of rows so that an all-gather moves equal chunks.
"""
from __future__ import annotations

import numpy as np


def local_rows(H: int, count: int, strip_rows: int) -> int:
    if count == 1:
        return ((H + 7) // 8) * 8
    nblocks = (H + strip_rows - 1) // strip_rows
    owned = (nblocks + count - 1) // count
    return owned * strip_rows


def effective_strip_rows(H: int, count: int, strip_rows: int) -> int:
    return ((H + 7) // 8) * 8 if count == 1 else strip_rows


def global_rows(H: int, rank: int, count: int, strip_rows: int) -> np.ndarray:
    """Global image row of every local row of `rank` (values >= H are padding rows)."""
    sr = effective_strip_rows(H, count, strip_rows)
    l = np.arange(local_rows(H, count, strip_rows))
    blk = l // sr
    return (blk * count + rank) * sr + (l - blk * sr)


def rows_of_blocks(H: int, blocks, strip_rows: int, n_local_rows: int) -> np.ndarray:
    """Global image row of every local row of a context that stores the image blocks `blocks` in this order (a dealt block table,
    evplp_set_blocks; or what the round-robin deal gives); local rows beyond them hold nothing (values >= H)."""
    rows = H + np.arange(n_local_rows)
    for l, b in enumerate(np.asarray(blocks).tolist()):
        rows[l * strip_rows:(l + 1) * strip_rows] = b * strip_rows + np.arange(strip_rows)
    return rows


def blocks_of_rank(owner, rank: int, cost=None) -> np.ndarray:
    """The blocks a deal (owner[b] = rank of image block b, evplp_deal_blocks) gives `rank`, in the order the rank stores and launches them
    (mirrors evplp_rank_blocks): the most expensive first when the costs are given -- ties by block index -- image order otherwise."""
    mine = np.nonzero(np.asarray(owner) == rank)[0].astype(np.int32)
    if cost is not None:
        mine = np.array(sorted(mine.tolist(), key=lambda b: (-int(cost[b]), b)), dtype=np.int32)
    return mine


def assemble_blocks(gathered: np.ndarray, H: int, owner, strip_rows: int, cost=None) -> np.ndarray:
    """gathered: [count, chunk_rows, W, C] all-gathered strips of a dealt partition -> [H, W, C] (mirrors assemble_strips_kernel).
    cost: what the ranks ordered their blocks by (blocks_of_rank)."""
    count, chunk_rows, W, Cc = gathered.shape
    out = np.zeros((H, W, Cc), dtype=gathered.dtype)
    for r in range(count):
        rows = rows_of_blocks(H, blocks_of_rank(owner, r, cost), strip_rows, chunk_rows)
        ok = rows < H
        out[rows[ok]] = gathered[r][ok]
    return out


def path_slice(n_paths: int, rank: int, count: int):
    """Light paths traced by `rank` when the record set is shared by an all-gather (needs equal chunks)."""
    if n_paths % count != 0:
        return 0, n_paths, False   # every rank traces everything (identical seeds, no collective)
    per = n_paths // count
    return rank * per, per, True


def assemble(gathered: np.ndarray, H: int, count: int, strip_rows: int) -> np.ndarray:
    """gathered: [count, local_rows, W, C] as produced by all_gather of the per-rank strips ->
    [H, W, C] full frame (y = 0 bottom)."""
    count_, lr, W, C = gathered.shape
    assert count_ == count and lr == local_rows(H, count, strip_rows)
    out = np.zeros((H, W, C), dtype=gathered.dtype)
    for r in range(count):
        rows = global_rows(H, r, count, strip_rows)
        ok = rows < H
        out[rows[ok]] = gathered[r][ok]
    return out
