"""Row-strip geometry of the multi-GPU partition (pure index math, no GPU needed).

Mirrors evplp::StripDev::global_row (evplp_amd/csrc/evplp_types.h): the image is cut into blocks
of `strip_rows` rows; rank r owns the blocks b with b % count == r and stores them compactly.
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
