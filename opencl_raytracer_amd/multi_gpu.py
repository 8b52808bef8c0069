"""Band ownership and image assembly for multi-GPU frames.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests).  The image is cut into bands of lcm(8, n)/n output rows
(n = supersample grid side) that are dealt round-robin to the ranks -- see
ocrt::Partition / rt_partition_global_row in include/rt_hip.h; cost per row is
very uneven (background vs model), hence interleaving.  Every rank renders and
box-filters its own bands into a compact uint8 buffer; ONE gather moves them to
rank 0, which scatters the rows to their place.  There is no other exchange
step in this path (the scene is replicated), so no other collective is used.

torch is used for the collective and for device memory only.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .api import Options, partition_rows


class BandLayout:
    """Where each rank's compact band rows go in the final image."""

    def __init__(self, options: Options, world: int):
        self.options = options
        self.world = world
        self.rows: List[np.ndarray] = [partition_rows(options, r, world) for r in range(world)]
        self.max_rows = max(int(r.size) for r in self.rows)
        src, dst = [], []
        for r, rows in enumerate(self.rows):
            keep = np.flatnonzero(rows < options.height)
            src.append(r * self.max_rows + keep)
            dst.append(rows[keep])
        self.src_index = np.concatenate(src)
        self.dst_index = np.concatenate(dst)
        if np.sort(self.dst_index).tolist() != list(range(options.height)):
            raise AssertionError("band partition does not tile the image")

        self._device_indices = {}

    def local_rows(self, rank: int) -> int:
        return int(self.rows[rank].size)

    def indices_on(self, device):
        """(src, dst) row-index tensors on `device`, uploaded once."""
        import torch

        key = str(device)
        if key not in self._device_indices:
            self._device_indices[key] = (torch.as_tensor(self.src_index, device=device),
                                         torch.as_tensor(self.dst_index, device=device))
        return self._device_indices[key]


def gather_bands(band, layout: BandLayout, rank: int, group=None):
    """Gathers every rank's (max_rows, width) uint8 band buffer to rank 0 and
    returns the assembled (height, width) image there (None elsewhere)."""
    import torch
    import torch.distributed as dist

    src, dst = layout.indices_on(band.device)
    if layout.world == 1:
        return band.index_select(0, src)
    gather_list: Optional[list] = None
    if rank == 0:
        gather_list = [torch.empty_like(band) for _ in range(layout.world)]
    dist.gather(band, gather_list, dst=0, group=group)
    if rank != 0:
        return None
    stacked = torch.cat(gather_list, dim=0)
    final = torch.empty((layout.options.height, layout.options.width), dtype=band.dtype, device=band.device)
    final.index_copy_(0, dst, stacked.index_select(0, src))
    return final
