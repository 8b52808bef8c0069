"""Band ownership and image assembly for multi-GPU frames.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests).  The image is cut into bands of lcm(8, n)/n output rows
(n = supersample grid side) that are dealt round-robin to the ranks -- see
ocrt::Partition / rt_partition_global_row in include/rt_hip_ring.h; cost per row is
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


class BandGatherer:
    """The one exchange step of a multi-GPU frame, with its buffers allocated once: every
    rank's (max_rows, width) uint8 band buffer goes to rank 0 by ONE gather (RCCL over
    xGMI under the "nccl" backend), straight into the slices of a stacked receive buffer;
    rank 0 then moves the rows to their place in the final image with one indexed copy.

    `collective=None` runs the gather whenever a process group is initialised -- also for
    a world of one, which is how the RCCL path is exercised on a one-GPU box."""

    def __init__(self, layout: BandLayout, rank: int, device, group=None, collective: Optional[bool] = None):
        import torch
        import torch.distributed as dist

        self.layout, self.rank, self.group = layout, rank, group
        self.collective = (dist.is_available() and dist.is_initialized()) if collective is None else collective
        if layout.world > 1 and not self.collective:
            raise RuntimeError("a frame split over several ranks needs an initialised process group")
        self.src, self.dst = layout.indices_on(device)
        width, height = layout.options.width, layout.options.height
        self.stacked = self.parts = self.final = None
        self._pending = self._band = None
        if rank == 0:
            self.final = torch.empty((height, width), dtype=torch.uint8, device=device)
            if self.collective:
                self.stacked = torch.empty((layout.world, layout.max_rows, width), dtype=torch.uint8, device=device)
                self.parts = list(self.stacked.unbind(0))  # views: the gather writes into `stacked` itself

    def __call__(self, band):
        """band: this rank's (max_rows, width) uint8 rows.  Returns the assembled image on rank 0, None elsewhere."""
        self.start(band)
        return self.finish()

    def start(self, band) -> None:
        """Issues the gather of `band` without waiting for it (the collective runs on the backend's own stream once
        the work queued on the current stream has produced the band); `finish` completes the frame.  A caller with two
        gatherers and two band buffers can render the next frame while this one is on the wire."""
        import torch.distributed as dist

        if self._pending is not None:
            raise RuntimeError("BandGatherer.start called again before finish")
        self._band = band
        self._pending = dist.gather(band, self.parts, dst=0, group=self.group, async_op=True) if self.collective else True

    def finish(self):
        """Waits for the gather issued by `start` (a stream-level wait under RCCL) and, on rank 0, moves the rows to
        their place.  Returns the assembled image on rank 0, None elsewhere."""
        import torch

        if self._pending is None:
            raise RuntimeError("BandGatherer.finish without start")
        if self.collective:
            self._pending.wait()
        band, self._pending, self._band = self._band, None, None
        if self.collective:
            if self.rank != 0:
                return None
            rows = self.stacked.view(-1, self.stacked.shape[-1])
        else:
            rows = band
        torch.index_select(rows, 0, self.src, out=self.final) if self._identity_dst() else \
            self.final.index_copy_(0, self.dst, rows.index_select(0, self.src))
        return self.final

    def _identity_dst(self) -> bool:
        # one rank: dst is 0..height-1 in order, so the selected rows ARE the image
        return self.layout.world == 1


def gather_bands(band, layout: BandLayout, rank: int, group=None):
    """Gathers every rank's (max_rows, width) uint8 band buffer to rank 0 and
    returns the assembled (height, width) image there (None elsewhere)."""
    return BandGatherer(layout, rank, band.device, group=group, collective=layout.world > 1)(band)
