"""Framebuffer sharding across the GPUs of one node and the single gather of final pixels.

Partitioning follows the tiling descriptor the reference sketched (BlockInfo{x,y,w,h}, raytracer_v4.glsl:70-79),
specialised to interleaved row blocks: block b of `tile_rows` rows belongs to rank (b mod world).  Interleaving keeps
the shards balanced although sky rows end at the first ray cast and ground rows bounce many times.  Every rank renders
its rows into a compact buffer; ONE `torch.distributed.gather` (RCCL over xGMI with backend "nccl", gloo on CPU)
brings the RGBA8 rows to rank 0, which scatters them to their frame rows.  No other collective exists on the path:
each pixel is owned by exactly one rank, and the RNG is keyed by frame coordinates, so the image does not depend on
the number of ranks.
"""
import torch
import torch.distributed as dist


class FrameGatherer:
    """Owns the padded tile buffer of this rank and, on rank 0, the gather list and the assembled frame."""

    def __init__(self, rt3, params_list, rank, device, force_collective=False):
        self.force_collective = force_collective           # run the gather even with one rank (rehearsal of the RCCL path)
        self.world = len(params_list)
        self.rank = rank
        self.rows = [rt3.rows_owned(p) for p in params_list]
        self.width = params_list[0].width
        self.height = params_list[0].height
        assert sum(self.rows) == self.height, "shards must cover every row exactly once"
        max_rows = max(self.rows)
        # every rank sends the same number of rows (padded): gather needs equal shapes
        self.tile = torch.zeros((max_rows, self.width), dtype=torch.int32, device=device)
        self.frame = None
        self.gather_list = None
        self.row_index = None
        if rank == 0:
            self.frame = torch.zeros((self.height, self.width), dtype=torch.int32, device=device)
            self.row_index = [torch.tensor([rt3.row_of_local(p, k) for k in range(n)], dtype=torch.long, device=device)
                              for p, n in zip(params_list, self.rows)]
            if self.world > 1 or force_collective:
                self.gather_list = [torch.zeros_like(self.tile) for _ in range(self.world)]

    def gather(self):
        """Collects every rank's tile on rank 0 and returns the assembled frame there (None elsewhere)."""
        if self.world > 1 or self.force_collective:
            dist.gather(self.tile, self.gather_list, dst=0)
            if self.rank == 0:
                for i in range(self.world):
                    self.frame.index_copy_(0, self.row_index[i], self.gather_list[i][: self.rows[i]])
        elif self.rank == 0:
            self.frame.index_copy_(0, self.row_index[0], self.tile[: self.rows[0]])
        return self.frame
