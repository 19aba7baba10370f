"""Framebuffer sharding across the GPUs of one node and the single gather of final pixels.

Partitioning follows the tiling descriptor the reference sketched (BlockInfo{x,y,w,h}, raytracer_v4.glsl:70-79),
specialised to interleaved row blocks: block b of `tile_rows` rows belongs to rank (b mod world).  Interleaving keeps
the shards balanced although sky rows end at the first ray cast and ground rows bounce many times.  Every rank renders
its rows into a compact buffer.

* one process per GPU (bench.py under torch.distributed.run): ONE `torch.distributed.gather` (RCCL over xGMI with
  backend "nccl", gloo on CPU) brings the RGBA8 rows into one buffer on rank 0, and ONE indexed copy puts them at
  their frame rows;
* one process, one GPU (world = 1): the rows go through the C ABI's rt3_gather_rows — the same device-to-device
  strided copy the C++ host uses between GPUs (host/HostApi.cpp) — so the native path is what bench.py times.

No other collective exists on the path: each pixel is owned by exactly one rank, and the RNG is keyed by frame
coordinates, so the image does not depend on the number of ranks.
"""
import torch
import torch.distributed as dist


class FrameGatherer:
    """Owns the tile buffer of this rank and, on rank 0, the receive buffer and the assembled frame."""

    def __init__(self, rt3, params_list, rank, device, force_collective=False, renderer=None, stage_host=False):
        self.force_collective = force_collective           # run the gather even with one rank (rehearsal of the RCCL path)
        # stage_host: the tile is rendered on `device` but travels through host memory and a CPU backend (gloo) — the rehearsal of the
        # multi-rank control flow with several ranks on ONE GPU (RCCL refuses two ranks on one device); never the measured path
        self.stage_host = stage_host
        self.world = len(params_list)
        self.rank = rank
        self.params = params_list[rank]
        self.renderer = renderer                           # HipRenderer of this rank: enables the native world-1 path
        self.rows = [rt3.rows_owned(p) for p in params_list]
        self.width = params_list[0].width
        self.height = params_list[0].height
        assert sum(self.rows) == self.height, "shards must cover every row exactly once"
        max_rows = max(self.rows)
        # every rank sends the same number of rows (padded): gather needs equal shapes
        self.tile = torch.zeros((max_rows, self.width), dtype=torch.int32, device=device)
        self.host_tile = torch.zeros((max_rows, self.width), dtype=torch.int32) if stage_host else None
        if stage_host:
            device = torch.device("cpu")                   # receive buffer, row maps and the assembled frame live where the backend works
        self.frame = None
        self.recv = None
        self.gather_list = None
        self.dst_rows = None
        self.src_rows = None
        if rank == 0:
            self.frame = torch.zeros((self.height, self.width), dtype=torch.int32, device=device)
            dst, src = [], []
            for i, (p, n) in enumerate(zip(params_list, self.rows)):
                dst += [rt3.row_of_local(p, k) for k in range(n)]
                src += [i * max_rows + k for k in range(n)]
            self.dst_rows = torch.tensor(dst, dtype=torch.long, device=device)
            # without padding (every rank owns max_rows rows) the receive buffer is read straight through
            self.src_rows = None if len(src) == self.world * max_rows else torch.tensor(src, dtype=torch.long, device=device)
            if self.world > 1 or force_collective:
                self.recv = torch.zeros((self.world, max_rows, self.width), dtype=torch.int32, device=device)
                self.gather_list = [self.recv[i] for i in range(self.world)]      # views: the gather fills one buffer

    def gather(self, stream_ptr=None):
        """Collects every rank's tile on rank 0 and returns the assembled frame there (None elsewhere)."""
        if self.world > 1 or self.force_collective:
            if self.stage_host:
                self.host_tile.copy_(self.tile)            # (synchronises with the render on the current stream)
            dist.gather(self.host_tile if self.stage_host else self.tile, self.gather_list, dst=0)
            if self.rank == 0:
                rows = self.recv.view(-1, self.width)
                self.frame.index_copy_(0, self.dst_rows, rows if self.src_rows is None else rows.index_select(0, self.src_rows))
        elif self.rank == 0:
            if self.renderer is not None and self.tile.is_cuda:
                self.renderer.gather_rows(self.frame.data_ptr(), self.renderer, self.tile.data_ptr(), self.params, stream_ptr)
            else:
                self.frame.index_copy_(0, self.dst_rows, self.tile[: self.rows[0]])
        return self.frame
