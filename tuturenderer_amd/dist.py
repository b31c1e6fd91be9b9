"""Multi-GPU sharding of a frame: one process per GPU, pixel tiles dealt round-robin, ONE collective per frame.

The reference splits image rows statically over 20 std::threads that share the framebuffer
(PathTracing.hpp:393-429).  Here every rank renders its own pixel list (the RNG is keyed by global pixel index and
sample index, so the image is bit-identical for any number of ranks) and the pieces are gathered to rank 0 with a
single grouped send/recv of exactly len(list) rows per rank -- RCCL over xGMI when the backend is "nccl", gloo in the
CPU tests.  Payload: W*H*12 B per frame (7.7 MB at 800x800), each peer on its own link to the root.
"""
import numpy as np

TILE = 32


def tile_pixel_lists(W, H, n_ranks, tile=TILE):
    """Pixel indices (y*W+x) per rank: tile x tile tiles dealt round-robin over ranks (contiguous row blocks, what
    the reference does, are load-imbalanced); inside a tile pixels are ordered in 8x8 blocks so that the 64 lanes
    of a wavefront cover a compact patch of the image."""
    lists = [[] for _ in range(n_ranks)]
    t = 0
    for ty in range(0, H, tile):
        for tx in range(0, W, tile):
            ys = np.arange(ty, min(ty + tile, H))
            xs = np.arange(tx, min(tx + tile, W))
            blk = []
            for by in range(0, len(ys), 8):
                for bx in range(0, len(xs), 8):
                    yy, xx = np.meshgrid(ys[by:by + 8], xs[bx:bx + 8], indexing="ij")
                    blk.append((yy * W + xx).ravel())
            lists[t % n_ranks].append(np.concatenate(blk))
            t += 1
    return [np.concatenate(l).astype(np.int32) if l else np.zeros(0, np.int32) for l in lists]


class FrameGather:
    """Owns the per-rank output piece and, on rank 0, the gathered pieces and the assembled frame.

    One collective per frame, exact sizes: every rank r > 0 sends its len(list_r) rows to rank 0, rank 0 posts the matching
    receives, all in ONE batch (`batch_isend_irecv`: ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd with backend
    "nccl" = RCCL) straight into one contiguous (W*H, 3) buffer in rank order -- rank 0's own piece IS the
    head of that buffer -- and ONE index_copy (one kernel) un-tiles it into the frame.

    Ordering without host synchronisation: FrameGather owns ONE non-default stream.  The library is handed that stream (the
    ABI's `stream` argument: its first kernel and its last -- the finalize that writes `piece` -- run on it), the collective
    and the un-tiling are issued on it, so the next frame's render is ordered after this frame's send / un-tiling, which read
    `piece`, by the stream itself.  (Torch's DEFAULT stream would not do: its handle is 0, which the ABI reads as "the
    context's own stream" -- a non-blocking stream that is ordered with nothing of torch's.)  A consumer of `frame` on another
    stream calls wait() (or synchronises the device, as bench.py's fences do).

    self_loop: rehearsal of the collective with ONE rank -- rank 0 sends its piece to ITSELF through the process group (one
    grouped send + recv, the same batch_isend_irecv call as with N ranks) into a second buffer and un-tiles from there: the
    RCCL code path on a one-GPU box (tests/test_hip_parity.py).
    """

    def __init__(self, W, H, rank, world, device, tile=TILE, host_staging=False, self_loop=False):
        import torch

        self.torch = torch
        self.W, self.H, self.rank, self.world = W, H, rank, world
        self.device = device
        self.host_staging = host_staging  # gloo rehearsal on GPUs: the collective runs on CPU copies of the pieces
        self.self_loop = bool(self_loop) and world == 1
        self.stream = torch.cuda.Stream(device) if device.type == "cuda" else None
        self.lists = tile_pixel_lists(W, H, world, tile)
        self.mine = self.lists[rank]
        self.sizes = [len(l) for l in self.lists]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        if rank == 0:
            self.gathered = torch.zeros((W * H, 3), dtype=torch.float32, device=device)  # pieces back to back, rank order
            self.piece = self.gathered[: self.sizes[0]]
            self.frame = torch.zeros((H * W, 3), dtype=torch.float32, device=device)
            self.index = torch.from_numpy(np.concatenate(self.lists).astype(np.int64)).to(device)  # row k of `gathered` -> pixel
        else:
            self.gathered = None
            self.piece = torch.zeros((max(self.sizes[rank], 1), 3), dtype=torch.float32, device=device)[: self.sizes[rank]]
            self.frame = None
            self.index = None
        self.looped = torch.zeros((W * H, 3), dtype=torch.float32, device=device) if self.self_loop else None
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(device))  # the buffers' zero fills ran on the caller's stream

    def render(self, ctx, spp, key0, key1, pixels=None, **kw):
        """ctx.render_device into this rank's piece, on FrameGather's own stream: ordered after the previous frame's consumers
        of `piece` (send / un-tiling) by the stream itself -- no host wait between frames"""
        return ctx.render_device(self.piece.data_ptr(), spp, key0, key1, pixels=self.mine if pixels is None else pixels,
                                 stream=self.stream.cuda_stream if self.stream is not None else None, **kw)

    def wait(self):
        """orders the caller's current stream after everything FrameGather has enqueued (the frame is then safe to read there)"""
        if self.stream is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def assemble(self):
        """the collective (world > 1) + ONE un-tiling kernel on rank 0, on FrameGather's stream"""
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):
                return self._assemble()
        return self._assemble()

    def _assemble(self):
        if self.self_loop:
            import torch.distributed as dist

            ops = [dist.P2POp(dist.isend, self.gathered, 0), dist.P2POp(dist.irecv, self.looped, 0)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            self.frame.index_copy_(0, self.index, self.looped)
            return self.frame
        if self.world > 1:
            import torch.distributed as dist

            if self.host_staging:
                piece = self.piece.cpu()
                if self.rank == 0:
                    host = self.torch.zeros((self.W * self.H, 3), dtype=self.torch.float32)
                    host[: self.sizes[0]] = piece
                    ops = [dist.P2POp(dist.irecv, host[self.offsets[r]: self.offsets[r + 1]], r) for r in range(1, self.world) if self.sizes[r]]
                else:
                    ops = [dist.P2POp(dist.isend, piece, 0)] if self.sizes[self.rank] else []
                for w in (dist.batch_isend_irecv(ops) if ops else []):
                    w.wait()
                if self.rank == 0:
                    self.gathered.copy_(host.to(self.device))
            else:
                if self.rank == 0:
                    ops = [dist.P2POp(dist.irecv, self.gathered[self.offsets[r]: self.offsets[r + 1]], r) for r in range(1, self.world)
                           if self.sizes[r]]
                else:
                    ops = [dist.P2POp(dist.isend, self.piece, 0)] if self.sizes[self.rank] else []
                for w in (dist.batch_isend_irecv(ops) if ops else []):
                    w.wait()  # nccl: orders the current stream after the transfer, no host block
        if self.rank == 0:
            self.frame.index_copy_(0, self.index, self.gathered)
        return self.frame
