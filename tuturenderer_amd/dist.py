"""Multi-GPU sharding of a frame: one process per GPU, pixel tiles dealt round-robin, ONE collective per frame.

The reference splits image rows statically over 20 std::threads that share the framebuffer
(PathTracing.hpp:393-429).  Here every rank renders its own pixel list (the RNG is keyed by global pixel index and
sample index, so the image is bit-identical for any number of ranks) and the pieces are gathered to rank 0 with a
single `torch.distributed.gather` -- RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.  Payload:
W*H*12 B per frame (7.7 MB at 800x800), each peer on its own link to the root.
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
    """Owns the per-rank output piece and, on rank 0, the assembled frame."""

    def __init__(self, W, H, rank, world, device, tile=TILE, host_staging=False):
        import torch

        self.torch = torch
        self.W, self.H, self.rank, self.world = W, H, rank, world
        self.host_staging = host_staging  # gloo rehearsal: the collective runs on CPU copies of the pieces
        self.lists = tile_pixel_lists(W, H, world, tile)
        self.mine = self.lists[rank]
        n_max = max(len(l) for l in self.lists)
        self.piece = torch.zeros((n_max, 3), dtype=torch.float32, device=device)  # equal size on every rank
        self.frame = torch.zeros((H * W, 3), dtype=torch.float32, device=device) if rank == 0 else None
        self.gather_list = [torch.zeros_like(self.piece) for _ in range(world)] if (world > 1 and rank == 0) else None
        self.index = [torch.from_numpy(l.astype(np.int64)).to(device) for l in self.lists] if rank == 0 else None
        # Ordering contract with tutu_hip_render_device: the library writes `piece` from ITS stream and
        # returns when that write is complete; what it cannot see is torch-stream work that still READS `piece` (the
        # gather / index_copy below are enqueued asynchronously).  `consumed` marks the end of that work; the next
        # render waits for it on the host before the library touches `piece` again.
        self.consumed = torch.cuda.Event() if device.type == "cuda" else None

    def render(self, ctx, spp, key0, key1, pixels=None, **kw):
        """ctx.render_device into this rank's piece, ordered after the previous frame's consumers of `piece`"""
        if self.consumed is not None:
            self.consumed.synchronize()
        return ctx.render_device(self.piece.data_ptr(), spp, key0, key1, pixels=self.mine if pixels is None else pixels, **kw)

    def assemble(self):
        """the collective (world > 1) + scatter of the pieces into the frame on rank 0"""
        frame = self._assemble()
        if self.consumed is not None:
            self.consumed.record()
        return frame

    def _assemble(self):
        if self.world > 1:
            import torch.distributed as dist

            if self.host_staging:
                piece = self.piece.cpu()
                glist = [self.torch.zeros_like(piece) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(piece, glist, dst=0)
                if self.rank == 0:
                    for r in range(self.world):
                        self.frame.index_copy_(0, self.index[r], glist[r][: len(self.lists[r])].to(self.frame.device))
            else:
                dist.gather(self.piece, self.gather_list, dst=0)
                if self.rank == 0:
                    for r in range(self.world):
                        self.frame.index_copy_(0, self.index[r], self.gather_list[r][: len(self.lists[r])])
        else:
            self.frame.index_copy_(0, self.index[0], self.piece[: len(self.mine)])
        return self.frame
