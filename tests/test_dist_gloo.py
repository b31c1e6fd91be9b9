"""CPU, world_size 2 over gloo: the multi-rank path of tuturenderer_amd.dist (tile sharding + the one gather)
assembles exactly the single-process frame.  The per-rank render is done by the CPU oracle here (no GPU in this
container); on GPUs bench.py drives the same FrameGather with the HIP context and backend "nccl" (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, key1, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle.pyoracle import Oracle
    from tuturenderer_amd import scenes
    from tuturenderer_amd.dist import FrameGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fg = FrameGather(W, H, rank, world, torch.device("cpu"), tile=16)
    S = Oracle("port").scene(scenes.cornell_box(W, H))
    # render only this rank's pixels: one trace per (pixel, sample), summed in sample order like the resolve kernel
    pix = np.repeat(fg.mine.astype(np.uint32), spp)
    smp = np.tile(np.arange(spp, dtype=np.uint32), len(fg.mine))
    L = S.trace_samples(pix, smp, 0x5EED0001, key1).reshape(len(fg.mine), spp, 3)
    acc = np.zeros((len(fg.mine), 3), np.float32)
    for s in range(spp):
        acc = acc + L[:, s]
    fg.piece[: len(fg.mine)] = torch.from_numpy(acc * np.float32(1.0 / spp))
    frame = fg.assemble()
    dist.barrier()
    if rank == 0:
        np.save(out_path, frame.numpy().reshape(H, W, 3))
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame(built, port, tmp_path):
    import torch.multiprocessing as mp

    from tuturenderer_amd import scenes
    from tuturenderer_amd.dist import tile_pixel_lists

    W, H, spp, key1 = 40, 24, 3, 9
    # the lists partition the frame: every pixel exactly once, for ragged sizes too
    for world in (1, 2, 3, 8):
        lists = tile_pixel_lists(W, H, world, tile=16)
        allp = np.concatenate(lists)
        assert len(allp) == W * H and len(np.unique(allp)) == W * H
        sizes = [len(l) for l in lists]
        assert max(sizes) - min(sizes) <= 16 * 16
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), W, H, spp, key1, out), nprocs=2, join=True)
    frame = np.load(out)
    S = port.scene(scenes.cornell_box(W, H))
    ref = S.render(spp, 0x5EED0001, key1, nthreads=2)
    S.close()
    assert frame.tobytes() == ref.tobytes()
