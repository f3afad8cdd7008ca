"""world_size=2 over gloo on the CPU: the row sharding + gather + de-interleave that bench.py
uses across GPUs, checked against a full-frame render.  The per-rank pixels come from the CPU
oracle here (no GPU in this tier); the GPU tier checks the same identity on one device
(test_gpu_parity.py::test_row_shards_assemble_the_full_frame)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden, Oracle, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, depth, name, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ndt_amd.multi import RowGather
    from ndt_amd import shard_rows
    g = golden(name)
    rows, st = Oracle().render(g.scene, width, height, depth, row_begin=rank, row_step=world, threads=2)
    rg = RowGather(height, width, 4, torch.float64, "cpu", rank, world, dist)
    n = shard_rows(height, rank, world)
    rg.local[:n] = torch.from_numpy(rows)
    # whole-job ray count the way bench.py aggregates it
    cnt = torch.tensor([float(st.rays_ref_equiv)], dtype=torch.float64)
    dist.all_reduce(cnt)
    img = rg.assemble()
    # the quantised image (what the reference writes to disk, pixel_d2c) gathered the same way: bench.py's default
    rg8 = RowGather(height, width, 4, torch.uint8, "cpu", rank, world, dist)
    rg8.local[:n] = torch.from_numpy(Oracle().quantize(rows))
    img8 = rg8.assemble()
    if rank == 0:
        np.save(out_path, img.numpy())
        np.save(out_path + ".cnt.npy", cnt.numpy())
        np.save(out_path + ".u8.npy", img8.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height", [72, 71])      # even split and ragged split
def test_two_ranks_assemble_the_reference_frame(tmp_path, oracle, height):
    name, world = "c3_random4d", 2
    g = golden(name)
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, g.width, g.depth, name, out), nprocs=world, join=True)
    got = np.load(out)
    want, st = oracle.render(g.scene, g.width, height, g.depth)
    assert np.array_equal(got, want)
    assert float(np.load(out + ".cnt.npy")[0]) == float(st.rays_ref_equiv)
    assert np.array_equal(np.load(out + ".u8.npy"), oracle.quantize(want))
    if height == g.height:
        assert np.array_equal(got, g.data["fb"])        # the compiled reference's framebuffer


def test_row_gather_single_rank_is_a_copy():
    from ndt_amd.multi import RowGather
    rg = RowGather(5, 3, 4, torch.float64, "cpu", 0, 1)
    rg.local[:5] = torch.arange(60, dtype=torch.float64).reshape(5, 3, 4)
    assert torch.equal(rg.assemble(), rg.local[:5])


def _pipelined_worker(rank, world, port, out_path):
    """bench.py's frame loop: the gather of frame k is in flight while frame k+1 is produced into the other buffer."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ndt_amd.multi import RowGather
    from ndt_amd import shard_rows
    height, width, frames = 37, 5, 6
    bufs = [RowGather(height, width, 4, torch.uint8, "cpu", rank, world, dist) for _ in range(2)]
    n = shard_rows(height, rank, world)
    rows = torch.arange(rank, height, world)
    got, pending = [], None
    for k in range(frames):
        g = bufs[k % 2]
        # frame k: pixel value = (row + 3 * k) mod 251, the same in every column and channel
        g.local[:n] = ((rows + 3 * k) % 251).to(torch.uint8)[:, None, None].expand(n, width, 4)
        if pending is not None:
            img = pending.finish()
            if rank == 0:
                got.append(img.clone())
        g.start()
        pending = g
    img = pending.finish()
    if rank == 0:
        got.append(img.clone())
        np.save(out_path, torch.stack(got).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_of_one_frame_overlaps_the_next(tmp_path):
    out = str(tmp_path / "frames.npy")
    mp.spawn(_pipelined_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (6, 37, 5, 4)
    for k in range(6):
        want = ((np.arange(37) + 3 * k) % 251).astype(np.uint8)[:, None, None] * np.ones((1, 5, 4), dtype=np.uint8)
        assert np.array_equal(got[k], want), "frame %d" % k
