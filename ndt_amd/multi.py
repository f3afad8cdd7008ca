"""Row-sharded rendering across ranks: one process per GPU, one gather to assemble the frame.

Rows are dealt cyclically exactly like the reference's MPI_ROW mode (ndt.c:812-820): rank r
renders image rows r, r+world, ...  Each rank holds its rows compactly; a single
`torch.distributed.gather` (RCCL over xGMI on GPUs, gloo in the CPU tests) brings the shards to
one rank, which de-interleaves them.  The reference instead sum-reduces full-size, mostly-zero
images up a binary tree (ndt.c:1277-1309); that is an MPI convenience, not reproduced here.
"""
import torch

from .flat_scene import shard_rows


def padded_rows(height, world):
    """Rows of the largest shard (every rank's buffer is padded to this for the gather)."""
    return shard_rows(height, 0, world)


class RowGather:
    """Pre-allocated buffers + the gather/de-interleave step for one frame geometry."""

    def __init__(self, height, width, channels, dtype, device, rank, world, dist=None, dst=0, always_collective=False):
        """always_collective: run the gather even in a world of one (bench.py --selftest-gather: the N>1 code path over
        RCCL on a single GPU)."""
        self.height, self.width, self.rank, self.world, self.dist, self.dst = height, width, rank, world, dist, dst
        self.collective = world > 1 or (always_collective and dist is not None)
        self.rows_max = padded_rows(height, world)
        self.local = torch.zeros((self.rows_max, width, channels), dtype=dtype, device=device)
        self.parts = None
        self.work = None
        self.image = None
        if self.collective and rank == dst:
            # one buffer for all shards and an image padded to world * rows_max rows: the de-interleave is then ONE strided
            # copy (shard r, row i -> image row i*world + r; the padding rows of a ragged split land behind the image)
            self._shards = torch.empty((world, self.rows_max, width, channels), dtype=dtype, device=device)
            self.parts = [self._shards[r] for r in range(world)]
            self._padded = torch.empty((self.rows_max * world, width, channels), dtype=dtype, device=device)
            self.image = self._padded[:height]
        elif rank == dst:
            self.image = torch.empty((height, width, channels), dtype=dtype, device=device)

    def start(self):
        """Begin gathering this buffer's `local` rows to `dst` (asynchronous for world > 1).

        The collective runs on the backend's own stream, so the caller may render the NEXT frame into
        ANOTHER RowGather's `local` meanwhile; this one's `local` and `parts` must stay untouched until
        `finish()` has returned and the current torch stream has been synchronised."""
        if self.collective:
            self.work = self.dist.gather(self.local, self.parts, dst=self.dst, async_op=True)

    def wait(self):
        """Order the current stream (gloo: the host) behind `start()`'s gather."""
        if self.work is not None:
            self.work.wait()
            self.work = None

    def deinterleave(self):
        """On dst: the gathered shards -> `image` (enqueued on the current stream); returns it, None elsewhere."""
        if not self.collective:
            self.image.copy_(self.local[:self.height])
            return self.image
        if self.rank != self.dst:
            return None
        self._padded.view(self.rows_max, self.world, self.width, -1).copy_(self._shards.permute(1, 0, 2, 3))
        return self.image

    def finish(self):
        """Wait for `start()`'s gather and de-interleave on dst; returns the frame on dst, None elsewhere."""
        self.wait()
        return self.deinterleave()

    def assemble(self):
        """Gather every rank's `local` rows to `dst` and de-interleave; returns the frame on dst."""
        self.start()
        return self.finish()
