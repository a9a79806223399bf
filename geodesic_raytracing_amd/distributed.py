"""Multi-GPU frame decomposition: one process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).

Rays are independent, so tracing needs no exchange.  Image rows are dealt to the ranks in blocks of
`block_rows` rows, block-cyclically (global block b -> rank b % world): the black-hole shadow, whose rays
are mostly skipped by the prepass, is spread over all ranks instead of landing on the middle strips.
Every rank traces one extra "halo" row under each of its blocks (the texture filter reads the pixel
below, cl.cl:5509-5520) instead of exchanging render_data rows.  The only collective is ONE gather of
the finished float4 rows to rank 0 (direct fan-in over the 7 xGMI links), followed by a local
un-permute of the block-cyclic layout.  The gather of frame n runs on RCCL's stream while frame n+1 is
traced (`submit` / `drain`, double-buffered strips).  The reference itself is single-GPU.
"""
import torch
import torch.distributed as dist


class StripPlan:
    def __init__(self, height, world, block_rows=16):
        if block_rows % 8:
            raise ValueError("block_rows must be a multiple of 8 (8x8 ray tiles)")
        if world > 1 and height > 1 and (height - 1) % block_rows == 0:
            raise ValueError("the last image row must not start a block")
        self.height, self.world, self.block_rows = height, world, block_rows
        self.total_blocks = (height + block_rows - 1) // block_rows
        # every rank ships the same number of blocks (the last ones may be padding)
        self.blocks_per_rank = (self.total_blocks + world - 1) // world

    def local_blocks(self, rank):
        """number of real blocks rank owns"""
        return len(range(rank, self.total_blocks, self.world))

    def blocks_of(self, rank):
        """[(row_begin, row_end)] of the blocks rank owns, in local order"""
        out = []
        for b in range(rank, self.total_blocks, self.world):
            out.append((b * self.block_rows, min((b + 1) * self.block_rows, self.height)))
        return out

    def owner_of_row(self, row):
        return (row // self.block_rows) % self.world


class FrameGather:
    """Gathers every rank's compact strip buffer ([blocks_per_rank, block_rows, W, 4] float32) on rank 0 and
    un-permutes it into the [H, W, 4] frame.

    run():              synchronous - gather + assemble, returns the frame on rank 0.
    submit() / drain(): pipelined - the gather of the strips just rendered is issued asynchronously and completes
                        while the next frame is traced into the other strip buffer; a frame is assembled on rank 0
                        when its buffer comes round again (or at drain())."""

    def __init__(self, plan, width, device, rank, world, group=None, depth=2):
        self.plan, self.width, self.rank, self.world, self.group = plan, width, rank, world, group
        shape = (plan.blocks_per_rank, plan.block_rows, width, 4)
        self.locals = [torch.zeros(shape, dtype=torch.float32, device=device) for _ in range(depth)]
        self.parts = [[torch.zeros(shape, dtype=torch.float32, device=device) for _ in range(world)] if rank == 0 else None
                      for _ in range(depth)]
        self.pending = [None] * depth
        self.rotation = [0] * depth   # which strip each rank rendered into this buffer: strip = (rank + rotation) % world
        self.slot = 0
        self.frames_done = 0

    def local_buffer(self):
        """the compact buffer gr_render_frame writes into (options.compact_out = 1)"""
        return self.locals[self.slot]

    def _collective(self):
        return dist.is_available() and dist.is_initialized()

    def frame_buffer(self, device=None):
        """a frame buffer padded to whole blocks for every rank ([blocks_per_rank * world * block_rows, W, 4]): passed as
        frame_out it lets the un-permute write the frame directly (one copy instead of two); rows [0, H) are the image"""
        p = self.plan
        return torch.zeros((p.blocks_per_rank * self.world * p.block_rows, self.width, 4), dtype=torch.float32,
                           device=device if device is not None else self.locals[0].device)

    def _assemble(self, parts, frame_out, rotation=0):
        p = self.plan
        if rotation % self.world:
            # rank r rendered strip (r + rotation) % world, so strip s came from rank (s - rotation) % world
            parts = [parts[(s - rotation) % self.world] for s in range(self.world)]
        padded_rows = p.blocks_per_rank * self.world * p.block_rows
        # parts[r][i] is global block i*world + r  ->  [blocks_per_rank, world, block_rows, W, 4] -> rows
        if frame_out is not None and frame_out.shape[0] == padded_rows and frame_out.is_contiguous():
            torch.stack(parts, dim=1, out=frame_out.view(p.blocks_per_rank, self.world, p.block_rows, self.width, 4))
            return frame_out[:p.height]
        stacked = torch.stack(parts, dim=1).reshape(padded_rows, self.width, 4)
        frame = stacked[:p.height]
        if frame_out is not None:
            frame_out.copy_(frame)
            return frame_out
        return frame

    def run(self, frame_out=None, rotation=0):
        """one collective: gather to rank 0; returns the assembled [H, W, 4] frame on rank 0, None elsewhere.
        rotation: this frame's strip assignment was rotated, rank r rendered strip (r + rotation) % world (strip_of)"""
        i = self.slot
        if self._collective():
            dist.gather(self.locals[i], self.parts[i], dst=0, group=self.group)
            parts = self.parts[i]
        else:
            parts = [self.locals[i]]
        if self.rank != 0:
            return None
        return self._assemble(parts, frame_out, rotation)

    def strip_of(self, rotation):
        """the strip_rank this rank renders in a frame whose assignment is rotated by `rotation` (e.g. the frame number):
        over `world` consecutive frames every rank renders every strip once, which evens out strips of different cost"""
        return (self.rank + rotation) % self.world

    def _finish(self, j, frame_out):
        work = self.pending[j]
        if work is None:
            return None
        self.pending[j] = None
        if work is not True:
            work.wait()          # orders the current stream after the collective; the host does not block
        self.frames_done += 1
        if self.rank != 0:
            return None
        return self._assemble(self.parts[j] if self._collective() else [self.locals[j]], frame_out, self.rotation[j])

    def submit(self, frame_out=None, rotation=0):
        """call after rendering into local_buffer(): starts its gather and moves on to the other buffer.  Returns the
        frame (rank 0) whose buffer is about to be reused, if one was in flight, else None."""
        i = self.slot
        self.rotation[i] = rotation
        if self._collective():
            self.pending[i] = dist.gather(self.locals[i], self.parts[i], dst=0, group=self.group, async_op=True)
        else:
            self.pending[i] = True
        self.slot = (i + 1) % len(self.locals)
        return self._finish(self.slot, frame_out)

    def drain(self, frame_out=None):
        """completes every gather still in flight; returns the last assembled frame on rank 0"""
        last = None
        n = len(self.locals)
        for k in range(n):
            r = self._finish((self.slot + k) % n, frame_out)
            if r is not None:
                last = r
        return last
