"""Multi-GPU frame decomposition: one process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).

Rays are independent, so tracing needs no exchange.  Image rows are dealt to the ranks in blocks of
`block_rows` rows, block-cyclically (global block b -> rank b % world): the black-hole shadow, whose rays
are mostly skipped by the prepass, is spread over all ranks instead of landing on the middle strips.
Every rank traces one extra "halo" row under each of its blocks (the texture filter reads the pixel
below, cl.cl:5509-5520) instead of exchanging render_data rows.  The only collective is ONE gather of
the finished float4 rows to rank 0 (direct fan-in over the 7 xGMI links), followed by a local
un-permute of the block-cyclic layout.  The reference itself is single-GPU.
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
    un-permutes it into the [H, W, 4] frame."""

    def __init__(self, plan, width, device, rank, world, group=None):
        self.plan, self.width, self.rank, self.world, self.group = plan, width, rank, world, group
        shape = (plan.blocks_per_rank, plan.block_rows, width, 4)
        self.local = torch.zeros(shape, dtype=torch.float32, device=device)
        self.parts = [torch.zeros(shape, dtype=torch.float32, device=device) for _ in range(world)] if rank == 0 else None
        self.frame = None

    def local_buffer(self):
        """the compact buffer gr_render_frame writes into (options.compact_out = 1)"""
        return self.local

    def run(self, frame_out=None):
        """one collective: gather to rank 0; returns the assembled [H, W, 4] frame on rank 0, None elsewhere"""
        if dist.is_available() and dist.is_initialized():
            dist.gather(self.local, self.parts, dst=0, group=self.group)
        else:
            self.parts = [self.local]
        if self.rank != 0:
            return None
        p = self.plan
        # parts[r][i] is global block i*world + r  ->  stack to [blocks_per_rank, world, block_rows, W, 4]
        stacked = torch.stack(self.parts, dim=1).reshape(p.blocks_per_rank * self.world * p.block_rows, self.width, 4)
        frame = stacked[:p.height]
        if frame_out is not None:
            frame_out.copy_(frame)
            return frame_out
        return frame
