"""Screen-space partition of one frame over the GPUs of a node.

Rank r of G renders the 8-pixel tile rows ty with ty % G == r of the frame on
its own replica of the scene (rt_render_tile_rows_device) into its own device
framebuffer; only the framebuffer rows are exchanged: one gather to rank 0 over
RCCL (backend "nccl" on ROCm = RCCL over xGMI; "gloo" for the CPU tests).  No
scene data and no rays ever cross GPUs.
"""
import torch
import torch.distributed as dist

TILE = 8


def rows_of(rank, world, height):
    """Pixel rows of the frame owned by `rank` (interleaved 8-row bands)."""
    idx = torch.arange(height)
    return idx[((idx // TILE) % world) == rank]


def max_rows(world, height):
    return max(int(rows_of(r, world, height).numel()) for r in range(world))


def gather_frame(fb, rank, world, group=None):
    """fb: [H, W, 3] uint8 tensor holding this rank's rows (other rows arbitrary).
    Returns the merged frame on rank 0 (None elsewhere).  One collective:
    a gather of each rank's packed rows, padded to the largest share."""
    if world == 1:
        return fb
    H, W, _ = fb.shape
    pad = max_rows(world, H)
    mine = rows_of(rank, world, H).to(fb.device)
    packed = torch.zeros((pad, W, 3), dtype=torch.uint8, device=fb.device)
    packed[: mine.numel()] = fb.index_select(0, mine)
    if rank == 0:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, gather_list=parts, dst=0, group=group)
        out = fb.clone()
        for r in range(1, world):
            rr = rows_of(r, world, H).to(fb.device)
            out.index_copy_(0, rr, parts[r][: rr.numel()])
        return out
    dist.gather(packed, gather_list=None, dst=0, group=group)
    return None
