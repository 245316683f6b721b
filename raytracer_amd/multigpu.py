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


_plans = {}


class _GatherPlan:
    """Index tensors and staging buffers of one (frame shape, world, device): built once, reused every frame."""

    def __init__(self, fb, rank, world):
        H, W, _ = fb.shape
        self.pad = max_rows(world, H)
        self.mine = rows_of(rank, world, H).to(fb.device)
        self.packed = torch.zeros((self.pad, W, 3), dtype=torch.uint8, device=fb.device)
        if rank == 0:
            self.parts = [torch.empty_like(self.packed) for _ in range(world)]
            self.rows = [rows_of(r, world, H).to(fb.device) for r in range(world)]
            self.out = torch.empty_like(fb)


def gather_frame(fb, rank, world, group=None):
    """fb: [H, W, 3] uint8 tensor holding this rank's rows (other rows arbitrary).
    Returns the merged frame on rank 0 (None elsewhere; the tensor is reused by the next call with the same
    frame shape).  One collective: a gather of each rank's packed rows, padded to the largest share."""
    if world == 1:
        return fb
    key = (tuple(fb.shape), rank, world, str(fb.device))
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = _GatherPlan(fb, rank, world)
    torch.index_select(fb, 0, plan.mine, out=plan.packed[: plan.mine.numel()])
    if rank == 0:
        dist.gather(plan.packed, gather_list=plan.parts, dst=0, group=group)
        out = plan.out
        out.index_copy_(0, plan.mine, plan.packed[: plan.mine.numel()])
        for r in range(1, world):
            rr = plan.rows[r]
            out.index_copy_(0, rr, plan.parts[r][: rr.numel()])
        return out
    dist.gather(plan.packed, gather_list=None, dst=0, group=group)
    return None
