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


def bands_of(rank, world, height):
    """Number of 8-row bands of the frame owned by `rank`."""
    n = (height + TILE - 1) // TILE
    return (n - rank + world - 1) // world if rank < n else 0


class _GatherPlan:
    """Staging buffers (and, for frames whose height is not a multiple of 8, index tensors) of one (frame shape,
    world, device): built once, reused every frame."""

    def __init__(self, fb, rank, world):
        H, W, _ = fb.shape
        self.banded = H % TILE == 0          # the frame is a whole number of bands: rank r's rows are the view [r::world] of them
        self.pad = max_rows(world, H)
        self.packed = torch.zeros((self.pad, W, 3), dtype=torch.uint8, device=fb.device)
        self.n_mine = int(rows_of(rank, world, H).numel())
        if not self.banded:
            self.mine = rows_of(rank, world, H).to(fb.device)
        if rank == 0:
            self.parts = [torch.empty_like(self.packed) for _ in range(world)]
            self.n_rows = [int(rows_of(r, world, H).numel()) for r in range(world)]
            if not self.banded:
                self.rows = [rows_of(r, world, H).to(fb.device) for r in range(world)]


def gather_frame(fb, rank, world, group=None, force=False):
    """fb: [H, W, 3] uint8 tensor holding this rank's rows (other rows arbitrary).
    Returns the merged frame on rank 0 — fb itself, with the other ranks' rows written into it — and None elsewhere.
    One collective: a gather of each rank's packed rows, padded to the largest share.  Packing is one strided copy
    (the rank's bands are the view [rank::world] of the frame's bands); rank 0 unpacks with one strided copy per rank
    straight into its own frame.  force: run the collective even for world == 1 (a one-rank RCCL communicator: the
    test that RCCL initialises and moves uint8 rows on the box at hand)."""
    if world == 1 and not force:
        return fb
    key = (tuple(fb.shape), rank, world, str(fb.device))
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = _GatherPlan(fb, rank, world)
    H, W, _ = fb.shape
    if plan.banded:
        bands = fb.view(H // TILE, TILE, W, 3)
        plan.packed[: plan.n_mine].view(-1, TILE, W, 3).copy_(bands[rank::world])
    else:
        torch.index_select(fb, 0, plan.mine, out=plan.packed[: plan.n_mine])
    if rank == 0:
        dist.gather(plan.packed, gather_list=plan.parts, dst=0, group=group)
        for r in range(0 if force and world == 1 else 1, world):      # (its own rows are in place already)
            if plan.banded:
                bands[r::world].copy_(plan.parts[r][: plan.n_rows[r]].view(-1, TILE, W, 3))
            else:
                fb.index_copy_(0, plan.rows[r], plan.parts[r][: plan.n_rows[r]])
        return fb
    dist.gather(plan.packed, gather_list=None, dst=0, group=group)
    return None
