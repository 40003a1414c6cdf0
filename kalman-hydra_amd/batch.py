"""Independent frame pairs across the GPUs of one node (BASELINE config 5).

Brox flow of one frame pair does not depend on any other pair (reference
src/optical_flow_ext.cpp:361-412 carries nothing but the previous frame), so a batch
is cut into contiguous blocks, one block per rank, one process per GPU.  Nothing is
exchanged while the blocks are computed; the only collective is the gather of the
results at the end (RCCL over xGMI when the backend is "nccl"; the same code runs on
"gloo" for CPU tests).  The EKF over the frames of ONE video is a recurrence and does
not shard; several videos shard the same way (bench.py).
"""
import numpy as np


def shard(n_items, rank, world):
    """Contiguous block of rank `rank`: the first n_items % world ranks get one item more."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def gather_blocks(local, n_items, group=None):
    """all_gather of per-rank blocks of unequal length along dim 0 -> tensor of n_items rows on
    every rank.  `local` is this rank's block (torch tensor, any device the backend supports)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    longest = len(shard(n_items, 0, world))
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: len(shard(n_items, r, world))] for r in range(world)], dim=0)


def gather_to_root(local, n_items, dst=0, group=None):
    """The batch path's one exchange: gather of per-rank blocks of unequal length along dim 0 to rank `dst`
    (dist.gather: RCCL over xGMI when the backend is "nccl", a fan-in over the direct links rather than a
    ring).  Returns the tensor of n_items rows on rank dst, None elsewhere; without a process group the
    block itself."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    longest = len(shard(n_items, 0, world))
    if local.shape[0] == longest:
        pad = local.contiguous()
    else:
        pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([parts[r][: len(shard(n_items, r, world))] for r in range(world)], dim=0)


def pair_seed(index):
    """BASELINE config 5: pair i of the batch is the `translate_leftup_stretch` warp of the
    noise texture of seed i, i = 0 .. 255."""
    return int(index)


def flow_batch_sharded(frames0, frames1, flow_fn, group=None, gather=True):
    """frames0/frames1: (n, H, W) u8 arrays present on every rank (or generated per rank by the
    caller); flow_fn(f0_block, f1_block) -> (u_block, v_block) as torch tensors.  Returns the
    (u, v) of all n pairs on every rank when gather is set, else this rank's block and its range."""
    import torch.distributed as dist
    n = int(np.shape(frames0)[0])
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = shard(n, rank, world)
    u, v = flow_fn(frames0[mine.start:mine.stop], frames1[mine.start:mine.stop])
    if not gather:
        return u, v, mine
    return gather_blocks(u, n, group), gather_blocks(v, n, group)
