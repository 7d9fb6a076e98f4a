"""Env sharding across ranks: rank r owns the contiguous block of env indices
[r*N/P, (r+1)*N/P); the only exchange is ONE optional all-gather per step of the compact observation records
(u8 screen + i32 status (+ u8 history) per env, packed by rg_pack_compact), expanded to f32 images on the consumer GPU by the HIP
encode kernels (rg_expand_compact)."""
from typing import Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) env indices owned by `rank` (ThreadConductor has no sharding: envs are independent,
    python/src/thread_impls.rs:17-31)."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def record_layout(height: int, width: int, with_hist: bool = False):
    """Byte offsets inside one compact record: (screen, status, hist or None, record size)."""
    hw = height * width
    return 0, hw, (hw + 40 if with_hist else None), hw + 40 + (hw if with_hist else 0)


def all_gather_packed(packed, group=None):
    """The one collective of the multi-GPU path (RCCL over xGMI with backend "nccl"; gloo on CPU): u8 [n, record] of every rank ->
    u8 [world * n, record] on every rank, rank slices in rank order.  Equal shard sizes."""
    import torch
    import torch.distributed as dist

    ws = dist.get_world_size(group)
    out = torch.empty((ws * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    return out


def unpack_records(packed, height: int, width: int, with_hist: bool = False):
    """Views into a gathered batch of records: (screen u8 [N,H,W], status i32 [N,10], hist u8 [N,H,W] or None)."""
    import torch

    o_scr, o_st, o_hist, rec = record_layout(height, width, with_hist)
    n = packed.shape[0]
    flat = packed.reshape(n, rec)
    screen = flat[:, o_scr:o_st].reshape(n, height, width)
    status = flat[:, o_st:o_st + 40].contiguous().view(torch.int32).reshape(n, 10)
    hist = flat[:, o_hist:o_hist + height * width].reshape(n, height, width) if with_hist else None
    return screen, status, hist
