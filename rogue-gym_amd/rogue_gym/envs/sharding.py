"""Env sharding across ranks: rank r owns the contiguous block of env indices
[r*N/P, (r+1)*N/P); the only exchange is ONE optional all-gather per step of the compact records
(u8 screen + i32 status + f32 reward + u32 flags (+ u8 history) per env, packed by rg_pack_compact) -- everything ThreadConductor::step returns
per env in one reply, state and terminal flag (python/src/thread_impls.rs:61-81) -- expanded to f32 images on the consumer GPU by the HIP
encode kernels (rg_expand_compact)."""
from typing import Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) env indices owned by `rank` (ThreadConductor has no sharding: envs are independent,
    python/src/thread_impls.rs:17-31)."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


FIXED_BYTES = 48        # status i32[10] + reward f32 + flags u32 (RG_COMPACT_FIXED_BYTES)
FLAG_TERMINAL = 0x1     # RG_FLAG_TERMINAL: the last key ended the episode (PlayerState.is_terminal)


def record_layout(height: int, width: int, with_hist: bool = False):
    """Byte offsets inside one compact record: (screen, status, hist or None, record size); reward / flags sit at status + 40 / + 44."""
    hw = height * width
    return 0, hw, (hw + FIXED_BYTES if with_hist else None), hw + FIXED_BYTES + (hw if with_hist else 0)


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


def unpack_step(packed, height: int, width: int, with_hist: bool = False):
    """The step half of a gathered batch of records: (reward f32 [N], done bool [N], flags i32 [N]) -- what parallel.py:59-64 derives from the
    states ParallelGameState::step returns, for the WHOLE job on every rank.  flags: the public RG_FLAG_* bits (terminal, dead, message bits
    8..14, error bits)."""
    import torch

    _, o_st, _, rec = record_layout(height, width, with_hist)
    n = packed.shape[0]
    tail = packed.reshape(n, rec)[:, o_st + 40:o_st + 48].contiguous()
    reward = tail[:, 0:4].contiguous().view(torch.float32).reshape(n)
    flags = tail[:, 4:8].contiguous().view(torch.int32).reshape(n)
    return reward, (flags & FLAG_TERMINAL) != 0, flags
