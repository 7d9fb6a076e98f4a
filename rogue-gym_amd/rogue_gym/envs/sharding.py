"""Env sharding across ranks: rank r owns the contiguous block of env indices
[r*N/P, (r+1)*N/P); the only exchange is an optional all-gather of the compact observation."""
from typing import Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) env indices owned by `rank` (ThreadConductor has no sharding: envs are independent,
    python/src/thread_impls.rs:17-31)."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def all_gather_compact(screen, status, group=None):
    """One all-gather (RCCL over xGMI with backend "nccl"; gloo on CPU) of the compact observation:
    u8 screen [n,H,W] and i32 status [n,10] -> [world*n, ...] on every rank.  Equal shard sizes."""
    import torch
    import torch.distributed as dist

    ws = dist.get_world_size(group)
    scr = torch.empty((ws * screen.shape[0],) + tuple(screen.shape[1:]), dtype=screen.dtype, device=screen.device)
    st = torch.empty((ws * status.shape[0],) + tuple(status.shape[1:]), dtype=status.dtype, device=status.device)
    dist.all_gather_into_tensor(scr, screen.contiguous(), group=group)
    dist.all_gather_into_tensor(st, status.contiguous(), group=group)
    return scr, st


def expand_gray(screen, symbols: int):
    """Consumer-side expansion of a gathered u8 screen to the f32 gray image (python/src/lib.rs:72-87):
    sym(glyph) / symbols, via a 256-entry table (core/src/symbol.rs:17-40)."""
    import torch

    lut = torch.zeros(256, dtype=torch.float32)
    for i, ch in enumerate(" @#.-%+^!?])/*:=,"):
        lut[ord(ch)] = i
    lut[ord("|")] = 4
    for i in range(26):
        lut[ord("A") + i] = 17 + i
    lut = (lut / float(symbols)).to(screen.device)
    return lut[screen.long()].unsqueeze(1)
