"""HipVecRogueEnv: the HBM-resident fast path.  Same engine and semantics as ParallelRogueEnv
(keys in, auto-reset on terminal, reward = max(0, gold delta)), but actions, observations, rewards
and done flags are PyTorch-ROCm tensors that never leave the GPU, and nothing synchronises with
the host.  With torch.distributed initialised (backend "nccl" = RCCL), envs are sharded across
ranks in contiguous index blocks and `all_gather_obs()` assembles the whole-job batch.
"""
import ctypes as C
import json
from typing import Iterable, Optional

import numpy as np

from rogue_gym_python import _rogue_gym as inner

from .rogue_env import DungeonType, ImageSetting, RogueEnv, StatusFlag


class _DevArray:
    """Zero-copy view of a device buffer owned by the C library (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipVecRogueEnv:
    ACTIONS = RogueEnv.ACTIONS

    def __init__(self, config_dicts: Iterable[dict], max_steps: int = 1000,
                 image_setting: ImageSetting = ImageSetting(DungeonType.GRAY, StatusFlag.EMPTY, False), device: Optional[int] = None):
        import torch

        self.torch = torch
        cfgs = [d if isinstance(d, str) else json.dumps(d) for d in config_dicts]
        self._h = inner._Handle(cfgs, max_steps, auto_reset=True, device=device)
        self.device = torch.device("cuda", self._h.device)
        self.num_envs = self._h.n
        self.image_setting = image_setting
        self.symbols = self._h.symbols
        self.height, self.width = self._h.height, self._h.width
        L, h = self._h.L, self._h.h
        with torch.cuda.device(self.device):
            self._h.check(L.rg_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            self._action_keys = torch.tensor([ord(a) for a in self.ACTIONS], dtype=torch.uint8, device=self.device)
            self._sym = image_setting.dungeon == DungeonType.SYMBOL
            self.channels = L.rg_obs_channels(h, int(self._sym), image_setting.status.value, int(image_setting.includes_hist))
            self.obs = torch.empty((self.num_envs, self.channels, self.height, self.width), dtype=torch.float32, device=self.device)
            p = C.c_void_p()
            self._h.check(L.rg_reward(h, C.byref(p)))
            self.reward = torch.as_tensor(_DevArray(p.value, (self.num_envs,), "<f4"), device=self.device)
            self._h.check(L.rg_done(h, C.byref(p)))
            self.done = torch.as_tensor(_DevArray(p.value, (self.num_envs,), "|b1"), device=self.device)
            self._h.check(L.rg_flags(h, C.byref(p)))
            self.flags = torch.as_tensor(_DevArray(p.value, (self.num_envs,), "<i4"), device=self.device)
            self._h.check(L.rg_status(h, C.byref(p)))
            self.status = torch.as_tensor(_DevArray(p.value, (self.num_envs, 10), "<i4"), device=self.device)
            self._h.check(L.rg_screen(h, C.byref(p)))
            self.screen = torch.as_tensor(_DevArray(p.value, (self.num_envs, self.height, self.width), "|u1"), device=self.device)
        self._encode()

    def _encode(self):
        L, h = self._h.L, self._h.h
        fn = L.rg_obs_symbol if self._sym else L.rg_obs_gray
        self._h.check(fn(h, self.image_setting.status.value, int(self.image_setting.includes_hist), C.c_void_p(self.obs.data_ptr())))
        return self.obs

    def reset(self):
        self._h.check(self._h.L.rg_reset(self._h.h))
        return self._encode()

    def seed(self, seeds):
        seeds = [int(s) for s in seeds]
        n = len(seeds)
        lo = (C.c_uint64 * n)(*[s & 0xFFFFFFFFFFFFFFFF for s in seeds])
        hi = (C.c_uint64 * n)(*[(s >> 64) & 0xFFFFFFFFFFFFFFFF for s in seeds])
        self._h.check(self._h.L.rg_seed(self._h.h, lo, hi, n))

    def step_keys(self, keys):
        """keys: uint8 CUDA tensor [num_envs] of key bytes (KeyMap::ai)."""
        self._h.check(self._h.L.rg_step(self._h.h, C.c_void_p(keys.data_ptr()), 1))
        obs = self._encode()
        return obs, self.reward, self.done

    def step(self, actions):
        """actions: integer CUDA tensor [num_envs] of indices into ACTIONS."""
        return self.step_keys(self._action_keys[actions.long()])

    def check_errors(self):
        """Synchronise and raise like the reference's PyRuntimeError if any env saw an invalid key."""
        self._h.check(self._h.L.rg_sync(self._h.h))

    def all_gather_obs(self, compact: bool = True):
        """Whole-job observation batch on every rank (one RCCL all-gather over xGMI).
        compact=True gathers the u8 screen + i32 status (1/4 .. 1/170 of the f32 payload) and
        expands on the consumer GPU; compact=False gathers the f32 observation itself."""
        import torch.distributed as dist

        torch = self.torch
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return self.obs
        ws = dist.get_world_size()
        if not compact:
            out = torch.empty((ws * self.num_envs,) + tuple(self.obs.shape[1:]), dtype=self.obs.dtype, device=self.device)
            dist.all_gather_into_tensor(out, self.obs)
            return out
        from .sharding import all_gather_compact

        _ = ws
        return all_gather_compact(self.screen, self.status)

    def close(self):
        self._h.close()


class HipVecStairReward(HipVecRogueEnv):
    """StairRewardParallel (python/rogue_gym/envs/wrappers.py:45-64) on device tensors: `stair_reward` is added whenever an env's
    dungeon level is above the level it reported one step earlier; the comparison level follows the reported one (so it falls back to 1
    with the auto-reset).  The reward is a new tensor; nothing leaves the GPU."""

    def __init__(self, *args, stair_reward: float = 50.0, **kwargs):
        super().__init__(*args, **kwargs)
        self.stair_reward = float(stair_reward)
        self.current_levels = self.torch.ones(self.num_envs, dtype=self.torch.int32, device=self.device)

    def step_keys(self, keys):
        obs, reward, done = super().step_keys(keys)
        level = self.status[:, 0]
        reward = reward + self.stair_reward * (self.current_levels < level).to(reward.dtype)
        self.current_levels.copy_(level)
        return obs, reward, done

    def reset(self):
        self.current_levels.fill_(1)
        return super().reset()
