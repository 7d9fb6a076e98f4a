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
                 image_setting: ImageSetting = ImageSetting(DungeonType.GRAY, StatusFlag.EMPTY, False), device: Optional[int] = None,
                 persistent_obs: bool = False):
        """persistent_obs (opt-in; image settings without status planes and history plane): `self.obs` is BOUND to the stepper (rg_obs_bind) -- every step
        keeps it current in place, rewriting only the envs whose screen changed; its contents are bit-identical to the unbound encode's.  The caller
        must not write to `self.obs`."""
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
            self._screen = torch.as_tensor(_DevArray(p.value, (self.num_envs, self.height, self.width), "|u1"), device=self.device)
        self._scratch = {}
        self.persistent_obs = bool(persistent_obs)
        if self.persistent_obs:
            self._h.check(L.rg_obs_bind(h, int(self._sym), image_setting.status.value, int(image_setting.includes_hist), C.c_void_p(self.obs.data_ptr())))
        self._encode()

    @property
    def screen(self):
        """u8 [num_envs, H, W] glyph mirror (PlayerState.map of every env).  Reading it flushes the pending render (a batch with several
        config groups assembles the groups' screens first); the tensor itself is the same device buffer every time."""
        p = C.c_void_p()
        self._h.check(self._h.L.rg_screen(self._h.h, C.byref(p)))
        return self._screen

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
        """keys: uint8 CUDA tensor [num_envs] of key bytes (KeyMap::ai), contiguous, on this env's device."""
        if keys.dtype != self.torch.uint8 or keys.device != self.device or not keys.is_contiguous() or keys.numel() != self.num_envs:
            raise ValueError("step_keys needs a contiguous uint8 tensor of %d keys on %s, got %s %s on %s"
                             % (self.num_envs, self.device, tuple(keys.shape), keys.dtype, keys.device))
        if self._sym:
            self._h.check(self._h.L.rg_step(self._h.h, C.c_void_p(keys.data_ptr()), 1))
            obs = self._encode()
        else:  # the step and the gray observation as one call: fused into one kernel where the config allows it (rg_step_obs_gray)
            self._h.check(self._h.L.rg_step_obs_gray(self._h.h, C.c_void_p(keys.data_ptr()), 1, self.image_setting.status.value, int(self.image_setting.includes_hist),
                                                      C.c_void_p(self.obs.data_ptr())))
            obs = self.obs
        return obs, self.reward, self.done

    def step(self, actions):
        """actions: integer CUDA tensor [num_envs] of indices into ACTIONS."""
        return self.step_keys(self._action_keys[actions.long()])

    def check_errors(self):
        """Synchronise and raise like the reference's PyRuntimeError if any env saw an invalid key."""
        self._h.check(self._h.L.rg_sync(self._h.h))

    def packed_records(self, with_hist: bool = False):
        """u8 [num_envs, record]: the compact observation record of every env of this rank (rg_pack_compact)."""
        L, h = self._h.L, self._h.h
        rec = L.rg_compact_record_bytes(h, int(with_hist))
        key = ("packed", bool(with_hist))
        buf = self._scratch.get(key)
        if buf is None:
            buf = self._scratch[key] = self.torch.empty((self.num_envs, rec), dtype=self.torch.uint8, device=self.device)
        self._h.check(L.rg_pack_compact(h, int(with_hist), C.c_void_p(buf.data_ptr())))
        return buf

    def expand_records(self, packed, image_setting: Optional[ImageSetting] = None, packed_has_hist: bool = False, out=None):
        """f32 [N, C, H, W] from N compact records (of any rank) under `image_setting`: the HIP encode kernels on the consumer GPU."""
        st = self.image_setting if image_setting is None else image_setting
        sym = st.dungeon == DungeonType.SYMBOL
        L, h = self._h.L, self._h.h
        n = int(packed.shape[0])
        c = L.rg_obs_channels(h, int(sym), st.status.value, int(st.includes_hist))
        if out is None:
            out = self.torch.empty((n, c, self.height, self.width), dtype=self.torch.float32, device=self.device)
        if packed.dtype != self.torch.uint8 or not packed.is_contiguous() or packed.device != self.device:
            raise ValueError("expand_records needs a contiguous uint8 tensor on %s" % (self.device,))
        self._h.check(L.rg_expand_compact(h, C.c_void_p(packed.data_ptr()), n, int(packed_has_hist), int(sym), st.status.value, int(st.includes_hist),
                                          C.c_void_p(out.data_ptr())))
        return out

    def init_comm(self, rank: Optional[int] = None, world: Optional[int] = None, unique_id: Optional[bytes] = None):
        """Give the handle its own RCCL communicator (rg_comm_init), so that the one collective of the sharded path runs behind the C-ABI
        (rg_allgather_compact: pack into this rank's slice, ncclAllGather in place on the handle's stream) -- the path a non-Python host
        binds.  Without arguments the rank / world come from torch.distributed and rank 0's ncclUniqueId is broadcast through it; a host
        without torch passes all three (the id from `HipVecRogueEnv.comm_unique_id()` on one rank)."""
        import torch.distributed as dist

        if rank is None or world is None:
            rank, world = dist.get_rank(), dist.get_world_size()
        if unique_id is None:
            box = [self.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            unique_id = box[0]
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._h.check(self._h.L.rg_comm_init(self._h.h, buf, int(rank), int(world)))
        self._comm = (int(rank), int(world))

    def comm_count(self):
        """(ranks, this rank) as the RCCL communicator itself reports them (rg_comm_count = ncclCommCount / ncclCommUserRank)."""
        c, r = C.c_int(), C.c_int()
        self._h.check(self._h.L.rg_comm_count(self._h.h, C.byref(c), C.byref(r)))
        return c.value, r.value

    @staticmethod
    def comm_unique_id() -> bytes:
        L = inner.load_library()
        buf = (C.c_uint8 * 128)()
        if L.rg_comm_unique_id(buf):
            raise RuntimeError("Error in rogue-gym: " + L.rg_last_error(None).decode())
        return bytes(buf)

    def all_gather_records(self, with_hist: bool = False):
        """u8 [world * num_envs, record] on every rank through the handle's own communicator (init_comm): rg_allgather_compact."""
        rank, world = self._comm
        L, h = self._h.L, self._h.h
        rec = L.rg_compact_record_bytes(h, int(with_hist))
        key = ("gathered", bool(with_hist))
        buf = self._scratch.get(key)
        if buf is None:
            buf = self._scratch[key] = self.torch.empty((world * self.num_envs, rec), dtype=self.torch.uint8, device=self.device)
        self._h.check(L.rg_allgather_compact(h, int(with_hist), C.c_void_p(buf.data_ptr())))
        return buf

    def all_gather_obs(self, compact: bool = True):
        """Whole-job observation batch f32 [world * num_envs, C, H, W] on every rank, env order = rank order -- the same type whatever the
        world size (world 1: this rank's own `obs`).  compact=True (default): ONE RCCL all-gather over xGMI of the packed records
        (560 B per mini env instead of 2 KB .. 330 KB of f32), expanded on the consumer GPU by the HIP encode kernels -- through the
        handle's own communicator when init_comm() was called (the C-ABI path), else through torch.distributed;
        compact=False gathers the f32 observation itself (xGMI-bound for the one-hot image)."""
        import torch.distributed as dist

        torch = self.torch
        if compact and getattr(self, "_comm", None) is not None:
            with_hist = bool(self.image_setting.includes_hist)
            return self.expand_records(self.all_gather_records(with_hist), packed_has_hist=with_hist)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return self.obs
        group = getattr(self, "process_group", None)  # None = the default group; a host whose default group is a CPU one sets its RCCL group here
        ws = dist.get_world_size(group)
        if not compact:
            out = torch.empty((ws * self.num_envs,) + tuple(self.obs.shape[1:]), dtype=self.obs.dtype, device=self.device)
            dist.all_gather_into_tensor(out, self.obs, group=group)
            return out
        from .sharding import all_gather_packed

        with_hist = bool(self.image_setting.includes_hist)
        gathered = all_gather_packed(self.packed_records(with_hist), group=group)
        return self.expand_records(gathered, packed_has_hist=with_hist)

    def all_gather_step(self):
        """(obs, reward, done, flags) of the WHOLE job on every rank from the ONE collective of the step: obs f32 [world * num_envs, C, H, W],
        reward f32 [N], done bool [N], flags i32 [N] (public RG_FLAG_* bits), env order = rank order.  ThreadConductor::step returns state and
        terminal flag of every env in one reply (python/src/thread_impls.rs:61-81) and parallel.py:59-64 derives reward and done from it; here the
        compact record carries them next to the screen, so a learner that wants the all-gathered batch needs no second collective."""
        import torch.distributed as dist
        from .sharding import all_gather_packed, unpack_step

        with_hist = bool(self.image_setting.includes_hist)
        if getattr(self, "_comm", None) is not None:
            packed = self.all_gather_records(with_hist)
        else:
            packed = self.packed_records(with_hist)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(getattr(self, "process_group", None)) > 1:
                packed = all_gather_packed(packed, group=getattr(self, "process_group", None))
        reward, done, flags = unpack_step(packed, self.height, self.width, with_hist)
        return self.expand_records(packed, packed_has_hist=with_hist), reward, done, flags

    def all_gather_compact(self, with_hist: bool = False):
        """The gathered records themselves, as views: (screen u8 [N,H,W], status i32 [N,10], hist u8 [N,H,W] or None)."""
        import torch.distributed as dist
        from .sharding import all_gather_packed, unpack_records

        packed = self.packed_records(with_hist)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            packed = all_gather_packed(packed, group=getattr(self, "process_group", None))
        return unpack_records(packed, self.height, self.width, with_hist)

    def status_vec(self, flag=None):
        """i32 [num_envs, popcount(flag)] device tensor: PlayerState::status_vec of every env (flags.rs:67-87)."""
        flag = self.image_setting.status.value if flag is None else int(getattr(flag, "value", flag))
        cols = [c for b, c in enumerate((0, 2, 3, 4, 5, 6, 7, 8, 9)) if flag & (1 << b)]
        return self.status[:, cols]

    def counters(self, reset: bool = False):
        """Workload counters since the last reset of the counters (rg_counters)."""
        out = (C.c_uint64 * 9)()
        self._h.check(self._h.L.rg_counters_ex(self._h.h, out, 9, int(reset)))
        names = ("resets", "descents", "dist_maps", "inline_generations", "spares_taken", "redraws", "keys", "partial_maps_continued", "next_level_structures_used")
        return dict(zip(names, (int(v) for v in out)))

    def enable_history(self, cap_per_env: int):
        self._h.check(self._h.L.rg_history_enable(self._h.h, int(cap_per_env)))

    def dump_history(self, env: int, previous: bool = False) -> str:
        """GameState::dump_history (python/src/lib.rs:245-250) of env `env`: the InputCode JSON of its running (or previous) episode."""
        return self._h.dump_history(int(env), previous)

    def history_keys(self, env: int, previous: bool = False) -> bytes:
        return self._h.history_keys(int(env), previous)

    def close(self):
        self._h.close()


class HipVecStairReward(HipVecRogueEnv):
    """StairRewardParallel (python/rogue_gym/envs/wrappers.py:45-64) on device tensors.  The rule -- `stair_reward` whenever an env reports a deeper
    level than one step earlier, the comparison level following the auto-reset back to 1 -- runs inside the step kernel (rg_set_stair_reward): `reward`
    is the same device tensor as without the wrapper, no extra launch, and the all-gathered records carry the bonus too."""

    def __init__(self, *args, stair_reward: float = 50.0, **kwargs):
        super().__init__(*args, **kwargs)
        self.stair_reward = float(stair_reward)
        self._h.check(self._h.L.rg_set_stair_reward(self._h.h, self.stair_reward))
