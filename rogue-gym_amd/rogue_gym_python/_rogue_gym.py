"""`rogue_gym_python._rogue_gym` -- host mirror of the reference's PyO3 module
(/root/reference/python/src/lib.rs:355-366) over the HIP C-ABI (include/rogue_gym_hip.h).

Same classes, method names, argument meaning and error behaviour as the Rust module:
GameState, ParallelGameState, PlayerState.  All game logic runs in librogue_gym_hip.so on an
MI355X; this file only marshals.  There is no CPU fallback: importing works anywhere, but
constructing a GameState without the built library or without a HIP device raises.
"""
import ctypes as C
import weakref
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ROGUE_GYM_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "librogue_gym_hip.so")  # (override: A/B runs of two builds)

RG_FLAG_TERMINAL = 0x1
RG_FLAG_DEAD = 0x2
RG_FLAG_MSG_SHIFT = 8
RG_FLAG_MSG_MASK = 0x7F00
RG_FLAG_ERR_MASK = 0x00FF0000

STATUS_KEYS = ("dungeon_level", "gold", "hp_current", "hp_max", "str_current", "str_max", "defense", "player_level", "exp", "hunger")


class RgDebugState(C.Structure):
    _fields_ = [
        ("px", C.c_int32), ("py", C.c_int32), ("dungeon_level", C.c_int32), ("hp", C.c_int32), ("hp_max", C.c_int32),
        ("player_level", C.c_int32), ("n_monsters", C.c_int32), ("n_gold", C.c_int32),
        ("exp", C.c_uint32), ("food_left", C.c_uint32), ("quiet", C.c_uint32), ("pack_gold", C.c_uint32), ("steps", C.c_uint32),
        ("rng", C.c_uint32 * 12),
        ("mon_x", C.c_int32 * 384), ("mon_y", C.c_int32 * 384), ("mon_type", C.c_int32 * 384), ("mon_active", C.c_int32 * 384), ("mon_hp", C.c_int32 * 384),
        ("mon_exp", C.c_uint32 * 384),
        ("gold_x", C.c_int32 * 384), ("gold_y", C.c_int32 * 384), ("gold_amount", C.c_int32 * 384),
        ("n_rooms", C.c_int32), ("room_rect", C.c_uint32 * 384), ("room_meta", C.c_int32 * 384),
    ]


_lib = None

_INT_FUNCS = (
    "rg_create", "rg_dims", "rg_env_dims", "rg_env_symbols", "rg_set_stream", "rg_seed", "rg_reset", "rg_step", "rg_step_prefix", "rg_step_obs_gray", "rg_step_fetch", "rg_sync", "rg_screen", "rg_hist", "rg_status", "rg_flags",
    "rg_reward", "rg_done", "rg_set_stair_reward", "rg_obs_bind", "rg_obs_gray", "rg_obs_symbol", "rg_obs_channels", "rg_fetch_states", "rg_encode_host", "rg_encode_host_batch", "rg_obs_host",
    "rg_host_alloc", "rg_dev_alloc", "rg_snapshot_take", "rg_dev_read", "rg_dev_read_rows", "rg_compact_record_bytes", "rg_pack_compact", "rg_expand_compact", "rg_comm_unique_id", "rg_comm_init", "rg_comm_destroy", "rg_comm_count", "rg_allgather_compact", "rg_status_vec", "rg_history_enable", "rg_history_keys",
    "rg_dump_history", "rg_counters", "rg_counters_ex", "rg_probe_sclk", "rg_dump_config", "rg_config_canonical", "rg_config_resolved", "rg_config_schema", "rg_debug_fetch", "rg_debug_descend", "rg_timing_enable", "rg_timing_read", "rg_timing_read_all", "rg_timing_read_samples",
)


def load_library():
    """Load librogue_gym_hip.so (built in-tree by __graft_entry__.build() / csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RuntimeError(
            "librogue_gym_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or rogue-gym_amd/csrc/build.sh -- there is no CPU fallback." % _SO
        )
    try:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname, libamdhip64.so.7).
        # Importing torch first makes the dynamic loader satisfy our NEEDED entry with that copy; loading ours
        # first would put two runtimes in the process and the second one sees no GPU.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_SO)
    vp, i32, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    sig = {
        "rg_create": [C.POINTER(C.c_char_p), i32, C.c_uint64, i32, i32, C.POINTER(vp)],
        "rg_destroy": [vp],
        "rg_last_error": [vp],
        "rg_dims": [vp] + [C.POINTER(i32)] * 4,
        "rg_env_symbols": [vp, vp], "rg_env_dims": [vp, vp, vp],
        "rg_set_stream": [vp, vp],
        "rg_seed": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i32],
        "rg_reset": [vp],
        "rg_step": [vp, vp, i32],
        "rg_step_prefix": [vp, vp, i32, i32],
        "rg_step_obs_gray": [vp, vp, i32, u32, i32, vp],
        "rg_step_fetch": [vp, vp, i32, vp, vp, vp, vp],
        "rg_sync": [vp],
        "rg_screen": [vp, C.POINTER(vp)], "rg_hist": [vp, C.POINTER(vp)], "rg_status": [vp, C.POINTER(vp)], "rg_flags": [vp, C.POINTER(vp)],
        "rg_reward": [vp, C.POINTER(vp)], "rg_done": [vp, C.POINTER(vp)], "rg_set_stair_reward": [vp, C.c_float],
        "rg_obs_bind": [vp, i32, u32, i32, vp], "rg_obs_gray": [vp, u32, i32, vp], "rg_obs_symbol": [vp, u32, i32, vp], "rg_obs_channels": [vp, i32, u32, i32],
        "rg_fetch_states": [vp, vp, vp, vp, vp],
        "rg_encode_host": [i32, vp, vp, vp, i32, i32, i32, u32, i32, i32, vp],
        "rg_encode_host_batch": [i32, i32, vp, vp, vp, i32, i32, i32, u32, i32, i32, vp],
        "rg_obs_host": [vp, i32, u32, i32, vp],
        "rg_host_alloc": [sz, C.POINTER(vp)], "rg_host_free": [vp],
        "rg_dev_alloc": [i32, sz, C.POINTER(vp)], "rg_dev_free": [i32, vp], "rg_snapshot_take": [vp, vp], "rg_dev_read": [vp, vp, vp, sz], "rg_dev_read_rows": [vp, vp, sz, vp, sz, i32],
        "rg_compact_record_bytes": [vp, i32], "rg_pack_compact": [vp, i32, vp], "rg_expand_compact": [vp, vp, i32, i32, i32, u32, i32, vp],
        "rg_comm_unique_id": [vp], "rg_comm_init": [vp, vp, i32, i32], "rg_comm_destroy": [vp], "rg_comm_count": [vp, vp, vp], "rg_allgather_compact": [vp, i32, vp],
        "rg_status_vec": [vp, u32, vp],
        "rg_history_enable": [vp, i32], "rg_history_keys": [vp, i32, i32, vp, sz, C.POINTER(u32)],
        "rg_dump_history": [vp, i32, i32, C.c_char_p, sz, C.POINTER(sz)],
        "rg_counters": [vp, C.POINTER(C.c_uint64), i32], "rg_counters_ex": [vp, C.POINTER(C.c_uint64), i32, i32], "rg_probe_sclk": [vp, C.POINTER(C.c_double)],
        "rg_timing_enable": [vp, i32], "rg_timing_read": [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
        "rg_timing_read_all": [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "rg_timing_read_samples": [vp, i32, C.POINTER(C.c_float), i32, C.POINTER(i32)],
        "rg_dump_config": [vp, i32, C.c_char_p, sz], "rg_config_canonical": [C.c_char_p, C.c_char_p, sz],
        "rg_config_resolved": [C.c_char_p, C.c_char_p, sz], "rg_config_schema": [C.c_char_p, sz, C.POINTER(sz)],
        "rg_debug_fetch": [vp, i32, C.POINTER(RgDebugState), vp], "rg_debug_descend": [vp],
    }
    # (entry points added in round 6: a library named by ROGUE_GYM_HIP_LIB -- an older build in a same-box A/B run -- may lack them; the product library
    # exports every symbol of the header, tests/test_cabi_load.py)
    optional = {"rg_timing_read_samples", "rg_obs_bind"} if os.environ.get("ROGUE_GYM_HIP_LIB") else set()
    for name, argtypes in sig.items():
        if name in optional and not hasattr(L, name):
            continue
        getattr(L, name).argtypes = argtypes
    L.rg_destroy.restype = None
    L.rg_host_free.restype = None
    L.rg_dev_free.restype = None
    L.rg_last_error.restype = C.c_char_p
    L.rg_build_id.restype = C.c_char_p
    L.rg_build_id.argtypes = []
    for name in _INT_FUNCS:
        if name in optional and not hasattr(L, name):
            continue
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _default_device():
    return int(os.environ.get("ROGUE_GYM_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


class _PinnedPool:
    """Page-locked host buffers for the per-step D2H copies of the value-object API.  A StateBatch borrows its buffers and hands them back
    when it is collected, so a stepping loop settles on two or three buffer sets and the copies run at full PCIe rate."""

    def __init__(self, L):
        self.L, self.free, self.closed = L, {}, False

    def take(self, nbytes):
        lst = self.free.get(nbytes)
        if lst:
            return lst.pop()
        p = C.c_void_p()
        if self.L.rg_host_alloc(nbytes, C.byref(p)):
            raise RuntimeError("Error in rogue-gym: " + self.L.rg_last_error(None).decode())
        return p.value

    def give(self, ptr, nbytes):
        if self.closed:
            self.L.rg_host_free(C.c_void_p(ptr))
        else:
            self.free.setdefault(nbytes, []).append(ptr)

    def close(self):
        self.closed = True
        for lst in self.free.values():
            for ptr in lst:
                self.L.rg_host_free(C.c_void_p(ptr))
        self.free = {}


class _DevPool:
    """Device buffers for the StateBatch snapshots (same take / give protocol as _PinnedPool)."""

    def __init__(self, L, device):
        self.L, self.device, self.free, self.closed = L, device, {}, False

    def take(self, nbytes):
        lst = self.free.get(nbytes)
        if lst:
            return lst.pop()
        p = C.c_void_p()
        if self.L.rg_dev_alloc(self.device, nbytes, C.byref(p)):
            raise RuntimeError("Error in rogue-gym: " + self.L.rg_last_error(None).decode())
        return p.value

    def give(self, ptr, nbytes):
        if self.closed:
            self.L.rg_dev_free(self.device, C.c_void_p(ptr))
        else:
            self.free.setdefault(nbytes, []).append(ptr)

    def close(self):
        self.closed = True
        for lst in self.free.values():
            for ptr in lst:
                self.L.rg_dev_free(self.device, C.c_void_p(ptr))
        self.free = {}


class _Lease:
    """One pooled pinned buffer on loan.  It goes back to the pool when the LAST array that views it is collected -- not when the StateBatch
    that asked for it is: `obs = env.images()` kept in a rollout buffer stays valid (and unchanged) for as long as the caller holds it, like the
    independent arrays the reference returns (python/src/lib.rs:78,95)."""
    __slots__ = ("pool", "ptr", "nbytes")

    def __init__(self, pool, nbytes):
        self.pool, self.nbytes = pool, nbytes
        self.ptr = pool.take(nbytes)

    def __del__(self):
        try:
            self.pool.give(self.ptr, self.nbytes)  # a closed pool frees it instead (hipHostFree)
        except Exception:
            pass


class _DevLease:
    """A pooled device buffer on loan to one StateBatch (its snapshot of the screen / history mirrors)."""
    __slots__ = ("pool", "ptr", "nbytes")

    def __init__(self, pool, nbytes):
        self.pool, self.nbytes = pool, nbytes
        self.ptr = pool.take(nbytes)

    def __del__(self):
        try:
            self.pool.give(self.ptr, self.nbytes)
        except Exception:
            pass


def _leased(pool, shape, dtype):
    """A numpy array over a freshly leased pinned buffer; every view / slice of it keeps the lease (numpy base -> ctypes array -> lease)."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    lease = _Lease(pool, nbytes)
    raw = (C.c_uint8 * nbytes).from_address(lease.ptr)
    raw._lease = lease
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def _leased_many(pool, specs):
    """Several arrays over ONE leased pinned buffer (a lease costs a few microseconds of Python; a small batch's step has four): 16-byte aligned sections."""
    sizes = [(int(np.prod(sh)) * np.dtype(dt).itemsize + 15) & ~15 for sh, dt in specs]
    lease = _Lease(pool, sum(sizes))
    raw = (C.c_uint8 * sum(sizes)).from_address(lease.ptr)
    raw._lease = lease
    whole, out, off = np.frombuffer(raw, dtype=np.uint8), [], 0
    for (sh, dt), sz in zip(specs, sizes):
        out.append(whole[off:off + int(np.prod(sh)) * np.dtype(dt).itemsize].view(dt).reshape(sh))
        off += sz
    return out


class _Handle:
    """Owns one rg_t."""

    def __init__(self, configs, max_steps, auto_reset, device=None):
        L = load_library()
        self.L = L
        self.device = _default_device() if device is None else device
        n = len(configs)
        arr = (C.c_char_p * n)()
        for i, cfg in enumerate(configs):
            arr[i] = None if cfg is None else cfg.encode("utf-8")
        h = C.c_void_p()
        rc = L.rg_create(arr, n, int(max_steps), self.device, int(auto_reset), C.byref(h))
        if rc:
            msg = L.rg_last_error(None).decode()
            if msg.startswith("Failed to parse config"):
                raise RuntimeError(msg)  # pyresult_with(.., "Failed to parse config") (python/src/lib.rs:219,275)
            raise RuntimeError("Error in rogue-gym: " + msg)
        self.h = h
        hh, ww, ss, nn = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.rg_dims(h, C.byref(hh), C.byref(ww), C.byref(ss), C.byref(nn))
        self.height, self.width, self.symbols, self.n = hh.value, ww.value, ss.value, nn.value
        self.env_symbols = np.empty(self.n, np.int32)  # per env: configs of one batch may differ (PlayerState.symbols is the env's own)
        L.rg_env_symbols(h, self.env_symbols.ctypes.data)
        self.uniform_symbols = bool((self.env_symbols == self.symbols).all())
        self.env_heights, self.env_widths = np.empty(self.n, np.int32), np.empty(self.n, np.int32)  # ... and in size (python/src/lib.rs:270-294)
        L.rg_env_dims(h, self.env_heights.ctypes.data, self.env_widths.ctypes.data)
        self.mixed_sizes = bool((self.env_heights != self.height).any() or (self.env_widths != self.width).any())
        self.pool = _PinnedPool(L)
        self.dev_pool = _DevPool(L, self.device)
        self.lazy = {}  # id -> weakref of the StateBatches whose screens still live only in their device snapshot (materialised before the handle goes)
        self.epoch = 0  # bumped by every call that changes the device-side states (a StateBatch remembers the epoch it was taken at)
        self.fast_step = True  # rg_step_fetch applies (cleared by the first refusal: a handle with config groups)

    def check(self, rc):
        if rc:
            raise RuntimeError("Error in rogue-gym: " + self.L.rg_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            for ref in list(self.lazy.values()):  # value semantics: states handed out earlier stay readable after close()
                b = ref()
                if b is not None:
                    b._materialise()
            self.L.rg_destroy(self.h)
            self.h = None
            self.pool.close()
            self.dev_pool.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fetch(self):
        """Fresh numpy copies of the mirrors (screen, hist, status, flags)."""
        n = self.n
        screen = np.empty((n, self.height, self.width), np.uint8)
        hist = np.empty((n, self.height, self.width), np.uint8)
        status = np.empty((n, 10), np.int32)
        flags = np.empty(n, np.uint32)
        self.check(self.L.rg_fetch_states(self.h, screen.ctypes.data, hist.ctypes.data, status.ctypes.data, flags.ctypes.data))
        return screen, hist, status, flags

    def snapshot(self):
        """The current states of all envs as one StateBatch (one D2H copy into pinned memory)."""
        return StateBatch(self)

    def step_keys_host(self, keys):
        """rg_step_prefix + rg_sync: the general form of a value-object step (config groups, mixed sizes); StateBatch(keys=...) uses rg_step_fetch where it applies."""
        self.check(self.L.rg_step_prefix(self.h, keys.ctypes.data, int(keys.shape[0]), 0))
        self.epoch += 1
        self.check(self.L.rg_sync(self.h))

    def states(self):
        return self.snapshot()

    def debug_state(self, env):
        out = RgDebugState()
        cells = np.empty((self.height, self.width), np.uint16)
        self.check(self.L.rg_debug_fetch(self.h, env, C.byref(out), cells.ctypes.data))
        return out, cells

    def history_keys(self, env, previous=False):
        n = C.c_uint32()
        rc = self.L.rg_history_keys(self.h, env, int(previous), None, 0, C.byref(n))
        self.check(rc)
        buf = (C.c_uint8 * max(1, n.value))()
        self.check(self.L.rg_history_keys(self.h, env, int(previous), buf, len(buf), C.byref(n)))
        return bytes(buf[: n.value])

    def dump_history(self, env, previous=False):
        need = C.c_size_t()
        self.check(self.L.rg_dump_history(self.h, env, int(previous), None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        self.check(self.L.rg_dump_history(self.h, env, int(previous), buf, need.value, None))
        return buf.value.decode()


_EAGER_MAX_BYTES = 1 << 16  # rg_step_fetch: batches of at most this many screen bytes get their screens written to pinned memory by the step itself
_LAZY_MIN_BYTES = 1 << 20   # batches with at least this many screen bytes keep their screens on the device until they are looked at
_LAZY_MAX_BYTES = 1 << 30   # device bytes all not-yet-read snapshots of a handle may pin together (65 536 mini envs: 8 batches); older ones move to the host
_IMAGE_BATCH_LIMIT = 1 << 28  # bytes: above this a batch does not cache whole-batch images (one-hot images of large batches are GBs)


class StateBatch:
    """What ParallelGameState::states/step/reset return (`Vec<PlayerState>`, python/src/lib.rs:308-328), kept as ONE snapshot of the batch
    instead of n Python objects: a read-only sequence whose items are PlayerState views, plus vector accessors (gold, dungeon_level,
    is_terminal) and whole-batch images computed by one kernel launch."""

    def __init__(self, handle, keys=None):
        """keys: step the batch with these keys first (ParallelGameState.step), through rg_step_fetch where it applies -- one call, one stream wait."""
        self._hd = handle
        n, h, w = handle.n, handle.height, handle.width
        self.n, self.symbols = n, handle.symbols
        # pinned, pooled buffers; each array owns its lease, so an array (or any slice of it) a caller keeps outlives this batch safely
        self.mixed_sizes = handle.mixed_sizes
        eager = keys is not None and handle.fast_step and not self.mixed_sizes and (h * w) % 4 == 0 and n * h * w <= _EAGER_MAX_BYTES
        if eager:  # (one lease for the four arrays of a small batch's step)
            self.status, self.flags, self._screen, self._hist = _leased_many(handle.pool, [((n, 10), np.int32), ((n,), np.uint32), ((n, h, w), np.uint8), ((n, h, w), np.uint8)])
        else:
            self.status = _leased(handle.pool, (n, 10), np.int32)
            self.flags = _leased(handle.pool, (n,), np.uint32)
        self._snap = None
        if keys is not None and not (handle.fast_step and not self.mixed_sizes and (h * w) % 4 == 0):
            handle.step_keys_host(keys)
            keys = None
        if keys is not None:
            # the step and its results in ONE call (rg_step_fetch): k_step reads the keys from pinned memory, one kernel writes status / flags -- and the
            # screens of a small batch -- straight into the pinned arrays below; a larger batch keeps its screens in a device snapshot, as without keys
            L = handle.L
            if eager:
                scr, hst = self._screen.ctypes.data, self._hist.ctypes.data
            else:
                self._screen = self._hist = None
                self._snap = _DevLease(handle.dev_pool, 2 * n * h * w)
                scr, hst = self._snap.ptr, self._snap.ptr + n * h * w
            rc = L.rg_step_fetch(handle.h, keys.ctypes.data, int(keys.shape[0]), C.c_void_p(scr), C.c_void_p(hst), self.status.ctypes.data, self.flags.ctypes.data)
            if rc and b"config groups" in L.rg_last_error(handle.h):  # (refused before anything was stepped: the general path from now on)
                handle.fast_step = False
                self._snap = None
                handle.step_keys_host(keys)
                keys = None
            else:
                handle.epoch += 1
                handle.check(rc)
                if self._snap is not None:
                    self._register_lazy()
        if keys is not None:
            pass
        elif self.mixed_sizes:  # envs of different width / height (ParallelGameState::new takes any config per env): ragged buffers, `screen` / `hist` are lists of per-env views
            off = np.concatenate(([0], np.cumsum(handle.env_heights.astype(np.int64) * handle.env_widths)))
            rs, rh = _leased(handle.pool, (int(off[-1]),), np.uint8), _leased(handle.pool, (int(off[-1]),), np.uint8)
            self._screen = [rs[off[i]:off[i + 1]].reshape(handle.env_heights[i], handle.env_widths[i]) for i in range(n)]
            self._hist = [rh[off[i]:off[i + 1]].reshape(handle.env_heights[i], handle.env_widths[i]) for i in range(n)]
            handle.check(handle.L.rg_fetch_states(handle.h, rs.ctypes.data, rh.ctypes.data, self.status.ctypes.data, self.flags.ctypes.data))
        elif n * h * w < _LAZY_MIN_BYTES:  # small batch: one call, four copies
            self._screen = _leased(handle.pool, (n, h, w), np.uint8)
            self._hist = _leased(handle.pool, (n, h, w), np.uint8)
            handle.check(handle.L.rg_fetch_states(handle.h, self._screen.ctypes.data, self._hist.ctypes.data, self.status.ctypes.data, self.flags.ctypes.data))
        else:
            # The RL loop reads gold and is_terminal of every state and nothing else (parallel.py:59-64): status + flags (44 B per env) come to the host
            # now, the screens (1 KB per env) are snapshotted device-to-device (~30 us for 65 536 envs) and cross PCIe only when somebody looks at them
            self._screen = self._hist = None
            self._snap = _DevLease(handle.dev_pool, 2 * n * h * w)
            handle.check(handle.L.rg_snapshot_take(handle.h, C.c_void_p(self._snap.ptr)))
            handle.check(handle.L.rg_fetch_states(handle.h, None, None, self.status.ctypes.data, self.flags.ctypes.data))  # (synchronises the stream)
            self._register_lazy()
        self._epoch = handle.epoch
        self._items = {}
        self._images = {}

    def _register_lazy(self):
        handle = self._hd
        key, lazy = id(self), handle.lazy
        lazy[key] = weakref.ref(self, lambda _r, key=key, lazy=lazy: lazy.pop(key, None))
        # Every lazy batch pins 2 * n * H * W bytes of HBM until it is collected or looked at: a caller that keeps the states of a whole rollout
        # (128 steps of 65 536 mini envs = 17 GB) must not exhaust the device silently.  Beyond _LAZY_MAX_BYTES the OLDEST batches get their host
        # copies now (what every batch cost before the lazy path existed) and hand their device buffer back.
        live = [b for b in (r() for r in list(lazy.values())) if b is not None and b._snap is not None]
        total = sum(b._snap.nbytes for b in live)
        for b in live:
            if total <= _LAZY_MAX_BYTES:
                break
            if b is not self:
                total -= b._snap.nbytes
                b._materialise()

    def _materialise(self):
        """Host copies of the whole snapshot (one D2H of screen + history into pinned memory); the device buffer goes back to the pool."""
        if self._snap is None:
            return
        hd = self._hd
        if hd.h is None:
            raise RuntimeError("Error in rogue-gym: the game was closed")
        n, h, w = self.n, hd.height, hd.width
        both = _leased(hd.pool, (2, n, h, w), np.uint8)
        hd.check(hd.L.rg_dev_read(hd.h, C.c_void_p(self._snap.ptr), both.ctypes.data, both.nbytes))
        self._screen, self._hist, self._snap = both[0], both[1], None
        hd.lazy.pop(id(self), None)

    @property
    def screen(self):
        """u8 [n, H, W] glyph mirrors (a list of per-env arrays when the batch mixes sizes)."""
        if self._screen is None:
            self._materialise()
        return self._screen

    @property
    def hist(self):
        if self._hist is None:
            self._materialise()
        return self._hist

    def _row(self, i):
        """Screen and history of env i: views of the host copies if they exist, else 2 x H*W bytes fetched from the device snapshot."""
        if self._snap is None:
            return self._screen[i], self._hist[i]
        hd = self._hd
        n, h, w = self.n, hd.height, hd.width
        both = np.empty((2, h, w), np.uint8)  # one 2-row copy: env i's screen and, n * H * W bytes further, its history
        hd.check(hd.L.rg_dev_read_rows(hd.h, C.c_void_p(self._snap.ptr + i * h * w), n * h * w, both.ctypes.data, h * w, 2))
        return both[0], both[1]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        st = self._items.get(i)
        if st is None:
            # one state costs one small synchronous copy (~25 us); the whole snapshot costs its bytes at PCIe rate -- cheaper from the first access
            # for a small batch, from the ~100th for a 65 536-env one (67 MB: ~2.7 ms)
            if self._snap is not None and (self._snap.nbytes <= (16 << 20) or len(self._items) >= 128):
                self._materialise()
            scr, hist = self._row(i)
            st = self._items[i] = PlayerState(scr, hist, self.status[i], int(self._hd.env_symbols[i]), int(self.flags[i]), self._hd.device,
                                              batch=self, index=i, batch_images=self._hd.uniform_symbols and not self.mixed_sizes)
        return st

    def __iter__(self):
        return (self[i] for i in range(self.n))

    def __eq__(self, other):
        if isinstance(other, StateBatch):
            other = list(other)
        return list(self) == other

    @property
    def gold(self):
        return self.status[:, 1]

    @property
    def dungeon_level(self):
        return self.status[:, 0]

    @property
    def is_terminal(self):
        return (self.flags & RG_FLAG_TERMINAL) != 0

    def status_vec(self, flag):
        cols = [c for b, c in enumerate((0, 2, 3, 4, 5, 6, 7, 8, 9)) if int(flag) & (1 << b)]  # StatusFlagInner::to_vector (flags.rs:67-87)
        return self.status[:, cols]

    def images(self, kind, flag, with_hist):
        """f32 [n, C, H, W]: PlayerState::{gray,symbol}_image[_with_hist] of every state of the batch."""
        flag = 0 if flag is None else int(flag) & 0x1FF
        key = (int(kind), flag, bool(with_hist))
        img = self._images.get(key)
        if img is not None:
            return img
        if self.mixed_sizes:  # no common [N, C, H, W]: one image per env, each in its own size (what the reference's per-state expand gives)
            img = self._images[key] = [self[i]._image(kind, flag, with_hist) for i in range(self.n)]
            return img
        hd, L = self._hd, self._hd.L
        h, w = hd.height, hd.width
        c = (self.symbols if kind else 1) + bin(flag).count("1") + (1 if with_hist else 0)
        nbytes = self.n * c * h * w * 4
        if nbytes <= _IMAGE_BATCH_LIMIT and not hd.pool.closed:  # pinned, pooled: a fresh 16 MB numpy array per step costs more in page faults than the copy
            img = _leased(hd.pool, (self.n, c, h, w), np.float32)
        else:
            img = np.empty((self.n, c, h, w), np.float32)
        if hd.h is not None and hd.epoch == self._epoch:
            hd.check(L.rg_obs_host(hd.h, int(kind), flag, int(with_hist), img.ctypes.data))  # the device still holds exactly these states
        elif hd.uniform_symbols:  # the envs have moved on: encode the snapshot itself
            rc = L.rg_encode_host_batch(hd.device, self.n, self.screen.ctypes.data, self.hist.ctypes.data, self.status.ctypes.data, h, w, self.symbols, flag,
                                        int(with_hist), int(kind), img.ctypes.data)
            if rc:
                raise RuntimeError("Error in rogue-gym: " + L.rg_last_error(None).decode())
        else:  # configs with different `symbols` in one batch: env by env, each with its own (gray values divide by the env's symbols)
            if kind:
                raise RuntimeError("Error in rogue-gym: a batch whose configs differ in `symbols` has no common symbol-image shape once the envs moved on")
            for i in range(self.n):
                img[i] = self[i]._image(kind, flag, with_hist)
        if img.nbytes <= _IMAGE_BATCH_LIMIT:
            self._images[key] = img
        return img


class PlayerState:
    """A memory efficient representation of Agent observation (python/src/lib.rs:27-206): a value
    object holding host copies of the mirror screen, history plane, status and flags."""

    def __init__(self, screen, hist, status, symbols, flags, device=0, batch=None, index=0, batch_images=True):
        self._map = np.ascontiguousarray(screen, np.uint8)
        self._hist = np.ascontiguousarray(hist, np.uint8)
        self._status = np.ascontiguousarray(status, np.int32)
        self._symbols = int(symbols)
        self._flags = int(flags)
        self._terminal = bool(flags & RG_FLAG_TERMINAL)
        self._device = device
        self._batch, self._index = batch, index  # keeps the batch (and its pinned buffers, which these arrays are views of) alive while this view exists
        self._batch_images = batch is not None and batch_images  # images of the whole batch by one launch (not when the batch mixes `symbols`)

    def __repr__(self):
        s = self._status  # Status::fmt (player.rs:433-449)
        hunger = {0: "", 1: "hungry", 2: "weak"}[int(s[9])]
        status = "Level: %2d Gold: %5d Hp: %2d(%2d) Str: %2d(%2d) Arm: %2d Exp: %2d/%2d %s" % (
            s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], hunger)
        return "".join(row + "\n" for row in self.dungeon) + status

    __str__ = __repr__

    def __eq__(self, other):
        return (isinstance(other, PlayerState) and np.array_equal(self._map, other._map) and np.array_equal(self._hist, other._hist)
                and np.array_equal(self._status, other._status) and self._symbols == other._symbols
                and (self._flags & RG_FLAG_MSG_MASK) == (other._flags & RG_FLAG_MSG_MASK) and self._terminal == other._terminal)

    @property
    def status(self):
        return {k: int(v) & 0xFFFFFFFF for k, v in zip(STATUS_KEYS, self._status)}

    @property
    def dungeon(self):
        return [bytes(row).decode("latin-1") for row in self._map]

    @property
    def dungeon_level(self):
        return int(self._status[0])

    @property
    def gold(self):
        return int(self._status[1])

    @property
    def symbols(self):
        return self._symbols

    @property
    def is_terminal(self):
        return self._terminal

    @property
    def message_flags(self):
        """MessageFlagInner bits (python/src/flags.rs:6-39); not exposed by the reference's Python API."""
        return (self._flags & RG_FLAG_MSG_MASK) >> RG_FLAG_MSG_SHIFT

    def status_vec(self, flag):
        order = (0, 2, 3, 4, 5, 6, 7, 8, 9)  # StatusFlagInner::to_vector (flags.rs:67-87)
        return [int(self._status[order[b]]) for b in range(9) if flag & (1 << b)]

    def _image(self, kind, flag, with_hist):
        flag = 0 if flag is None else int(flag)
        h, w = self._map.shape
        c = (self._symbols if kind else 1) + bin(flag & 0x1FF).count("1") + (1 if with_hist else 0)
        b = self._batch
        if self._batch_images and b.n * c * h * w * 4 <= _IMAGE_BATCH_LIMIT:
            return b.images(kind, flag, with_hist)[self._index].copy()  # one launch serves every state of the batch
        L = load_library()
        out = np.empty((c, h, w), np.float32)
        rc = L.rg_encode_host(self._device, self._map.ctypes.data, self._hist.ctypes.data, self._status.ctypes.data, h, w, self._symbols,
                              flag, int(with_hist), kind, out.ctypes.data)
        if rc:
            raise RuntimeError("Error in rogue-gym: " + L.rg_last_error(None).decode())
        return out

    def gray_image(self, flag=None):
        return self._image(0, flag, False)

    def gray_image_with_hist(self, flag=None):
        return self._image(0, flag, True)

    def symbol_image(self, flag=None):
        return self._image(1, flag, False)

    def symbol_image_with_hist(self, flag=None):
        return self._image(1, flag, True)


_HISTORY_CAP_MAX = 1 << 20


def _keys_array(input):
    if isinstance(input, (bytes, bytearray)):
        return np.frombuffer(bytes(input), np.uint8)
    if isinstance(input, np.ndarray):
        return np.ascontiguousarray(input.astype(np.int64) & 0xFF, dtype=np.uint8)
    return np.ascontiguousarray(np.fromiter((int(k) & 0xFF for k in input), dtype=np.int64), dtype=np.uint8)


class GameState:
    """python/src/lib.rs:208-258 (one env, no auto-reset)."""

    def __init__(self, max_steps, config_str=None, device=None):
        self._h = _Handle([config_str], max_steps, auto_reset=False, device=device)
        self._max_steps = int(max_steps)
        # saved_inputs lives on the device: react() beyond max_steps + 1 keys is a no-op, so the log never outgrows that
        self._h.check(self._h.L.rg_history_enable(self._h.h, min(self._max_steps + 2, _HISTORY_CAP_MAX)))
        self._prev = None

    def screen_size(self):
        return (self._h.height, self._h.width)

    def set_seed(self, seed):
        lo = (C.c_uint64 * 1)(int(seed) & 0xFFFFFFFFFFFFFFFF)
        hi = (C.c_uint64 * 1)(0)  # `seed as u128` of a u64 (python/src/lib.rs:229-232)
        self._h.check(self._h.L.rg_seed(self._h.h, lo, hi, 1))

    def reset(self):
        self._h.check(self._h.L.rg_reset(self._h.h))
        self._h.epoch += 1
        self._prev = None

    def prev(self):
        if self._prev is None:
            self._prev = self._h.snapshot()[0]
        return self._prev

    def react(self, input):
        keys = (C.c_uint8 * 1)(int(input) & 0xFF)
        self._h.check(self._h.L.rg_step(self._h.h, keys, 0))
        self._h.epoch += 1
        self._prev = None
        self._h.check(self._h.L.rg_sync(self._h.h))

    def dump_history(self):
        return self._h.dump_history(0)

    def dump_config(self):
        buf = C.create_string_buffer(1 << 16)
        self._h.check(self._h.L.rg_dump_config(self._h.h, 0, buf, len(buf)))
        return buf.value.decode()

    def symbols(self):
        return self._h.symbols


class ParallelGameState:
    """python/src/lib.rs:260-335: the reference's one-OS-thread-per-env ThreadConductor becomes one
    batched kernel launch; envs auto-reset on terminal (thread_impls.rs:69-79).  states/step/reset return a StateBatch
    (a sequence of PlayerState, like the reference's Vec<PlayerState>)."""

    def __init__(self, max_steps, configs, device=None, history=0):
        self._h = _Handle(list(configs), max_steps, auto_reset=True, device=device)
        if history:
            self.enable_history(history)

    def screen_size(self):
        return (self._h.height, self._h.width)

    def symbols(self):
        return self._h.symbols

    def seed(self, seed):
        seed = [int(s) for s in seed][: self._h.n]  # zip (thread_impls.rs:45-50)
        n = len(seed)
        if n == 0:
            return
        lo = (C.c_uint64 * n)(*[s & 0xFFFFFFFFFFFFFFFF for s in seed])
        hi = (C.c_uint64 * n)(*[(s >> 64) & 0xFFFFFFFFFFFFFFFF for s in seed])
        self._h.check(self._h.L.rg_seed(self._h.h, lo, hi, n))

    def states(self):
        return self._h.snapshot()

    def step(self, input):
        """One key per env.  Like ThreadConductor::step the keys are zipped with the envs (thread_impls.rs:62-64): surplus keys are dropped,
        and with fewer keys than envs only that prefix is stepped (the reference would then wait forever for the others' replies)."""
        keys = _keys_array(input)
        return StateBatch(self._h, keys=keys)

    def reset(self):
        self._h.check(self._h.L.rg_reset(self._h.h))
        self._h.epoch += 1
        return self._h.snapshot()

    def dump_config(self, env=0):
        buf = C.create_string_buffer(1 << 16)
        self._h.check(self._h.L.rg_dump_config(self._h.h, int(env), buf, len(buf)))
        return buf.value.decode()

    def enable_history(self, cap_per_env):
        """Keep the keys of the running and of the previous episode of every env on the device (not part of the reference's API)."""
        self._h.check(self._h.L.rg_history_enable(self._h.h, int(cap_per_env)))

    def dump_history(self, env, previous=False):
        """GameState::dump_history for env `env` of the batch; previous=True: the episode that ended with the last auto-reset."""
        return self._h.dump_history(int(env), previous)

    def close(self):
        self._h.close()


def replay(game, interval_ms=100):
    raise RuntimeError("replay needs the reference's terminal UI (devui), which is out of scope for the HIP stepper")


def play_cli(game):
    raise RuntimeError("play_cli needs the reference's terminal UI (devui), which is out of scope for the HIP stepper")
