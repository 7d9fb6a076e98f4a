"""`rogue_gym_python._rogue_gym` -- host mirror of the reference's PyO3 module
(/root/reference/python/src/lib.rs:355-366) over the HIP C-ABI (include/rogue_gym_hip.h).

Same classes, method names, argument meaning and error behaviour as the Rust module:
GameState, ParallelGameState, PlayerState.  All game logic runs in librogue_gym_hip.so on an
MI355X; this file only marshals.  There is no CPU fallback: importing works anywhere, but
constructing a GameState without the built library or without a HIP device raises.
"""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "librogue_gym_hip.so")

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
        ("mon_x", C.c_int32 * 32), ("mon_y", C.c_int32 * 32), ("mon_type", C.c_int32 * 32), ("mon_active", C.c_int32 * 32), ("mon_hp", C.c_int32 * 32),
        ("mon_exp", C.c_uint32 * 32),
        ("gold_x", C.c_int32 * 32), ("gold_y", C.c_int32 * 32), ("gold_amount", C.c_int32 * 32),
    ]


_lib = None


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
    vp, u8p, i32p, u32p, f32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.rg_create.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(vp)]
    L.rg_destroy.argtypes = [vp]
    L.rg_destroy.restype = None
    L.rg_last_error.argtypes = [vp]
    L.rg_last_error.restype = C.c_char_p
    L.rg_dims.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
    L.rg_set_stream.argtypes = [vp, vp]
    L.rg_seed.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    L.rg_reset.argtypes = [vp]
    L.rg_step.argtypes = [vp, vp, C.c_int]
    L.rg_sync.argtypes = [vp]
    L.rg_screen.argtypes = [vp, C.POINTER(vp)]
    L.rg_hist.argtypes = [vp, C.POINTER(vp)]
    L.rg_status.argtypes = [vp, C.POINTER(vp)]
    L.rg_flags.argtypes = [vp, C.POINTER(vp)]
    L.rg_reward.argtypes = [vp, C.POINTER(vp)]
    L.rg_done.argtypes = [vp, C.POINTER(vp)]
    L.rg_obs_gray.argtypes = [vp, C.c_uint32, C.c_int, vp]
    L.rg_obs_symbol.argtypes = [vp, C.c_uint32, C.c_int, vp]
    L.rg_obs_channels.argtypes = [vp, C.c_int, C.c_uint32, C.c_int]
    L.rg_fetch_states.argtypes = [vp, vp, vp, vp, vp]
    L.rg_encode_host.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, vp]
    L.rg_timing_enable.argtypes = [vp, C.c_int]
    L.rg_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.rg_dump_config.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    L.rg_config_canonical.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.rg_debug_fetch.argtypes = [vp, C.c_int, C.POINTER(RgDebugState), vp]
    for f in ("rg_create", "rg_dims", "rg_set_stream", "rg_seed", "rg_reset", "rg_step", "rg_sync", "rg_screen", "rg_hist", "rg_status", "rg_flags",
              "rg_reward", "rg_done", "rg_obs_gray", "rg_obs_symbol", "rg_obs_channels", "rg_fetch_states", "rg_encode_host", "rg_dump_config", "rg_config_canonical", "rg_debug_fetch", "rg_timing_enable", "rg_timing_read"):
        getattr(L, f).restype = C.c_int
    _ = (u8p, i32p, u32p, f32p)
    _lib = L
    return L


def _default_device():
    return int(os.environ.get("ROGUE_GYM_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


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

    def check(self, rc):
        if rc:
            raise RuntimeError("Error in rogue-gym: " + self.L.rg_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.rg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fetch(self):
        n, hw = self.n, self.height * self.width
        screen = np.empty((n, self.height, self.width), np.uint8)
        hist = np.empty((n, self.height, self.width), np.uint8)
        status = np.empty((n, 10), np.int32)
        flags = np.empty(n, np.uint32)
        self.check(self.L.rg_fetch_states(self.h, screen.ctypes.data, hist.ctypes.data, status.ctypes.data, flags.ctypes.data))
        _ = hw
        return screen, hist, status, flags

    def states(self):
        screen, hist, status, flags = self.fetch()
        return [PlayerState(screen[i], hist[i], status[i], self.symbols, int(flags[i]), self.device) for i in range(self.n)]

    def debug_state(self, env):
        out = RgDebugState()
        cells = np.empty((self.height, self.width), np.uint16)
        self.check(self.L.rg_debug_fetch(self.h, env, C.byref(out), cells.ctypes.data))
        return out, cells


class PlayerState:
    """A memory efficient representation of Agent observation (python/src/lib.rs:27-206): a value
    object holding host copies of the mirror screen, history plane, status and flags."""

    def __init__(self, screen, hist, status, symbols, flags, device=0):
        self._map = np.ascontiguousarray(screen, np.uint8)
        self._hist = np.ascontiguousarray(hist, np.uint8)
        self._status = np.ascontiguousarray(status, np.int32)
        self._symbols = int(symbols)
        self._flags = int(flags)
        self._terminal = bool(flags & RG_FLAG_TERMINAL)
        self._device = device

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
        L = load_library()
        h, w = self._map.shape
        c = (self._symbols if kind else 1) + bin(flag & 0x1FF).count("1") + (1 if with_hist else 0)
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


_KEY_TO_INPUT = {  # KeyMap::ai (input.rs:73-100) -> serde form of InputCode, for dump_history
    "l": "Right", "k": "Up", "j": "Down", "h": "Left", "u": "RightUp", "y": "LeftUp", "n": "RightDown", "b": "LeftDown",
}


def _input_code(key):
    ch = chr(key)
    if ch in _KEY_TO_INPUT:
        return {"Act": {"Move": _KEY_TO_INPUT[ch]}}
    if ch.lower() in _KEY_TO_INPUT and ch.isupper():
        return {"Act": {"MoveUntil": _KEY_TO_INPUT[ch.lower()]}}
    return {"Act": {".": "NoOp", "s": "Search", ">": "DownStair"}[ch]}


class GameState:
    """python/src/lib.rs:208-258 (one env, no auto-reset)."""

    def __init__(self, max_steps, config_str=None, device=None):
        self._h = _Handle([config_str], max_steps, auto_reset=False, device=device)
        self._max_steps = int(max_steps)
        self._steps = 0
        self._history = []
        self._prev = None

    def screen_size(self):
        return (self._h.height, self._h.width)

    def set_seed(self, seed):
        lo = (C.c_uint64 * 1)(int(seed) & 0xFFFFFFFFFFFFFFFF)
        hi = (C.c_uint64 * 1)(0)  # `seed as u128` of a u64 (python/src/lib.rs:229-232)
        self._h.check(self._h.L.rg_seed(self._h.h, lo, hi, 1))

    def reset(self):
        self._h.check(self._h.L.rg_reset(self._h.h))
        self._steps = 0
        self._history = []
        self._prev = None

    def prev(self):
        if self._prev is None:
            self._prev = self._h.states()[0]
        return self._prev

    def react(self, input):
        key = int(input) & 0xFF
        keys = (C.c_uint8 * 1)(key)
        if self._steps > self._max_steps:
            return
        self._h.check(self._h.L.rg_step(self._h.h, keys, 0))
        rc = self._h.L.rg_sync(self._h.h)
        if rc:
            if chr(key) in "hjklyubnHJKLYUBN.s>":
                self._history.append(_input_code(key))  # saved_inputs.push precedes the IgnoredInput error (core/src/lib.rs:288)
            self._h.check(rc)
        self._history.append(_input_code(key))
        self._steps += 1
        self._prev = None

    def dump_history(self):
        return json.dumps(self._history, indent=2)

    def dump_config(self):
        buf = C.create_string_buffer(1 << 16)
        self._h.check(self._h.L.rg_dump_config(self._h.h, 0, buf, len(buf)))
        return buf.value.decode()

    def symbols(self):
        return self._h.symbols


class ParallelGameState:
    """python/src/lib.rs:260-335: the reference's one-OS-thread-per-env ThreadConductor becomes one
    batched kernel launch; envs auto-reset on terminal (thread_impls.rs:69-79)."""

    def __init__(self, max_steps, configs, device=None):
        self._h = _Handle(list(configs), max_steps, auto_reset=True, device=device)

    def screen_size(self):
        return (self._h.height, self._h.width)

    def symbols(self):
        return self._h.symbols

    def seed(self, seed):
        seed = list(seed)[: self._h.n]
        n = len(seed)
        lo = (C.c_uint64 * n)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in seed])
        hi = (C.c_uint64 * n)(*[(int(s) >> 64) & 0xFFFFFFFFFFFFFFFF for s in seed])
        self._h.check(self._h.L.rg_seed(self._h.h, lo, hi, n))

    def states(self):
        return self._h.states()

    def step(self, input):
        keys = np.ascontiguousarray(np.asarray(list(input), dtype=np.int64) & 0xFF, dtype=np.uint8)
        if keys.shape[0] < self._h.n:  # zip semantics (thread_impls.rs:62-64): missing keys leave an env unstepped; we require all
            raise RuntimeError("Error in rogue-gym: expected %d keys, got %d" % (self._h.n, keys.shape[0]))
        self._h.check(self._h.L.rg_step(self._h.h, keys.ctypes.data, 0))
        self._h.check(self._h.L.rg_sync(self._h.h))
        return self._h.states()

    def reset(self):
        self._h.check(self._h.L.rg_reset(self._h.h))
        return self._h.states()

    def close(self):
        self._h.close()


def replay(game, interval_ms=100):
    raise RuntimeError("replay needs the reference's terminal UI (devui), which is out of scope for the HIP stepper")


def play_cli(game):
    raise RuntimeError("play_cli needs the reference's terminal UI (devui), which is out of scope for the HIP stepper")
