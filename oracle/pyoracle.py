"""ctypes binding of the CPU oracle (oracle/rogue_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (rogue-gym_amd/) never imports this module.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librogue_oracle.so")
_SO_OVERRIDE = os.environ.get("ROGUE_ORACLE_SO")  # tests/test_oracle_mutations.py: a mutant build of the same source (-DORC_MUTANT=k), loaded instead


class OrcMonStat(C.Structure):
    _fields_ = [("n_attack", C.c_int32), ("att_times", C.c_int32 * 4), ("att_max", C.c_int32 * 4), ("attr", C.c_int32), ("defense", C.c_int32),
                ("exp", C.c_uint32), ("level", C.c_int32), ("rarity", C.c_int32), ("tile", C.c_int32)]


NAME_CAP, MAX_STATS, MAX_INIT_ITEMS = 32, 16, 16


class OrcWeaponStat(C.Structure):
    _fields_ = [("name", C.c_char * NAME_CAP), ("wield_times", C.c_uint64), ("wield_max", C.c_int64), ("init_lo", C.c_uint32), ("init_hi", C.c_uint32),
                ("attr", C.c_uint32)]


class OrcArmorStat(C.Structure):
    _fields_ = [("name", C.c_char * NAME_CAP), ("def_", C.c_int32)]


class OrcItem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("how_many", C.c_uint32), ("attr", C.c_uint32), ("name", C.c_char * NAME_CAP), ("wield_times", C.c_uint64),
                ("wield_max", C.c_int64), ("hit_plus", C.c_int64), ("dam_plus", C.c_int64), ("def_", C.c_int32), ("def_plus", C.c_int32)]


class OrcInitItem(C.Structure):
    _fields_ = [("tag", C.c_int32), ("name", C.c_char * NAME_CAP), ("num_plus", C.c_uint32), ("hit_plus", C.c_int32), ("dam_plus", C.c_int32),
                ("def_plus", C.c_int32), ("item", OrcItem)]


ITEM_KINDS = ["Armor", "Food", "Gold", "Potion", "Ring", "Scroll", "Wand", "Weapon"]  # ItemKind order (item/mod.rs:32-41)


class OrcConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("seed_lo", C.c_uint64), ("seed_hi", C.c_uint64),
        ("hide_dungeon", C.c_int32),
        ("room_num_x", C.c_int32), ("room_num_y", C.c_int32),
        ("min_room_x", C.c_int32), ("min_room_y", C.c_int32),
        ("max_empty_rooms", C.c_uint32), ("amulet_level", C.c_uint32),
        ("maze_rate_inv", C.c_uint32), ("dark_level", C.c_uint32),
        ("hidden_passage_rate_inv", C.c_uint32), ("locked_door_rate_inv", C.c_uint32),
        ("max_extra_edges", C.c_uint32),
        ("door_unlock_rate_inv", C.c_uint32), ("passage_unlock_rate_inv", C.c_uint32),
        ("gold_rate_inv", C.c_uint32), ("gold_base", C.c_uint32),
        ("gold_per_level", C.c_uint32), ("gold_minimum", C.c_uint32),
        ("hunger_time", C.c_uint32),
        ("init_hp", C.c_int64),
        ("appear_rate_gold", C.c_uint32), ("appear_rate_nogold", C.c_uint32),
        ("n_enemies", C.c_int32),
        ("enemy_builtin", C.c_int32 * 32),
        ("enemy_custom", OrcMonStat * 32),
        ("choose_width", C.c_int32),
        ("n_weapons", C.c_int32), ("n_armors", C.c_int32), ("n_init_items", C.c_int32),
        ("weapons", OrcWeaponStat * MAX_STATS), ("armors", OrcArmorStat * MAX_STATS), ("init_items", OrcInitItem * MAX_INIT_ITEMS),
        ("max_items", C.c_uint64),
    ]


class OrcMonster(C.Structure):
    _fields_ = [
        ("x", C.c_int32), ("y", C.c_int32), ("type", C.c_int32), ("active", C.c_int32), ("running", C.c_int32),
        ("hp", C.c_int64), ("max_hp", C.c_int64), ("level", C.c_int64),
        ("defense", C.c_int32), ("exp", C.c_uint32),
    ]


def build(force=False):
    if _SO_OVERRIDE:
        return _SO_OVERRIDE
    src = [os.path.join(_HERE, f) for f in ("rogue_oracle.c", "rogue_oracle.h")]
    if force or not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def build_native():
    """bench.py's cpu_baseline leg only: the same C restatement compiled ON THE HOST IT IS TIMED ON with -O3 -march=native, into
    its own file; the portable build stays what the tests check.  Returns the path, or None when this host has no compiler (the portable library is
    timed then, and the bench line says so).  Must be called before the first lib()."""
    global _SO
    out = os.path.join(_HERE, "_build", "librogue_oracle_native.so")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-std=gnu11", "-fPIC", "-shared", "-o", out, os.path.join(_HERE, "rogue_oracle.c"), "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:  # noqa: BLE001
        return None
    if _lib is None:
        _SO = out
        return out
    return None


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO_OVERRIDE or _SO)
        L.orc_new.restype = C.c_void_p
        L.orc_new.argtypes = [C.POINTER(OrcConfig), C.c_uint64]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_set_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_debug_descend.argtypes = [C.c_void_p]
        L.orc_debug_descend.restype = None
        L.orc_react.argtypes = [C.c_void_p, C.c_uint8]
        L.orc_step_autoreset.argtypes = [C.c_void_p, C.c_uint8]
        for f in ("orc_screen", "orc_hist", "orc_status", "orc_flags", "orc_scalars"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_grid.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_monsters.argtypes = [C.c_void_p, C.POINTER(OrcMonster), C.c_int]
        L.orc_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rooms.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_move_enemy_kat.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2
        L.orc_status_vec.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        for f in ("orc_gray_image", "orc_symbol_image"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_kat_u32.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_kat_range64.restype = C.c_uint64
        L.orc_kat_range64.argtypes = [C.c_uint64] * 4 + [C.c_int]
        L.orc_batch_new.restype = C.c_void_p
        L.orc_batch_new.argtypes = [C.POINTER(OrcConfig), C.c_int, C.c_uint64, C.c_int]
        L.orc_batch_free.argtypes = [C.c_void_p]
        L.orc_batch_env.restype = C.c_void_p
        L.orc_batch_env.argtypes = [C.c_void_p, C.c_int]
        L.orc_batch_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_config_default.argtypes = [C.POINTER(OrcConfig)]
        _lib = L
    return _lib


def config_from_dict(d, seed=None, choose_width=64):
    """Flatten a reference GameConfig JSON dict (core/src/lib.rs:42-86) into OrcConfig."""
    c = OrcConfig()
    lib().orc_config_default(C.byref(c))
    c.width = d.get("width", 80)
    c.height = d.get("height", 24)
    s = d.get("seed") if seed is None else seed
    if s is None:
        raise ValueError("oracle needs an explicit seed")
    c.seed_lo = s & 0xFFFFFFFFFFFFFFFF
    c.seed_hi = (s >> 64) & 0xFFFFFFFFFFFFFFFF
    c.hide_dungeon = int(d.get("hide_dungeon", True))
    dg = d.get("dungeon", {})
    c.room_num_x = dg.get("room_num_x", 3)
    c.room_num_y = dg.get("room_num_y", 3)
    mrs = dg.get("min_room_size", {"x": 4, "y": 4})
    c.min_room_x, c.min_room_y = mrs["x"], mrs["y"]
    for k in ("max_empty_rooms", "amulet_level", "maze_rate_inv", "dark_level", "hidden_passage_rate_inv",
              "locked_door_rate_inv", "max_extra_edges", "door_unlock_rate_inv", "passage_unlock_rate_inv"):
        if k in dg:
            setattr(c, k, dg[k])
    gold = d.get("item", {}).get("gold", {})
    for k, f in (("rate_inv", "gold_rate_inv"), ("base", "gold_base"), ("per_level", "gold_per_level"), ("minimum", "gold_minimum")):
        if k in gold:
            setattr(c, f, gold[k])
    pl = d.get("player", {})
    if "hunger_time" in pl:
        c.hunger_time = pl["hunger_time"]
    if "init_hp" in pl:
        c.init_hp = pl["init_hp"]
    en = d.get("enemies", {})
    if "enemies" in en:
        ids = en["enemies"]
        c.n_enemies = len(ids)
        for i, v in enumerate(ids):
            if isinstance(v, int):
                c.enemy_builtin[i] = v
                continue
            c.enemy_builtin[i] = -1  # Preset::Custom(Status)
            m = c.enemy_custom[i]
            m.n_attack = len(v["attack"])
            for k, die in enumerate(v["attack"]):
                m.att_times[k], m.att_max[k] = die["times"], die["max"]
            m.attr, m.defense, m.exp, m.level, m.rarity, m.tile = v["attr"], v["defense"], v["exp"], v["level"], v["rarelity"], v["tile"]
    if "appear_rate_gold" in en:
        c.appear_rate_gold = en["appear_rate_gold"]
    if "appear_rate_nogold" in en:
        c.appear_rate_nogold = en["appear_rate_nogold"]
    c.choose_width = choose_width
    _items_from_dict(c, d)
    return c


def _name(s):
    b = s.encode()
    if len(b) >= NAME_CAP:
        raise ValueError("oracle: item name longer than %d bytes" % (NAME_CAP - 1))
    return b


def _items_from_dict(c, d):
    """item.weapon / item.armor presets and player.{init_items,max_items} (weapon.rs:12-81, armor.rs:10-86, player.rs:26-29) into the
    oracle's tables.  orc_config_default filled in the builtin tables and the default pack; builtin presets are copied from there."""
    item = d.get("item")
    if item is not None:
        for k in ("armor", "gold", "weapon"):  # item::Config has no serde defaults (item/mod.rs:24-29)
            if k not in item:
                raise ValueError("missing field `%s`" % k)
        if "weapons" in item["weapon"]:
            builtin = [OrcWeaponStat.from_buffer_copy(w) for w in c.weapons[:9]]
            ws = item["weapon"]["weapons"]
            if len(ws) > MAX_STATS:
                raise ValueError("oracle: at most %d weapon presets" % MAX_STATS)
            c.n_weapons = len(ws)
            for i, w in enumerate(ws):
                if isinstance(w, int):
                    c.weapons[i] = builtin[w]
                else:
                    c.weapons[i] = OrcWeaponStat(_name(w["name"]), w["at_weild"]["times"], w["at_weild"]["max"], w["init_num"]["start"],
                                                 w["init_num"]["end"], w["attr"])
        if "armors" in item["armor"]:
            builtin = [OrcArmorStat.from_buffer_copy(a) for a in c.armors[:8]]
            ar = item["armor"]["armors"]
            if len(ar) > MAX_STATS:
                raise ValueError("oracle: at most %d armor presets" % MAX_STATS)
            c.n_armors = len(ar)
            for i, a in enumerate(ar):
                c.armors[i] = builtin[a] if isinstance(a, int) else OrcArmorStat(_name(a["name"]), a["def"])
    pl = d.get("player", {})
    if "max_items" in pl:
        c.max_items = pl["max_items"]
    if "init_items" in pl:
        items = pl["init_items"]
        if len(items) > MAX_INIT_ITEMS:
            raise ValueError("oracle: at most %d init_items" % MAX_INIT_ITEMS)
        c.n_init_items = len(items)
        for i, it in enumerate(items):
            (tag, b), = it.items()
            o = OrcInitItem()
            if tag == "Weapon":
                o.tag, o.name, o.num_plus, o.hit_plus, o.dam_plus = 2, _name(b["name"]), b["num_plus"], b["hit_plus"], b["dam_plus"]
            elif tag == "Armor":
                o.tag, o.name, o.def_plus = 1, _name(b["name"]), b["def_plus"]
            elif tag == "Noinit":
                o.tag = 0
                kind = b["kind"]
                o.item.how_many, o.item.attr = b["how_many"], b["attr"]
                if isinstance(kind, str):
                    o.item.kind = ITEM_KINDS.index(kind)
                else:
                    (kt, kb), = kind.items()
                    o.item.kind = ITEM_KINDS.index(kt)
                    if kt == "Weapon":
                        o.item.name, o.item.wield_times, o.item.wield_max = _name(kb["name"]), kb["at_weild"]["times"], kb["at_weild"]["max"]
                        o.item.hit_plus, o.item.dam_plus = kb["hit_plus"], kb["dam_plus"]
                    elif kt == "Armor":
                        o.item.name, o.item.def_, o.item.def_plus = _name(kb["name"]), kb["def"], kb["def_plus"]
            else:
                raise ValueError("unknown InitItem variant " + tag)
            c.init_items[i] = o


STATUS_KEYS = ["dungeon_level", "gold", "hp_current", "hp_max", "str_current", "str_max", "defense", "player_level", "exp", "hunger"]


class OracleEnv:
    """One environment == GameStateImpl (python/src/state_impls.rs)."""

    def __init__(self, config, max_steps=1000, seed=None, choose_width=64):
        if isinstance(config, str):
            config = json.loads(config)
        self.cfg = config_from_dict(config, seed=seed, choose_width=choose_width)
        self.h, self.w = self.cfg.height, self.cfg.width
        self._L = lib()
        self._e = self._L.orc_new(C.byref(self.cfg), max_steps)
        if not self._e:
            raise RuntimeError("oracle: invalid config")

    def __del__(self):
        if getattr(self, "_e", None):
            self._L.orc_free(self._e)
            self._e = None

    def set_seed(self, s):
        self._L.orc_set_seed(self._e, s & 0xFFFFFFFFFFFFFFFF, (s >> 64) & 0xFFFFFFFFFFFFFFFF)

    def reset(self):
        self._L.orc_reset(self._e)

    def debug_descend(self):
        self._L.orc_debug_descend(self._e)

    def react(self, key):
        rc = self._L.orc_react(self._e, key if isinstance(key, int) else ord(key))
        if rc:
            raise RuntimeError("Error in rogue-gym: oracle react rc=%d" % rc)

    def react_str(self, keys):
        for k in keys:
            self.react(k)

    def step_autoreset(self, key):
        rc = self._L.orc_step_autoreset(self._e, key if isinstance(key, int) else ord(key))
        if rc:
            raise RuntimeError("Error in rogue-gym: oracle react rc=%d" % rc)

    # mirrors
    def screen(self):
        a = np.empty((self.h, self.w), np.uint8)
        self._L.orc_screen(self._e, a.ctypes.data)
        return a

    def dungeon(self):
        return [bytes(r).decode("ascii") for r in self.screen()]

    def hist(self):
        a = np.empty((self.h, self.w), np.uint8)
        self._L.orc_hist(self._e, a.ctypes.data)
        return a

    def status_arr(self):
        a = np.empty(10, np.uint32)
        self._L.orc_status(self._e, a.ctypes.data)
        return a

    def status(self):
        return dict(zip(STATUS_KEYS, (int(v) for v in self.status_arr())))

    def flags(self):
        a = np.empty(5, np.uint32)
        self._L.orc_flags(self._e, a.ctypes.data)
        return dict(is_terminal=bool(a[0]), message=int(a[1]), steps=int(a[2]), dead=bool(a[3]), symbols=int(a[4]))

    @property
    def symbols(self):
        return self.flags()["symbols"]

    # internals
    def grid(self):
        n = (self.h, self.w)
        s, a, d = (np.empty(n, np.uint8) for _ in range(3))
        g = np.empty(n, np.int32)
        self._L.orc_grid(self._e, s.ctypes.data, a.ctypes.data, d.ctypes.data, g.ctypes.data)
        return s, a, d, g

    def scalars(self):
        a = np.zeros(16, np.int64)
        self._L.orc_scalars(self._e, a.ctypes.data)
        keys = ["px", "py", "level", "hp", "hp_max", "exp", "plevel", "food_left", "quiet", "gold", "n_monsters", "n_pack", "weapon_slot", "armor_slot"]
        return dict(zip(keys, (int(v) for v in a)))

    def monsters(self):
        buf = (OrcMonster * 512)()
        n = self._L.orc_monsters(self._e, buf, 512)
        return [dict(x=m.x, y=m.y, type=m.type, active=m.active, running=m.running, hp=m.hp, max_hp=m.max_hp,
                     level=m.level, defense=m.defense, exp=m.exp) for m in buf[:n]]

    def rooms(self):
        """Rooms of the current level (orc_rooms): kind 0 normal / 1 maze / 2 empty, flags, range and assigned area as half-open (x0, y0, x1, y1)."""
        buf = np.zeros(12 * 512, np.int32)
        n = self._L.orc_rooms(self._e, buf.ctypes.data, 512)
        r = buf[:12 * n].reshape(n, 12)
        return [dict(kind=int(q[0]), dark=bool(q[1]), visited=bool(q[2]), has_gold=bool(q[3]), range=tuple(int(v) for v in q[4:8]), assigned=tuple(int(v) for v in q[8:12])) for q in r]

    def rng(self):
        s = np.empty(12, np.uint32)
        c = np.empty(3, np.uint64)
        self._L.orc_rng(self._e, s.ctypes.data, c.ctypes.data)
        return s.reshape(3, 4), [int(v) for v in c]

    def move_enemy_kat(self, fx, fy, tx, ty):
        nx, ny = C.c_int(), C.c_int()
        r = self._L.orc_move_enemy_kat(self._e, fx, fy, tx, ty, C.byref(nx), C.byref(ny))
        return r, nx.value, ny.value

    # observations (PlayerState methods, python/src/lib.rs:158-205)
    def status_vec(self, flag):
        out = np.empty(9, np.int32)
        st = self.status_arr()
        n = self._L.orc_status_vec(st.ctypes.data, flag, out.ctypes.data)
        return [int(v) for v in out[:n]]

    def _image(self, fn, base, flag, with_hist):
        c = base + bin(flag).count("1") + (1 if with_hist else 0)
        out = np.empty((c, self.h, self.w), np.float32)
        scr, st = self.screen(), self.status_arr()
        hist = self.hist() if with_hist else None
        rc = fn(scr.ctypes.data, self.h, self.w, self.symbols, st.ctypes.data, flag,
                hist.ctypes.data if with_hist else None, out.ctypes.data)
        if rc:
            raise RuntimeError("Error in rogue-gym: Invalid tile")
        return out

    def gray_image(self, flag=0, with_hist=False):
        return self._image(self._L.orc_gray_image, 1, flag, with_hist)

    def symbol_image(self, flag=0, with_hist=False):
        return self._image(self._L.orc_symbol_image, self.symbols, flag, with_hist)


def kat_edges(xr, yr, direction, inclusive=True):
    """passages::edges on RectRange::from_ranges(xr, yr); direction in ("Up", "Down", "Left", "Right")."""
    xs, ys = (C.c_int * 256)(), (C.c_int * 256)()
    L = lib()
    L.orc_kat_edges.argtypes = [C.c_int] * 6 + [C.c_void_p] * 2
    n = L.orc_kat_edges(xr[0], yr[0], xr[1], yr[1], ["Up", "Down", "Left", "Right"].index(direction), int(inclusive), xs, ys)
    return [[xs[i], ys[i]] for i in range(n)]


def kat_u32(seed, n):
    out = np.empty(n, np.uint32)
    lib().orc_kat_u32(seed & 0xFFFFFFFFFFFFFFFF, seed >> 64, n, out.ctypes.data)
    return out


class OracleBatch:
    """ThreadConductor semantics (python/src/thread_impls.rs) over a pthread pool."""

    def __init__(self, configs, max_steps=1000, n_threads=1, seeds=None):
        n = len(configs)
        arr = (OrcConfig * n)()
        for i, d in enumerate(configs):
            arr[i] = config_from_dict(d, seed=None if seeds is None else seeds[i])
        self._L = lib()
        self.n = n
        self.h, self.w = arr[0].height, arr[0].width
        self._b = self._L.orc_batch_new(arr, n, max_steps, n_threads)
        if not self._b:
            raise RuntimeError("oracle: invalid config")

    def __del__(self):
        if getattr(self, "_b", None):
            self._L.orc_batch_free(self._b)
            self._b = None

    def step(self, keys, obs=None):
        keys = np.ascontiguousarray(keys, np.uint8)
        rc = self._L.orc_batch_step(self._b, keys.ctypes.data, None if obs is None else obs.ctypes.data)
        if rc:
            raise RuntimeError("oracle batch step error")

    def env(self, i):
        e = OracleEnv.__new__(OracleEnv)
        e._L = self._L
        e._e = None  # borrowed: do not free
        e._borrowed = self._L.orc_batch_env(self._b, i)
        e.h, e.w = self.h, self.w
        # route calls through the borrowed pointer
        object.__setattr__(e, "_e_view", e._borrowed)
        return _Borrowed(e, e._borrowed)


class _Borrowed(OracleEnv):
    def __init__(self, proto, ptr):
        self._L = proto._L
        self.h, self.w = proto.h, proto.w
        self._e = ptr

    def __del__(self):
        self._e = None
