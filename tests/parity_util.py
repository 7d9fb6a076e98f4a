"""Helpers shared by the GPU parity tests: drive the HIP stepper through the C-ABI and compare it,
field by field, with the CPU oracle on identical seeds and keys."""
import ctypes as C
import json

import numpy as np

from oracle.pyoracle import OracleEnv
from rogue_gym_python import _rogue_gym as inner

ACTION_KEYS = np.frombuffer(b".hjklnbuy>s", np.uint8)
ALL_KEYS = np.frombuffer(b".hjklnbuy>shjklnbuyHJKLYUBN", np.uint8)  # move-heavy mix incl. run keys


class HipBatch:
    def __init__(self, cfg, seeds, max_steps=1000, auto_reset=True):
        cfgs = []
        for s in seeds:
            d = dict(cfg)
            d["seed"] = int(s)
            cfgs.append(json.dumps(d))
        self.h = inner._Handle(cfgs, max_steps, auto_reset=auto_reset)
        self.n = self.h.n

    def step(self, keys):
        keys = np.ascontiguousarray(keys, np.uint8)
        self.h.check(self.h.L.rg_step(self.h.h, keys.ctypes.data, 0))

    def sync(self):
        self.h.check(self.h.L.rg_sync(self.h.h))

    def fetch(self):
        return self.h.fetch()

    def debug(self, i):
        return self.h.debug_state(i)

    def obs(self, kind, flag, with_hist):
        import torch

        c = self.h.L.rg_obs_channels(self.h.h, kind, flag, int(with_hist))
        out = torch.empty((self.n, c, self.h.height, self.h.width), dtype=torch.float32, device="cuda:%d" % self.h.device)
        fn = self.h.L.rg_obs_symbol if kind else self.h.L.rg_obs_gray
        self.h.check(fn(self.h.h, flag, int(with_hist), C.c_void_p(out.data_ptr())))
        torch.cuda.synchronize()
        return out.cpu().numpy()


def make_oracles(cfg, seeds, max_steps=1000):
    return [OracleEnv(cfg, max_steps=max_steps, seed=int(s)) for s in seeds]


def compare_mirrors(hip, oracles, where=""):
    screen, hist, status, flags = hip.fetch()
    for i, o in enumerate(oracles):
        osc = o.screen()
        if not np.array_equal(screen[i], osc):
            a = "\n".join(bytes(r).decode() for r in screen[i])
            b = "\n".join(bytes(r).decode() for r in osc)
            raise AssertionError("%s env %d screen differs\nHIP:\n%s\nORACLE:\n%s" % (where, i, a, b))
        assert np.array_equal(hist[i], o.hist()), "%s env %d hist differs" % (where, i)
        assert [int(v) & 0xFFFFFFFF for v in status[i]] == [int(v) for v in o.status_arr()], "%s env %d status %s vs %s" % (where, i, status[i], o.status_arr())
        f = o.flags()
        assert bool(flags[i] & 1) == f["is_terminal"], "%s env %d terminal" % (where, i)
        assert ((int(flags[i]) >> 8) & 0x7F) == f["message"], "%s env %d message %x vs %x" % (where, i, (int(flags[i]) >> 8) & 0x7F, f["message"])
        assert bool(flags[i] & 2) == f["dead"], "%s env %d dead flag" % (where, i)


def compare_internal(hip, oracles, envs, where=""):
    for i in envs:
        o = oracles[i]
        d, cells = hip.debug(i)
        sc = o.scalars()
        got = dict(px=d.px, py=d.py, level=d.dungeon_level, hp=d.hp, hp_max=d.hp_max, exp=d.exp, plevel=d.player_level, food_left=d.food_left,
                   quiet=d.quiet, gold=d.pack_gold, n_monsters=d.n_monsters)
        sc = {k: sc[k] for k in got}  # (the oracle also reports its pack: n_pack / weapon_slot / armor_slot have no device-side counterpart)
        assert got == sc, "%s env %d scalars %s vs %s" % (where, i, got, sc)
        assert d.steps == o.flags()["steps"], "%s env %d steps" % (where, i)
        rs, _ = o.rng()
        assert list(d.rng) == [int(v) for v in rs.reshape(-1)], "%s env %d rng state" % (where, i)
        surf, attr, doors, gold = o.grid()
        assert np.array_equal(cells & 7, surf), "%s env %d surface" % (where, i)
        assert np.array_equal((cells >> 4) & 0x3F, attr), "%s env %d attr" % (where, i)
        assert np.array_equal((cells >> 3) & 1, doors), "%s env %d doors" % (where, i)
        assert np.array_equal(((cells >> 11) & 1).astype(bool), gold >= 0), "%s env %d gold bits" % (where, i)
        gl = sorted((d.gold_x[k], d.gold_y[k], d.gold_amount[k]) for k in range(d.n_gold))
        ys, xs = np.nonzero(gold >= 0)
        assert gl == sorted((int(x), int(y), int(gold[y, x])) for y, x in zip(ys, xs)), "%s env %d gold table" % (where, i)
        mons = o.monsters()
        got_m = [(d.mon_x[k], d.mon_y[k], d.mon_type[k], d.mon_active[k], d.mon_hp[k], d.mon_exp[k]) for k in range(d.n_monsters)]
        exp_m = [(m["x"], m["y"], m["type"], m["active"], m["hp"], m["exp"]) for m in mons]
        assert got_m == exp_m, "%s env %d monsters %s vs %s" % (where, i, got_m, exp_m)


def lockstep(cfg, seeds, keys_per_step, max_steps=1000, check_every=1, internal_every=8, auto_reset=True):
    """Run both engines on the same keys; compare mirrors every `check_every` steps."""
    hip = HipBatch(cfg, seeds, max_steps=max_steps, auto_reset=auto_reset)
    oracles = make_oracles(cfg, seeds, max_steps=max_steps)
    compare_mirrors(hip, oracles, "t=0")
    compare_internal(hip, oracles, range(len(seeds)), "t=0")
    for t, keys in enumerate(keys_per_step):
        hip.step(keys)
        for i, o in enumerate(oracles):
            if auto_reset:
                o.step_autoreset(int(keys[i]))
            else:
                o.react(int(keys[i]))
        if (t + 1) % check_every == 0:
            compare_mirrors(hip, oracles, "t=%d" % (t + 1))
        if (t + 1) % internal_every == 0:
            compare_internal(hip, oracles, range(len(seeds)), "t=%d" % (t + 1))
    hip.sync()
    compare_mirrors(hip, oracles, "end")
    compare_internal(hip, oracles, range(len(seeds)), "end")
    return hip, oracles


def custom_enemy_config(base):
    """A config mixing builtin presets with Preset::Custom(Status) objects (character/enemies.rs:87-121)."""
    cfg = dict(base)
    cfg["enemies"] = {"enemies": [
        1,   # bat (random mover)
        18,  # snake
        {"attack": [{"times": 1, "max": 3}, {"times": 1, "max": 3}], "attr": 513, "defense": 6, "exp": 4, "gold": 0, "level": 2,
         "name": "gremlin", "tile": 71, "rarelity": 0},           # mean + random, two dice, glyph 'G'
        {"attack": [], "attr": 0, "defense": 9, "exp": 3, "gold": 5, "level": 1, "name": "ooze", "tile": 79, "rarelity": 1},   # harmless sleeper 'O'
        {"attack": [{"times": 2, "max": 4}], "attr": 1, "defense": 2, "exp": 30, "gold": 0, "level": 3, "name": "warg", "tile": 87, "rarelity": 3},
        10,  # kestrel
    ], "appear_rate_gold": 90, "appear_rate_nogold": 60}
    return cfg
