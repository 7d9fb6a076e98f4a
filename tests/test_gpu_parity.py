"""GPU parity tests proper: the HIP stepper (through the C-ABI) against the CPU oracle and the
reference's golden vectors.  Bit-exact for all integer state; float-exact for observations
(tolerance 0: the encode is one correctly-rounded f32 division per cell)."""
import ctypes
import json

import numpy as np
import pytest

from parity_util import ACTION_KEYS, ALL_KEYS, HipBatch, compare_internal, compare_mirrors, lockstep, make_oracles

pytestmark = pytest.mark.gpu


def rand_keys(rng, table, n, steps):
    return [table[rng.randint(0, len(table), n)] for _ in range(steps)]


def test_first_screens_10000_seeds_mini(goldens):
    """BASELINE config 5's parity sweep, on the mini dungeon: first screen + internals of 10 000 seeds."""
    cfg = goldens["configs"]["mini"]
    seeds = list(range(10000))
    hip = HipBatch(cfg, seeds)
    oracles = make_oracles(cfg, seeds)
    compare_mirrors(hip, oracles, "build")
    compare_internal(hip, oracles, range(0, 10000, 7), "build")


def test_first_screens_default_and_nohide(goldens):
    for name in ("default", "nohide", "ff", "st"):
        cfg = goldens["configs"][name]
        seeds = list(range(1500))
        hip = HipBatch(cfg, seeds)
        oracles = make_oracles(cfg, seeds)
        compare_mirrors(hip, oracles, name)
        compare_internal(hip, oracles, range(0, 1500, 3), name)


def test_seed_width_u128(goldens):
    cfg = goldens["configs"]["mini"]
    seeds = [2**32 + 5, 2**64 + 1, 2**100 + 12345, 2**127 + 99, 0]
    hip = HipBatch(cfg, seeds)
    oracles = make_oracles(cfg, seeds)
    compare_mirrors(hip, oracles, "u128")
    compare_internal(hip, oracles, range(len(seeds)), "u128")


@pytest.mark.parametrize("name,n,steps", [("mini", 512, 400), ("default", 256, 300)])
def test_lockstep_random_policy(goldens, name, n, steps):
    """Uniform-random policy over the 11-action table with auto-reset (the benchmark workload)."""
    rng = np.random.RandomState(7)
    lockstep(goldens["configs"][name], list(range(n)), rand_keys(rng, ACTION_KEYS, n, steps), max_steps=150, check_every=1, internal_every=25)


@pytest.mark.parametrize("name,n,steps", [("mini", 384, 500), ("default", 192, 400), ("nohide", 128, 300)])
def test_lockstep_run_keys(goldens, name, n, steps):
    """Move-heavy key mix incl. run keys (MoveUntil): deaths, fights, level-ups, descents, auto-resets."""
    rng = np.random.RandomState(11)
    lockstep(goldens["configs"][name], [1000 + i for i in range(n)], rand_keys(rng, ALL_KEYS, n, steps), max_steps=400, check_every=1, internal_every=50)


def test_lockstep_long_episode_hunger(goldens):
    """No enemies, long horizon: hunger thresholds (130/260), food wrap below zero, many descents."""
    rng = np.random.RandomState(3)
    keys = rand_keys(rng, np.frombuffer(b"hjklyubnHJKLYUBN>>s", np.uint8), 64, 1500)
    lockstep(goldens["configs"]["st"], list(range(64)), keys, max_steps=100000, check_every=10, internal_every=100)


def test_single_env_semantics_no_autoreset(goldens):
    """GameState semantics: no auto-reset; a dead env rejects action keys (IgnoredInput) without changing."""
    cfg = goldens["configs"]["mini"]
    n = 256
    rng = np.random.RandomState(5)
    hip = HipBatch(cfg, list(range(n)), max_steps=100000, auto_reset=False)
    oracles = make_oracles(cfg, list(range(n)), max_steps=100000)
    dead_seen = 0
    for t in range(600):
        keys = ALL_KEYS[rng.randint(0, len(ALL_KEYS), n)]
        hip.step(keys)
        for i, o in enumerate(oracles):
            if o.flags()["dead"]:
                dead_seen += 1
                with pytest.raises(RuntimeError):
                    o.react(int(keys[i]))
            else:
                o.react(int(keys[i]))
        if t % 20 == 19:
            compare_mirrors(hip, oracles, "t=%d" % t)
    assert dead_seen > 0
    with pytest.raises(RuntimeError):
        hip.sync()  # error flag raised by the dead envs
    compare_mirrors(hip, oracles, "end")
    compare_internal(hip, oracles, range(n), "end")


def test_invalid_key_is_an_error(goldens):
    hip = HipBatch(goldens["configs"]["mini"], [1, 2, 3])
    before = hip.fetch()
    hip.step(np.frombuffer(b"hZh", np.uint8))
    with pytest.raises(RuntimeError):
        hip.sync()
    after = hip.fetch()
    assert np.array_equal(before[0][1], after[0][1]) and np.array_equal(before[2][1], after[2][1])


@pytest.mark.parametrize("name", ["mini", "default", "st"])
def test_observation_encoders(goldens, name):
    """gray / symbol images with every status-flag combination used by the reference + history planes."""
    cfg = goldens["configs"][name]
    n = 96
    rng = np.random.RandomState(2)
    hip, oracles = lockstep(cfg, list(range(n)), rand_keys(rng, ALL_KEYS, n, 60), max_steps=1000, check_every=60, internal_every=60)
    for flag, with_hist in [(0, False), (0x1FF, False), (0b010000011, True), (1, False), (0, True)]:
        g = hip.obs(0, flag, with_hist)
        for i, o in enumerate(oracles):
            assert np.array_equal(g[i], o.gray_image(flag, with_hist)), "gray env %d flag %x" % (i, flag)
        # symbol image errors on 'Z' in the reference; compare only envs whose oracle succeeds
        s = hip.obs(1, flag, with_hist)
        for i, o in enumerate(oracles):
            try:
                exp = o.symbol_image(flag, with_hist)
            except RuntimeError:
                continue
            assert np.array_equal(s[i], exp), "symbol env %d flag %x" % (i, flag)
    hip.h.L.rg_sync(hip.h.h)  # drain a possible tile-error flag


def test_full_size_invariants(goldens):
    """BASELINE config 2 size (65 536 envs): properties that do not need the oracle at full size, plus
    oracle parity on a 512-env stride sample after 64 random steps."""
    cfg = goldens["configs"]["mini"]
    n = 65536
    rng = np.random.RandomState(9)
    hip = HipBatch(cfg, list(range(n)), max_steps=1000)
    keys = rand_keys(rng, ACTION_KEYS, n, 64)
    for k in keys:
        hip.step(k)
    hip.sync()
    screen, hist, status, flags = hip.fetch()
    assert (screen[:, 0, :] == 32).all() and (screen[:, -1, :] == 32).all()      # rows 0 and H-1 are never drawn
    n_at = (screen == ord("@")).sum(axis=(1, 2))
    assert (n_at <= 1).all() and (n_at == 1).mean() > 0.99   # one player glyph (none only when standing on a HIDDEN maze cell)
    assert (status[:, 2] >= 0).all() and (status[:, 2] <= status[:, 3]).all()      # 0 <= hp <= hp_max
    assert (status[:, 0] >= 1).all() and (hist.max() <= 1)
    sample = list(range(0, n, 128))
    oracles = make_oracles(cfg, sample)
    for k in keys:
        for j, o in enumerate(oracles):
            o.step_autoreset(int(k[sample[j]]))
    for j, o in enumerate(oracles):
        i = sample[j]
        assert np.array_equal(screen[i], o.screen()), "env %d" % i
        assert [int(v) for v in status[i]] == [int(v) for v in o.status_arr()]
    g = hip.obs(0, 0, False)
    assert g.shape == (n, 1, 16, 32)
    sym = g * 43.0
    assert np.abs(sym - np.round(sym)).max() < 1e-4


def test_inline_generation_path_without_spares(goldens, monkeypatch):
    """The spare-level pipeline (k_regen) is an optimisation: with it disabled every reset generates inline.
    Both paths must give identical states."""
    monkeypatch.setenv("ROGUE_GYM_HIP_NO_SPARES", "1")
    rng = np.random.RandomState(21)
    lockstep(goldens["configs"]["mini"], list(range(256)), rand_keys(rng, ALL_KEYS, 256, 200), max_steps=60, check_every=1, internal_every=40)


def test_spare_hand_off_at_full_batch(goldens):
    """The hand-off of regenerated spare levels (k_regen on the side stream writes them through to memory behind every k_step launch, the
    resets of later launches take them) under load: 65 536 envs with 30-step episodes -- about 2 200 resets per step, every one served
    from a spare written a few launches earlier -- against a handle that generates every reset inline.  Any stale or half-written spare
    shows up as a differing env."""
    import os

    from rogue_gym_python import _rogue_gym as inner

    n, steps = 65536, 240
    cfgs = [json.dumps(dict(goldens["configs"]["mini"], seed=i % 5000)) for i in range(n)]
    a = inner._Handle(cfgs, 30, auto_reset=True)
    os.environ["ROGUE_GYM_HIP_NO_SPARES"] = "1"
    try:
        b = inner._Handle(cfgs, 30, auto_reset=True)
    finally:
        del os.environ["ROGUE_GYM_HIP_NO_SPARES"]
    rng = np.random.RandomState(3)
    table = np.frombuffer(b"hjklyubn>s.", np.uint8)
    for t in range(steps):
        keys = np.ascontiguousarray(table[rng.randint(0, len(table), n)])
        for h in (a, b):
            h.check(h.L.rg_step(h.h, keys.ctypes.data, 0))
        if t % 4 == 3:
            for x, y, what in zip(a.fetch(), b.fetch(), ("screen", "hist", "status", "flags")):
                assert np.array_equal(x, y), (t, what, [i for i in range(n) if not np.array_equal(x[i], y[i])][:8])
    cnt = (ctypes.c_uint64 * 8)()
    a.check(a.L.rg_counters(a.h, cnt, 0))
    assert cnt[4] > 0.9 * cnt[0] > 400000, list(cnt)   # the resets did take spares
    a.close()
    b.close()


@pytest.mark.parametrize("name,n", [("mini", 65536), ("default", 16384)], ids=["mini", "default 80x24 (nine rooms)"])
def test_mirror_kept_by_the_turn_equals_the_mirror_drawn_from_the_tiles(goldens, name, n):
    """An ordinary Redraw (one turn, no whole-room reveal, no new level, history plane in step) is applied to the screen and history mirrors by k_step itself
    -- the window cells it wrote back, the overlays where they stood (S.ovl) and where they stand -- and raises no Redraw flag; everything else is drawn from
    the tiles by the observation / render pass.  Against a handle that draws EVERY Redraw from the tiles (ROGUE_GYM_HIP_NO_MIRROR_UPDATE): screen, history,
    status and the public flag bits of every env, every second step, run keys and 50-step episodes (resets, descents, rooms entered and left) included."""
    import os

    from rogue_gym_python import _rogue_gym as inner

    steps = 160
    cfgs = [json.dumps(dict(goldens["configs"][name], seed=i % 9000)) for i in range(n)]
    a = inner._Handle(cfgs, 50, auto_reset=True)
    os.environ["ROGUE_GYM_HIP_NO_MIRROR_UPDATE"] = "1"
    try:
        b = inner._Handle(cfgs, 50, auto_reset=True)
    finally:
        del os.environ["ROGUE_GYM_HIP_NO_MIRROR_UPDATE"]
    rng = np.random.RandomState(31)
    table = np.frombuffer(b"hjklyubnhjklyubnhjklyubnHJKL>s.", np.uint8)
    public = 0x00FF7F03  # terminal, dead, message bits, error bits (the mirror bookkeeping bits differ by construction)
    for t in range(steps):
        keys = np.ascontiguousarray(table[rng.randint(0, len(table), n)])
        for h in (a, b):
            h.check(h.L.rg_step(h.h, keys.ctypes.data, 0))
        if t % 2 == 1:
            xa, xb = a.fetch(), b.fetch()
            for x, y, what in zip(xa[:3], xb[:3], ("screen", "hist", "status")):
                assert np.array_equal(x, y), (t, what, [i for i in range(n) if not np.array_equal(x[i], y[i])][:8])
            assert np.array_equal(xa[3] & public, xb[3] & public), (t, "flags")
    a.close()
    b.close()


def test_urgent_spares_of_envs_that_die_fast(goldens):
    """The bulk of the consumed spares is rebuilt every 16th step (one level per lane, rg_regen_lanes.hip); an env that is down to its last ready spare gets
    one built beside the very next step by the wave-per-level producer (k_regen, spares == 2).  Episodes of 3 steps consume a spare every 3 steps -- five
    between two bulk launches, more than the four slots -- so the resets here depend on the urgent path, with both producers claiming slots of
    the same envs.  Same bits as a handle that generates every reset inline, and the resets did take spares."""
    import os

    from rogue_gym_python import _rogue_gym as inner

    n, steps = 4096, 330
    cfgs = [json.dumps(dict(goldens["configs"]["mini"], seed=i % 700)) for i in range(n)]
    a = inner._Handle(cfgs, 3, auto_reset=True)
    os.environ["ROGUE_GYM_HIP_NO_SPARES"] = "1"
    try:
        b = inner._Handle(cfgs, 3, auto_reset=True)
    finally:
        del os.environ["ROGUE_GYM_HIP_NO_SPARES"]
    rng = np.random.RandomState(11)
    table = np.frombuffer(b"hjklyubn>s.", np.uint8)
    for t in range(steps):
        keys = np.ascontiguousarray(table[rng.randint(0, len(table), n)])
        for h in (a, b):
            h.check(h.L.rg_step(h.h, keys.ctypes.data, 0))
        if t % 3 == 2:
            for x, y, what in zip(a.fetch(), b.fetch(), ("screen", "hist", "status", "flags")):
                assert np.array_equal(x, y), (t, what, [i for i in range(n) if not np.array_equal(x[i], y[i])][:8])
    cnt = (ctypes.c_uint64 * 8)()
    a.check(a.L.rg_counters(a.h, cnt, 0))
    assert cnt[0] > 400000 and cnt[4] > 0.4 * cnt[0], list(cnt)   # resets; the bulk launch (4 x 4096 spares per 16 steps) + one urgent build per wave and step serve most of them, the rest generate inline
    a.close()
    b.close()


@pytest.mark.parametrize("name,n,steps", [("mini", 16384, 500), ("default", 4096, 700), ((160, 48, 8, 5), 2048, 300), ((50, 21, 3, 2), 4096, 400)],
                         ids=["mini", "default", "160x48, 40 rooms (the 64-room generator instance)", "50x21 (cells not a multiple of 8)"])
def test_next_level_structures_match_inline_generation(goldens, name, n, steps):
    """A descent whose next-level structure is ready (asked when a staircase came into the player's window, generated by k_regen from the env's dungeon
    and item streams, valid iff the dungeon stream has not moved since) loads it and only runs the monster half of the generator; every other descent
    generates the whole level inside the turn.  Same bits either way: the product against a handle with ROGUE_GYM_HIP_NO_NEXT_LEVELS, every env compared
    -- and the structures were in fact used for most descents."""
    import os

    from rogue_gym_python import _rogue_gym as inner

    if isinstance(name, tuple):
        base = dict(goldens["configs"]["default"], width=name[0], height=name[1])
        base["dungeon"] = dict(base.get("dungeon", {}), style="rogue", room_num_x=name[2], room_num_y=name[3])
    else:
        base = goldens["configs"][name]
    cfgs = [json.dumps(dict(base, seed=i)) for i in range(n)]
    a = inner._Handle(cfgs, 400, auto_reset=True)
    os.environ["ROGUE_GYM_HIP_NO_NEXT_LEVELS"] = "1"
    try:
        b = inner._Handle(cfgs, 400, auto_reset=True)
    finally:
        del os.environ["ROGUE_GYM_HIP_NO_NEXT_LEVELS"]
    rng = np.random.RandomState(13)
    table = np.frombuffer(b"hjklyubnhjklyubn>>s.HJKL", np.uint8)
    for t in range(steps):
        keys = np.ascontiguousarray(table[rng.randint(0, len(table), n)])
        for h in (a, b):
            h.check(h.L.rg_step(h.h, keys.ctypes.data, 0))
        if t % 5 == 4 or t > steps - 4:
            for x, y, what in zip(a.fetch(), b.fetch(), ("screen", "hist", "status", "flags")):
                assert np.array_equal(x, y), (t, what, [i for i in range(n) if not np.array_equal(x[i], y[i])][:8])
    ca, cb = (ctypes.c_uint64 * 9)(), (ctypes.c_uint64 * 9)()
    a.check(a.L.rg_counters_ex(a.h, ca, 9, 0))
    b.check(b.L.rg_counters_ex(b.h, cb, 9, 0))
    assert list(ca)[:3] == list(cb)[:3] and ca[1] > (300 if not isinstance(name, tuple) else 30), (list(ca), list(cb))  # (resets, descents, dist maps: the same game; [3] / [4] depend on when a spare was ready)
    assert cb[8] == 0 and ca[8] > 0.5 * ca[1], (list(ca), list(cb))  # [8]: descents that loaded their structure
    a.close()
    b.close()


def test_frequent_descents_with_monsters(goldens):
    """A policy that presses '>' often: descents with monsters around, and descents in consecutive steps (the player can be placed on the
    stairs) -- after the second one the reference's history plane shows the level that was just discarded, not the one before it."""
    cfg = dict(goldens["configs"]["mini"], hide_dungeon=False)
    n = 256
    rng = np.random.RandomState(77)
    pool = np.frombuffer(b"hjklyubn>>>>s.", np.uint8)
    keys = [pool[rng.randint(0, len(pool), n)] for _ in range(400)]
    hip, oracles = lockstep(cfg, list(range(n)), keys, max_steps=400, check_every=2, internal_every=100)
    assert max(int(o.status_arr()[0]) for o in oracles) >= 2


@pytest.mark.parametrize("period", [1, 9])
def test_descents_many_and_few_lanes_per_wave(goldens, period):
    """The DDQN key log (data/learned/ddqn-minidungeon) takes seed 5 down a level.  Every `period`-th env plays seed 5, the others play
    other seeds with the same keys: period 1 = every lane of a wave descends on the same step (generation rounds over the LDS slots),
    period 9 = a handful per wave."""
    cfg = goldens["configs"]["ddqn"]
    n = 192
    seeds = [5 if i % period == 0 else 1000 + i for i in range(n)]
    keys = [np.full(n, ord(ch), np.uint8) for ch in goldens["ddqn_keys"][:400]]
    hip, oracles = lockstep(cfg, seeds, keys, max_steps=1000, check_every=4, internal_every=50)
    levels = [int(o.status_arr()[0]) for o in oracles]
    assert max(levels) >= 2, "the log must descend at least once within 400 keys"


GEOMETRIES = [
    # width, height, room_num_x, room_num_y  -- which code paths the shape selects
    (40, 20, 2, 2),    # 32 < W <= 64: one 64-bit mask per row; H = 20: 32-row BFS groups (2 maps per round)
    (50, 21, 3, 2),    # H*W = 1050 is not a multiple of 8: unfused k_render + scalar encode, scalar grid copies
    (96, 32, 4, 3),    # 64 < W <= 128: two mask words per row; 12 rooms
    (160, 48, 4, 4),   # the reference's maximum screen (core/src/lib.rs:134-140): three mask words, 64-row BFS, 16 rooms
    (64, 16, 4, 1),    # a single row of rooms
    (32, 48, 1, 3),    # narrowest x tallest screen: a single column of rooms
    (160, 48, 6, 5),   # 30 rooms (the stepper's limit is 32; the reference has none)
]


@pytest.mark.parametrize("w,h,rx,ry", GEOMETRIES)
def test_lockstep_other_geometries(goldens, w, h, rx, ry):
    """Screen sizes and room grids other than the two benchmark configs: first screens, lock-step play with every key,
    internal state and both observation encoders."""
    cfg = {"width": w, "height": h, "dungeon": {"style": "rogue", "room_num_x": rx, "room_num_y": ry, "min_room_size": {"x": 4, "y": 4}}}
    n = 96
    rng = np.random.RandomState(w * 1000 + h)
    hip, oracles = lockstep(cfg, list(range(500, 500 + n)), rand_keys(rng, ALL_KEYS, n, 160), max_steps=90, check_every=1, internal_every=40)
    for kind in (0, 1):
        for flag, with_hist in ((0, False), (0x1ff, True)):
            got = hip.obs(kind, flag, with_hist)
            for i in (0, n // 2, n - 1):
                o = oracles[i]
                try:
                    exp = o.symbol_image(flag, with_hist) if kind else o.gray_image(flag, with_hist)
                except RuntimeError:  # symbol image with a 'Z' on screen errors in the reference
                    continue
                assert np.array_equal(got[i], exp), (kind, flag, with_hist, i)
    hip.h.L.rg_sync(hip.h.h)  # drain a possible tile-error flag


def _stair_seeker_keys(oracles, rng, stuck):
    """One key per env from the oracle's own state: '>' on the stairs, else a greedy step down the BFS distance to the stairs (8 directions,
    hidden / locked cells impassable), else search or wander.  Test policy only -- it exists to reach deep levels quickly."""
    dirs = {(0, -1): "k", (0, 1): "j", (-1, 0): "h", (1, 0): "l", (-1, -1): "y", (1, -1): "u", (-1, 1): "b", (1, 1): "n"}
    out = []
    for i, o in enumerate(oracles):
        surf, attr, _, _ = o.grid()
        sc = o.scalars()
        px, py = sc["px"], sc["py"]
        h, w = surf.shape
        if surf[py, px] == 4:
            out.append(ord(">")); continue
        walk = ~np.isin(surf, (2, 3, 7)) & ((attr & 0x12) == 0)   # not a wall / nothing, not hidden (0x02) or locked (0x10)
        ys, xs = np.nonzero(surf == 4)
        key = None
        if len(xs) and stuck[i] < 6:
            dist = np.full((h, w), 1 << 20, np.int32)
            dist[ys[0], xs[0]] = 0
            front = [(xs[0], ys[0])]
            while front:
                nxt = []
                for (x, y) in front:
                    for (dx, dy) in dirs:
                        nx, ny = x + dx, y + dy
                        if 0 <= nx < w and 0 <= ny < h and walk[ny, nx] and dist[ny, nx] > dist[y, x] + 1:
                            if dx and dy and not (walk[y, nx] and walk[ny, x]):
                                continue
                            dist[ny, nx] = dist[y, x] + 1
                            nxt.append((nx, ny))
                front = nxt
            best = dist[py, px]
            for (dx, dy), k in dirs.items():
                nx, ny = px + dx, py + dy
                if 0 <= nx < w and 0 <= ny < h and walk[ny, nx] and dist[ny, nx] < best and (not (dx and dy) or (walk[py, nx] and walk[ny, px])):
                    best, key = dist[ny, nx], k
        if key is None:
            stuck[i] = (stuck[i] + 1) % 24
            key = "s" if stuck[i] % 3 == 0 else "hjklyubn"[rng.randint(0, 8)]
        else:
            stuck[i] = 0
        out.append(ord(key))
    return np.array(out, np.uint8)


def test_deep_levels_without_enemies(goldens):
    """A stairs-seeking policy without monsters goes dozens of levels down, where every room is dark and mazes and hidden passages / locked
    doors are common (dark_level 10, maze_rate_inv 15, amulet_level 25) -- the part of the generator and of the field-of-view code the
    level-1 goldens never reach.  Every step compares the mirrors, every 100th the whole internal state."""
    cfg = dict(goldens["configs"]["mini"], enemies={"enemies": []})
    n = 48
    seeds = list(range(900, 900 + n))
    hip = HipBatch(cfg, seeds, max_steps=100000)
    oracles = make_oracles(cfg, seeds, max_steps=100000)
    rng = np.random.RandomState(5)
    stuck = [0] * n
    for t in range(900):
        keys = _stair_seeker_keys(oracles, rng, stuck)
        hip.step(keys)
        for i, o in enumerate(oracles):
            o.step_autoreset(int(keys[i]))
        compare_mirrors(hip, oracles, "t=%d" % t)
        if t % 100 == 99:
            compare_internal(hip, oracles, range(n), "t=%d" % t)
    levels = sorted(int(o.status_arr()[0]) for o in oracles)
    assert levels[n // 2] >= 8 and levels[-1] >= 20, levels


def test_stair_seekers_with_enemies(goldens):
    """The same policy with the 26 builtin monsters: players get a few levels down before they die (auto-reset), so monster tables, combat
    and level-ups run at dungeon levels the random policy never sees."""
    cfg = goldens["configs"]["mini"]
    n = 64
    seeds = list(range(1300, 1300 + n))
    hip = HipBatch(cfg, seeds, max_steps=400)
    oracles = make_oracles(cfg, seeds, max_steps=400)
    rng = np.random.RandomState(6)
    stuck = [0] * n
    deepest = 1
    for t in range(700):
        keys = _stair_seeker_keys(oracles, rng, stuck)
        hip.step(keys)
        for i, o in enumerate(oracles):
            o.step_autoreset(int(keys[i]))
        compare_mirrors(hip, oracles, "t=%d" % t)
        if t % 100 == 99:
            compare_internal(hip, oracles, range(n), "t=%d" % t)
        deepest = max(deepest, max(int(o.status_arr()[0]) for o in oracles))
    assert deepest >= 5, deepest


def test_stair_seekers_default_dungeon(goldens):
    """Stairs-seeking policy on the default 80x24 / 3x3-room dungeon with monsters: deeper levels, mazes and dark rooms on the wide-grid
    code paths (two-word row masks in the BFS, 256-thread observation blocks)."""
    cfg = dict(goldens["configs"]["default"])
    n = 12
    seeds = list(range(40, 40 + n))
    hip = HipBatch(cfg, seeds, max_steps=600)
    oracles = make_oracles(cfg, seeds, max_steps=600)
    rng = np.random.RandomState(8)
    stuck = [0] * n
    deepest = 1
    for t in range(450):
        keys = _stair_seeker_keys(oracles, rng, stuck)
        hip.step(keys)
        for i, o in enumerate(oracles):
            o.step_autoreset(int(keys[i]))
        compare_mirrors(hip, oracles, "t=%d" % t)
        if t % 150 == 149:
            compare_internal(hip, oracles, range(n), "t=%d" % t)
        deepest = max(deepest, max(int(o.status_arr()[0]) for o in oracles))
    assert deepest >= 4, deepest


def test_spares_survive_reseeding(goldens):
    """rg_seed invalidates the pre-generated spares: after seed() + reset() and further auto-resets the envs follow the new seeds."""
    cfg = goldens["configs"]["mini"]
    n = 128
    hip = HipBatch(cfg, list(range(n)), max_steps=25)
    rng = np.random.RandomState(4)
    for k in rand_keys(rng, ACTION_KEYS, n, 40):
        hip.step(k)
    import ctypes as C
    new_seeds = [1000 + 7 * i for i in range(n)]
    lo = (C.c_uint64 * n)(*new_seeds)
    hi = (C.c_uint64 * n)(*([0] * n))
    hip.h.check(hip.h.L.rg_seed(hip.h.h, lo, hi, n))
    hip.h.check(hip.h.L.rg_reset(hip.h.h))
    oracles = make_oracles(cfg, new_seeds, max_steps=25)
    compare_mirrors(hip, oracles, "after reseed+reset")
    for t, k in enumerate(rand_keys(rng, ACTION_KEYS, n, 80)):
        hip.step(k)
        for i, o in enumerate(oracles):
            o.step_autoreset(int(k[i]))
        compare_mirrors(hip, oracles, "reseeded t=%d" % t)
    compare_internal(hip, oracles, range(n), "reseeded end")


def test_full_size_default_and_symbol_obs(goldens):
    """BASELINE configs 3 and 4 at their per-GPU sizes (32 768 envs of the 80x24 dungeon; nohide + 43-channel symbol
    image): run, check size-independent invariants, and compare a strided sample with the oracle."""
    import torch
    for name, sym in (("default", False), ("nohide", True)):
        cfg = goldens["configs"][name]
        n = 32768
        rng = np.random.RandomState(13)
        hip = HipBatch(cfg, list(range(n)), max_steps=1000)
        keys = rand_keys(rng, ACTION_KEYS, n, 24)
        for k in keys:
            hip.step(k)
        hip.sync()
        screen, hist, status, flags = hip.fetch()
        assert (screen[:, 0, :] == 32).all() and (screen[:, -1, :] == 32).all()
        assert ((screen == ord("@")).sum(axis=(1, 2)) <= 1).all()
        assert (status[:, 2] <= status[:, 3]).all() and (status[:, 0] >= 1).all()
        sample = list(range(0, n, 256))
        oracles = make_oracles(cfg, sample)
        for k in keys:
            for j, o in enumerate(oracles):
                o.step_autoreset(int(k[sample[j]]))
        for j, o in enumerate(oracles):
            i = sample[j]
            assert np.array_equal(screen[i], o.screen()), "%s env %d" % (name, i)
            assert [int(v) for v in status[i]] == [int(v) for v in o.status_arr()]
        obs = hip.obs(1 if sym else 0, 0x1FF if not sym else 0, False)
        assert obs.shape == (n, (43 if sym else 1 + 9), 24, 80)
        for j, o in enumerate(oracles[:32]):
            i = sample[j]
            if sym:
                try:
                    exp = o.symbol_image(0, False)
                except RuntimeError:
                    continue
            else:
                exp = o.gray_image(0x1FF, False)
            assert np.array_equal(obs[i], exp), "%s obs env %d" % (name, i)
        if sym:  # one-hot: every cell has exactly one active channel among 0..41 (unless it shows 'Z')
            assert (obs[:, :42].sum(axis=1) <= 1.0).all()
        hip.h.L.rg_sync(hip.h.h)
        del obs, hip
        torch.cuda.empty_cache()


def test_batch_sizes_where_the_wave_count_is_capped(goldens):
    """on a W > 32 config (one step wave per SIMD) rgk_step_epw caps the index-order blocks at 1008 -- and for n = 64 513..65 472 that gave 65
    envs per 64-lane wave: env b * 65 + 64 of every block was never stepped.  65 000 envs of the 80x24 dungeon: every env receives its keys (the launch's
    key counter), and the envs at the old blocks' 65th places play like the oracle."""
    cfg = goldens["configs"]["default"]
    n = 65000
    rng = np.random.RandomState(29)
    hip = HipBatch(cfg, list(range(n)), max_steps=1000)
    keys = rand_keys(rng, ACTION_KEYS, n, 12)
    cnt = (ctypes.c_uint64 * 8)()
    hip.h.check(hip.h.L.rg_counters(hip.h.h, cnt, 1))
    for k in keys:
        hip.step(k)
    hip.sync()
    hip.h.check(hip.h.L.rg_counters(hip.h.h, cnt, 0))
    assert cnt[6] == n * len(keys), (cnt[6], n * len(keys))   # [6] keys processed
    screen, hist, status, flags = hip.fetch()
    sample = [b * 65 + 64 for b in range(0, 1000, 25)] + [0, 63, 64, 65, n - 1]
    oracles = make_oracles(cfg, sample)
    for k in keys:
        for j, o in enumerate(oracles):
            o.step_autoreset(int(k[sample[j]]))
    for j, o in enumerate(oracles):
        i = sample[j]
        assert np.array_equal(screen[i], o.screen()), "env %d" % i
        assert [int(v) for v in status[i]] == [int(v) for v in o.status_arr()], "env %d" % i


def test_custom_enemy_presets(goldens):
    """Full GameConfig coverage: custom monster statuses mixed with builtin presets, custom appear rates."""
    from parity_util import custom_enemy_config
    cfg = custom_enemy_config(goldens["configs"]["mini"])
    rng = np.random.RandomState(17)
    hip, oracles = lockstep(cfg, list(range(384)), rand_keys(rng, ALL_KEYS, 384, 400), max_steps=300, check_every=1, internal_every=50)
    assert oracles[0].symbols == ord("W") - ord("A") + 18
    g = hip.obs(0, 0, False)
    for i, o in enumerate(oracles[:64]):
        assert np.array_equal(g[i], o.gray_image(0, False))
    seen = set()
    for o in oracles:
        seen.update(chr(65 + m["type"]) for m in o.monsters())
    assert {"G", "W"} <= seen  # the custom monsters actually spawn


def test_config5_10000_seeds_default_dungeon(goldens):
    """BASELINE config 5 at its own shape: 10 000 distinct seeds of the DEFAULT 80x24 dungeon, one env each -- first screen, history, status,
    internals incl. RNG words vs the oracle, plus the observation the config names: status_vec(FULL) + gray image for every env."""
    import torch
    from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, StatusFlag

    cfg = goldens["configs"]["default"]
    seeds = list(range(10000))
    hip = HipBatch(cfg, seeds)
    oracles = make_oracles(cfg, seeds)
    compare_mirrors(hip, oracles, "config5")
    compare_internal(hip, oracles, range(0, 10000, 11), "config5")
    gray = hip.obs(0, 0, False)
    assert gray.shape == (10000, 1, 24, 80)
    for i in range(0, 10000, 3):
        assert np.array_equal(gray[i], oracles[i].gray_image(0, False)), "gray env %d" % i
    del hip
    venv = HipVecRogueEnv([dict(cfg, seed=s) for s in seeds], image_setting=ImageSetting(DungeonType.GRAY, StatusFlag.FULL, False))
    sv = venv.status_vec(StatusFlag.FULL).cpu().numpy()
    obs = venv.obs.cpu().numpy()
    assert obs.shape == (10000, 10, 24, 80)
    for i in range(0, 10000, 7):
        assert sv[i].tolist() == oracles[i].status_vec(0x1FF)
        assert np.array_equal(obs[i], oracles[i].gray_image(0x1FF, False))
    # the compact records every rank would contribute to the one all-gather, expanded again: identical to the direct observation
    packed = venv.packed_records()
    out = venv.expand_records(packed)
    torch.cuda.synchronize()
    assert torch.equal(out, venv.obs)
    venv.close()


def test_full_size_descent_heavy(goldens):
    """65 536 envs, half of all keys are '>': every step several hundred envs descend in waves of their own (k_step's stair waves) while the
    index-order waves hold their lanes -- oracle parity on a 1 024-env stride sample every step guards the hand-over between the two kinds
    of wave (the first version let an index-order wave that started late re-play an env its stair wave had already reset)."""
    cfg = dict(goldens["configs"]["mini"], enemies={"enemies": [1, 10, 18]})
    n = 65536
    rng = np.random.RandomState(31)
    table = np.frombuffer(b">>>>>>>>hjklyubn", np.uint8)
    hip = HipBatch(cfg, list(range(n)), max_steps=400)
    sample = list(range(0, n, 64))
    oracles = make_oracles(cfg, sample, max_steps=400)
    for t in range(90):
        k = table[rng.randint(0, len(table), n)]
        hip.step(k)
        for j, o in enumerate(oracles):
            o.step_autoreset(int(k[sample[j]]))
        if t % 6 == 5:
            screen, hist, status, flags = hip.fetch()
            for j, o in enumerate(oracles):
                i = sample[j]
                assert np.array_equal(screen[i], o.screen()), "t=%d env %d" % (t, i)
                assert [int(v) & 0xFFFFFFFF for v in status[i]] == [int(v) for v in o.status_arr()], "t=%d env %d status" % (t, i)
    hip.sync()
    import ctypes as C
    cnt = (C.c_uint64 * 8)()
    hip.h.check(hip.h.L.rg_counters(hip.h.h, cnt, 0))
    assert cnt[1] > 3000, "only %d descents: the test does not exercise the stair waves" % cnt[1]


def test_config1_plumbing_shape(goldens):
    """BASELINE config 1 at its own shape: data/config-mini.json as it is (seed 4 in the file), 64 envs all on that seed, max_steps = 1000, actions =
    LCG-generated indices into the 11-action table (seeded 0) -- 1 000 lock-step keys through the drop-in ParallelRogueEnv, every env equal to the
    oracle (and to each other while they receive the same key)."""
    from oracle.pyoracle import OracleEnv
    from rogue_gym.envs import ParallelRogueEnv

    cfg = goldens["configs"]["mini"]
    assert cfg["seed"] == 4
    n = 64
    env = ParallelRogueEnv([cfg] * n, max_steps=1000)
    oracles = [OracleEnv(cfg, max_steps=1000) for _ in range(4)]
    lcg = 0
    for t in range(1000):
        acts = []
        for i in range(n):
            if i < 60:  # 60 envs share one action stream (they must stay identical), 4 get streams of their own
                a = None
            else:
                lcg = (lcg * 1103515245 + 12345) & 0x7FFFFFFF
                a = (lcg >> 16) % 11
            acts.append(a)
        lcg = (lcg * 1103515245 + 12345) & 0x7FFFFFFF
        shared = (lcg >> 16) % 11
        acts = [shared if a is None else a for a in acts]
        states, rewards, dones, _ = env.step(acts)
        keys = [ord(ParallelRogueEnv.ACTIONS[a]) for a in acts]
        oracles[0].step_autoreset(keys[0])
        for j in range(1, 4):
            oracles[j].step_autoreset(keys[60 + j])
        if t % 20 == 19 or dones[0]:
            assert all(np.array_equal(states.screen[i], states.screen[0]) for i in range(1, 60))
            assert states[0].dungeon == oracles[0].dungeon() and [int(v) for v in states.status[0].astype(np.uint32)] == [int(v) for v in oracles[0].status_arr()]
            for j in range(1, 4):
                assert states[60 + j].dungeon == oracles[j].dungeon(), "t=%d env %d" % (t, 60 + j)
    env.close()
