"""CPU-only checks of the drop-in boundary: librogue_gym_hip.so loads, exports every symbol that
include/rogue_gym_hip.h declares, parses configs, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from rogue_gym_python import _rogue_gym as inner
    return inner.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "rogue_gym_hip.h")).read()
    names = sorted(set(re.findall(r"\b(rg_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n


def test_loaded_library_is_built_from_the_checked_out_sources(lib):
    """The .so is git-ignored and ships prebuilt to the GPU box: its compiled-in build id must be the hash of the sources in the tree."""
    import __graft_entry__ as g
    assert lib.rg_build_id().decode() == g.source_id()
    assert g.library_id(os.path.join(ROOT, "rogue-gym_amd", "librogue_gym_hip.so")) == g.source_id()


def _create(lib, cfgs, n=None):
    n = len(cfgs) if n is None else n
    arr = (C.c_char_p * n)(*[c if c is None else c.encode() for c in cfgs])
    h = C.c_void_p()
    rc = lib.rg_create(arr, n, 1000, 0, 1, C.byref(h))
    return rc, lib.rg_last_error(None).decode(), h


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_config_errors_come_before_device_errors(lib):
    rc, msg, _ = _create(lib, ['{"width": 31}'])
    assert rc != 0 and "too narrow" in msg
    rc, msg, _ = _create(lib, ['{"height": 49}'])
    assert rc != 0 and "too wide" in msg
    rc, msg, _ = _create(lib, ['{"seed": "abc"}'])
    assert rc != 0 and msg.startswith("Failed to parse config")
    rc, msg, _ = _create(lib, ['{"seed": 1,}'])
    assert rc != 0 and msg.startswith("Failed to parse config")
    rc, msg, _ = _create(lib, ['{"dungeon": {"style": "nethack"}}'])
    assert rc != 0 and "nethack" in msg
    rc, msg, _ = _create(lib, ['{"seed": 1}', '{"seed": 2, "width": 40}'])  # per-env configs may differ in anything, the screen size included
    assert rc == 0 or "no HIP device" in msg
    rc, msg, _ = _create(lib, ['{"seed": 1}', '{"seed": 2, "enemies": {"enemies": []}}'])  # a legal mixed batch: only the missing GPU stops it here
    assert rc == 0 or "no HIP device" in msg


def test_no_cpu_fallback(lib):
    if _has_gpu():
        pytest.skip("GPU present")
    rc, msg, _ = _create(lib, [json.dumps({"seed": 1})])
    assert rc != 0 and "no HIP device" in msg
    from rogue_gym.envs import RogueEnv
    with pytest.raises(RuntimeError, match="no HIP device"):
        RogueEnv(config_dict={}, seed=1)


def test_product_does_not_import_oracle():
    """The product package must never reach into oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "rogue-gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower() or f == "rg_config.cpp" and False, "%s mentions the oracle" % os.path.join(dirpath, f)


def test_python_surface_imports():
    import rogue_gym
    from rogue_gym.envs import (DungeonType, FirstFloorEnv, HipVecRogueEnv, ImageSetting, ParallelRogueEnv, RogueEnv, StairRewardEnv,
                                StairRewardParallel, StatusFlag)
    assert RogueEnv.ACTIONS == [".", "h", "j", "k", "l", "n", "b", "u", "y", ">", "s"]
    assert StatusFlag.FULL.count_one() == 9
    assert ImageSetting(DungeonType.GRAY, StatusFlag.DUNGEON_LEVEL | StatusFlag.EXP, True).dim(43) == 4
    assert ImageSetting().dim(17) == 26
    assert len(RogueEnv.SYMBOLS) == 43
    _ = (rogue_gym, FirstFloorEnv, HipVecRogueEnv, ParallelRogueEnv, StairRewardEnv, StairRewardParallel)


def _canon(lib, cfg):
    buf = C.create_string_buffer(1 << 16)
    rc = lib.rg_config_canonical(None if cfg is None else json.dumps(cfg).encode(), buf, len(buf))
    if rc:
        raise RuntimeError(lib.rg_last_error(None).decode())
    return json.loads(buf.value.decode())


def test_config_canonical_roundtrip(lib, goldens):
    """GameState.dump_config must satisfy `json.loads(dump) == config_dict` for the reference's test configs
    (python/tests/test_ff_env.py:22): default-valued sections are skipped, `hide_dungeon` is always written."""
    assert _canon(lib, goldens["configs"]["ff"]) == goldens["configs"]["ff"]
    assert _canon(lib, {"seed": 1}) == {"seed": 1, "hide_dungeon": True}
    assert _canon(lib, None) == {"hide_dungeon": True}
    st = _canon(lib, goldens["configs"]["st"])
    assert st["width"] == 32 and st["height"] == 16 and st["seed"] == 5 and st["hide_dungeon"] is False
    assert st["dungeon"]["room_num_x"] == 2 and st["dungeon"]["style"] == "rogue" and st["enemies"] == {"enemies": []}
    big = 2**100 + 12345
    assert _canon(lib, {"seed": big})["seed"] == big                      # u128 seeds survive
    d = _canon(lib, goldens["configs"]["default"])                          # data/config-default.json: all defaults except `exps` (last entry 0)
    assert set(d) == {"player", "hide_dungeon"} and d["player"]["exps"][-1] == 0   # (tests/test_config_schema.py looks at the rest)
    m = _canon(lib, goldens["configs"]["mini"])
    assert m["dungeon"]["min_room_size"] == {"x": 4, "y": 4} and m["seed"] == 4
    assert _canon(lib, _canon(lib, goldens["configs"]["st"])) == st        # idempotent
    rng = _canon(lib, {"seed_range": [3, 2**70]})                           # seed_range survives the dump (u128 bounds), with and without a seed
    assert rng == {"seed_range": [3, 2**70], "hide_dungeon": True}
    assert _canon(lib, {"seed": 7, "seed_range": [0, 40]}) == {"seed": 7, "seed_range": [0, 40], "hide_dungeon": True}


def test_config_validation(lib):
    for bad, frag in [({"width": 160, "height": 48, "dungeon": {"style": "rogue", "room_num_x": 41, "room_num_y": 10}}, "room_num"),      # 410 rooms: beyond any geometry
                      ({"width": 160, "height": 48, "dungeon": {"style": "rogue", "room_num_x": 33, "room_num_y": 8}}, "min_room_size"),  # rooms of 4 x 6 < min 4 + 1
                      ({"dungeon": {"style": "rogue", "room_num_x": 5, "room_num_y": 5}}, "min_room_size"),
                      ({"enemies": {"enemies": [99]}}, "builtin"),
                      ({"enemies": {"enemies": [{"name": "x"}]}}, "missing field"),
                      ({"enemies": {"enemies": ["kestrel"]}}, "invalid enemy preset"),
                      ({"dungeon": {"style": "rogue", "dark_level": 0}}, "zero rate"),
                      ({"width": "wide"}, "width")]:
        with pytest.raises(RuntimeError) as ei:
            _canon(lib, bad)
        assert frag in str(ei.value), (bad, str(ei.value))


def test_custom_enemy_config_roundtrip(lib, goldens):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_util import custom_enemy_config
    cfg = custom_enemy_config(goldens["configs"]["mini"])
    c1 = _canon(lib, cfg)
    assert c1["enemies"] == cfg["enemies"]
    assert _canon(lib, c1) == c1
    bad = json.loads(json.dumps(cfg))
    bad["enemies"]["enemies"][2]["tile"] = 38  # '&' has no symbol id (core/src/symbol.rs:17-40)
    with pytest.raises(RuntimeError):
        _canon(lib, bad)
    del bad["enemies"]["enemies"][2]["tile"]
    with pytest.raises(RuntimeError, match="missing field"):
        _canon(lib, bad)


def test_default_build_reads_only_the_documented_knobs():
    """the generator placements that were measured and rejected are development-build knobs (-DRG_DEV_KNOBS); the product's launch
    path reads seven environment variables, each with a purpose and each exercised by a GPU test."""
    allowed = {"ROGUE_GYM_HIP_NO_SPARES", "ROGUE_GYM_HIP_NO_STAIR_WAVES", "ROGUE_GYM_HIP_KEEP_SPARES", "ROGUE_GYM_HIP_FULL_BFS", "ROGUE_GYM_HIP_EPW",
               "ROGUE_GYM_HIP_NO_NEXT_LEVELS", "ROGUE_GYM_HIP_WAVE_REGEN", "ROGUE_GYM_HIP_NO_MIRROR_UPDATE", "ROGUE_GYM_HIP_SP_SLOTS"}
    csrc = os.path.join(ROOT, "rogue-gym_amd", "csrc")
    seen, sites = set(), 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cpp", ".hip", ".h")):
            continue
        dev = 0  # nesting depth inside `#ifdef RG_DEV_KNOBS` (up to its #else / #endif)
        for line in open(os.path.join(csrc, f)):
            t = line.strip()
            if t.startswith("#ifdef RG_DEV_KNOBS"):
                dev = 1
            elif dev and (t.startswith("#else") or t.startswith("#endif")):
                dev = 0
            code = line.split("//")[0]
            if dev or "define RG_DEV_ENV" in code:
                continue
            names = re.findall(r'(?<![A-Za-z_])getenv\("([A-Z_0-9]+)"\)', code)
            if names:
                sites += 1
                seen.update(names)
    assert seen == allowed, seen ^ allowed
    assert sites <= 9
    tests = "".join(open(os.path.join(ROOT, "tests", f)).read() for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py"))
    for k in allowed:
        assert k in tests, "knob %s is not exercised by any test" % k


def test_step_wave_occupancy_never_exceeds_a_wave(lib):
    """rgk_step_epw (envs per index-order wave of k_step): a wave has 64 lanes, so 16 <= epw <= 64 for every batch size and both slot classes, and
    the index-order blocks cover every env.  (n = 64 513..65 472 with one slot per SIMD once gave 65: env 64 of every block was never stepped.)"""
    f = lib.rgk_step_epw
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int]
    sizes = list(range(1, 4097)) + list(range(60000, 70000)) + [1 << k for k in range(12, 21)] + [32768 + d for d in range(-70, 70)] + [262144, 1000000]
    for slots in (1, 2):
        for n in sizes:
            epw = f(n, slots)
            assert 16 <= epw <= 64, (n, slots, epw)
            assert ((n + epw - 1) // epw) * epw >= n
