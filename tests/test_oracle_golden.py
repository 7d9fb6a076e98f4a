"""Pin the CPU oracle against every golden the reference's own tests hold for the hot path
(SURVEY.md section 8c).  CPU-only; the oracle is test infrastructure, never the product."""
import numpy as np
import pytest

from oracle.pyoracle import OracleBatch, OracleEnv, kat_u32


def diff_cells(a, b):
    assert len(a) == len(b)
    return sum(x != y for r1, r2 in zip(a, b) for x, y in zip(r1, r2))


def test_xorshift_known_answers():
    # SURVEY.md App. A-5 (probe-derived; seed -> first eight next_u32)
    assert [int(v) for v in kat_u32(0, 8)] == [0x6A963343, 0x0BAD52CA, 0x6A963F64, 0x0BAD5EED, 0xD07A78F0, 0xB1205B37, 0x6A963F68, 0x0BAD5EE1]
    assert [int(v) for v in kat_u32(1, 8)] == [0x809, 0x809, 0x809, 0x809, 0x400840, 0x801, 0x400848, 0x809]
    assert [int(v) for v in kat_u32(5, 8)] == [0x282D, 0x282D, 0x282D, 0x282D, 0x1402940, 0x2805, 0x1402968, 0x282D]


def test_seed1_clear_map(goldens):
    """python/tests/data.py:83-108 -- full level-1 map, seed 1, no enemies, nohide (1920 cells)."""
    e = OracleEnv(goldens["configs"]["ff"])
    assert diff_cells(e.dungeon(), goldens["screens"]["SEED1_DUNGEON_CLEAR"]) == 0


def test_seed1_clear_map_probe_answers(goldens):
    """SURVEY.md App. A-5 (probe-derived secondary fixture, not held by the reference): the player's cell and the item stream's position after the build."""
    e = OracleEnv(goldens["configs"]["ff"])
    assert (e.scalars()["px"], e.scalars()["py"]) == (14, 20)
    assert e.rng()[1][1] == 21  # item rng: 18 level-1 gold draws + 3 init-weapon draws


def test_choose_is_64bit(goldens):
    """The u32 `gen_index` variant of SliceRandom::choose does NOT reproduce the golden."""
    e = OracleEnv(goldens["configs"]["ff"], choose_width=32)
    assert diff_cells(e.dungeon(), goldens["screens"]["SEED1_DUNGEON_CLEAR"]) > 0


def test_first_floor_env(goldens):
    """python/tests/test_ff_env.py:14-22 -- CMD_STR2 reaches level 2 with gold 2 (reward 102 = 2 + 100)."""
    e = OracleEnv(goldens["configs"]["ff"])
    e.react_str(goldens["keys"]["CMD_STR2"])
    st = e.status()
    assert st["dungeon_level"] == 2
    assert st["gold"] + 100 == goldens["expect"]["ff"]["reward_with_stair100"]
    assert list(e.symbol_image(flag=1).shape) == goldens["expect"]["ff"]["symbol_image_shape"]


def test_stair_reward_env(goldens):
    """python/tests/test_st_env.py:27-37"""
    ex = goldens["expect"]["st"]
    e = OracleEnv(goldens["configs"]["st"])
    e.react_str(goldens["keys"]["CMD_STR3"])
    assert e.status()["gold"] + 100.0 == ex["first"]["reward_with_stair100"]
    assert e.status()["dungeon_level"] == 2
    g = e.status()["gold"]
    e.react_str(goldens["keys"]["CMD_STR4"])
    assert (e.status()["gold"] - g) + 100.0 == ex["second"]["reward_with_stair100"]
    # ImageSetting(SYMBOL, DUNGEON_LEVEL|HP_CURRENT|EXP, hist)
    img = e.symbol_image(flag=0b010000011, with_hist=True)
    assert list(img.shape) == ex["second"]["image_shape"]
    assert img[17][0][0] == ex["second"]["plane17"]
    assert img[18][0][0] == ex["second"]["plane18"]
    assert e.status_vec(0b111111111) == ex["second"]["status_vec_full"]


@pytest.mark.parametrize("keys,screen", [("CMD_STR", "SEED1_DUNGEON2"), ("CMD_STR5", "SEED1_DUNGEON3")])
def test_seed1_enemies_screens(goldens, keys, screen):
    """python/tests/data.py:28-81 -- seed 1 with default enemies, hidden dungeon: spawn, activation,
    BFS chase, monster attack draws, dark-room FoV, runs, Redraw-only mirror refresh."""
    e = OracleEnv(goldens["configs"]["seed1"])
    e.react_str(goldens["keys"][keys])
    assert diff_cells(e.dungeon(), goldens["screens"][screen]) == 0


def test_noaction(goldens):
    """python/tests/test_rogue_env.py:31-36"""
    e = OracleEnv(goldens["configs"]["seed1"])
    d, s = e.dungeon(), e.status()
    e.react(".")
    assert e.dungeon() == d and e.status() == s


def test_max_steps_terminal(goldens):
    """python/tests/test_rogue_env.py:39-42, test_parallel.py:51-60"""
    e = OracleEnv(goldens["configs"]["seed1"], max_steps=5)
    first = e.dungeon()
    for i, c in enumerate(goldens["keys"]["CMD_STR"]):
        if i >= 5:
            break
        e.step_autoreset(c)
        assert e.flags()["is_terminal"] == (i == 4)
    assert e.dungeon() == first  # post-reset state is returned with is_terminal forced true


def test_move_enemy_kat(goldens):
    """core/src/dungeon/rogue/mod.rs:566-578 -- seed 5 mini + enemies: (9,9) chasing (28,4) steps Right."""
    k = goldens["expect"]["move_enemy_kat"]
    e = OracleEnv(goldens["configs"]["move_enemy_kat"])
    r, nx, ny = e.move_enemy_kat(*k["from"], *k["to"])
    assert r == 1 and [nx, ny] == k["next"]


def test_mini_known_answers(goldens):
    """SURVEY.md App. A-5 (probe-derived secondary fixture): data/config-mini.json (seed 4)."""
    e = OracleEnv(goldens["configs"]["mini"])
    sc = e.scalars()
    assert (sc["px"], sc["py"]) == (9, 12)
    assert [(m["x"], m["y"], chr(65 + m["type"])) for m in e.monsters()] == [(6, 5, "K"), (9, 11, "B"), (19, 9, "S"), (24, 4, "K")]
    assert e.rng()[1] == [379, 11, 92]
    assert e.dungeon()[10:14] == [r.ljust(32) for r in (" --------+--", " |.......B.|", " |.*.....@.+", " -----------")]


def test_ddqn_trajectory(goldens):
    """the 1000-action DDQN replay ends on level 2 with gold 4, hp 12/12."""
    e = OracleEnv(goldens["configs"]["ddqn"], max_steps=2000)
    e.react_str(goldens["ddqn_keys"])
    st = e.status()
    assert (st["dungeon_level"], st["gold"], st["hp_current"], st["hp_max"]) == (2, 4, 12, 12)


def test_shapes(goldens):
    """python/tests/test_rogue_env.py:47-68"""
    sh = goldens["expect"]["shapes"]
    e = OracleEnv(goldens["configs"]["seed1_noenem"])
    e.react("H")
    img = e.symbol_image(flag=0, with_hist=True)
    assert list(img.shape) == sh["symbol_hist_noenem"]
    assert (img[-1][20][2:15] == 1.0).all()
    assert list(e.gray_image(0).shape) == sh["gray"]
    assert list(e.gray_image(0, with_hist=True).shape) == sh["gray_hist"]
    assert e.symbols + 9 == sh["space_noenem_full"][0]
    assert OracleEnv(goldens["configs"]["seed1"]).symbols == 43


def test_dead_env_rejects_actions(goldens):
    """core/src/lib.rs:301-315: action keys in the Grave modal are IgnoredInput errors."""
    rng = np.random.RandomState(0)
    keys = "hjklyubn"
    for seed in range(30):
        e = OracleEnv(goldens["configs"]["mini"], seed=seed, max_steps=100000)
        for _ in range(3000):
            e.react(keys[rng.randint(8)])
            if e.flags()["dead"]:
                break
        if e.flags()["dead"]:
            assert e.flags()["is_terminal"]
            with pytest.raises(RuntimeError):
                e.react("h")
            return
    pytest.fail("no death in 30 seeds")


def test_batch_matches_single(goldens):
    cfg = goldens["configs"]["mini"]
    n = 16
    rng = np.random.RandomState(1)
    acts = np.frombuffer(b".hjklnbuy>s", np.uint8)
    b = OracleBatch([cfg] * n, max_steps=50, n_threads=4, seeds=list(range(n)))
    singles = [OracleEnv(cfg, max_steps=50, seed=i) for i in range(n)]
    obs = np.zeros((n, 1, 16, 32), np.float32)
    for _ in range(120):
        keys = acts[rng.randint(0, 11, n)]
        b.step(keys, obs)
        for i, e in enumerate(singles):
            e.step_autoreset(int(keys[i]))
    for i, e in enumerate(singles):
        be = b.env(i)
        assert (be.screen() == e.screen()).all()
        assert be.status() == e.status()
        assert np.array_equal(obs[i], e.gray_image(0))


def test_inclusive_edges_kat(goldens):
    """The reference's own deterministic KAT of passages::edges (passages.rs:272-296), transcribed into the goldens: pins the oracle's
    edges() -- and with it rect-iter's corner naming -- directly, not only through SEED1_DUNGEON_CLEAR."""
    from oracle.pyoracle import kat_edges
    k = goldens["expect"]["inclusive_edges"]
    for d in ("Down", "Up", "Left", "Right"):
        assert kat_edges(k["range"]["x"], k["range"]["y"], d, True) == k[d], d
