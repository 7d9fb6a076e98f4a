"""The oracle's mutants (oracle/rogue_oracle.c, -DORC_MUTANT=k): ONE plausible misreading of the reference per RNG call site of SURVEY.md App. B and per
quirk of App. C.  Shared by tests/test_oracle_mutations.py (which goldens notice which mutant) and tools/pin_map.py (profiles/r05_pin_map.txt)."""

# k: (reference site, what the mutant does instead)
MUTANTS = {
    1: ("rooms.rs:179", "empty-room count drawn as 64 bits instead of u32"),
    2: ("rooms.rs:189 (rng.rs:134)", "empty-room ids selected with 32-bit draws instead of usize"),
    3: ("rooms.rs:226", "empty-room anchor drawn as 64 bits instead of i32"),
    4: ("rooms.rs:226", "empty-room anchor: y drawn before x"),
    5: ("rooms.rs:237", "dark roll drawn as 64 bits"),
    6: ("rooms.rs:238", "maze roll drawn as 64 bits"),
    7: ("rooms.rs:258,262", "room size / offset drawn as 64 bits instead of i32"),
    8: ("rooms.rs:258,262", "room draws in the order size x, offset x, size y, offset y"),
    9: ("maze.rs:73", "maze reservoir rolls drawn as 64 bits"),
    10: ("maze.rs:73", "maze direction: the FIRST successful roll wins instead of the last"),
    11: ("floor.rs:430,437", "gen_attr level roll drawn as 64 bits"),
    12: ("floor.rs:431,438", "gen_attr hidden / locked roll drawn as 64 bits"),
    13: ("floor.rs:430-438", "gen_attr: the hidden / locked roll is taken whatever the level roll said (no short circuit)"),
    14: ("passages.rs:30", "first room of the spanning tree drawn as 32 bits instead of usize"),
    15: ("passages.rs:56", "extra-edge room drawn as 32 bits instead of usize"),
    16: ("passages.rs:79", "neighbour reservoir rolls drawn as 64 bits"),
    17: ("passages.rs:49", "re-pick of a connected room drawn as 32 bits instead of usize"),
    18: ("passages.rs:54", "number of extra-edge tries drawn as 64 bits"),
    19: ("passages.rs:54", "number of extra-edge tries in 0..=max instead of 0..max"),
    20: ("passages.rs:146,156", "door cell chosen with a 32-bit index (the u32 `gen_index` form of SliceRandom::choose)"),
    21: ("passages.rs:98-99", "the END room's door is drawn before the start room's"),
    22: ("passages.rs:106,115", "corridor bend drawn as 64 bits instead of i32"),
    23: ("passages.rs:106,115", "corridor bend range includes the end door's row / column"),
    24: ("rooms.rs:134 via floor.rs:122,144,337-339", "free cell of a room drawn as 32 bits instead of usize"),
    25: ("floor.rs:337-339", "room id of Floor::select_cell drawn as 32 bits instead of usize"),
    26: ("gold.rs:19", "gold 1-in-2 roll drawn as 64 bits"),
    27: ("gold.rs:22", "gold amount drawn as 64 bits"),
    28: ("gold.rs:19-22", "gold amount drawn BEFORE the 1-in-2 roll"),
    29: ("gold.rs:22", "gold amount drawn on the dungeon stream instead of the item stream"),
    30: ("enemies.rs:297", "monster appearance roll drawn as 64 bits"),
    31: ("enemies.rs:266", "monster type index drawn as 64 bits"),
    32: ("enemies.rs:303 (character/mod.rs:218)", "monster hp dice drawn as 32 bits instead of i64"),
    33: ("enemies.rs:303", "monster hp dice roll 1..level instead of 1..=level"),
    34: ("enemies.rs:401", "monster turn: the 1-in-2 roll drawn as 64 bits"),
    35: ("enemies.rs:402", "monster turn: the 1-in-5 roll drawn as 64 bits"),
    36: ("enemies.rs:401-402", "monster turn: both rolls always taken (no short circuit)"),
    37: ("rogue/mod.rs:383", "random direction drawn as 32 bits instead of usize"),
    38: ("rogue/mod.rs:383", "random direction drawn on the enemy stream instead of the dungeon stream"),
    39: ("fight.rs:61", "to-hit roll drawn as 64 bits"),
    40: ("player.rs:193", "level-up hp gain drawn as 32 bits instead of i64"),
    41: ("player.rs:228", "heal amount (player level >= 8) drawn as 32 bits instead of i64"),
    42: ("player.rs:225", "the random heal starts at player level 9 instead of 8"),
    43: ("character/mod.rs:232", "damage dice drawn as 32 bits instead of i64"),
    44: ("passages.rs:33-51 (App. C-4)", "the spanning-tree walk advances to the room it has just connected"),
    45: ("floor.rs:92-100 (App. C-3)", "a hidden passage / locked door still paints its surface"),
    46: ("enemies.rs:366-424 (App. C-10)", "a monster that cannot move does not overwrite one that moved onto its cell"),
    47: ("rogue/mod.rs:492-518 (App. C-11)", "the DistCache is dropped on a new level"),
    48: ("rogue/mod.rs:339-375 (App. C-10)", "chasing monsters only consider legal moves (no corner cutting)"),
    49: ("rogue/mod.rs:339-375", "chasing monsters take the LAST minimum of the 3x3 instead of the first"),
    50: ("enemies.rs:205-213 (App. C-13)", "Enemy::get_damage stores cur - damage (the sane subtraction)"),
    51: ("fight.rs:74-78 (App. C-13)", "+4 to hit applies (the monster counts as not running when attacked)"),
    52: ("rooms.rs:237-238", "the maze roll is taken for lit rooms too"),
    53: ("rooms.rs:179", "empty-room count in 0..max instead of 0..=max"),
    56: ("floor.rs:359", "search: hidden-passage roll drawn as 64 bits"),
    57: ("floor.rs:363", "search: locked-door roll drawn as 64 bits"),
    58: ("floor.rs:264-295 (App. C-7)", "diagonal passage cells are revealed around the player too"),
    59: ("weapon.rs:159", "initial weapon counts drawn as 64 bits"),
    60: ("core/src/lib.rs:199-216 (App. C-1)", "the pack's item-stream draws come after the player's placement"),
    62: ("actions.rs:44-57 (App. C-8)", "the stopping iteration of a run costs a turn too"),
    63: ("actions.rs:62 (App. C-8)", "NoOp costs a turn"),
    64: ("actions.rs:58-61 (App. C-8)", "Search costs no turn"),
    65: ("player.rs:224 (App. C-9)", "natural healing one turn earlier (quiet + 2 level >= 20)"),
    66: ("floor.rs:298-312 (App. C-7)", "dark floor stays visible behind the player"),
    67: ("floor.rs:231-247 (App. C-7)", "dark rooms are lit up on entry like lit ones"),
    68: ("actions.rs:75 (App. C-9)", "starvation kills (PlayerEvent::Dead of turn_passed is honoured)"),
}

# The checks of tests/test_oracle_golden.py by who holds the expected values.  REFERENCE: data the reference's own tests / sources hold (python/tests/data.py
# screens, the constants of test_ff_env / test_st_env / test_rogue_env / test_parallel, the two Rust KATs).  SECONDARY: probe-derived answers of SURVEY.md
# App. A-5 and the oracle's own consistency checks -- they notice a change, but the reference does not vouch for the expected value.
REFERENCE = ["seed1_clear_map", "first_floor_env", "stair_reward_env", "seed1_enemies_screens[CMD_STR]", "seed1_enemies_screens[CMD_STR5]", "noaction",
             "max_steps_terminal", "move_enemy_kat", "shapes", "inclusive_edges_kat"]
SECONDARY = ["xorshift_known_answers", "seed1_clear_map_probe_answers", "mini_known_answers", "ddqn_trajectory", "dead_env_rejects_actions", "batch_matches_single"]


def build_mutant(k, out_dir):
    """gcc -DORC_MUTANT=k of the oracle's one source file -> out_dir/librogue_oracle_m<k>.so"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(out_dir, "librogue_oracle_m%d.so" % k)
    subprocess.check_call([os.environ.get("CC", "gcc"), "-O1", "-std=gnu11", "-fPIC", "-shared", "-DORC_MUTANT=%d" % k, "-o", so,
                           os.path.join(root, "oracle", "rogue_oracle.c"), "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so


def run_mutant(k, so, timeout=120):
    """{check: True (passed) / False (failed) / 'crash' (the oracle aborted or hung inside it)} for every check, one child process per crash."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    todo, res = list(REFERENCE + SECONDARY), {}
    while todo:
        env = dict(os.environ, ROGUE_ORACLE_SO=so)
        try:
            p = subprocess.run([sys.executable, os.path.join(root, "tests", "oracle_mutant_probe.py")] + todo, env=env, capture_output=True, text=True, timeout=timeout)
            out = p.stdout
        except subprocess.TimeoutExpired as e:
            out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        started = None
        for line in out.splitlines():
            w = line.split()
            if w[:1] == ["PROBE"] and w[1] == "mutant":
                assert int(w[2]) == k, "library %s reports mutant %s" % (so, w[2])
            elif w[:1] == ["PROBE-START"]:
                started = w[1]
            elif w[:1] == ["PROBE"]:
                res[w[1]] = w[2] == "1"
                started = None
        if started is not None:
            res[started] = "crash"
        left = [c for c in todo if c not in res]
        if len(left) == len(todo):
            raise RuntimeError("mutant %d: the probe made no progress:\n%s" % (k, out[-2000:]))
        todo = left
    return res


def kill_matrix(ks=None, jobs=None):
    """{k: {check: result}} for the mutants `ks` (default: all, plus 0 = the restatement itself), built and probed in parallel."""
    import os
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    ks = sorted(MUTANTS) + [0] if ks is None else list(ks)
    jobs = jobs or max(2, (os.cpu_count() or 4))
    with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(jobs) as ex:
        return dict(zip(ks, ex.map(lambda k: run_mutant(k, build_mutant(k, d)), ks)))


def summarise(matrix):
    """{k: {"reference": [checks of REFERENCE that notice], "secondary": [...]}} (k as str: JSON keys)"""
    out = {}
    for k, res in sorted(matrix.items()):
        out[str(k)] = {"reference": [c for c in REFERENCE if res[c] is not True], "secondary": [c for c in SECONDARY if res[c] is not True]}
    return out
