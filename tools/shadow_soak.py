"""10^6 env-steps of the C oracle against the second restatement of the turn (tests/shadow_turn.py), in parallel processes; then the same harness against every
oracle mutant no reference golden notices (profiles/r05_pin_map.txt), to record which of them the differential catches.  Writes profiles/r05_shadow_diff.txt.
CPU only: python tools/shadow_soak.py [--steps 1000000] [--jobs N]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def configs():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))
    mini = dict(g["configs"]["mini"])
    strong = dict(mini, player={"init_hp": 60, "init_items": [{"Weapon": {"name": "two-handed-sword", "num_plus": 0, "hit_plus": 3, "dam_plus": 3}}, {"Armor": {"name": "plate mail", "def_plus": 2}}]})
    tank = dict(mini, player={"init_hp": 400, "init_items": [{"Weapon": {"name": "two-handed-sword", "num_plus": 0, "hit_plus": 3, "dam_plus": 3}}, {"Armor": {"name": "plate mail", "def_plus": 2}}]})
    hard = dict(mini, dungeon=dict(mini["dungeon"], hidden_passage_rate_inv=2, locked_door_rate_inv=2, dark_level=1))
    default = dict(g["configs"]["seed1"])
    quiet = dict(mini, enemies={"enemies": []})
    return dict(mini=(mini, {}), strong=(strong, dict(weapon=(4, 4, 3, 3), armor=9)), tank=(tank, dict(weapon=(4, 4, 3, 3), armor=9)), hard=(hard, {}), default=(default, {}),
                quiet=(quiet, {}))


def chunk(args):
    """One work item: (config name, first seed, seeds, keys per seed, max_steps, policy name, numpy seed) -> (env-steps, stats) or the first difference."""
    name, seed0, n_seeds, steps, max_steps, policy, rs = args
    import numpy as np
    import test_oracle_shadow as T
    cfg, kw = configs()[name]
    stats = {}
    try:
        n = T.run_differential(cfg, range(seed0, seed0 + n_seeds), steps, max_steps, getattr(T, policy), np.random.RandomState(rs), grid_every=3, stats=stats, **kw)
        return ("ok", n, stats)
    except AssertionError as e:
        return ("differ", str(e)[:300], stats)


def plan(total):
    """Work items adding up to ~`total` env-steps: random and stairs-seeking play on the mini dungeon, the strong pack (kills, level-ups, deep levels), a 400-hp
    player (player level >= 8: the random heal), many hidden passages / locked doors (search), the 80x24 dungeon, and 3 000-key episodes without monsters (hunger wrap)."""
    items, rs = [], 1000
    per = total // 100
    for i in range(30):
        items.append(("mini", 10000 + 50 * i, 25, per // 25, 300, "random_policy", rs + i))
    for i in range(20):
        items.append(("mini", 20000 + 50 * i, 20, per // 20, 500, "mixed_policy", rs + 100 + i))
    for i in range(20):
        items.append(("strong", 30000 + 50 * i, 10, per // 10, 1500, "mixed_policy", rs + 200 + i))
    for i in range(12):
        items.append(("tank", 40000 + 50 * i, 4, per // 4, 4000, "mixed_policy", rs + 300 + i))
    for i in range(8):
        items.append(("hard", 50000 + 50 * i, 20, per // 20, 400, "mixed_policy", rs + 400 + i))
    for i in range(6):
        items.append(("default", 60000 + 50 * i, 5, per // 5, 600, "mixed_policy", rs + 500 + i))
    for i in range(4):
        items.append(("quiet", 70000 + 50 * i, 3, per // 3, 3000, "random_policy", rs + 600 + i))
    return items


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000000)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 4)
    a = ap.parse_args()
    from oracle import pyoracle
    pyoracle.build()
    t0 = time.time()
    items = plan(a.steps)
    with ProcessPoolExecutor(a.jobs) as ex:
        res = list(ex.map(chunk, items))
    total, stats, bad = 0, {}, []
    for it, r in zip(items, res):
        if r[0] == "ok":
            total += r[1]
        else:
            bad.append((it, r[1]))
        for k, v in r[2].items():
            stats[k] = max(stats.get(k, 0), v) if k.startswith("max") else stats.get(k, 0) + v
    lines = ["# C oracle (oracle/rogue_oracle.c) vs the second restatement of the turn (tests/shadow_turn.py): tools/shadow_soak.py --steps %d, %.0f s on %d processes" % (a.steps, time.time() - t0, a.jobs),
             "env-steps compared (whole state after every key): %d   differences: %d" % (total, len(bad)),
             "covered: %s" % json.dumps(stats, sort_keys=True)]
    for it, msg in bad:
        lines.append("DIFFERENCE in %s: %s" % (it, msg))
    # the mutants no reference golden notices: does the differential catch them?
    from oracle_mutants import MUTANTS, build_mutant
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "mutant_pins.json")))
    turn_mutants = [k for k in sorted(MUTANTS) if not pins[str(k)]["reference"] and k not in (2, 26, 27, 28, 45, 59, 60)]
    probe = plan(max(a.steps // 2, 100000))
    probe.sort(key=lambda it: {"tank": 0, "quiet": 1, "hard": 2}.get(it[0], 3))  # the rare branches first: level >= 8 heals, the hunger wrap, locked doors

    def hunt(k):
        with tempfile.TemporaryDirectory() as d:
            so = build_mutant(k, d)
            code = ("import sys\nsys.path[:0] = [%r, %r, %r]\nimport shadow_soak as S\nfor it in %r:\n    r = S.chunk(it)\n    if r[0] == 'differ':\n        print('CAUGHT', it[0], r[1][:160].replace(chr(10), ' '))\n        break\nelse:\n    print('SURVIVES')\n"
                    % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), probe))
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROGUE_ORACLE_SO=so), capture_output=True, text=True)
            return k, (r.stdout.strip().splitlines() or ["ERROR " + r.stderr[-300:]])[-1]
    with ThreadPoolExecutor(a.jobs) as ex:
        hunted = list(ex.map(hunt, turn_mutants))
    lines.append("")
    lines.append("# every TURN mutant the reference's goldens miss, played against the second restatement (up to %d env-steps each):" % sum(i[2] * i[3] for i in probe))
    for k, verdict in hunted:
        lines.append("M%-3d %-40s %s   -> %s" % (k, MUTANTS[k][0], MUTANTS[k][1], verdict))
    lines.append("# generator sites without a reference pin (M2, M28, M45) and the two no CPU golden can see (M60: the two draws are on different streams and commute):")
    lines.append("# see tests/test_oracle_shadow.py::test_unpinned_generator_sites")
    out = "\n".join(lines) + "\n"
    open(os.path.join(ROOT, "profiles", "r05_shadow_diff.txt"), "w").write(out)
    print(out)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
