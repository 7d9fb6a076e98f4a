"""Random soak of whatever oracle library ROGUE_ORACLE_SO names (tests/test_oracle_sanitizers.py runs it on the ASan + UBSan build): argv = env-steps.
Random keys of the ai keymap (run keys included) over the mini, default and nohide configs of the reference's tests, through the single-env entry points
(react with manual reset, step_autoreset), the debug descent, every mirror and both image encoders, and the threaded batch.  Prints `SOAK ok <steps>`.
A child process: the binding loads one library per process."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KEYS = np.frombuffer(b".hjklnbuy>sHJKLNBUY", np.uint8)


def main():
    from oracle import pyoracle
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    goldens = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))
    cfgs = [goldens["configs"][k] for k in ("mini", "default", "nohide") if k in goldens["configs"]]
    rng = np.random.RandomState(606)
    done, per_env = 0, 500
    seed = 1
    while done < total * 3 // 4:
        cfg = cfgs[seed % len(cfgs)]
        env = pyoracle.OracleEnv(cfg, max_steps=int(rng.choice([30, 200, 1000])), seed=seed)
        if seed % 7 == 0:
            for _ in range(int(rng.randint(1, 12))):
                env.debug_descend()
        keys = KEYS[rng.randint(0, len(KEYS), per_env)]
        for t, k in enumerate(keys):
            if seed % 2:
                env.step_autoreset(int(k))
            else:
                try:
                    env.react(int(k))
                except RuntimeError:  # dead / past max_steps: GameStateImpl::react's errors; the caller resets
                    env.reset()
            if t % 97 == 0:
                env.screen(); env.hist(); env.status(); env.flags()
                env.gray_image(flag=7)
                try:
                    env.symbol_image(flag=7)
                except RuntimeError:  # the reference's `Z` glyph error
                    pass
        done += per_env
        seed += 1
    # the threaded batch (cpu_baseline's entry point)
    n = 64
    b = pyoracle.OracleBatch([cfgs[0]] * n, max_steps=100, n_threads=4, seeds=list(range(1000, 1000 + n)))
    steps = max(1, (total - done) // n)
    for _ in range(steps):
        b.step(KEYS[rng.randint(0, 11, n)])
    done += steps * n
    print("SOAK ok %d" % done, flush=True)


if __name__ == "__main__":
    main()
