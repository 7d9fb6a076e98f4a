"""A fixed-seed slice of tools/fuzz_parity.py in the GPU suite: 30 random valid GameConfigs (sizes 32x16 .. 160x48 incl. widths that are not multiples
of 8, 1 .. 100+ rooms, random rates / monster subsets / packs), 32 envs x 100 random keys each, HIP vs the C oracle in lock step (mirrors every step,
tiles / doors / gold / monsters / RNG words at intervals).  The soak run of round 3 (546 configs, 48 envs x 150 steps, 0 differences) is recorded in
profiles/r03_fuzz_parity.txt."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_configs_in_lock_step(seed):
    import fuzz_parity as F
    from parity_util import lockstep
    from rogue_gym_python import _rogue_gym as inner

    L = inner.load_library()
    rng = np.random.RandomState(seed)
    done = 0
    while done < 10:
        cfg = F.random_config(rng)
        buf = (inner.C.c_char * 65536)()
        if L.rg_config_canonical(json.dumps(cfg).encode(), buf, len(buf)):
            continue
        done += 1
        try:
            F.run_one(cfg, 32, 100, rng, inner, lockstep)
        except AssertionError as e:
            raise AssertionError("config %s\n%s" % (json.dumps(cfg), e))
