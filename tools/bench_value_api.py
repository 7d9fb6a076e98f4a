"""Throughput of the drop-in VALUE-OBJECT API (PCIe and Python included): ParallelRogueEnv.step(actions) followed by the images of all envs,
the way the reference's callers use it (python/rogue_gym/envs/parallel.py:44-66 + ImageSetting.expand).  VERDICT r1 target: >= 5 M env-steps/s
at 8 192 envs.  Usage: python tools/bench_value_api.py [n_envs ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))
import numpy as np

from rogue_gym.envs import DungeonType, ImageSetting, ParallelRogueEnv, StatusFlag

cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["configs"]["mini"]
sizes = [int(a) for a in sys.argv[1:]] or [64, 1024, 8192, 65536]
for n in sizes:
    st = ImageSetting(DungeonType.GRAY, StatusFlag.EMPTY, False)
    env = ParallelRogueEnv([dict(cfg, seed=i) for i in range(n)], max_steps=1000, image_setting=st)
    rng = np.random.RandomState(0)
    acts = [rng.randint(0, 11, n) for _ in range(16)]
    for t in range(20):
        env.step(acts[t % 16]); env.images()
    res = {}
    for mode in ("step only", "step + images", "step + images + per-state access of 64 states"):
        steps = max(20, min(400, 2_000_000 // n))
        t0 = time.perf_counter()
        for t in range(steps):
            states, rewards, dones, _ = env.step(acts[t % 16])
            if mode != "step only":
                img = env.images()
            if mode.endswith("states"):
                for i in range(0, n, max(1, n // 64)):
                    _ = states[i].gold
        dt = time.perf_counter() - t0
        res[mode] = n * steps / dt
    print("ParallelRogueEnv %6d envs: " % n + "  ".join("%s %.2f M env-steps/s" % (k, v / 1e6) for k, v in res.items()), flush=True)
    env.close()
