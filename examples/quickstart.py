"""Quick start: the three ways in, on one MI355X.

    python examples/quickstart.py

1. `RogueEnv`           -- the reference's gym.Env, one game (python/rogue_gym/envs/rogue_env.py): same constructor, same step() tuple.
2. `ParallelRogueEnv`   -- the reference's batched executor (parallel.py): N games, one kernel launch per key, value-object states.
3. `HipVecRogueEnv`     -- the tensor-native form: observations, rewards and dones stay in HBM as PyTorch-ROCm tensors (what bench.py measures).

Needs the built library (`python -c "import __graft_entry__ as g; g.build()"`) and a GPU; there is no CPU fallback."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))

import numpy as np
import torch

from rogue_gym.envs import DungeonType, HipVecRogueEnv, ImageSetting, ParallelRogueEnv, RogueEnv, StatusFlag

MINI = {"width": 32, "height": 16, "seed": 4, "hide_dungeon": True,
        "dungeon": {"style": "rogue", "room_num_x": 2, "room_num_y": 2}, "enemies": {"enemies": []}}  # the reference's data/config-mini.json

# 1. one game, the gym surface -------------------------------------------------------------------------------------------------------------
env = RogueEnv(config_dict=MINI, max_steps=50, image_setting=ImageSetting(DungeonType.SYMBOL, StatusFlag.DUNGEON_LEVEL | StatusFlag.HP_CURRENT))
state = env.reset()
total = 0.0
for key in "hjklyubn>s" * 3:
    state, reward, done, _ = env.step(key)          # a key of RogueEnv.ACTIONS or its index
    total += reward
    if done:
        state = env.reset()
print("RogueEnv: obs", env.image_setting.expand(state).shape, "gold so far", total)
print(env)                                           # the screen, like the reference's __repr__

# 2. many games, value objects -------------------------------------------------------------------------------------------------------------
n = 1024
penv = ParallelRogueEnv([dict(MINI, seed=i) for i in range(n)], max_steps=100)
rng = np.random.RandomState(0)
states = penv.states
for _ in range(50):
    states, rewards, dones, _ = penv.step(rng.randint(0, penv.ACTION_LEN, n))   # terminal envs are reset inside step()
print("ParallelRogueEnv: %d envs, %d finished an episode in the last step, images %s" % (n, sum(dones), states.images(0, 0, False).shape))
penv.close()

# 3. tensors that never leave HBM -----------------------------------------------------------------------------------------------------------
n = 65536
venv = HipVecRogueEnv([dict(MINI, seed=i) for i in range(n)], max_steps=1000, image_setting=ImageSetting(DungeonType.GRAY, StatusFlag.EMPTY, False), device=0)
actions = torch.randint(0, 11, (64, n), device=venv.device)
for t in range(100):
    obs, reward, done = venv.step(actions[t % 64])   # f32 [N, C, H, W], f32 [N], bool / u8 [N] -- device tensors, reused every step
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(500):
    obs, reward, done = venv.step(actions[t % 64])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("HipVecRogueEnv: %d envs, obs %s on %s, %.0f M env-steps/s" % (n, tuple(obs.shape), obs.device, n * 500 / dt / 1e6))
venv.close()

# 3b. the same with the observation tensor BOUND to the stepper (opt-in): `obs` is kept current in place -- only the envs whose screen changed are rewritten;
#     contents identical, the caller must not write to it ----------------------------------------------------------------------------------------------
venv = HipVecRogueEnv([dict(MINI, seed=i) for i in range(n)], max_steps=1000, image_setting=ImageSetting(DungeonType.GRAY, StatusFlag.EMPTY, False), device=0,
                      persistent_obs=True)
for t in range(100):
    obs, reward, done = venv.step(actions[t % 64])
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(500):
    obs, reward, done = venv.step(actions[t % 64])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("HipVecRogueEnv(persistent_obs=True): %.0f M env-steps/s" % (n * 500 / dt / 1e6))
venv.close()
