"""Reward-shaping wrappers: a bonus whenever the player reaches a deeper dungeon level.

Public names and semantics follow /root/reference/python/rogue_gym/envs/wrappers.py:12-64 (pinned by python/tests/test_ff_env.py,
test_st_env.py, test_parallel.py): `StairRewardEnv` pays once per new deepest level of the episode, `FirstFloorEnv` additionally ends the
episode on level 2, `StairRewardParallel` compares every env with the level it reported one step earlier (so the bookkeeping follows the
auto-reset back to level 1).  The batched wrapper is vectorised over the StateBatch; `HipVecStairReward` (device.py) is the same rule on
device tensors.
"""
from typing import Iterable, List, Tuple, Union

import numpy as np

from rogue_gym_python._rogue_gym import StateBatch

from ._gym_compat import Env, Wrapper
from .parallel import ParallelRogueEnv
from .rogue_env import PlayerState, RogueEnv


def check_rogue_env(env: Env) -> None:
    """The wrappers read `env.unwrapped.result`, which only a RogueEnv has."""
    base = getattr(env, "unwrapped", None)
    if not isinstance(base, RogueEnv):
        raise ValueError("env have to be a wrapper of RoguEnv, got {}".format(type(base).__name__))


class StairRewardEnv(Wrapper):
    def __init__(self, env: Env, stair_reward: float = 50.0) -> None:
        check_rogue_env(env)
        Wrapper.__init__(self, env)
        self.stair_reward, self.current_level = stair_reward, 1

    def step(self, action: Union[int, str]) -> Tuple[PlayerState, float, bool, dict]:
        state, reward, done, info = self.env.step(action)
        level = self.unwrapped.result.status["dungeon_level"]
        if level > self.current_level:  # one bonus per step, however many levels the key string descended
            reward, self.current_level = reward + self.stair_reward, level
        return state, reward, done, info

    def reset(self) -> PlayerState:
        self.current_level = 1
        return self.env.reset()

    def __repr__(self) -> str:
        return repr(self.env)


class FirstFloorEnv(StairRewardEnv):
    """The episode is over as soon as the second level is reached."""

    def step(self, action: Union[int, str]) -> Tuple[PlayerState, float, bool, dict]:
        state, reward, done, info = super().step(action)
        return state, reward, done or self.current_level == 2, info


class StairRewardParallel(ParallelRogueEnv):
    def __init__(self, *args, stair_reward: float = 50.0, **kwargs) -> None:
        ParallelRogueEnv.__init__(self, *args, **kwargs)
        self.stair_reward = float(stair_reward)
        self.current_levels = np.ones(self.num_workers, np.int64)

    def step(self, action: Union[Iterable[int], str]) -> Tuple[StateBatch, List[float], List[bool], List[dict]]:
        states, rewards, dones, infos = super().step(action)
        levels = states.dungeon_level.astype(np.int64)
        deeper = levels > self.current_levels
        if deeper.any():
            rewards = (np.asarray(rewards, np.float64) + self.stair_reward * deeper).tolist()
        self.current_levels = levels
        return states, rewards, dones, infos
